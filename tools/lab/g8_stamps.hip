// Phase timing of the product eight-phase GEMM (diagnostic library built with -DMVOC_G8_STAMPS): s_memtime at kernel entry,
// end of prologue, end of the K loop, end of each epilogue pass, stores drained -- block 0, waves 0 (group 0) and 4 (group 1).
//   usage: g8_stamps M N K tile(81|82) resid(0|1) [form: 0 plain, 2 LayerNorm fold, 3 LayerNorm fold + GEGLU; + 8: with bias and a per-256-row row-add]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mvoc_hip.h"

typedef _Float16 half_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
extern "C" int mvoc_g8_stamps_read(unsigned long long*);

__global__ void fill(half_t* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (half_t)(((int)(x & 255) - 128) / 256.0f);
  }
}

int main(int argc, char** argv) {
  const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), tile = atoi(argv[4]), use_r = atoi(argv[5]);
  half_t *W, *A, *O, *R;
  CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&O, (size_t)M * N * 2)); CK(hipMalloc(&R, (size_t)M * N * 2));
  fill<<<2048, 256>>>(W, (size_t)N * K, 1u); fill<<<2048, 256>>>(A, (size_t)M * K, 2u); fill<<<2048, 256>>>(R, (size_t)M * N, 3u);
  mvoc_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.a = A; d.w = W; d.out = O; d.m = M; d.n = N; d.k = K; d.n_store = N; d.ldo = N; d.lda = K; d.c1 = K; d.cin = K;
  d.a_mode = 0; d.tile = tile; d.split_k = 1;
  const int form0 = argc > 6 ? atoi(argv[6]) : 0;
  const int form = form0 & 7;
  half_t *B, *RA;
  CK(hipMalloc(&B, (size_t)N * 2)); CK(hipMalloc(&RA, (size_t)(M / 256 + 1) * N * 2));
  fill<<<64, 256>>>(B, (size_t)N, 4u); fill<<<2048, 256>>>(RA, (size_t)(M / 256 + 1) * N, 5u);
  if (form0 & 8) { d.bias = B; d.rowadd = RA; d.ld_rowadd = N; d.rowadd_div = 256; }
  float *LS, *LC, *ST;
  CK(hipMalloc(&LS, (size_t)N * 4)); CK(hipMalloc(&LC, (size_t)N * 4)); CK(hipMalloc(&ST, (size_t)M * 8));
  CK(hipMemset(LS, 0, (size_t)N * 4)); CK(hipMemset(LC, 0, (size_t)N * 4)); CK(hipMemset(ST, 0, (size_t)M * 8));
  if (form >= 2) { d.ln_rowsum = LS; d.ln_bias = LC; d.ln_stats = ST; d.ln_eps = 1e-5f; }
  if (form == 3) { d.act = MVOC_ACT_GEGLU; d.n_store = N / 2; d.ldo = N / 2; }
  if (use_r) { d.resid = R; d.ldr = d.ldo; }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) if (mvoc_gemm_f16(&d, 0)) { fprintf(stderr, "%s\n", mvoc_last_error()); return 1; }
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 5; ++i) mvoc_gemm_f16(&d, 0);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long t[32];
  if (mvoc_g8_stamps_read(t)) { fprintf(stderr, "no stamps\n"); return 1; }
  const int bx = tile == 82 ? 320 : 256;
  const long blocks = (long)((M + 255) / 256) * ((N + bx - 1) / bx);
  printf("M=%d N=%d K=%d tile %d resid %d form %d: %.1f us per launch, %.0f TF/s, %ld blocks = %.2f rounds, %d K tiles\n", M, N, K, tile, use_r, form,
         ms / 5 * 1e3, 2.0 * M * N * K / (ms / 5 * 1e-3) * 1e-12, blocks, blocks / 256.0, K / 64);
  {  // the clock the chip held during the K loop of block 0: shader cycles per 100 MHz reference tick
    const double clk = (double)(t[2] - t[1]) / (double)(t[16 + 2] - t[16 + 1]) * 0.1;
    const int bxw = tile == 82 ? 320 : 256;
    const double slots = (double)(K / 64) * 4 * (bxw / 16) * 16 * 2;  // K tiles x phases x MFMAs per wave-phase x 16 cycles x 2 waves per SIMD
    printf("  K loop: held clock %.3f GHz; MFMA slots filled %.1f %%; every slot filled at this clock = %.0f TFLOP/s on the chip (2 500 nominal at 2.4 GHz)\n",
           clk, 100.0 * slots / (double)(t[2] - t[1]), 2.0 * 256 * bxw * 64 / (4.0 * (bxw / 16) * 16 * 2) * 256 * clk * 1e-3);
  }
  for (int g = 0; g < 2; ++g) {
    const unsigned long long* s = t + 8 * g;
    printf("  group %d (ticks): prologue %llu | K loop %llu (%.0f per tile) | epilogue pass 0 %llu (arithmetic + LDS write %llu, readback + stores %llu) | pass 1 %llu | store drain %llu | block %llu\n", g,
           s[1] - s[0], s[2] - s[1], (double)(s[2] - s[1]) / (K / 64), s[3] - s[2], s[6] - s[2], s[3] - s[6], s[4] - s[3], s[5] - s[4], s[5] - s[0]);
  }
  return 0;
}
