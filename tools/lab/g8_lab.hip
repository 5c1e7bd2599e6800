// GEMM main-loop laboratory (GPU box only; diagnostic, not part of libmvoc_hip.so).
//   out[y][x] = sum_k X[x][k] * Y[y][k]      X = weights [NX][K], Y = activations [NY][K], fp16, fp32 accumulate
// g8_kernel: 256 x 256 tile per 8-wave block, K tile 64, eight half-tile LDS slots (2 K tiles x {Yh0, Xh0, Yh1, Xh1}),
// LDS-DMA three half-tiles ahead behind counted vmcnt, raw s_barrier, the two wave groups one barrier apart
// (cdna_hip_programming.md "The 256^2 8-phase template").  Baseline in the same process: mvoc_gemm_f16 tiles 67 / 66.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/mvoc_hip.h"

typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GLDS(gp, lp)                                                                               \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp),             \
                                   (__attribute__((address_space(3))) void*)(lp), 16, 0, 0)
#define BAR()                                \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)
#define LBAR()                               \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    if (!(ABL & 1)) __builtin_amdgcn_s_barrier(); \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, x = bid & 7u, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

constexpr int HT = 16384;  // bytes per half-tile slot: 128 rows x 128 B

// SHAPE 0: v_mfma_f32_16x16x32_f16, 1: v_mfma_f32_32x32x16_f16
template <int SHAPE, bool STAG, bool PRIO, int ABL = 0>  // ABL (timing only, wrong results): 1 = no barriers in the loop, 2 = no LDS-DMA in the loop, 4 = no fragment reads in the loop
__global__ __launch_bounds__(512) void g8_kernel(const half_t* __restrict__ X, const half_t* __restrict__ Y,
                                                 half_t* __restrict__ out, int NX, int NY, int K, int ldo, int nxt) {
  __shared__ __attribute__((aligned(1024))) char smem[8 * HT];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const int x0 = (int)(logical % (unsigned)nxt) * 256;
  const int y0 = (int)(logical / (unsigned)nxt) * 256;
  const int nk = K / 64;

  // ---- LDS-DMA sources: piece j = wave + 8 p covers local rows 8 j .. 8 j + 7 of a half-tile ------------------------
  const int lrow = lane >> 3, pos = lane & 7;
  const half_t* px[2][2];
  const half_t* py[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int lr = 8 * (wave + 8 * p) + lrow;
      const int c = pos ^ ((lr >> 1) & 7);
      const int xr = x0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
      const int yr = y0 + (lr >> 5) * 64 + h * 32 + (lr & 31);
      px[h][p] = X + (size_t)xr * K + c * 8;
      py[h][p] = Y + (size_t)yr * K + c * 8;
    }
  // kind 0: Y half 0, 1: X half 0, 2: Y half 1, 3: X half 1
#define ISSUE(KIND, SLOT)                                                     \
  do {                                                                        \
    char* b_ = smem + (SLOT) * HT + wave * 1024;                              \
    if ((KIND) == 0) { GLDS(py[0][0], b_); GLDS(py[0][1], b_ + 8192); py[0][0] += 64; py[0][1] += 64; } \
    if ((KIND) == 1) { GLDS(px[0][0], b_); GLDS(px[0][1], b_ + 8192); px[0][0] += 64; px[0][1] += 64; } \
    if ((KIND) == 2) { GLDS(py[1][0], b_); GLDS(py[1][1], b_ + 8192); py[1][0] += 64; py[1][1] += 64; } \
    if ((KIND) == 3) { GLDS(px[1][0], b_); GLDS(px[1][1], b_ + 8192); px[1][0] += 64; px[1][1] += 64; } \
  } while (0)

  // ---- fragment read addresses ------------------------------------------------------------------------------------------
  // SHAPE 0: lane reads row (l & 15) of a 16-row tile, chunk 4 s + (l >> 4);  SHAPE 1: row (l & 31), chunk 2 s + (l >> 5)
  constexpr int TR = SHAPE == 0 ? 16 : 32;      // rows per MFMA tile
  constexpr int NS = SHAPE == 0 ? 2 : 4;        // k-steps per K tile
  constexpr int XT = 64 / TR, YT = 32 / TR;     // tiles per quadrant
  const int fr = lane & (TR - 1), fg = lane / TR;
  const int sw = (fr >> 1) & 7;
  const char* xrd = smem + (wr * 64 + fr) * 128;
  const char* yrd = smem + (wc * 32 + fr) * 128;
  int coff[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) coff[s] = (((SHAPE == 0 ? 4 * s : 2 * s) + fg) ^ sw) * 16;

  half8_t xf[XT][NS], yf0[YT][NS], yf1[YT][NS];
  typedef float accv __attribute__((ext_vector_type(SHAPE == 0 ? 4 : 16)));
  accv acc[2][2][XT][YT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < XT; ++i)
#pragma unroll
        for (int j = 0; j < YT; ++j)
#pragma unroll
          for (int e = 0; e < (SHAPE == 0 ? 4 : 16); ++e) acc[a][b][i][j][e] = 0.f;

#define RDX(SLOT)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < XT; ++i) _Pragma("unroll") for (int s = 0; s < NS; ++s)          \
      xf[i][s] = *reinterpret_cast<const half8_t*>(xrd + (SLOT) * HT + i * TR * 128 + coff[s]);
#define RDY(DST, SLOT)                                                                                    \
  _Pragma("unroll") for (int j = 0; j < YT; ++j) _Pragma("unroll") for (int s = 0; s < NS; ++s)          \
      DST[j][s] = *reinterpret_cast<const half8_t*>(yrd + (SLOT) * HT + j * TR * 128 + coff[s]);
#define MMA(A, B, YF)                                                                                     \
  do {                                                                                                    \
    if (PRIO) __builtin_amdgcn_s_setprio(1);                                                              \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) _Pragma("unroll") for (int i = 0; i < XT; ++i)        \
        _Pragma("unroll") for (int j = 0; j < YT; ++j) {                                                  \
      if constexpr (SHAPE == 0)                                                                           \
        acc[A][B][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[i][s], YF[j][s], acc[A][B][i][j], 0, 0, 0); \
      else                                                                                                \
        acc[A][B][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[i][s], YF[j][s], acc[A][B][i][j], 0, 0, 0); \
    }                                                                                                     \
    if (PRIO) __builtin_amdgcn_s_setprio(0);                                                              \
  } while (0)

  // ---- prologue: half-tiles 0..6 in flight, K tile 0 landed ---------------------------------------------------------------
  const int nh = 4 * nk;
  ISSUE(0, 0); ISSUE(1, 1); ISSUE(2, 2); ISSUE(3, 3);
  if (nk > 1) {
    ISSUE(0, 4); ISSUE(1, 5); ISSUE(2, 6);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  BAR();
  if (STAG && wr == 1) BAR();

  // one K tile = 4 phases; P = buffer parity of the tile (slots 4P .. 4P+3), phase j issues half-tile 4 t + 6 + j
#define KTILE(P, t)                                                                                       \
  do {                                                                                                    \
    const int hb_ = 4 * (t) + 6;                                                                          \
    /* phase 1: (X0, Y0) */                                                                               \
    if (!(ABL & 4)) { RDY(yf0, 4 * (P) + 0); }                                                                                \
    asm volatile("" ::: "memory");                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    if (!(ABL & 4)) { RDX(4 * (P) + 1); }                                                                                     \
    asm volatile("" ::: "memory");                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    if (!(ABL & 2) && hb_ + 1 < nh) ISSUE(3, (4 * (P) + 7) & 7);                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    if (!(ABL & 4)) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(XT * NS) : "memory");                                      \
    LBAR();                                                                                                \
    MMA(0, 0, yf0);                                                                                       \
    LBAR();                                                                                                \
    /* phase 2: (X0, Y1) */                                                                               \
    if (!(ABL & 4)) { RDY(yf1, 4 * (P) + 2); }                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    if (!(ABL & 2) && hb_ + 2 < nh) ISSUE(0, (4 * (P) + 8) & 7);                                                        \
    LBAR();                                                                                                \
    MMA(0, 1, yf1);                                                                                       \
    LBAR();                                                                                                \
    /* phase 3: (X1, Y1) */                                                                               \
    if (!(ABL & 4)) { RDX(4 * (P) + 3); }                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    if (!(ABL & 2) && hb_ + 3 < nh) ISSUE(1, (4 * (P) + 9) & 7);                                                        \
    LBAR();                                                                                                \
    MMA(1, 1, yf1);                                                                                       \
    LBAR();                                                                                                \
    /* phase 4: (X1, Y0) */                                                                               \
    if (!(ABL & 2) && hb_ + 4 < nh) ISSUE(2, (4 * (P) + 10) & 7);                                                       \
    if ((t) + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                    \
    else if ((t) + 2 == nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                              \
    LBAR();                                                                                                \
    MMA(1, 0, yf0);                                                                                       \
    LBAR();                                                                                                \
  } while (0)

  if (ABL & 4) { RDY(yf0, 0); RDX(1); RDY(yf1, 2); }
  int t = 0;
#pragma unroll 1
  for (; t + 1 < nk; t += 2) {
    KTILE(0, t);
    KTILE(1, t + 1);
  }
  if (t < nk) KTILE(0, t);
  if (STAG && wr == 0) BAR();

  // ---- epilogue: lane owns 4 consecutive x of one y --------------------------------------------------------------------------
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < XT; ++i)
#pragma unroll
        for (int j = 0; j < YT; ++j) {
          const int xb = x0 + wr * 128 + a * 64 + i * TR, yb = y0 + wc * 64 + b * 32 + j * TR;
          if constexpr (SHAPE == 0) {
            const int y = yb + (lane & 15), x = xb + 4 * (lane >> 4);
            half4_t o = {(half_t)acc[a][b][i][j][0], (half_t)acc[a][b][i][j][1], (half_t)acc[a][b][i][j][2], (half_t)acc[a][b][i][j][3]};
            *reinterpret_cast<half4_t*>(out + (size_t)y * ldo + x) = o;
          } else {
            const int y = yb + (lane & 31);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int x = xb + 8 * q + 4 * (lane >> 5);
              half4_t o = {(half_t)acc[a][b][i][j][4 * q], (half_t)acc[a][b][i][j][4 * q + 1], (half_t)acc[a][b][i][j][4 * q + 2],
                           (half_t)acc[a][b][i][j][4 * q + 3]};
              *reinterpret_cast<half4_t*>(out + (size_t)y * ldo + x) = o;
            }
          }
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
__global__ void fill_kernel(half_t* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned h = (unsigned)(i * 2654435761u) ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    p[i] = (half_t)(((float)(h & 0xffff) / 32768.0f - 1.0f));
  }
}

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

struct Variant {
  const char* name;
  void (*run)(const half_t*, const half_t*, half_t*, int, int, int, hipStream_t);
};

template <int TILE>
void run_base(const half_t* X, const half_t* Y, half_t* out, int NX, int NY, int K, hipStream_t s);
template <int SHAPE, bool STAG, bool PRIO, int ABL = 0>
void run_g8(const half_t* X, const half_t* Y, half_t* out, int NX, int NY, int K, hipStream_t s) {
  if (NX % 256) { run_base<67>(X, Y, out, NX, NY, K, s); return; }
  const int nxt = NX / 256, nyt = NY / 256;
  hipLaunchKernelGGL((g8_kernel<SHAPE, STAG, PRIO, ABL>), dim3(nxt * nyt), dim3(512), 0, s, X, Y, out, NX, NY, K, NX, nxt);
}
template <int TILE>
void run_base(const half_t* X, const half_t* Y, half_t* out, int NX, int NY, int K, hipStream_t s) {
  mvoc_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.a = Y; d.w = X; d.out = out; d.m = NY; d.n = NX; d.k = K; d.n_store = NX; d.ldo = NX; d.lda = K; d.c1 = K; d.cin = K;
  d.a_mode = 0; d.tile = TILE; d.split_k = 1;
  if (mvoc_gemm_f16(&d, s) != 0) {
    fprintf(stderr, "baseline tile %d: %s\n", TILE, mvoc_last_error());
  }
}

int main(int argc, char** argv) {
  struct Shape { int NY, NX, K; };
  std::vector<Shape> shapes = {{81920, 1280, 11520}, {327680, 1280, 1280}, {81920, 1280, 2560}, {20480, 1280, 11520},
                               {327680, 320, 2880}, {81920, 640, 5760}, {81920, 2560, 640}, {327680, 960, 320}, {65536, 320, 2880},
                               {16384, 640, 5760}};
  if (argc >= 4) shapes = {{atoi(argv[1]), atoi(argv[2]), atoi(argv[3])}};
  std::vector<Variant> vars = {
      {"base t67 256x256", run_base<67>},
      {"base t66 320x256", run_base<66>},
      {"base auto", run_base<0>},
      {"lib t81 g8 256", run_base<81>},
      {"lib t82 g8 320", run_base<82>},
      {"lab g8 16x16", run_g8<0, true, true>},
      {"lab g8 ABL no barriers", run_g8<0, true, true, 1>},
      {"lab g8 ABL no DMA", run_g8<0, true, true, 2>},
      {"lab g8 ABL no reads", run_g8<0, true, true, 4>},
      {"lab g8 ABL no DMA no reads", run_g8<0, true, true, 6>},
      {"lab g8 ABL MFMA only", run_g8<0, true, true, 7>},
  };
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (auto& sh : shapes) {
    const size_t nX = (size_t)sh.NX * sh.K, nY = (size_t)sh.NY * sh.K, nO = (size_t)sh.NY * sh.NX;
    half_t *X, *Y, *O, *R;
    CK(hipMalloc(&X, nX * 2)); CK(hipMalloc(&Y, nY * 2)); CK(hipMalloc(&O, nO * 2)); CK(hipMalloc(&R, nO * 2));
    fill_kernel<<<2048, 256, 0, st>>>(X, nX, 0x1234u);
    fill_kernel<<<2048, 256, 0, st>>>(Y, nY, 0xabcdu);
    CK(hipMemsetAsync(R, 0, nO * 2, st));
    vars[0].run(X, Y, R, sh.NX, sh.NY, sh.K, st);
    CK(hipStreamSynchronize(st));
    printf("== NY(m)=%d NX(n)=%d K=%d  (%.1f GFLOP)\n", sh.NY, sh.NX, sh.K, 2.0 * sh.NY * sh.NX * sh.K * 1e-9);
    std::vector<half_t> hr(nO), ho(nO);
    CK(hipMemcpy(hr.data(), R, nO * 2, hipMemcpyDeviceToHost));
    const int rounds = 7, reps = 3;
    std::vector<std::vector<float>> tms(vars.size());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t v = 0; v < vars.size(); ++v) {  // correctness
      CK(hipMemsetAsync(O, 0xff, nO * 2, st));
      vars[v].run(X, Y, O, sh.NX, sh.NY, sh.K, st);
      CK(hipStreamSynchronize(st));
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) { printf("   %-24s launch error %s\n", vars[v].name, hipGetErrorString(e)); continue; }
      CK(hipMemcpy(ho.data(), O, nO * 2, hipMemcpyDeviceToHost));
      double mx = 0, ref = 0;
      size_t bad = 0;
      for (size_t i = 0; i < nO; i += 7) {
        const double d = fabs((double)(float)ho[i] - (double)(float)hr[i]);
        if (!(d <= 1e30)) ++bad;
        mx = std::max(mx, d);
        ref = std::max(ref, fabs((double)(float)hr[i]));
      }
      printf("   %-24s max|diff| vs base %.4g (max|ref| %.4g, nan %zu)\n", vars[v].name, mx, ref, bad);
    }
    for (int r = 0; r < rounds; ++r)
      for (size_t v = 0; v < vars.size(); ++v) {
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < reps; ++k) vars[v].run(X, Y, O, sh.NX, sh.NY, sh.K, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) tms[v].push_back(ms / reps);
      }
    for (size_t v = 0; v < vars.size(); ++v) {
      std::sort(tms[v].begin(), tms[v].end());
      const double med = tms[v][tms[v].size() / 2], mn = tms[v][0];
      const double fl = 2.0 * sh.NY * sh.NX * sh.K;
      printf("   %-24s median %8.1f us %7.0f TF/s   best %8.1f us %7.0f TF/s\n", vars[v].name, med * 1e3, fl / med * 1e-9, mn * 1e3,
             fl / mn * 1e-9);
    }
    CK(hipFree(X)); CK(hipFree(Y)); CK(hipFree(O)); CK(hipFree(R));
  }
  return 0;
}
