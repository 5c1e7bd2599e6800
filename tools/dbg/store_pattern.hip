// Store-pattern microbenchmark: M x N fp16 output written by waves that own 32 rows and walk N in steps of 32 channels
// (the activation-stationary kernels' epilogue), with SEG contiguous bytes per row per store instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int SEG>  // bytes per row per instruction: 32 (cur), 64, 128, 1024 (= one row)
__global__ __launch_bounds__(256) void k(uint4* out, long M, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row0 = (long)blockIdx.x * 128 + wave * 32;
  const int lpr = SEG / 16;            // lanes per row
  const int rpi = 64 / lpr;            // rows per instruction
  const uint4 v = {1u, 2u, 3u, (unsigned)lane};
  // each "stage pair" produces 32 rows x 128 B; emit it as 4096 / (64*16) = 4 instructions
  for (int n0 = 0; n0 < N * 2; n0 += 128) {            // byte offset of the 128-B column block
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int r, cb;
      if (SEG <= 128) {
        const int seg_per_row = 128 / SEG;
        const int idx = i * rpi + lane / lpr;          // (row, seg) index in [0, 32*seg_per_row)
        r = idx % 32; cb = (idx / 32) * SEG + (lane % lpr) * 16;
        if (SEG == 128) { r = i * 8 + lane / 8; cb = (lane % 8) * 16; }
        else if (SEG == 64) { r = (i & 1) * 16 + lane / 4; cb = (i >> 1) * 64 + (lane % 4) * 16; }
        else { r = lane & 31; cb = i * 32 + (lane >> 5) * 16; }
      } else { r = 0; cb = 0; }
      out[((row0 + r) * (long)N * 2 + n0 + cb) / 16] = v;
    }
  }
}
int main() {
  const long M = 327680; const int N = 2560;
  uint4* d; hipMalloc(&d, M * N * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(M / 128), dim3(256), 0, 0, d, M, N);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(M / 128), dim3(256), 0, 0, d, M, N);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.1f us  %6.2f TB/s\n", name, ms / 5 * 1e3, (double)M * N * 2 / (ms / 5 * 1e-3) / 1e12);
  };
  run(k<32>, "32 rows x 32 B per instr");
  run(k<64>, "16 rows x 64 B per instr");
  run(k<128>, "8 rows x 128 B per instr");
  return 0;
}
