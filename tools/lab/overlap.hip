// Do vector instructions of one wave overlap matrix instructions of another wave of the same SIMD on gfx950?  Wall time (HIP
// events) of 1024 blocks (4 per CU in turn... one resident at a time is not needed: every mode runs the same number of blocks) of
// W waves: roles per wave are M (32x32x16 MFMA chain on 4 independent accumulators), V (v_fma_f32 on 8 independent registers),
// E (v_exp_f32), X (one wave interleaving 1 MFMA : K vector instructions).  Waves w, w+4, w+8 ... share SIMD (w & 3).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct Roles { char r[16]; int nw; int iters; int k; };

template <int K>
__global__ __launch_bounds__(1024) void mix(float* out, Roles ro, float seed) {
  const int wave = threadIdx.x >> 6;
  const char role = ro.r[wave];
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  f16v acc[4];
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  h8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(seed + e); y[e] = (_Float16)(seed - e); }
  if (role == 'M') {
#pragma unroll 1
    for (int it = 0; it < ro.iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[j], 0, 0, 0);  // 16 MFMAs = 512 cycles
    }
  } else if (role == 'V') {
#pragma unroll 1
    for (int it = 0; it < ro.iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));  // 128 instructions
    }
  } else if (role == 'E') {
#pragma unroll 1
    for (int it = 0; it < ro.iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
    }
  } else if (role == 'X') {  // one wave: 16 MFMAs with K vector instructions after each
#pragma unroll 1
    for (int it = 0; it < ro.iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < K; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
        }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K>
static double run(const char* roles, int iters, int k, float* out) {
  Roles ro; memset(&ro, 0, sizeof ro);
  ro.nw = (int)strlen(roles); memcpy(ro.r, roles, ro.nw); ro.iters = iters; ro.k = k;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(mix<K>, dim3(256), dim3(ro.nw * 64), 0, 0, out, ro, 0.5f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3;
}

int main() {
  float* out; CK(hipMalloc(&out, 256 * 1024 * 4));
  const int IT = 2000;
  const char* modes[] = {"MMMM", "VVVV", "EEEE", "MMMMVVVV", "MMMMEEEE", "VVVVVVVV", "VVVVVVVVVVVV", "VVVVVVVVVVVVVVVV", "MMMMMMMM",
                         "MMMMVVVVVVVV", "MMMMMMMMVVVV", "MMMMVVVVEEEE", "EEEEEEEE", "EEEEEEEEEEEE"};
  for (const char* m : modes) {
    double us = run<0>(m, IT, 0, out);
    printf("%-18s %9.1f us   (per iteration: %6.1f ns; M = 16 MFMA 32x32x16 [512 matrix cycles], V / E = 128 instructions)\n", m, us, us * 1e3 / IT);
  }
#define XR(k) { double us = run<k>("XXXX", IT, k, out); \
    printf("XXXX k=%d           %9.1f us   (per iteration: %6.1f ns; 16 MFMA + %d v_fma in one wave)\n", k, us, us * 1e3 / IT, 16 * k); \
    us = run<k>("XXXXXXXX", IT, k, out); \
    printf("XXXXXXXX k=%d       %9.1f us   (per iteration: %6.1f ns; two such waves per SIMD)\n", k, us, us * 1e3 / IT); \
    us = run<k>("XXXXXXXXXXXX", IT, k, out); \
    printf("XXXXXXXXXXXX k=%d   %9.1f us   (per iteration: %6.1f ns; three such waves per SIMD)\n", k, us, us * 1e3 / IT); }
  XR(0) XR(2) XR(4) XR(6) XR(8)
  return 0;
}
