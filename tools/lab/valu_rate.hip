// Issue rate of vector instructions on gfx950 (ticks of s_memtime per instruction and wave), one block of W waves per CU:
// v_exp_f32, v_fma_f32, v_pk_fma_f32, v_cvt_pk_f16_f32, v_max3_f32, v_rcp_f32 -- with 1, 2 and 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void rate(float* out, unsigned long long* ticks, float seed) {
  float a[8];
  f2 p[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = f2{a[i], a[i] + 1}; }
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
        if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
        if (OP == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(a[i]));
        if (OP == 4) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
        if (OP == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 6) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
        if (OP == 7) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
      }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int OP>
void run(const char* name, float* out, unsigned long long* tk) {
  for (int waves = 4; waves <= 12; waves += 4) {
    hipLaunchKernelGGL(rate<OP>, dim3(256), dim3(waves * 64), 0, 0, out, tk, 0.5f);
    CK(hipDeviceSynchronize());
    unsigned long long h;
    CK(hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost));
    printf("%-18s %d wave(s)/SIMD: %6.2f ticks per instruction and wave\n", name, waves / 4, (double)h / (256.0 * 32));
  }
}

int main() {
  float* out; unsigned long long* tk;
  CK(hipMalloc(&out, 256 * 768 * 4)); CK(hipMalloc(&tk, 8));
  run<0>("v_exp_f32", out, tk); run<7>("v_exp_f16", out, tk); run<5>("v_rcp_f32", out, tk); run<1>("v_fma_f32", out, tk);
  run<2>("v_pk_fma_f32", out, tk); run<6>("v_pk_add_f32", out, tk); run<3>("v_cvt_pk_f16_f32", out, tk); run<4>("v_max3_f32", out, tk);
  return 0;
}
