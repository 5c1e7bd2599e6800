// Shared device/host helpers for the gfx950 kernels of libmvoc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mvoc_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MVOC_WAVE 64

// ---- error plumbing (host) ------------------------------------------------------------------
void mvoc_set_error(const char* fmt, ...);
int mvoc_check_launch(const char* what);  // returns 0 or -3

#define MVOC_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      mvoc_set_error(__VA_ARGS__);     \
      return (code);                   \
    }                                  \
  } while (0)

// ---- profiling brackets (host) --------------------------------------------------------------
struct MvocProfScope {
  int fam;
  hipStream_t stream;
  void* rec;
  MvocProfScope(int fam, hipStream_t s, double work);
  ~MvocProfScope();
};

// ---- device helpers ----------------------------------------------------------------------------
// v_exp + v_rcp (1 ulp) instead of an IEEE division (~10 VALU ops): GroupNorm+SiLU apply is VALU-co-limited otherwise
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// erf-GELU for GEMM epilogues: Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, far below the fp16 output ulp)
// with one v_exp + one v_rcp instead of libm erff's ~40-instruction polynomial ladder
__device__ __forceinline__ float gelu_fast_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// round-to-nearest-even fp32 -> fp16 -> fp32 (one "eager op" rounding of the reference's fp16 chain)
__device__ __forceinline__ float r16(float x) { return (float)(half_t)x; }

// XCD-aware bijective remap of a 1-D block id: blocks b and b+8 share an XCD (round-robin dispatch), so
// give each XCD a contiguous range of logical tiles (speed only; any placement is correct).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, x = bid & 7u, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
