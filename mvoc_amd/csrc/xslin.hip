// Activation-stationary linear for the finest level (K = 320, also 64 / 128 for the toy networks): out = epi(x @ W^T).
//
// The K = 320 projections of the two finest levels (to_out / proj_in / proj_out 320 -> 320, fused QKV 320 -> 960, GEGLU ff1
// 320 -> 2560: F.linear at pnp_utils.py:191, 206, 438, 505, 604-612, 692, 335) have only 5-10 K steps per output tile in the
// tiled GEMM: its per-tile prologue / epilogue and the operand staging (both operands through LDS, ~35 GB/s per CU) held them at
// 280-520 TFLOP/s.  Here a wave owns 32 CONSECUTIVE rows for the whole kernel and keeps them in registers as MFMA operand
// fragments (K/16 x 4 VGPRs = 80); only the weights stream, pre-packed on the host in fragment order (every LDS-DMA piece 1 KB
// contiguous in memory, every fragment read 1 KB contiguous in LDS: no swizzle, no bank conflict), in stages of 32 output
// channels through a 3-stage ring.  One wave per SIMD, two blocks per CU (64 KB of LDS each): the blocks are not coupled, one
// block's HBM prologue / epilogues overlap the other's MFMAs.  Per staged byte a block does 128 flop and the activation operand
// is never staged at all.
//   normalize != 0: LayerNorm folded -- the row's mean / rstd come from the fragments (one shuffle), the fragments are normalised
//                   in place once, gamma rides on the weights and beta on the per-channel constant (fp32 `cvec`)
//   act GEGLU     : weight rows packed in (value, gate) blocks of 32 (as for mvoc_gemm_f16): stage pairs -> 32 output channels
//   epilogue      : + cvec / bias, activation, fp16 rounding points of the reference's eager chain, residual add, 16-byte stores
//                   (v_permlane32_swap pairs the 8-byte channel quads of the accumulator layout)
#include <stdlib.h>

#include "common.h"

namespace {

struct XsArgs {
  const half_t* x;
  const half_t* wp;     // [N/32 tiles][NK][64 lanes][8]
  const half_t* bias;   // fp16 [N] or NULL
  const float* cvec;    // fp32 [N] (LayerNorm-folded constant) or NULL
  const half_t* resid;
  half_t* out;
  long M;
  int N, n_store, ldo, ldr, act, normalize;
  float eps;
};

template <int N_>
__device__ __forceinline__ void xs_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

template <int NK>
__global__ __launch_bounds__(256) void xslin_kernel(const XsArgs p) {
  constexpr int NS = 3, NW = 4;
  constexpr int STAGE = NK * 1024;
  constexpr int K = NK * 16;
  constexpr int NMAX = 2560;            // per-channel constants of the whole projection live in LDS (fp32)
  __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE + NMAX * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const long m = (long)blockIdx.x * (NW * 32) + wave * 32 + r;
  const bool live = m < p.M;
  const long ms = live ? m : p.M - 1;

  const int pw = (NK - wave + NW - 1) / NW;  // LDS-DMA pieces of this wave per stage
  const int T = p.N / 32;
  auto issue = [&](int st) {
    const char* src = reinterpret_cast<const char*>(p.wp) + (size_t)st * STAGE + lane * 16;
    char* dst = smem + (st % NS) * STAGE;
#pragma unroll
    for (int i = 0; i < (NK + NW - 1) / NW; ++i) {
      const int j = wave + NW * i;
      if (j < NK)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
    }
  };
  auto wait_stage = [&](int st) {  // this wave's pieces of stage st have landed; one later stage may stay in flight
    const int n = (T - 1 - st < 1 ? T - 1 - st : 1) * pw;
    if (n >= 5) xs_wait<5>(); else if (n == 4) xs_wait<4>(); else if (n == 3) xs_wait<3>(); else if (n == 2) xs_wait<2>();
    else if (n == 1) xs_wait<1>(); else xs_wait<0>();
  };

  // bias / folded LayerNorm constant -> LDS (an ordinary global load in a tile epilogue would make hipcc drain the in-flight
  // LDS-DMA with vmcnt(0))
  float* cv_lds = reinterpret_cast<float*>(smem + NS * STAGE);
  for (int i = tid; i < p.N; i += NW * 64) cv_lds[i] = p.cvec ? p.cvec[i] : (p.bias ? (float)p.bias[i] : 0.f);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // written before the first (raw) barrier below releases any reader

  half8_t xf[NK];
  {
    const half_t* xr = p.x + ms * K + 8 * h;
#pragma unroll
    for (int s = 0; s < NK; ++s) xf[s] = *reinterpret_cast<const half8_t*>(xr + 16 * s);
  }
  for (int st = 0; st < NS - 1 && st < T; ++st) issue(st);
  if (p.normalize) {
    float s1 = 0.f, s2 = 0.f;
    const half2_t one2 = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
    for (int s = 0; s < NK; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const half2_t v2 = {xf[s][2 * e], xf[s][2 * e + 1]};
        s1 = __builtin_amdgcn_fdot2(v2, one2, s1, false);
        s2 = __builtin_amdgcn_fdot2(v2, v2, s2, false);
      }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    const float mu = s1 / (float)K;
    const float rs = rsqrtf(fmaxf(s2 / (float)K - mu * mu, 0.f) + p.eps);
#pragma unroll
    for (int s = 0; s < NK; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) xf[s][e] = (half_t)(((float)xf[s][e] - mu) * rs);
  }

  int stage = 0;
  constexpr int PD = NK < 8 ? NK : 8;
  uint4 rv[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};  // residual of the tile being multiplied (requested at the stage's start)
  // one stage: acc[channel][row] = (32 weight rows) x (this wave's 32 rows)
  auto run_stage = [&](f32x16& acc) {
    wait_stage(stage);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();   // stage landed for every wave; stage-1's readers are done: its slot takes stage+2
    __builtin_amdgcn_sched_barrier(0);
    if (stage + NS - 1 < T) issue(stage + NS - 1);
    if (p.resid && live) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = 32 * stage + 16 * j + 8 * h;
        if (n < p.n_store) rv[j] = *reinterpret_cast<const uint4*>(p.resid + m * p.ldr + n);
      }
    }
    const char* wl = smem + (stage % NS) * STAGE + lane * 16;
    half8_t wf[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) wf[i] = *reinterpret_cast<const half8_t*>(wl + i * 1024);
    __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % PD], xf[s], acc, 0, 0, 0);
      if (s + PD < NK) wf[s % PD] = *reinterpret_cast<const half8_t*>(wl + (s + PD) * 1024);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    ++stage;
  };
  // channel-quad vector of the per-channel constant for accumulator rows 8q + 4h + {0..3} of tile n0
  auto cvec4 = [&](int n) -> f32x4 { return *reinterpret_cast<const f32x4*>(cv_lds + n); };
  // 32 output channels of this lane's row, as 4 quads of fp16 pairs -> 16-byte stores (+ residual)
  auto store_tile = [&](unsigned (&pk)[4][2], int nout) {
#pragma unroll
    for (int qq = 0; qq < 4; qq += 2)
#pragma unroll
      for (int w2 = 0; w2 < 2; ++w2) {
        const auto sw = __builtin_amdgcn_permlane32_swap(pk[qq][w2], pk[qq + 1][w2], false, false);
        pk[qq][w2] = sw[0];
        pk[qq + 1][w2] = sw[1];
      }
    // lane (r, h) holds channels nout + 16 j + 8 h + {0..7} of row r for j = 0, 1
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = nout + 16 * j + 8 * h;
      if (!live || n >= p.n_store) continue;
      uint4 v = {pk[2 * j][0], pk[2 * j][1], pk[2 * j + 1][0], pk[2 * j + 1][1]};
      if (p.resid) {
        half8_t a = __builtin_bit_cast(half8_t, v);
        const half8_t b = __builtin_bit_cast(half8_t, rv[j]);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (half_t)((float)a[e] + (float)b[e]);
        v = __builtin_bit_cast(uint4, a);
      }
      *reinterpret_cast<uint4*>(p.out + m * p.ldo + n) = v;
    }
  };

  if (p.act == MVOC_ACT_GEGLU) {
    for (int t = 0; t + 1 < T; t += 2) {  // (value tile, gate tile) -> 32 output channels
      f32x16 av, ag;
#pragma unroll
      for (int e = 0; e < 16; ++e) av[e] = ag[e] = 0.f;
      run_stage(av);
      run_stage(ag);
      unsigned pk[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 cvl = cvec4(32 * t + 8 * q + 4 * h), cgt = cvec4(32 * t + 32 + 8 * q + 4 * h);
        half_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float hv = r16(av[4 * q + e] + cvl[e]);
          const float gv = r16(ag[4 * q + e] + cgt[e]);
          o[e] = (half_t)(hv * r16(gelu_fast_f(gv)));
        }
        const half2_t lo = {o[0], o[1]}, hi = {o[2], o[3]};
        pk[q][0] = __builtin_bit_cast(unsigned, lo);
        pk[q][1] = __builtin_bit_cast(unsigned, hi);
      }
      store_tile(pk, 16 * t);
    }
  } else {
    for (int t = 0; t < T; ++t) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      run_stage(acc);
      unsigned pk[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 cv = cvec4(32 * t + 8 * q + 4 * h);
        half_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = r16(acc[4 * q + e] + cv[e]);
          if (p.act == MVOC_ACT_SILU) v = r16(silu_f(v));
          else if (p.act == MVOC_ACT_GELU) v = r16(gelu_fast_f(v));
          o[e] = (half_t)v;
        }
        const half2_t lo = {o[0], o[1]}, hi = {o[2], o[3]};
        pk[q][0] = __builtin_bit_cast(unsigned, lo);
        pk[q][1] = __builtin_bit_cast(unsigned, hi);
      }
      store_tile(pk, 32 * t);
    }
  }
}

}  // namespace

extern "C" int mvoc_xs_linear_f16(const mvoc_xs_desc* d, void* stream) {
  MVOC_REQUIRE(d && d->x && d->wp && d->out, -1, "xs_linear: null operand");
  MVOC_REQUIRE(d->m > 0 && d->n > 0 && d->n % 32 == 0 && d->n <= 2560, -1, "xs_linear: m > 0, n %% 32 == 0, n <= 2560");
  MVOC_REQUIRE(!(d->resid && d->act == MVOC_ACT_GEGLU), -2, "xs_linear: no residual with GEGLU");
  MVOC_REQUIRE(d->k == 64 || d->k == 128 || d->k == 320, -2, "xs_linear: k (%d) must be 64, 128 or 320", d->k);
  MVOC_REQUIRE(d->act != MVOC_ACT_GEGLU || d->n % 64 == 0, -2, "xs_linear: GEGLU needs n %% 64 == 0");
  const int ns = d->n_store > 0 ? d->n_store : (d->act == MVOC_ACT_GEGLU ? d->n / 2 : d->n);
  MVOC_REQUIRE(ns % 8 == 0 && d->ldo % 8 == 0 && ((uintptr_t)d->out & 15) == 0 &&
                   (d->resid == nullptr || (d->ldr % 8 == 0 && ((uintptr_t)d->resid & 15) == 0)),
               -2, "xs_linear: outputs / residual must be 16-byte addressable per 8 channels");
  MVOC_REQUIRE(!(d->normalize && !d->cvec), -1, "xs_linear: normalize needs the folded constant vector");
  XsArgs a;
  a.x = (const half_t*)d->x; a.wp = (const half_t*)d->wp; a.bias = (const half_t*)d->bias; a.cvec = (const float*)d->cvec;
  a.resid = (const half_t*)d->resid; a.out = (half_t*)d->out;
  a.M = d->m; a.N = d->n; a.n_store = ns; a.ldo = d->ldo; a.ldr = d->ldr; a.act = d->act; a.normalize = d->normalize; a.eps = d->ln_eps;
  const long nblk = (d->m + 127) / 128;
  MVOC_REQUIRE(nblk < 0x7fffffffL, -2, "xs_linear: grid too large");
  hipStream_t s = (hipStream_t)stream;
  MvocProfScope prof(MVOC_FAM_GEMM, s, 2.0 * (double)d->m * d->n * d->k);
  if (d->k == 320) hipLaunchKernelGGL(xslin_kernel<20>, dim3((unsigned)nblk), dim3(256), 0, s, a);
  else if (d->k == 128) hipLaunchKernelGGL(xslin_kernel<8>, dim3((unsigned)nblk), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(xslin_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, s, a);
  return mvoc_check_launch("xslin_kernel");
}
