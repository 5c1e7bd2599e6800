// Activation-stationary linear for the finest level (K = 320, also 64 / 128 for the toy networks): out = epi(x @ W^T).
//
// The K = 320 projections of the two finest levels (to_out / proj_in / proj_out 320 -> 320, fused QKV 320 -> 960, GEGLU ff1
// 320 -> 2560: F.linear at pnp_utils.py:191, 206, 438, 505, 604-612, 692, 335) have only 5-10 K steps per output tile in the
// tiled GEMM: its per-tile prologue / epilogue and the operand staging (both operands through LDS, ~35 GB/s per CU) held them at
// 280-520 TFLOP/s.  Here a wave owns 32 CONSECUTIVE rows for the whole kernel and keeps them in registers as MFMA operand
// fragments (K/16 x 4 VGPRs = 80); only the weights stream, pre-packed on the host in fragment order (every LDS-DMA piece 1 KB
// contiguous in memory, every fragment read 1 KB contiguous in LDS: no swizzle, no bank conflict), in stages of 32 output
// channels through a 3-stage ring.  One wave per SIMD, two blocks per CU (79 KB of LDS each): the blocks are not coupled, one
// block's HBM prologue / epilogues overlap the other's MFMAs.  Per staged byte a block does 128 flop and the activation operand
// is never staged at all.
//   constants     : the 32 per-channel fp32 constants of a tile (bias, or beta @ W^T + bias with LayerNorm folded) are one more
//                   1 KB piece of the tile's packed stream -- they arrive with the weights
//   normalize != 0: LayerNorm folded -- the row's mean / rstd come from the fragments (one shuffle), the fragments are normalised
//                   in place once, gamma rides on the weights and beta on the constants
//   act GEGLU     : weight rows packed in (value, gate) blocks of 32 (as for mvoc_gemm_f16): stage pairs -> 32 output channels
//   residual      : the wave's 32 x 32 residual tile is LDS-DMA'd one stage ahead into a wave-private double buffer
//   epilogue      : + constant, activation, fp16 rounding points of the reference's eager chain, residual add, 16-byte stores
//                   (v_permlane32_swap pairs the 8-byte channel quads of the accumulator layout)
// No ordinary vector-memory load is left in the loop: every s_waitcnt vmcnt there is written by hand with an exact count.
#include <stdlib.h>

#include "common.h"

namespace {

struct XsArgs {
  const half_t* x;
  const half_t* wp;     // [N/32 tiles][NK + 1][64 lanes][8]: NK fragment pieces + one piece whose first 128 B are the tile's 32 fp32 constants
  const half_t* resid;
  half_t* out;
  long M;
  int N, n_store, ldo, ldr, act, normalize;
  float eps;
  unsigned long long* stamps;  // diagnostic builds only
  int lab;                     // diagnostic builds only: 1 = skip the output stores, 2 = store row-contiguous garbage instead
  long set_rows;               // > 0: a weight set per set_rows consecutive rows (mvoc_groupnorm_fold_xs_f16), set_bytes apart
  long set_bytes;
};

#ifdef MVOC_PP_LAB
#define XS_STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define XS_ACC(slot, a, b) (acc_t[slot] += (b) - (a))
#else
#define XS_STAMP(t) do { } while (0)
#define XS_ACC(slot, a, b) do { } while (0)
#endif

template <int N_>
__device__ __forceinline__ void xs_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
// s_waitcnt takes an immediate: dispatch a wave-uniform count (a smaller count than the true one only waits longer)
__device__ __forceinline__ void xs_wait_n(int n) {
  switch (n) {
    case 0: xs_wait<0>(); break;   case 1: xs_wait<1>(); break;   case 2: xs_wait<2>(); break;   case 3: xs_wait<3>(); break;
    case 4: xs_wait<4>(); break;   case 5: xs_wait<5>(); break;   case 6: xs_wait<6>(); break;   case 7: xs_wait<7>(); break;
    case 8: xs_wait<8>(); break;   case 9: xs_wait<9>(); break;   case 10: xs_wait<10>(); break; case 11: xs_wait<11>(); break;
    case 12: xs_wait<12>(); break; case 13: xs_wait<13>(); break; case 14: xs_wait<14>(); break; case 15: xs_wait<15>(); break;
    case 16: xs_wait<16>(); break; case 17: xs_wait<17>(); break; case 18: xs_wait<18>(); break; case 19: xs_wait<19>(); break;
    case 20: xs_wait<20>(); break; case 21: xs_wait<21>(); break; default: xs_wait<22>(); break;
  }
}

// K / 16 ; row groups (of 32 rows) per wave ; activation (a template parameter: tested per element at run time it left two
// scalar branches per output element in the epilogue, which then took 60 % of a plain projection's time)
template <int NK, int RG, int ACT>
__global__ __launch_bounds__(256, 2) void xslin_kernel(const XsArgs p) {
  constexpr int NS = 3, NW = 4;
  constexpr int NP = NK + 1;            // pieces (1 KB) per stage: NK weight fragments + the constants
  constexpr int STAGE = NP * 1024;
  constexpr int K = NK * 16;
  constexpr int RBUF = 4096;            // per-wave residual tiles: RG 1: two parities x 2 KB; RG 2: one buffer of 4 KB
  // ring | per-wave residual tiles: everything the loop reads arrives by LDS-DMA, so no compiler-managed vector-memory load sits
  // beside the in-flight DMA (hipcc drains vmcnt to 0 for those) and every wait below is an exact count
  __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE + NW * RBUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  long ms[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    const long m = (long)blockIdx.x * (NW * 32 * RG) + wave * (32 * RG) + 32 * g + r;
    ms[g] = m < p.M ? m : p.M - 1;
  }

  // (per-sample weight sets: the block's rows lie inside one set -- set_rows is a multiple of the rows per block, host check)
  const char* wp_set = reinterpret_cast<const char*>(p.wp) +
                       (p.set_rows > 0 ? ((long)blockIdx.x * (NW * 32 * RG) / p.set_rows) * p.set_bytes : 0);
  const int pw = (NP - wave + NW - 1) / NW;  // LDS-DMA pieces of this wave per stage
  const int T = p.N / 32;
  auto issue = [&](int st) {
    const char* src = wp_set + (size_t)st * STAGE + lane * 16;
    char* dst = smem + (st % NS) * STAGE;
#pragma unroll
    for (int i = 0; i < (NP + NW - 1) / NW; ++i) {
      const int j = wave + NW * i;
      if (j < NP)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
    }
  };
  char* resbuf = smem + NS * STAGE + wave * RBUF;
  auto res_slot = [&](int tile, int g, int j) -> char* {
    return resbuf + (RG == 1 ? (tile & 1) * 2048 : g * 2048) + j * 1024;
  };
  auto issue_resid = [&](int tile) {  // this lane's 16-byte pieces of the residual tile -> its own slots of the wave's buffer
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int n = 32 * tile + 16 * j + 8 * h;
        n = n < p.n_store ? n : p.n_store - 8;  // always issued (the counts below assume it); a clamped piece is never used
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.resid + ms[g] * p.ldr + n),
                                         (__attribute__((address_space(3))) void*)res_slot(tile, g, j), 16, 0, 0);
      }
  };
  // vmcnt retires in issue order and counts loads AND stores: a wait names how many vector-memory instructions issued AFTER the
  // awaited one may still be in flight.  (The first version left the epilogue stores out of the count: every stage then waited
  // for the acknowledgement of stores issued a moment before and for part of the next stage's DMA -- a ring one stage deep, 2.9 us
  // per stage on a plain 320 -> 2560 projection.)  Issue order -- prologue: residual(0), DMA(0), DMA(1);
  //   RG 1, stage s: DMA(s + 2) [pw], residual(s + 1) [R] (other parity), MFMAs, stores(s) [S]
  //   RG 2, stage s: DMA(s + 2) [pw], MFMAs, stores(s) [S], residual(s + 1) [R] (the one buffer, after the epilogue has read it)
  // GEGLU stores after odd stages only and has no residual.
  const int S = 2 * RG, R = p.resid ? 2 * RG : 0;
  constexpr bool GEGLU = ACT == MVOC_ACT_GEGLU;
  constexpr bool geglu = GEGLU;
  constexpr int PW_LO = NP / NW, PW_HI = (NP + NW - 1) / NW;  // pieces per stage of the waves without / with a remainder piece
  auto wait_stage = [&](int st) {  // DMA(st) has landed
    if (st >= 2 && st + 1 < T) {
      // steady state: the count is one of two compile-time constants per kernel variant (no dispatch tree on the stage's path)
      if (geglu) {
        if (pw == PW_HI) xs_wait<PW_HI + 2 * RG>(); else xs_wait<PW_LO + 2 * RG>();
      } else if (p.resid) {
        if (pw == PW_HI) xs_wait<PW_HI + 8 * RG>(); else xs_wait<PW_LO + 8 * RG>();
      } else {
        if (pw == PW_HI) xs_wait<PW_HI + 4 * RG>(); else xs_wait<PW_LO + 4 * RG>();
      }
      return;
    }
    const int dma_next = st + 1 < T ? pw : 0;
    int n;
    if (st == 0) n = dma_next;
    else if (geglu) n = dma_next + (st >= 2 ? S : 0);
    else n = dma_next + (st >= 2 ? 2 * (R + S) : R + S);
    xs_wait_n(n);
  };
  auto wait_resid = [&](int st) {  // residual(st) has landed
    if (RG == 1) xs_wait_n((st >= 1 ? S : 0) + (st + 2 < T ? pw : 0) + (st + 1 < T ? R : 0));  // newer: stores(st-1), DMA(st+2), residual(st+1)
    else xs_wait_n(st >= 1 && st + 2 < T ? pw : 0);                                             // newer: DMA(st+2)
  };

  half8_t xf[RG][NK];
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    const half_t* xr = p.x + ms[g] * K + 8 * h;
#pragma unroll
    for (int s = 0; s < NK; ++s) xf[g][s] = *reinterpret_cast<const half8_t*>(xr + 16 * s);
  }
  if (p.resid) issue_resid(0);
  for (int st = 0; st < NS - 1 && st < T; ++st) issue(st);
  if (p.normalize) {
    const half2_t one2 = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      // two passes over the register-resident row, like F.layer_norm: the mean first, then the moments of x - c with c the mean
      // rounded to fp16 (packed fp16 subtraction: exact up to half an ulp of the small difference; fdot2 accumulates in fp32).
      // The one-pass E[x^2] - mu^2 form loses the variance of rows whose mean is large against their spread; converting the
      // row to fp32 for a textbook second pass made hipcc spill 350 registers in the 64-row form.
      float s1 = 0.f;
#pragma unroll
      for (int s = 0; s < NK; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const half2_t v2 = {xf[g][s][2 * e], xf[g][s][2 * e + 1]};
          s1 = __builtin_amdgcn_fdot2(v2, one2, s1, false);
        }
      s1 += __shfl_xor(s1, 32);
      const float mu = s1 / (float)K;
      const half_t c16 = (half_t)mu;
      const half2_t c2 = {c16, c16};
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int s = 0; s < NK; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const half2_t v2 = {xf[g][s][2 * e], xf[g][s][2 * e + 1]};
          const half2_t d2 = v2 - c2;
          t1 = __builtin_amdgcn_fdot2(d2, one2, t1, false);
          t2 = __builtin_amdgcn_fdot2(d2, d2, t2, false);
        }
      t1 += __shfl_xor(t1, 32);
      t2 += __shfl_xor(t2, 32);
      const float dm = t1 / (float)K;  // mean of x - c: at most half an fp16 ulp of the mean
      const float rs = rsqrtf(fmaxf(t2 / (float)K - dm * dm, 0.f) + p.eps);
#pragma unroll
      for (int s = 0; s < NK; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[g][s][e] = (half_t)(((float)xf[g][s][e] - mu) * rs);
    }
  }

  int stage = 0;
  constexpr int PD = NK < 8 ? NK : (RG == 1 ? 8 : (GEGLU ? 2 : 4));
  const char* cslot = smem;  // ring slot of the stage just multiplied (its constants piece)
  // one stage: acc[g][channel][row] = (32 weight rows) x (row group g of this wave); a weight fragment feeds RG MFMAs
#ifdef MVOC_PP_LAB
  unsigned long long acc_t[6] = {0, 0, 0, 0, 0, 0}, ta = 0, tb = 0, tc = 0, td = 0, te = 0, tstart = 0;
  XS_STAMP(tstart);
  te = tstart;
#endif
  auto run_stage = [&](f32x16 (&acc)[RG]) {
    XS_STAMP(ta);
    XS_ACC(3, te, ta);  // epilogue of the previous stage (or prologue)
    wait_stage(stage);
    XS_STAMP(tb);
    XS_ACC(0, ta, tb);  // DMA wait
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();   // stage landed for every wave; stage-1's readers are done: its slot takes stage+2
    __builtin_amdgcn_sched_barrier(0);
    XS_STAMP(tc);
    XS_ACC(1, tb, tc);  // barrier
    if (stage + NS - 1 < T) issue(stage + NS - 1);
    if (RG == 1 && p.resid && stage + 1 < T) issue_resid(stage + 1);
    cslot = smem + (stage % NS) * STAGE;
    const char* wl = cslot + lane * 16;
    half8_t wf[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) wf[i] = *reinterpret_cast<const half8_t*>(wl + i * 1024);
    __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
    for (int s = 0; s < NK; ++s) {
#pragma unroll
      for (int g = 0; g < RG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % PD], xf[g][s], acc[g], 0, 0, 0);
      if (s + PD < NK) wf[s % PD] = *reinterpret_cast<const half8_t*>(wl + (s + PD) * 1024);
      __builtin_amdgcn_sched_group_barrier(0x008, RG, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#ifdef MVOC_PP_LAB
    asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[RG - 1][15]));  // the stamp below follows the last MFMA's result
#endif
    XS_STAMP(te);
    XS_ACC(2, tc, te);  // issue + fragment reads + MFMAs
    ++stage;
  };
  // channel-quad vector of the tile's constants for accumulator rows 8q + 4h + {0..3} (read before the slot is recycled: the
  // next stage's barrier comes after)
  auto cvec4 = [&](const char* slot, int c) -> f32x4 { return *reinterpret_cast<const f32x4*>(slot + NK * 1024 + c * 4); };
  // 32 output channels of one row group's row, as 4 quads of fp16 pairs -> 16-byte stores (+ residual)
  auto store_tile = [&](unsigned (&pk)[4][2], int nout, int tile, int g) {
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
      uint4 v = {pk[2 * j][0], pk[2 * j][1], pk[2 * j + 1][0], pk[2 * j + 1][1]};
      if (p.resid) {
        half8_t a = __builtin_bit_cast(half8_t, v);
        const half8_t b = *reinterpret_cast<const half8_t*>(res_slot(tile, g, j) + lane * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (half_t)((float)a[e] + (float)b[e]);
        v = __builtin_bit_cast(uint4, a);
      }
      if (n >= p.n_store) continue;  // only in the last tile (n_store > N - 32)
      // a lane past the last row carries row M - 1's data (clamped loads) and rewrites that row with identical values: every
      // wave issues the same number of stores, which the vmcnt accounting relies on
#ifdef MVOC_PP_LAB
      if (p.lab & 1) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); continue; }
      if (p.lab & 2) {  // same bytes, one 1 KB run per store instruction
        *reinterpret_cast<uint4*>(p.out + ((long)blockIdx.x * 1024 + (wave * 2 * RG + 2 * g + j)) * 512 + lane * 8 + (long)(nout / 32) * 0) = v;
        continue;
      }
#endif
      *reinterpret_cast<uint4*>(p.out + ms[g] * p.ldo + n) = v;
    }
  };

  if constexpr (GEGLU) {
    for (int t = 0; t + 1 < T; t += 2) {  // (value tile, gate tile) -> 32 output channels
      f32x16 acc[RG];
#pragma unroll
      for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
      run_stage(acc);
      // value half, rounded to fp16 as the reference's chunk does (exact in 16 bits: kept packed while the gate tile runs)
      half2_t hv[RG][8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 cvl = cvec4(cslot, 8 * q + 4 * h);  // the value tile's slot is recycled during the gate stage
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          hv[g][2 * q] = half2_t{(half_t)(acc[g][4 * q] + cvl[0]), (half_t)(acc[g][4 * q + 1] + cvl[1])};
          hv[g][2 * q + 1] = half2_t{(half_t)(acc[g][4 * q + 2] + cvl[2]), (half_t)(acc[g][4 * q + 3] + cvl[3])};
        }
      }
#pragma unroll
      for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
      run_stage(acc);
#pragma unroll
      for (int g = 0; g < RG; ++g) {
        unsigned pk[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 cgt = cvec4(cslot, 8 * q + 4 * h);
          half_t o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float gv = r16(acc[g][4 * q + e] + cgt[e]);
            o[e] = (half_t)((float)hv[g][2 * q + (e >> 1)][e & 1] * r16(gelu_fast_f(gv)));
          }
          const half2_t lo = {o[0], o[1]}, hi = {o[2], o[3]};
          pk[q][0] = __builtin_bit_cast(unsigned, lo);
          pk[q][1] = __builtin_bit_cast(unsigned, hi);
        }
        store_tile(pk, 16 * t, t + 1, g);
      }
    }
  } else {
    for (int t = 0; t < T; ++t) {
      f32x16 acc[RG];
#pragma unroll
      for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
      run_stage(acc);
      if (p.resid) wait_resid(t);
#pragma unroll
      for (int g = 0; g < RG; ++g) {
        unsigned pk[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 cv = cvec4(cslot, 8 * q + 4 * h);
          half_t o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = r16(acc[g][4 * q + e] + cv[e]);
            if constexpr (ACT == MVOC_ACT_SILU) v = r16(silu_f(v));
            else if constexpr (ACT == MVOC_ACT_GELU) v = r16(gelu_fast_f(v));
            o[e] = (half_t)v;
          }
          const half2_t lo = {o[0], o[1]}, hi = {o[2], o[3]};
          pk[q][0] = __builtin_bit_cast(unsigned, lo);
          pk[q][1] = __builtin_bit_cast(unsigned, hi);
        }
        store_tile(pk, 32 * t, t, g);
      }
      if (RG == 2 && p.resid && t + 1 < T) {
        // the one residual buffer is refilled for the next tile once this tile's reads are done (their values were consumed by
        // the stores above)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_resid(t + 1);
      }
    }
  }
#ifdef MVOC_PP_LAB
  XS_STAMP(ta);
  XS_ACC(3, te, ta);
  if (p.stamps && blockIdx.x == 0 && wave == 0 && lane == 0) {
    for (int i = 0; i < 4; ++i) p.stamps[i] = acc_t[i];
    p.stamps[4] = ta - tstart;
    p.stamps[5] = (unsigned long long)T;
  }
#endif
}

}  // namespace

extern "C" int mvoc_xs_linear_f16(const mvoc_xs_desc* d, void* stream) {
  MVOC_REQUIRE(d && d->x && d->wp && d->out, -1, "xs_linear: null operand");
  MVOC_REQUIRE(d->m > 0 && d->n > 0 && d->n % 32 == 0, -1, "xs_linear: m > 0, n %% 32 == 0");
  MVOC_REQUIRE(d->act >= MVOC_ACT_NONE && d->act <= MVOC_ACT_GELU, -1, "xs_linear: bad act %d", d->act);
  MVOC_REQUIRE(!(d->resid && d->act == MVOC_ACT_GEGLU), -2, "xs_linear: no residual with GEGLU");
  MVOC_REQUIRE(d->k == 64 || d->k == 128 || d->k == 320, -2, "xs_linear: k (%d) must be 64, 128 or 320", d->k);
  MVOC_REQUIRE(d->act != MVOC_ACT_GEGLU || d->n % 64 == 0, -2, "xs_linear: GEGLU needs n %% 64 == 0");
  const int ns = d->n_store > 0 ? d->n_store : (d->act == MVOC_ACT_GEGLU ? d->n / 2 : d->n);
  MVOC_REQUIRE(ns > (d->act == MVOC_ACT_GEGLU ? d->n / 2 : d->n) - 32, -2, "xs_linear: n_store may only trim the last 32-channel tile");
  MVOC_REQUIRE(ns % 8 == 0 && d->ldo % 8 == 0 && ((uintptr_t)d->out & 15) == 0 &&
                   (d->resid == nullptr || (d->ldr % 8 == 0 && ((uintptr_t)d->resid & 15) == 0)),
               -2, "xs_linear: outputs / residual must be 16-byte addressable per 8 channels");
  XsArgs a;
  a.x = (const half_t*)d->x; a.wp = (const half_t*)d->wp;
  a.resid = (const half_t*)d->resid; a.out = (half_t*)d->out;
  a.M = d->m; a.N = d->n; a.n_store = ns; a.ldo = d->ldo; a.ldr = d->ldr; a.act = d->act; a.normalize = d->normalize; a.eps = d->ln_eps;
  // two row groups per wave (each weight fragment read from LDS feeds two MFMAs, each staged byte 256 flop) once there are
  // enough rows to fill the chip with 256-row blocks; MVOC_XS_RG=1|2 forces a form (experiments)
  static const int force_rg = getenv("MVOC_XS_RG") ? atoi(getenv("MVOC_XS_RG")) : 0;
  const int rg = force_rg ? force_rg : (d->m >= 131072 && d->k == 320 ? 2 : 1);
  a.stamps = nullptr;
  a.lab = 0;
  a.set_rows = d->wp_set_rows;
  a.set_bytes = (long)(d->n / 32) * (d->k / 16 + 1) * 1024;
  MVOC_REQUIRE(d->wp_set_rows >= 0 && (d->wp_set_rows == 0 || (d->wp_set_rows % (128 * rg) == 0 && d->m % d->wp_set_rows == 0)), -2,
               "xs_linear: wp_set_rows (%ld) must be a multiple of the %d rows of a block and divide m", (long)d->wp_set_rows, 128 * rg);
#ifdef MVOC_PP_LAB
  if (const char* e = getenv("MVOC_XS_STAMPS")) a.stamps = (unsigned long long*)strtoull(e, nullptr, 10);
  a.lab = getenv("MVOC_XS_LAB") ? atoi(getenv("MVOC_XS_LAB")) : 0;
#endif
  const long nblk = (d->m + 128 * rg - 1) / (128 * rg);
  MVOC_REQUIRE(nblk < 0x7fffffffL, -2, "xs_linear: grid too large");
  hipStream_t s = (hipStream_t)stream;
  MvocProfScope prof(MVOC_FAM_GEMM, s, 2.0 * (double)d->m * d->n * d->k);
  const dim3 grid((unsigned)nblk), blk(256);
#define XS_LAUNCH(NK_, RG_)                                                                                        \
  do {                                                                                                             \
    if (d->act == MVOC_ACT_GEGLU) hipLaunchKernelGGL((xslin_kernel<NK_, RG_, MVOC_ACT_GEGLU>), grid, blk, 0, s, a);  \
    else if (d->act == MVOC_ACT_SILU) hipLaunchKernelGGL((xslin_kernel<NK_, RG_, MVOC_ACT_SILU>), grid, blk, 0, s, a); \
    else if (d->act == MVOC_ACT_GELU) hipLaunchKernelGGL((xslin_kernel<NK_, RG_, MVOC_ACT_GELU>), grid, blk, 0, s, a); \
    else hipLaunchKernelGGL((xslin_kernel<NK_, RG_, MVOC_ACT_NONE>), grid, blk, 0, s, a);                          \
  } while (0)
  if (d->k == 320 && rg == 2) XS_LAUNCH(20, 2);
  else if (d->k == 320) XS_LAUNCH(20, 1);
  else if (d->k == 128 && rg == 2) XS_LAUNCH(8, 2);
  else if (d->k == 128) XS_LAUNCH(8, 1);
  else if (rg == 2) XS_LAUNCH(4, 2);
  else XS_LAUNCH(4, 1);
#undef XS_LAUNCH
  return mvoc_check_launch("xslin_kernel");
}
