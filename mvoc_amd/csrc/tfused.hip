// Fused temporal self-attention front half for the finest level (C <= 320): LayerNorm -> QKV projection -> attention over
// the frame axis in ONE kernel; Q, K, V never exist in HBM (reference: TransformerTemporalModel's two self-attentions,
// i2vgen-xl/pnp_utils.py:170-220, 222-346 [norm1/norm2 -> attn], 720-887 [to_q/to_k/to_v + SDPA over frames]; the
// unfused form is three kernels -- fused-QKV GEMM [rows, 3C] written and re-read, mvoc_temporal_attn_f16 -- whose SDPA
// core has an arithmetic intensity of 8 flop/B and whose K = 320 GEMM is staging-bound).
//
// Why this shape maps to the hardware: temporal attention is point-wise in (h, w), a sequence is the F <= 32 frames of one
// pixel, and a 32-row MFMA tile holds 32/F whole sequences.  A wave therefore owns 32 rows = 32/F pixels x F frames for the
// whole kernel and keeps their RAW activations in registers as MFMA operand fragments (C/16 x 4 VGPRs = 80 at C = 320);
// only the (gamma-scaled) weights stream, pre-packed on the host in fragment order so that every LDS-DMA piece is 1 KB
// contiguous in memory and every fragment read is 1 KB contiguous in LDS (no swizzle, no bank conflict).  Per staged weight
// byte the block does 256 flop (the activation operand is never staged), against 128-142 for the best GEMM tile.
//   per head:  q, k tiles  D[ch][row]  = W'  x^T   (weights as the row operand)        4 stages of 32 weight rows (q0 k0 q1 k1)
//              S^T = K Q^T  from the fp16-packed accumulators (both carry the same channel permutation)
//              block-diagonal softmax (a row attends only to the rows of its own pixel), P^T packed
//              v tiles     D[row][d]   = x  W'^T   (activations as the row operand)    2 stages
//              O^T = V^T P^T, scaled by 1/rowsum, stored as 8-byte channel quads
//   LayerNorm: the row's mean / rstd come from the register fragments (one shuffle) and the fragments are normalised in place,
//   once; gamma is folded into the weights and beta into a per-channel constant added before the fp16 rounding that the
//   reference's q / k / v tensors have.
#include <stdlib.h>

#include "common.h"

namespace {

struct TfArgs {
  const half_t* x;
  const half_t* wp;      // [heads*6 tiles][NK][64 lanes][8] fragment-major
  const float* ln_s;     // [3C] row sums of the gamma-scaled weights
  const float* ln_c;     // [3C] beta @ W^T
  half_t* out;
  int nsample, frames, logf, hw, c, heads;
  float eps, scale_log2;
  unsigned long long* stamps;  // diagnostic builds only
};

// (round 6: s_setprio(1) around the 20 MFMAs of a stage -- the wave against its SIMD partner, a wave of the CU's OTHER block in its
// vector-only epilogue -- moved nothing: 292 -> 294-297 us at batch 5, profiles/r6/tfused_setprio_ab.txt; not kept)
// (round 6, second experiment: the two blocks a CU holds start together and, doing identical work, stay in lockstep -- their activation
// load + LayerNorm prologues coincide instead of hiding under each other's MFMAs.  Holding half of the FIRST round's blocks back by 12 /
// 25 us (ids 256..511, every second id, every second group of 8) moved nothing or lost 4 %: 285 us -> 285-298 us at batch 5,
// profiles/r6/tfused_stagger_ab.txt; not kept)
template <int N>
__device__ __forceinline__ void tf_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifdef MVOC_PP_LAB
#define TF_STAMP(t)                                                                        \
  do {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");               \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  } while (0)
#else
#define TF_STAMP(t) do { } while (0)
#endif

// NW = 8: one block per CU, its two wave groups ping-pong (below).  NW = 4: one wave per SIMD and TWO blocks per CU (3-stage
// ring, 64 KB of LDS each): the blocks are not coupled by barriers, so one block's prologue (its rows' activations come
// straight from HBM: ~20 % of a block's life) and epilogues overlap the other block's MFMAs.
template <int NK, int NW>
__global__ __launch_bounds__(NW * 64) void tfused_kernel(const TfArgs p) {
  constexpr int NS = NW == 8 ? 4 : 3;   // weight-stage ring (one stage = 32 weight rows x C = NK KB)
  constexpr bool PP = NW == 8;
  constexpr int STAGE = NK * 1024;
  __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE + 3 * NK * 16 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int F = p.frames, C = p.c;
#ifdef MVOC_PP_LAB
  unsigned long long tb = 0;
  TF_STAMP(tb);
#endif
  const int ppw = 32 >> p.logf;         // pixels per wave
  // LayerNorm-fold vectors in LDS (an ordinary global load next to in-flight LDS-DMA makes hipcc drain vmcnt(0))
  float* lnc = reinterpret_cast<float*>(smem + NS * STAGE);                 // [3C] beta @ W^T (+ bias)
  for (int i = tid; i < 3 * NK * 16; i += NW * 64) lnc[i] = p.ln_c[i];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // written before the first (raw) barrier below releases any reader

  // ---- this lane's row: tile row r = (pixel r / F, frame r % F) ---------------------------------------------------------
  const int px = (int)blockIdx.x * (NW * ppw) + wave * ppw + (r >> p.logf);
  const int fr = r & (F - 1);
  const bool live = px < p.hw;
  const long grow = ((long)blockIdx.y * F + fr) * p.hw + (live ? px : 0);

  // ---- weight stages: piece j (1 KB) of a stage is issued by wave j % 8 ---------------------------------------------------
  const int pw = (NK - wave + NW - 1) / NW;   // pieces of this wave per stage (wave-uniform)
  const int T = p.heads * 6;
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
  auto wait_stage = [&](int st, int issued_beyond) {  // this wave's pieces of stage st have landed (later stages may stay in flight)
    const int ahead = T - 1 - st < issued_beyond ? T - 1 - st : issued_beyond;  // stages issued beyond st at this point
    const int n = ahead * pw;
    if (n >= 10) tf_wait<10>(); else if (n == 6) tf_wait<6>(); else if (n == 5) tf_wait<5>(); else if (n == 4) tf_wait<4>();
    else if (n == 3) tf_wait<3>(); else if (n == 2) tf_wait<2>(); else if (n == 1) tf_wait<1>(); else tf_wait<0>();
  };

  // ---- raw activations of the row -> registers (operand fragments), row statistics ----------------------------------------
  half8_t xf[NK];
  {
    const half_t* xr = p.x + grow * C + 8 * h;
#pragma unroll
    for (int s = 0; s < NK; ++s) xf[s] = *reinterpret_cast<const half8_t*>(xr + 16 * s);
  }
  for (int st = 0; st < NS - 1 && st < T; ++st) issue(st);
  float s1 = 0.f, s2 = 0.f;
  {
    const half2_t one2 = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
    for (int s = 0; s < NK; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const half2_t v2 = {xf[s][2 * e], xf[s][2 * e + 1]};
        s1 = __builtin_amdgcn_fdot2(v2, one2, s1, false);
      }
  }
  s1 += __shfl_xor(s1, 32);
  const float mu = s1 / (float)C;
  // second pass over the register-resident row: the moments of x - c with c the mean rounded to fp16, in PACKED fp16 (the difference of
  // two nearby fp16 values is exact up to half an ulp of the small difference; v_dot2 accumulates in fp32) -- as xslin.hip's LayerNorm
  // fold; E[x^2] - mu^2 cancels for rows whose mean is large against their spread, and the fp32 form of this pass (convert, subtract,
  // fma per element: 480 vector instructions per wave, 640 more for the normalisation below) was a third of the prologue that the
  // CU's other block has to hide (round 6)
  float q2 = 0.f;
  {
    const half_t c16 = (half_t)mu;
    const half2_t c2 = {c16, c16};
#pragma unroll
    for (int s = 0; s < NK; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const half2_t v2 = {xf[s][2 * e], xf[s][2 * e + 1]};
        const half2_t d2 = v2 - c2;
        q2 = __builtin_amdgcn_fdot2(d2, d2, q2, false);
      }
    q2 += __shfl_xor(q2, 32);
    const float dm = mu - (float)c16;  // mean of x - c (at most half an fp16 ulp of the mean): known without a pass of its own
    s2 = fmaxf(q2 / (float)C - dm * dm, 0.f);
  }
  const float rs = rsqrtf(s2 + p.eps);
  // normalise the row in place, once: x^ = (x - mean) * rstd as ONE fused multiply-add per element with fp16 input and output
  // (v_fma_mixlo / mixhi_f16: x * rstd - mean * rstd in fp32, rounded once).  gamma rides on the weights (W' = W * gamma) and beta in the
  // per-channel constant c = beta @ W^T, so a projection tile's fix-up is ONE add per value -- no row-sum correction, no
  // per-row statistics in the epilogues (the earlier rstd * (acc - mean * rowsum) + c form cost ~1000 cycles per stage beside
  // the partner's MFMAs)
  const float nb = -mu * rs;
#pragma unroll
  for (int s = 0; s < NK; ++s) {
    // (written out: hipcc turns the C form into convert / packed fp32 fma / convert, twice the instructions)
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    u32x4_ w = __builtin_bit_cast(u32x4_, xf[s]);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      unsigned o;
      asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
          : "=&v"(o) : "v"(w[d]), "v"(rs), "v"(nb));
      w[d] = o;
    }
    xf[s] = __builtin_bit_cast(half8_t, w);
  }

  // The two waves of a SIMD (wave w and w + 4) run ONE BARRIER APART: while one multiplies a stage (20 MFMAs on registers and
  // LDS fragments), its partner does the previous stage's LayerNorm fix-up / softmax / stores (VALU, LDS, VMEM) -- matrix pipe
  // beside vector pipe by construction.  Two barriers per stage:
  //     bar_M ; issue stage+3 ; MFMAs(stage) ; wait(stage+1 landed) ; bar_E ; epilogue(stage)
  // group B (waves 4-7) executes one extra barrier up front, group A one at the end.  A stage's pieces are waited for (by every
  // wave, counted vmcnt) before the barrier that precedes any wave's MFMAs on it; a ring slot is refilled two barriers after
  // its last reader finished.
  int stage = 0;
  const int grp = PP ? wave >> 2 : 0;
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
  unsigned long long s_bm = 0, s_is = 0, s_mf = 0, s_wt = 0, s_be = 0;
  (void)t0; (void)t1; (void)t2; (void)t3; (void)t4; (void)t5; (void)s_bm; (void)s_is; (void)s_mf; (void)s_wt; (void)s_be;
  wait_stage(0, NS - 2);
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();
#ifdef MVOC_PP_LAB
  unsigned long long tpro = 0;
  TF_STAMP(tpro);
  tpro -= tb;
#endif
  // fragment reads run PD ahead of the MFMA that consumes them; the first PD of a stage are issued in the PREVIOUS stage's
  // epilogue phase (the stage is known to have landed there), so an MFMA phase starts on registers
  constexpr int PD = NK < 8 ? NK : 8;
  half8_t wf[PD];
  auto prefetch = [&](int st) {
    const char* wl = smem + (st % NS) * STAGE + lane * 16;
#pragma unroll
    for (int i = 0; i < PD; ++i) wf[i] = *reinterpret_cast<const half8_t*>(wl + i * 1024);
  };
  if constexpr (PP) prefetch(0);
  // one weight stage: acc += (32 weight rows) x (this wave's 32 activation rows); w_rows: weights are the row operand.
  //     bar_M ; MFMAs(stage) ; wait(stage+1 landed) ; bar_E ; first reads of stage+1 ; issue stage+3   | caller: epilogue(stage)
  // vmcnt retires in issue order and counts stores too: the wait for DMA(stage) must allow the output stores issued after it (the
  // previous head's: 2 after each value tile) to stay in flight, or the first stages of a head wait for their acknowledgement and
  // for part of the next stage's DMA.  `extra` = those stores (position in the head: v1 -> 2, next q0 -> 4, next k0 -> 2); a wave
  // whose lanes are all past the last pixel issues none and keeps the plain count.
  const bool wave_stores = __builtin_amdgcn_readfirstlane((int)__any(live));
  auto run_stage = [&](f32x16& acc, bool w_rows, int extra) {
    if constexpr (!PP) {
      // one wave per SIMD, one barrier per stage: stage landed for every wave (and stage-1's readers are done) -> refill the
      // slot of stage-1 with stage+2, multiply, then the caller's epilogue; the CU's other block fills the gaps
      if constexpr (NK % NW == 0) {
        constexpr int PWC = NK / NW;
        if (stage + 1 < T && extra && wave_stores) {
          if (extra == 2) tf_wait<PWC + 2>(); else tf_wait<PWC + 4>();
        } else {
          wait_stage(stage, NS - 2);
        }
      } else {
        wait_stage(stage, NS - 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (stage + NS - 1 < T) issue(stage + NS - 1);
      const char* wl = smem + (stage % NS) * STAGE + lane * 16;
#pragma unroll
      for (int i = 0; i < PD; ++i) wf[i] = *reinterpret_cast<const half8_t*>(wl + i * 1024);
      __builtin_amdgcn_sched_group_barrier(0x100, PD, 0);
#pragma unroll
      for (int s = 0; s < NK; ++s) {
        acc = w_rows ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % PD], xf[s], acc, 0, 0, 0)
                     : __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[s], wf[s % PD], acc, 0, 0, 0);
        if (s + PD < NK) wf[s % PD] = *reinterpret_cast<const half8_t*>(wl + (s + PD) * 1024);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      ++stage;
      return;
    }
    TF_STAMP(t3);
    // stage+1 must have landed for EVERY wave before the barrier after which the partner group pre-reads it (its bar_E is this
    // group's bar_M): pieces are issued through stage+2 at this point
    if (stage + 1 < T) wait_stage(stage + 1, 1);
    TF_STAMP(t0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();               // bar_M
    __builtin_amdgcn_sched_barrier(0);
    TF_STAMP(t1);
#ifdef MVOC_PP_LAB
    s_wt += t0 - t3;
#endif
    const char* wl = smem + (stage % NS) * STAGE + lane * 16;
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      acc = w_rows ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % PD], xf[s], acc, 0, 0, 0)
                   : __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[s], wf[s % PD], acc, 0, 0, 0);
      if (s + PD < NK) wf[s % PD] = *reinterpret_cast<const half8_t*>(wl + (s + PD) * 1024);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
    }
    ++stage;
    TF_STAMP(t4);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();               // bar_E: the partner group starts its MFMAs
    __builtin_amdgcn_sched_barrier(0);
    TF_STAMP(t5);
    if (stage < T) prefetch(stage);             // (landed: waited for by every wave before the previous barrier)
    // the slot of the stage before the one just multiplied is free: the partner multiplied it one barrier interval ago
    if (stage + NS - 2 < T) issue(stage + NS - 2);
    TF_STAMP(t2);
#ifdef MVOC_PP_LAB
    s_bm += t1 - t0; s_mf += t4 - t1; s_be += t5 - t4; s_is += t2 - t5;
#endif
  };

  // a q or k tile: D[channel][row] = W' x^T, LayerNorm fix-up, fp16 -> the two k16 operand fragments of the 32 channels
  auto proj_qk = [&](int nbase, half8_t (&pk)[2], int extra) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    run_stage(acc, true, extra);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = nbase + 8 * q + 4 * h;
      const f32x4 cv = *reinterpret_cast<const f32x4*>(lnc + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) pk[q >> 1][4 * (q & 1) + e] = (half_t)(acc[4 * q + e] + cv[e]);
    }
  };

  for (int hd = 0; hd < p.heads; ++hd) {
    // ---- S^T = K Q^T over the head's 64 channels, 32 at a time (stage order q0 k0 q1 k1) -----------------------------------
    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      half8_t qp[2], kp[2];
      proj_qk(hd * 64 + 32 * i, qp, (i == 0 && hd > 0) ? 4 : 0);
      proj_qk(C + hd * 64 + 32 * i, kp, (i == 0 && hd > 0) ? 2 : 0);
#pragma unroll
      for (int s = 0; s < 2; ++s) st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kp[s], qp[s], st, 0, 0, 0);
    }
    // ---- block-diagonal softmax: a row attends only to the rows of its own pixel ------------------------------------------
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = 8 * (e >> 2) + 4 * h + (e & 3);
      const float sv = ((key >> p.logf) == (r >> p.logf)) ? st[e] * p.scale_log2 : -INFINITY;
      st[e] = sv;
      mx = fmaxf(mx, sv);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float ps = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pv = __builtin_amdgcn_exp2f(st[e] - mx);
      st[e] = pv;
      ps += pv;
    }
    ps += __shfl_xor(ps, 32);
    const float inv = 1.0f / ps;
    half8_t pp[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) pp[s][j] = (half_t)st[8 * s + j];
    // ---- v : D[row][d] = x W'^T, then O^T = V^T P^T, 32 value channels at a time ---------------------------------------------
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      run_stage(acc, false, dt == 1 ? 2 : 0);
      const int n = 2 * C + hd * 64 + dt * 32 + r;  // this lane's value channel
      const float cv = lnc[n];
      half8_t vp[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) vp[e >> 3][e & 7] = (half_t)(acc[e] + cv);
      f32x16 ot;
#pragma unroll
      for (int e = 0; e < 16; ++e) ot[e] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) ot = __builtin_amdgcn_mfma_f32_32x32x16_f16(vp[s], pp[s], ot, 0, 0, 0);
      // In the accumulator layout a lane owns 4 consecutive channels (8 B) of its row per quad, the row's other 4 sit in lane
      // +-32: v_permlane32_swap pairs quads (0,1) and (2,3) so that each lane ends up with 8 consecutive channels -> two 16-byte
      // stores per tile instead of four 8-byte ones (the 8-byte row-strided stores were issue-bound: half the per-head time)
      unsigned pk[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
          const half2_t v2 = {(half_t)(ot[q * 4 + 2 * w2] * inv), (half_t)(ot[q * 4 + 2 * w2 + 1] * inv)};
          pk[q][w2] = __builtin_bit_cast(unsigned, v2);
        }
#pragma unroll
      for (int qq = 0; qq < 4; qq += 2)
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
          const auto sw = __builtin_amdgcn_permlane32_swap(pk[qq][w2], pk[qq + 1][w2], false, false);
          pk[qq][w2] = sw[0];       // lanes < 32: quad qq of this lane     | lanes >= 32: quad qq+1 of lane - 32
          pk[qq + 1][w2] = sw[1];   // lanes < 32: quad qq of lane + 32     | lanes >= 32: quad qq+1 of this lane
        }
      if (live) {
        // lane (r, h) now holds channels 16 j + 8 h + {0..7} of row r for j = 0, 1
        half_t* op = p.out + grow * C + hd * 64 + 32 * dt + 8 * h;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint4 v = {pk[2 * j][0], pk[2 * j][1], pk[2 * j + 1][0], pk[2 * j + 1][1]};
          *reinterpret_cast<uint4*>(op + 16 * j) = v;
        }
      }
    }
  }
  if (PP && grp == 0) __builtin_amdgcn_s_barrier();  // group A's balancing barrier
#ifdef MVOC_PP_LAB
  TF_STAMP(t0);
  if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && (wave == 0 || wave == 4) && lane == 0) {
    unsigned long long* o = p.stamps + (wave >> 2) * 8;
    o[0] = s_bm; o[1] = s_is; o[2] = s_mf; o[3] = s_wt; o[4] = s_be; o[5] = (t0 - tb) - (s_bm + s_is + s_mf + s_wt + s_be) - tpro; o[6] = tpro; o[7] = t0 - tb;
  }
#endif
}

}  // namespace

extern "C" int mvoc_temporal_qkv_attn_f16(const mvoc_tfused_desc* d, void* stream) {
  MVOC_REQUIRE(d && d->x && d->wp && d->ln_rowsum && d->ln_bias && d->out, -1, "temporal_qkv_attn: null operand");
  MVOC_REQUIRE(d->nsample > 0 && d->hw > 0 && d->heads > 0 && d->c == d->heads * 64, -1, "temporal_qkv_attn: c must be heads * 64");
  MVOC_REQUIRE(d->frames == 8 || d->frames == 16 || d->frames == 32, -2, "temporal_qkv_attn: frames (%d) must be 8, 16 or 32", d->frames);
  MVOC_REQUIRE(d->c == 64 || d->c == 128 || d->c == 320, -2, "temporal_qkv_attn: c (%d) must be 64, 128 or 320", d->c);
  MVOC_REQUIRE(d->nsample <= 65535, -2, "temporal_qkv_attn: grid too large");
  TfArgs a;
  a.x = (const half_t*)d->x; a.wp = (const half_t*)d->wp; a.ln_s = (const float*)d->ln_rowsum; a.ln_c = (const float*)d->ln_bias;
  a.out = (half_t*)d->out;
  a.nsample = d->nsample; a.frames = d->frames; a.hw = d->hw; a.c = d->c; a.heads = d->heads;
  a.logf = d->frames == 8 ? 3 : d->frames == 16 ? 4 : 5;
  a.eps = d->ln_eps;
  a.scale_log2 = 0.125f * 1.4426950408889634f;
  a.stamps = nullptr;
#ifdef MVOC_PP_LAB
  if (const char* e = getenv("MVOC_TF_STAMPS")) a.stamps = (unsigned long long*)strtoull(e, nullptr, 10);
#endif
  static const int mode = getenv("MVOC_TFUSED_WAVES") ? atoi(getenv("MVOC_TFUSED_WAVES")) : 4;  // 8: one ping-pong block per CU
  const int nw = mode == 8 ? 8 : 4;
  const int ppb = nw * (32 / d->frames);  // pixels per block
  dim3 grid((unsigned)((d->hw + ppb - 1) / ppb), (unsigned)d->nsample);
  hipStream_t s = (hipStream_t)stream;
  const double rows = (double)d->nsample * d->frames * d->hw;
  MvocProfScope prof(MVOC_FAM_TFUSED, s, 2.0 * rows * 3.0 * d->c * d->c + 4.0 * rows * d->frames * d->c);
  if (nw == 8) {
    if (d->c == 320) hipLaunchKernelGGL((tfused_kernel<20, 8>), grid, dim3(512), 0, s, a);
    else if (d->c == 128) hipLaunchKernelGGL((tfused_kernel<8, 8>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((tfused_kernel<4, 8>), grid, dim3(512), 0, s, a);
  } else {
    if (d->c == 320) hipLaunchKernelGGL((tfused_kernel<20, 4>), grid, dim3(256), 0, s, a);
    else if (d->c == 128) hipLaunchKernelGGL((tfused_kernel<8, 4>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((tfused_kernel<4, 4>), grid, dim3(256), 0, s, a);
  }
  return mvoc_check_launch("tfused_kernel");
}
