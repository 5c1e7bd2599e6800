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
//   per head:  q, k tiles  D[ch][row]  = W'  x^T   (weights as the row operand)        4 stages of 32 weight rows
//              S^T = K Q^T  from the fp16-packed accumulators (both carry the same channel permutation)
//              block-diagonal softmax (a row attends only to the rows of its own pixel), P^T packed
//              v tiles     D[row][d]   = x  W'^T   (activations as the row operand)    2 stages
//              O^T = V^T P^T, scaled by 1/rowsum, stored as 8-byte channel quads
//   LayerNorm is folded: the MFMAs run on raw rows, the per-row mean / rstd come from the register fragments (one shuffle),
//   and the accumulators are fixed up as rstd * (acc - mean * rowsum(W')) + beta @ W^T before the fp16 rounding that the
//   reference's q / k / v tensors have.
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
};

template <int N>
__device__ __forceinline__ void tf_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NK>
__global__ __launch_bounds__(512) void tfused_kernel(const TfArgs p) {
  constexpr int NS = 4;                 // weight-stage ring (one stage = 32 weight rows x C = NK KB)
  constexpr int STAGE = NK * 1024;
  __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE + 8 * 256 + 2 * 3 * NK * 16 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int F = p.frames, C = p.c;
  const int ppw = 32 >> p.logf;         // pixels per wave
  float* stats = reinterpret_cast<float*>(smem + NS * STAGE + wave * 256);  // [32 rows][mean, rstd]
  // LayerNorm-fold vectors in LDS (an ordinary global load next to in-flight LDS-DMA makes hipcc drain vmcnt(0))
  float* lns = reinterpret_cast<float*>(smem + NS * STAGE + 8 * 256);       // [3C] row sums of W'
  float* lnc = lns + 3 * NK * 16;                                            // [3C] beta @ W^T
  for (int i = tid; i < 3 * NK * 16; i += 512) {
    lns[i] = p.ln_s[i];
    lnc[i] = p.ln_c[i];
  }

  // ---- this lane's row: tile row r = (pixel r / F, frame r % F) ---------------------------------------------------------
  const int px = (int)blockIdx.x * (8 * ppw) + wave * ppw + (r >> p.logf);
  const int fr = r & (F - 1);
  const bool live = px < p.hw;
  const long grow = ((long)blockIdx.y * F + fr) * p.hw + (live ? px : 0);

  // ---- weight stages: piece j (1 KB) of a stage is issued by wave j % 8 ---------------------------------------------------
  const int pw = (NK - wave + 7) / 8;   // pieces of this wave per stage (wave-uniform)
  const int T = p.heads * 6;
  auto issue = [&](int st) {
    const char* src = reinterpret_cast<const char*>(p.wp) + (size_t)st * STAGE + lane * 16;
    char* dst = smem + (st & (NS - 1)) * STAGE;
#pragma unroll
    for (int i = 0; i < (NK + 7) / 8; ++i) {
      const int j = wave + 8 * i;
      if (j < NK)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
    }
  };
  auto wait_stage = [&](int st) {  // this wave's pieces of stage st have landed (later stages may stay in flight)
    const int ahead = T - 1 - st < 2 ? T - 1 - st : 2;  // stages issued beyond st at this point
    const int n = ahead * pw;
    if (n >= 6) tf_wait<6>(); else if (n == 4) tf_wait<4>(); else if (n == 3) tf_wait<3>(); else if (n == 2) tf_wait<2>();
    else if (n == 1) tf_wait<1>(); else tf_wait<0>();
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
        s2 = __builtin_amdgcn_fdot2(v2, v2, s2, false);
      }
  }
  s1 += __shfl_xor(s1, 32);
  s2 += __shfl_xor(s2, 32);
  const float mu = s1 / (float)C;
  const float rs = rsqrtf(fmaxf(s2 / (float)C - mu * mu, 0.f) + p.eps);
  if (h == 0) {
    stats[2 * r] = mu;
    stats[2 * r + 1] = rs;
  }

  int stage = 0;
  bool waited = false;
  // one weight stage: acc += (32 weight rows) x (this wave's 32 activation rows); w_rows: weights are the row operand
  auto run_stage = [&](f32x16& acc, bool w_rows) {
    if (!waited) wait_stage(stage);
    waited = false;
    __builtin_amdgcn_s_barrier();               // stage landed for every wave; the slot of stage-1 is free
    if (stage + NS - 1 < T) issue(stage + NS - 1);
    const char* wl = smem + (stage & (NS - 1)) * STAGE + lane * 16;
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      const half8_t wf = *reinterpret_cast<const half8_t*>(wl + s * 1024);
      acc = w_rows ? __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf[s], acc, 0, 0, 0)
                   : __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[s], wf, acc, 0, 0, 0);
    }
    ++stage;
  };

  for (int hd = 0; hd < p.heads; ++hd) {
    // ---- q, k : D[channel][row] ------------------------------------------------------------------------------------------
    half8_t qp[2][2], kp[2][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      run_stage(acc, true);
      const int nbase = (t >> 1) * C + hd * 64 + (t & 1) * 32;  // q rows, then k rows of the [3C] projection
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = nbase + 8 * q + 4 * h;
        const f32x4 sv = *reinterpret_cast<const f32x4*>(lns + n);
        const f32x4 cv = *reinterpret_cast<const f32x4*>(lnc + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const half_t v = (half_t)(rs * (acc[4 * q + e] - mu * sv[e]) + cv[e]);
          if (t < 2) qp[t & 1][q >> 1][4 * (q & 1) + e] = v; else kp[t & 1][q >> 1][4 * (q & 1) + e] = v;
        }
      }
    }
    // ---- S^T = K Q^T, block-diagonal softmax ------------------------------------------------------------------------------
    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kp[i][s], qp[i][s], st, 0, 0, 0);
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
    // ---- v : D[row][d], then O^T = V^T P^T ----------------------------------------------------------------------------------
    f32x16 ot[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      run_stage(acc, false);
      const int n = 2 * C + hd * 64 + dt * 32 + r;  // this lane's value channel
      const float sv = lns[n], cv = lnc[n];
      half8_t vp[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = 8 * (e >> 2) + 4 * h + (e & 3);
        const float m_ = stats[2 * row], r_ = stats[2 * row + 1];
        vp[e >> 3][e & 7] = (half_t)(r_ * (acc[e] - m_ * sv) + cv);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) ot[dt][e] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vp[s], pp[s], ot[dt], 0, 0, 0);
    }
    // retire the next stage's pieces BEFORE the stores enter the in-order vmcnt queue behind them
    if (stage < T) { wait_stage(stage); waited = true; }
    if (live) {
      half_t* op = p.out + grow * C + hd * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[dt][q * 4 + e] * inv);
          *reinterpret_cast<half4_t*>(op + 32 * dt + 8 * q + 4 * h) = o;
        }
    }
  }
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
  const int ppb = 8 * (32 / d->frames);  // pixels per block
  dim3 grid((unsigned)((d->hw + ppb - 1) / ppb), (unsigned)d->nsample);
  hipStream_t s = (hipStream_t)stream;
  const double rows = (double)d->nsample * d->frames * d->hw;
  MvocProfScope prof(MVOC_FAM_TATTN, s, 2.0 * rows * 3.0 * d->c * d->c + 4.0 * rows * d->frames * d->c);
  if (d->c == 320) hipLaunchKernelGGL(tfused_kernel<20>, grid, dim3(512), 0, s, a);
  else if (d->c == 128) hipLaunchKernelGGL(tfused_kernel<8>, grid, dim3(512), 0, s, a);
  else hipLaunchKernelGGL(tfused_kernel<4>, grid, dim3(512), 0, s, a);
  return mvoc_check_launch("tfused_kernel");
}
