// Implicit-GEMM on MFMA for gfx950: every linear / 3x3 conv / 1x1 conv / (3,1,1) temporal conv of the
// I2VGen-XL UNet runs through this one kernel family.
//
//   out[m, n] = epilogue( sum_k A(m, k) * W[n, k] )       fp16 operands, fp32 accumulate (v_mfma_f32_32x32x16_f16)
//
// * A(m, k) is gathered on the fly from channels-last activations:
//     PLAIN      A(m,k) = a[m*lda + k]                                         (linear, 1x1 conv)
//     CONV3X3    k = tap*cin + c, tap = 3*ky+kx: pixel (oy*stride+ky-1, ox*stride+kx-1), zero outside;
//                optional nearest-upsample of the source folded into the index (Upsample2D + conv)
//     TEMPORAL3  k = tap*cin + c: same pixel of frame f+tap-1, zero outside [0, frames)
//   In all modes channels c < c1 come from `a`, the rest from `a2` (torch.cat([x, skip], dim=1) never
//   materialised).
// * MFMA orientation is "weights as the row operand": D[row = out-channel n][col = pixel m].  In the
//   32x32 accumulator layout (col = lane&31, row = 8*(reg>>2) + 4*(lane>>5) + (reg&3)) every lane then owns 4
//   CONSECUTIVE output channels of one pixel per register quad -> 8-byte vector stores / bias / residual loads
//   with no LDS transpose, and GEGLU's (value, gate) pairs sit in the same lane.
// * Block = WN x WM waves, each wave TN x TM tiles of 32x32; K step 32 (2 MFMA k-steps); global -> register ->
//   LDS staging, double-buffered, one barrier per K step; LDS rows padded to 80 B so the ds_read_b128 fragment
//   reads are bank-conflict free (stride 20 dwords: 5r mod 16 is a bijection over a 16-lane group).
// * 1-D grid with an XCD-aware remap: the n-tiles of one m-tile are neighbours on one XCD, so the activation
//   tile they share is served by that XCD's L2.
#include <stdlib.h>

#include "gemm_args.h"

namespace {

constexpr int BK = 32;
constexpr int ROWB = 80;  // LDS row pitch in bytes (64 B of data + 16 B pad)

struct RowInfo {   // per staged activation row (fixed for the whole K loop)
  int m;           // global row, -1 if beyond M
  int y0, x0;      // conv: oy*stride-1, ox*stride-1
  int img;         // conv: image index; temporal: frame index
};

template <int WN, int WM, int TN, int TM>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[TN][TM], int n0, int m0, int wn, int wm,
                                              int r, int h, const float* lnstat = nullptr) {
  // ---- epilogue: lane owns pixel m = tile_m + r and, per register quad q, channels n..n+3 ---------------
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + (wm * TM + j) * 32 + r;
    if (m >= p.M) continue;
    const half_t* ra = p.rowadd ? p.rowadd + (size_t)(m / p.rowadd_div) * p.ld_rowadd : nullptr;
    const half_t* rs = p.resid ? p.resid + (size_t)m * p.ldr : nullptr;
    half_t* orow = p.out + (size_t)m * p.ldo;
    // folded LayerNorm: the MFMAs multiplied the RAW rows by gamma-scaled weights; with mu/rstd of this row
    //   LN(x) @ W^T + b  =  rstd * (acc - mu * rowsum(W')) + (beta @ W^T + b)      (applied per quad below)
    float ln_mu = 0.f, ln_rs = 1.f;
    if (lnstat) {
      ln_mu = lnstat[2 * ((wm * TM + j) * 32 + r)];
      ln_rs = lnstat[2 * ((wm * TM + j) * 32 + r) + 1];
    }
    const bool use_bias = p.bias && !lnstat;
    if (p.act == MVOC_ACT_GEGLU) {
      if constexpr (TN % 2 == 0) {
#pragma unroll
        for (int i = 0; i < TN; i += 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nh = n0 + (wn * TN + i) * 32 + 8 * q + 4 * h;  // packed row of the value half
            if (nh >= p.N) continue;
            const int no = (n0 + (wn * TN + i) * 32) / 2 + 8 * q + 4 * h;
            half4_t bh = {0, 0, 0, 0}, bg = {0, 0, 0, 0};
            if (use_bias) {
              bh = *reinterpret_cast<const half4_t*>(p.bias + nh);
              bg = *reinterpret_cast<const half4_t*>(p.bias + nh + 32);
            }
            f32x4 sh = {0.f, 0.f, 0.f, 0.f}, sg = sh, ch = sh, cg = sh;
            if (lnstat) {
              sh = *reinterpret_cast<const f32x4*>(p.ln_s + nh);
              sg = *reinterpret_cast<const f32x4*>(p.ln_s + nh + 32);
              ch = *reinterpret_cast<const f32x4*>(p.ln_c + nh);
              cg = *reinterpret_cast<const f32x4*>(p.ln_c + nh + 32);
            }
            half4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float hv = r16(lnstat ? ln_rs * (acc[i][j][q * 4 + e] - ln_mu * sh[e]) + ch[e] : acc[i][j][q * 4 + e] + (float)bh[e]);
              const float gv = r16(lnstat ? ln_rs * (acc[i + 1][j][q * 4 + e] - ln_mu * sg[e]) + cg[e]
                                          : acc[i + 1][j][q * 4 + e] + (float)bg[e]);
              float v = r16(hv * r16(gelu_fast_f(gv)));
              if (rs) v = r16(v + (float)rs[no + e]);
              o[e] = (half_t)v;
            }
            *reinterpret_cast<half4_t*>(orow + no) = o;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < TN; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + (wn * TN + i) * 32 + 8 * q + 4 * h;
          if (n >= p.n_store) continue;
          half4_t b4 = {0, 0, 0, 0};
          if (use_bias) b4 = *reinterpret_cast<const half4_t*>(p.bias + n);
          float v[4];
          if (lnstat) {
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.ln_s + n);
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(p.ln_c + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = r16(ln_rs * (acc[i][j][q * 4 + e] - ln_mu * s4[e]) + c4[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = r16(acc[i][j][q * 4 + e] + (float)b4[e]);
          }
          if (ra) {
            const half4_t t4 = *reinterpret_cast<const half4_t*>(ra + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = r16(v[e] + (float)t4[e]);
          }
          if (p.act == MVOC_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = r16(silu_f(v[e]));
          } else if (p.act == MVOC_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = r16(gelu_fast_f(v[e]));
          }
          if (rs) {
            const half4_t r4 = *reinterpret_cast<const half4_t*>(rs + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = r16(v[e] + (float)r4[e]);
          }
          if (n + 4 <= p.n_store) {
            half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<half4_t*>(orow + n) = o;
          } else {
            for (int e = 0; e < 4 && n + e < p.n_store; ++e) orow[n + e] = (half_t)v[e];
          }
        }
      }
    }
  }
}

// Epilogue of the direct-to-LDS kernels: same arithmetic (and the same fp16 rounding points) as gemm_epilogue, but the
// output tile goes through LDS so that global stores and residual loads are 16 bytes per lane with consecutive lanes on
// consecutive chunks of a row.  In the accumulator layout a lane owns 4 channels (8 B) of one pixel and the 32 lanes of
// a store instruction hit 32 different rows: 8-byte row-strided requests, which bounded the K = 320 GEMMs
// (M x 320 x 320 moved 2 TB/s).  Everything up to the activation is applied in the accumulator layout (bias / LN-fold /
// row-add vectors are per channel there: broadcast loads), the fp16-rounded result is parked in a wave-private LDS
// tile [32 pixels][TO*32 channels] (pitch + 16 B: conflict-free 8-byte writes), read back as 16-byte row chunks, the
// residual added, stored.  `lds` = this wave's region (32 * (TO*64+16) bytes), free once every wave left the K loop.
// EPI 1: bias / row-add / residual only (no LayerNorm fold, no activation) -- the form most launches take, as its own code
template <int WN, int WM, int TN, int TM, bool GEGLU, int EPI = 0>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmArgs& p, f32x16 (&acc)[TN][TM], int n0, int m0, int wn, int wm,
                                                  int r, int h, int lane, char* lds, const float* lnstat_) {
  const float* lnstat = EPI == 1 ? nullptr : lnstat_;
  constexpr bool geglu = GEGLU;
  constexpr int TO = GEGLU ? TN / 2 : TN;  // 32-channel output tiles per wave
  const bool use_bias = p.bias && !lnstat;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int mrow = (wm * TM + j) * 32;
    const int m_own = m0 + mrow + r;
    const int m_safe = m_own < p.M ? m_own : p.M - 1;
    const half_t* ra = p.rowadd ? p.rowadd + (size_t)(m_safe / p.rowadd_div) * p.ld_rowadd : nullptr;
    float ln_mu = 0.f, ln_rs = 1.f;
    if (lnstat && m_own < p.M) {  // the statistics buffer ends at row M
      ln_mu = lnstat[2 * (mrow + r)];
      ln_rs = lnstat[2 * (mrow + r) + 1];
    }
    if constexpr (geglu) {
      if constexpr (TN % 2 == 0) {
        const int pitch = TO * 64 + 16;
#pragma unroll
        for (int i = 0; i < TN; i += 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nh = n0 + (wn * TN + i) * 32 + 8 * q + 4 * h;  // packed row of the value half
            half4_t o = {0, 0, 0, 0};
            if (nh < p.N) {
              half4_t bh = {0, 0, 0, 0}, bg = {0, 0, 0, 0};
              if (use_bias) {
                bh = *reinterpret_cast<const half4_t*>(p.bias + nh);
                bg = *reinterpret_cast<const half4_t*>(p.bias + nh + 32);
              }
              f32x4 sh = {0.f, 0.f, 0.f, 0.f}, sg = sh, ch = sh, cg = sh;
              if (lnstat) {
                sh = *reinterpret_cast<const f32x4*>(p.ln_s + nh);
                sg = *reinterpret_cast<const f32x4*>(p.ln_s + nh + 32);
                ch = *reinterpret_cast<const f32x4*>(p.ln_c + nh);
                cg = *reinterpret_cast<const f32x4*>(p.ln_c + nh + 32);
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float hv = r16(lnstat ? ln_rs * (acc[i][j][q * 4 + e] - ln_mu * sh[e]) + ch[e] : acc[i][j][q * 4 + e] + (float)bh[e]);
                const float gv = r16(lnstat ? ln_rs * (acc[i + 1][j][q * 4 + e] - ln_mu * sg[e]) + cg[e]
                                            : acc[i + 1][j][q * 4 + e] + (float)bg[e]);
                o[e] = (half_t)(hv * r16(gelu_fast_f(gv)));
              }
            }
            *reinterpret_cast<half4_t*>(lds + r * pitch + ((i / 2) * 32 + 8 * q + 4 * h) * 2) = o;
          }
        }
      }
    } else {
      const int pitch = TN * 64 + 16;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + (wn * TN + i) * 32 + 8 * q + 4 * h;
          half4_t o = {0, 0, 0, 0};
          if (n < p.n_store) {
            half4_t b4 = {0, 0, 0, 0};
            if (use_bias) b4 = *reinterpret_cast<const half4_t*>(p.bias + n);
            float v[4];
            if (lnstat) {
              const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.ln_s + n);
              const f32x4 c4 = *reinterpret_cast<const f32x4*>(p.ln_c + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = r16(ln_rs * (acc[i][j][q * 4 + e] - ln_mu * s4[e]) + c4[e]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = r16(acc[i][j][q * 4 + e] + (float)b4[e]);
            }
            if (ra) {
              const half4_t t4 = *reinterpret_cast<const half4_t*>(ra + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = r16(v[e] + (float)t4[e]);
            }
            if (EPI != 1 && p.act == MVOC_ACT_SILU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = r16(silu_f(v[e]));
            } else if (EPI != 1 && p.act == MVOC_ACT_GELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = r16(gelu_fast_f(v[e]));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
          }
          *reinterpret_cast<half4_t*>(lds + r * pitch + (i * 32 + 8 * q + 4 * h) * 2) = o;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's tile is in LDS (DS ops of a wave execute in order)
    // ---- read back as rows: chunk = 8 consecutive channels (16 B) of one pixel ---------------------------
    // in batches of NB chunks per lane: the batch's LDS reads and residual loads are issued before its first add
    // (NB = 2 keeps the 160-wide tiles at their register budget; a full unroll cost 40-70 VGPRs and an occupancy step)
    constexpr int nchunk = 4 * TO, pitch = TO * 64 + 16;
    constexpr int NIT = (32 * nchunk + 63) / 64;
    constexpr int NB = NIT <= 4 ? NIT : 2;
    const int nbase = geglu ? (n0 + wn * TN * 32) / 2 : n0 + wn * TN * 32;
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += NB) {
      half8_t v[NB], r8[NB];
      bool on[NB];
      size_t orow[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int idx = lane + (it0 + u) * 64;
        const int px = idx / nchunk, c = idx - px * nchunk;
        const int m = m0 + mrow + px, n = nbase + c * 8;
        on[u] = it0 + u < NIT && idx < 32 * nchunk && m < p.M && n < p.n_store;
        orow[u] = (size_t)m * p.ldo + n;
        v[u] = *reinterpret_cast<const half8_t*>(lds + (on[u] ? px * pitch + c * 16 : 0));
        if (p.resid && on[u]) r8[u] = *reinterpret_cast<const half8_t*>(p.resid + (size_t)m * p.ldr + n);
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        if (!on[u]) continue;
        half8_t o = v[u];
        if (p.resid) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)o[e] + (float)r8[u][e]);
        }
        *reinterpret_cast<half8_t*>(p.out + orow[u]) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next j overwrites the tile
  }
}

template <int WN, int WM, int TN, int TM>
__global__ __launch_bounds__(WN* WM * 64) void gemm_kernel(const GemmArgs p) {
  constexpr int NT = WN * WM * 64;
  constexpr int BN = WN * TN * 32;
  constexpr int BM = WM * TM * 32;
  constexpr int CHW = (BN * 4 + NT - 1) / NT;  // 16-byte chunks of W per thread per K step
  constexpr int CHA = (BM * 4 + NT - 1) / NT;
  constexpr int STAGE = (BN + BM) * ROWB;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave / WM, wm = wave % WM;
  const int r = lane & 31, h = lane >> 5;

  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const int n0 = (int)(logical % (unsigned)p.n_tiles) * BN;
  const int m0 = (int)(logical / (unsigned)p.n_tiles) * BM;

  const int kc = tid & 3;  // NT is a multiple of 4: every chunk of this thread has the same k sub-offset
  // ---- per-thread staging bookkeeping ---------------------------------------------------------
  RowInfo ri[CHA];
#pragma unroll
  for (int i = 0; i < CHA; ++i) {
    const int row = (tid + i * NT) >> 2;
    const int m = m0 + row;
    ri[i].m = (row < BM && m < p.M) ? m : -1;
    ri[i].y0 = ri[i].x0 = ri[i].img = 0;
    if (ri[i].m >= 0) {
      if (p.a_mode == MVOC_A_CONV3X3) {
        const int hwout = p.hout * p.wout;
        const int img = m / hwout;
        const int rem = m - img * hwout;
        const int oy = rem / p.wout;
        ri[i].img = img;
        ri[i].y0 = oy * p.stride - p.pad;
        ri[i].x0 = (rem - oy * p.wout) * p.stride - p.pad;
      } else if (p.a_mode == MVOC_A_TEMPORAL3) {
        ri[i].img = (m / p.hw) % p.frames;
      }
    }
  }
  // running (tap, channel) of this thread's k position kk = k0 + kc*8
  int tap = 0, ch = kc * 8;
  while (ch >= p.cin) { ch -= p.cin; ++tap; }

  half8_t regW[CHW], regA[CHA];
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < CHW; ++i) {
      const int c = tid + i * NT;
      const int row = c >> 2;
      regW[i] = zero8;
      if (row < BN && n0 + row < p.N)
        regW[i] = *reinterpret_cast<const half8_t*>(p.w + (size_t)(n0 + row) * p.K + k0 + kc * 8);
    }
    const half_t* src = p.a;
    int ld = p.lda, cc = ch;
    if (ch >= p.c1) { src = p.a2; ld = p.lda2; cc = ch - p.c1; }
    int ky = 0, kx = 0;
    if (p.a_mode == MVOC_A_CONV3X3) { ky = tap / 3; kx = tap - ky * 3; }
#pragma unroll
    for (int i = 0; i < CHA; ++i) {
      regA[i] = zero8;
      const int m = ri[i].m;
      if (m < 0) continue;
      long srow;
      if (p.a_mode == MVOC_A_PLAIN) {
        srow = m;
      } else if (p.a_mode == MVOC_A_TEMPORAL3) {
        const int f2 = ri[i].img + tap - 1;
        if (f2 < 0 || f2 >= p.frames || tap > 2) continue;
        srow = (long)m + (long)(tap - 1) * p.hw;
      } else {
        if (tap > 8) continue;
        int iy = ri[i].y0 + ky, ix = ri[i].x0 + kx;
        if (iy < 0 || ix < 0 || iy >= p.hup || ix >= p.wup) continue;
        if (p.upsample) {
          iy = min((int)floorf(iy * p.ups_sh), p.hsrc - 1);
          ix = min((int)floorf(ix * p.ups_sw), p.wsrc - 1);
        }
        srow = ((long)ri[i].img * p.hsrc + iy) * p.wsrc + ix;
      }
      regA[i] = *reinterpret_cast<const half8_t*>(src + srow * ld + cc);
    }
    ch += BK;
    while (ch >= p.cin) { ch -= p.cin; ++tap; }
  };

  auto store_tile = [&](int buf) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < CHW; ++i) {
      const int c = tid + i * NT;
      const int row = c >> 2;
      if (row < BN) *reinterpret_cast<half8_t*>(base + row * ROWB + kc * 16) = regW[i];
    }
#pragma unroll
    for (int i = 0; i < CHA; ++i) {
      const int c = tid + i * NT;
      const int row = c >> 2;
      if (row < BM) *reinterpret_cast<half8_t*>(base + BN * ROWB + row * ROWB + kc * 16) = regA[i];
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = p.K / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
    const char* wl = smem + cur * STAGE + (wn * TN * 32 + r) * ROWB + h * 16;
    const char* al = smem + cur * STAGE + BN * ROWB + (wm * TM * 32 + r) * ROWB + h * 16;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      half8_t wf[TN], af[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const half8_t*>(wl + i * 32 * ROWB + s * 32);
#pragma unroll
      for (int j = 0; j < TM; ++j) af[j] = *reinterpret_cast<const half8_t*>(al + j * 32 * ROWB + s * 32);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  gemm_epilogue<WN, WM, TN, TM>(p, acc, n0, m0, wn, wm, r, h);
#undef MVOC_SWZ
}

template <int WN, int WM, int TN, int TM>
int launch(const GemmArgs& a0, hipStream_t s) {
  GemmArgs a = a0;
  constexpr int BN = WN * TN * 32, BM = WM * TM * 32;
  a.n_tiles = (a.N + BN - 1) / BN;
  a.m_tiles = (a.M + BM - 1) / BM;
  const long nblk = (long)a.n_tiles * a.m_tiles;
  if (nblk <= 0 || nblk > 0x7fffffffL) {
    mvoc_set_error("gemm: grid of %ld blocks", nblk);
    return -2;
  }
  hipLaunchKernelGGL((gemm_kernel<WN, WM, TN, TM>), dim3((unsigned)nblk), dim3(WN * WM * 64), 0, s, a);
  return mvoc_check_launch("gemm_kernel");
}


// ------------------------------------------------------------------------------------------------------------
// Direct-to-LDS variant (global_load_lds_dwordx4): no staging registers, no ds_write pass; the loads of K-tile
// t+1 are in flight while tile t is multiplied.  K step BKK (64): one barrier per 64-deep step.
// An LDS-DMA wave-instruction writes 64 x 16 B CONTIGUOUSLY (wave-uniform base + lane*16), so rows cannot be
// padded; bank conflicts are removed by an XOR swizzle applied to the per-lane SOURCE address and again on
// the fragment read: 16-byte chunk c of row r sits at position c ^ ((r >> 1) & 7)  (BKK = 64: 128-byte rows;
// ds_read_b128 slot = 8*(r&1) + position -> 16 distinct slots over any 16 rows with distinct r mod 16).
// Zero padding of the conv / temporal gathers and M/N tails: the lane's source is a 16-byte zero constant.
// ------------------------------------------------------------------------------------------------------------
__device__ const uint4 g_zero16 = {0u, 0u, 0u, 0u};

// PLAIN = 1: single-source plain linear (no taps, no second source, no upsampling) -- the per-K-step address work is one
// pointer bump per LDS-DMA instead of the general gather bookkeeping (the K <= 1280 projections were VALU-issue bound:
// ~22 non-MFMA instructions per MFMA in the general loop)
template <int WN, int WM, int TN, int TM, int NST, int PF = 0, int BKK = 64, int PLAIN = 0, int EPI = 0>
__global__ __launch_bounds__(WN* WM * 64) void gemm_glds_kernel(const GemmArgs p) {
  static_assert(BKK == 64 || BKK == 32, "K step of 64 (128-byte staged rows) or 32 (64-byte rows)");
  constexpr int NCH = BKK / 8;           // 16-byte chunks per row
  constexpr int RPI = 64 / NCH;          // rows one LDS-DMA wave-instruction covers
  // XOR swizzle of the chunk position: 128-byte rows -> (row >> 1) & 7, 64-byte rows -> (row >> 2) & 3; either way
  // 16 consecutive rows reading the same logical chunk touch 16 distinct 16-byte slots of a 256-byte bank window
#define MVOC_SWZ(row) (BKK == 64 ? (((row) >> 1) & 7) : (((row) >> 2) & 3))
  constexpr int NW = WN * WM;            // waves
  constexpr int BN = WN * TN * 32;
  constexpr int BM = WM * TM * 32;
  constexpr int IW = BN * NCH / 64;      // wave-instructions per K step for the weight tile
  constexpr int IA = BM * NCH / 64;
  static_assert(IA % NW == 0, "every wave issues the same number of activation LDS-DMA instructions");
  static_assert(NST == 2 || IW % NW == 0, "the counted-vmcnt ring needs a uniform LDS-DMA count per wave");
  constexpr int PW = (IW + NW - 1) / NW;
  constexpr int PA = IA / NW;
  constexpr int ROW = BKK * 2;           // bytes
  constexpr int STAGE = (BN + BM) * ROW;
  constexpr int EPI_BYTES = 32 * (TN * 64 + 16);  // per-wave LDS tile of the transposing epilogue (gemm_epilogue_lds)
  constexpr int SMEM = NST * STAGE > NW * EPI_BYTES ? NST * STAGE : NW * EPI_BYTES;
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave / WM, wm = wave % WM;
  const int r = lane & 31, h = lane >> 5;

  const unsigned logical0 = xcd_remap(blockIdx.x, gridDim.x);
  const int slice = (int)(logical0 % (unsigned)p.split_k);
  const unsigned logical = logical0 / (unsigned)p.split_k;
  const int n0 = (int)(logical % (unsigned)p.n_tiles) * BN;
  const int m0 = (int)(logical / (unsigned)p.n_tiles) * BM;
  const int kbeg = slice * p.k_per_split;
  const half_t* zsrc = reinterpret_cast<const half_t*>(&g_zero16);

  // lane -> (row within the 8-row group of an instruction, position); source chunk = pos ^ swizzle(row)
  const int lrow = lane / NCH, pos = lane % NCH;
  // ---- weight side: instruction j = wave + i*NW covers rows j*8 .. j*8+7 -------------------------------
  const half_t* wsrc[PW];
  bool wok[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int row = (wave + i * NW) * RPI + lrow;
    const int c = pos ^ MVOC_SWZ(row);
    wok[i] = row < BN && n0 + row < p.N;
    wsrc[i] = p.w + (size_t)(wok[i] ? n0 + row : 0) * p.K + c * 8;
  }
  // ---- activation side (cin and c1 are multiples of 64: a K step never straddles a tap or a source) ----
  // Per staged row everything tap-independent is hoisted out of the K loop (the loop body was issue-bound on this
  // address arithmetic: ~7 VALU + 6 SALU per MFMA before): a signed source-row offset of the row for tap (0,0) and a
  // 9-bit mask of the taps that fall inside the image.  Per K step the wave-uniform tap adds a scalar row offset.
  int rowoff[PA], cch[PA];
  unsigned vmask[PA];
  int ry0[PA], rx0[PA], rimg[PA];  // only the upsample path still needs coordinates
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = (wave + i * NW) * RPI + lrow;
    const int m = m0 + row;
    const bool live = m < p.M;
    const int mm = live ? m : 0;
    ry0[i] = rx0[i] = rimg[i] = 0;
    rowoff[i] = mm;
    vmask[i] = live ? 1u : 0u;
    if (p.a_mode == MVOC_A_CONV3X3) {
      const int hwout = p.hout * p.wout;
      const int img = mm / hwout;
      const int rem = mm - img * hwout;
      const int oy = rem / p.wout;
      const int y0 = oy * p.stride - p.pad, x0 = (rem - oy * p.wout) * p.stride - p.pad;
      rimg[i] = img; ry0[i] = y0; rx0[i] = x0;
      rowoff[i] = (img * p.hsrc + y0) * p.wsrc + x0;
      unsigned mk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = y0 + t / 3, ix = x0 + t % 3;
        if (live && iy >= 0 && ix >= 0 && iy < p.hup && ix < p.wup) mk |= 1u << t;
      }
      vmask[i] = mk;
    } else if (p.a_mode == MVOC_A_TEMPORAL3) {
      const int f = (mm / p.hw) % p.frames;
      unsigned mk = 0;
#pragma unroll
      for (int t = 0; t < 3; ++t)
        if (live && f + t - 1 >= 0 && f + t - 1 < p.frames) mk |= 1u << t;
      vmask[i] = mk;
    }
    cch[i] = (pos ^ MVOC_SWZ(row)) * 8;
  }
  const half_t* aptr[PA];  // PLAIN: the lane's source pointer per staged row, bumped by one K step per issue
#pragma unroll
  for (int i = 0; i < PA; ++i) aptr[i] = (vmask[i] & 1u) ? p.a + (size_t)rowoff[i] * p.lda + cch[i] + kbeg : zsrc;
  int astep[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) astep[i] = 0;
  bool regather = true;
  int tap = kbeg / p.cin, ch0 = kbeg - tap * p.cin;  // wave-uniform position of the current K step: k0 = tap*cin + ch0
  // weight pointers advance by one K step per issue; rows beyond N read the zero constant with stride 0
  const half_t* wptr[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) wptr[i] = wok[i] ? wsrc[i] + kbeg : zsrc;

  auto issue = [&](int buf) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      if (IW % NW == 0 || wave + i * NW < IW)  // wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wptr[i],
                                         (__attribute__((address_space(3))) void*)(base + (wave + i * NW) * 1024), 16, 0, 0);
      wptr[i] += wok[i] ? BKK : 0;
    }
    if constexpr (PLAIN) {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)aptr[i],
                                         (__attribute__((address_space(3))) void*)(base + BN * ROW + (wave + i * NW) * 1024),
                                         16, 0, 0);
        aptr[i] += (vmask[i] & 1u) ? BKK : 0;
      }
      return;
    }
    // General gather.  The lane's source pointer only changes non-trivially when the K position crosses a tap or moves to
    // the second source (every cin/BKK or c1/BKK steps): those wave-uniform events rebuild the pointers; every other step
    // is a pointer bump, as in the plain path.
    if (regather) {
      const bool second = ch0 >= p.c1;
      const half_t* sbase = second ? p.a2 : p.a;
      const int ld = second ? p.lda2 : p.lda;
      const int cbase = second ? ch0 - p.c1 : ch0;
      const int ky = tap / 3, kx = tap - ky * 3;
      int tap_rows = 0;
      if (p.a_mode == MVOC_A_CONV3X3) tap_rows = ky * p.wsrc + kx;
      else if (p.a_mode == MVOC_A_TEMPORAL3) tap_rows = (tap - 1) * p.hw;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const bool ok = (vmask[i] >> tap) & 1u;
        long srow = rowoff[i] + tap_rows;
        if (p.upsample) {  // nearest-upsampled source: the row is not affine in the tap
          const int iy = min((int)floorf((ry0[i] + ky) * p.ups_sh), p.hsrc - 1);
          const int ix = min((int)floorf((rx0[i] + kx) * p.ups_sw), p.wsrc - 1);
          srow = ((long)rimg[i] * p.hsrc + iy) * p.wsrc + ix;
        }
        aptr[i] = ok ? sbase + srow * ld + (cbase + cch[i]) : zsrc;
        astep[i] = ok ? BKK : 0;
      }
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)aptr[i],
                                       (__attribute__((address_space(3))) void*)(base + BN * ROW + (wave + i * NW) * 1024),
                                       16, 0, 0);
      aptr[i] += astep[i];
    }
    ch0 += BKK;
    regather = ch0 == p.c1;  // the next step starts reading the second source (never true for single-source calls: c1 == cin)
    if (ch0 >= p.cin) { ch0 = 0; ++tap; regather = true; }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = p.k_per_split / BKK;
  const int swz = MVOC_SWZ(r);  // tile bases are multiples of 32 rows: the swizzle depends on r only
  static_assert(BKK != 64 || TM != 1 || WN * WM * 64 == 2 * BM, "in-kernel LayerNorm statistics assume two threads per staged activation row");
  issue(0);
  if (NST == 3 && nk > 1) issue(1);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (NST == 2) {
      __syncthreads();  // tile kt has landed (the barrier's fence drains this wave's LDS-DMA), buffer cur^1 is free
      if (kt + 1 < nk) issue(cur ^ 1);
    } else {
      // 3-stage ring, two K steps in flight: wait until only the NEWEST tile's PW+PA LDS-DMAs of this wave are
      // outstanding (vmcnt counts in issue order), then a raw barrier (no fence => no vmcnt(0) drain): every wave's
      // share of tile kt has landed and every wave is done reading the buffer tile kt+2 is about to overwrite.
      if (kt + 1 < nk)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW + PA) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 2 < nk) issue(cur == 0 ? 2 : cur - 1);
    }
    const char* wl = smem + cur * STAGE + (wn * TN * 32 + r) * ROW;
    const char* al = smem + cur * STAGE + BN * ROW + (wm * TM * 32 + r) * ROW;
    if constexpr (PF != 1) {
#pragma unroll
      for (int s = 0; s < BKK / 16; ++s) {
        const int off = ((2 * s + h) ^ swz) * 16;
        half8_t wf[TN], af[TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const half8_t*>(wl + i * 32 * ROW + off);
#pragma unroll
        for (int j = 0; j < TM; ++j) af[j] = *reinterpret_cast<const half8_t*>(al + j * 32 * ROW + off);
        if constexpr (PF == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], af[j], acc[i][j], 0, 0, 0);
        if constexpr (PF == 2) __builtin_amdgcn_s_setprio(0);
      }
    } else {
      static_assert(PF != 1 || BKK == 64, "the fragment-prefetch variant is written for K step 64");
      // software-pipelined fragments: the ds_reads of k-step s+1 are issued before the MFMAs of k-step s
      half8_t wf[2][TN], af[2][TM];
      {
        const int off = (h ^ swz) * 16;
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[0][i] = *reinterpret_cast<const half8_t*>(wl + i * 32 * ROW + off);
#pragma unroll
        for (int j = 0; j < TM; ++j) af[0][j] = *reinterpret_cast<const half8_t*>(al + j * 32 * ROW + off);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < 3) {
          const int off = ((2 * (s + 1) + h) ^ swz) * 16;
#pragma unroll
          for (int i = 0; i < TN; ++i) wf[(s + 1) & 1][i] = *reinterpret_cast<const half8_t*>(wl + i * 32 * ROW + off);
#pragma unroll
          for (int j = 0; j < TM; ++j) af[(s + 1) & 1][j] = *reinterpret_cast<const half8_t*>(al + j * 32 * ROW + off);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s & 1][i], af[s & 1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    cur = (cur + 1 == NST) ? 0 : cur + 1;
  }
  if (p.split_k > 1) {  // raw fp32 partials; bias / activation / residual happen in splitk_reduce_kernel
    float* slab = p.ws + (size_t)slice * p.M * p.N;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + (wm * TM + j) * 32 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + (wn * TN + i) * 32 + 8 * q + 4 * h;
          if (n >= p.N) continue;
          f32x4 v = {acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
          *reinterpret_cast<f32x4*>(slab + (size_t)m * p.N + n) = v;
        }
    }
    return;
  }
  if constexpr (EPI == 1) {  // (the host launches this form only for epi_lds launches without LayerNorm fold / activation)
    __syncthreads();
    gemm_epilogue_lds<WN, WM, TN, TM, false, 1>(p, acc, n0, m0, wn, wm, r, h, lane, smem + wave * EPI_BYTES, nullptr);
    return;
  }
  {
    if (p.epi_lds) {
      __syncthreads();  // every wave is out of the K loop: the staging buffers are free
      const float* st = p.ln_s ? p.ln_stats + 2 * (size_t)m0 : nullptr;
      if (p.act == MVOC_ACT_GEGLU) {
        if constexpr (TN % 2 == 0) gemm_epilogue_lds<WN, WM, TN, TM, true>(p, acc, n0, m0, wn, wm, r, h, lane, smem + wave * EPI_BYTES, st);
      } else {
        gemm_epilogue_lds<WN, WM, TN, TM, false>(p, acc, n0, m0, wn, wm, r, h, lane, smem + wave * EPI_BYTES, st);
      }
      return;
    }
  }
  if (p.ln_s && p.ln_stats) {
    gemm_epilogue<WN, WM, TN, TM>(p, acc, n0, m0, wn, wm, r, h, p.ln_stats + 2 * (size_t)m0);
    return;
  }
  gemm_epilogue<WN, WM, TN, TM>(p, acc, n0, m0, wn, wm, r, h);
}

// sums the split-K slabs in slice order (deterministic) and applies the GEMM epilogue; thread = 4 consecutive n of a row
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = p.N / 4;
  if (idx >= (long)p.M * n4) return;
  const int m = (int)(idx / n4), n = (int)(idx % n4) * 4;
  if (n >= p.n_store) return;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < p.split_k; ++s) a += *reinterpret_cast<const f32x4*>(p.ws + ((size_t)s * p.M + m) * p.N + n);
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = r16(a[e] + (p.bias ? (float)p.bias[n + e] : 0.f));
  if (p.rowadd) {
    const half_t* ra = p.rowadd + (size_t)(m / p.rowadd_div) * p.ld_rowadd;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = r16(v[e] + (float)ra[n + e]);
  }
  if (p.act == MVOC_ACT_SILU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = r16(silu_f(v[e]));
  } else if (p.act == MVOC_ACT_GELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = r16(gelu_fast_f(v[e]));
  }
  if (p.resid) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = r16(v[e] + (float)p.resid[(size_t)m * p.ldr + n + e]);
  }
  half_t* o = p.out + (size_t)m * p.ldo + n;
  if (n + 4 <= p.n_store) {
    half4_t o4 = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    *reinterpret_cast<half4_t*>(o) = o4;
  } else {
    for (int e = 0; e < 4 && n + e < p.n_store; ++e) o[e] = (half_t)v[e];
  }
}

// The same reduction for launches whose consumer is a GroupNorm (chan_sums requested; no activation): one block per 256-row slab
// and 32-channel strip -- thread = one row, its 32 channels in eight 16-byte groups -- so that the per-slab channel sums of the
// STORED values (the contract of gemm8's stats_pass) come out of the pass that writes them: the 8 x 8 / 16 x 16 levels, whose
// under-filled grids run split-K, then skip gn_partial like the levels above.  Sums in a fixed order (row by row): deterministic.
__global__ __launch_bounds__(256) void splitk_reduce_sums_kernel(const GemmArgs p) {
  // block = one 256-row slab x 32 channels; thread (row group rg = tid >> 3, channel quad cq = tid & 7) walks rows rg, rg + 32, ...:
  // 8 threads read 128 contiguous bytes of a row, a wave 8 rows.  Its 8 rows' stored values are summed in registers, the 32 row
  // groups meet in LDS and are added in a fixed order (deterministic sums).
  __shared__ float ls[32][33], lq[32][33];
  const int tid = threadIdx.x, rg = tid >> 3, cq = tid & 7;
  const int n = blockIdx.y * 32 + 4 * cq;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bs[e] = (float)p.bias[n + e];
  }
  float sx[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int i = 0; i < 8; ++i) {
    const int m = blockIdx.x * 256 + rg + 32 * i;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.split_k; ++s) a += *reinterpret_cast<const f32x4*>(p.ws + ((size_t)s * p.M + m) * p.N + n);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = r16(a[e] + bs[e]);
    if (p.rowadd) {
      const half4_t ra = *reinterpret_cast<const half4_t*>(p.rowadd + (size_t)(m / p.rowadd_div) * p.ld_rowadd + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = r16(v[e] + (float)ra[e]);
    }
    if (p.resid) {
      const half4_t rs = *reinterpret_cast<const half4_t*>(p.resid + (size_t)m * p.ldr + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = r16(v[e] + (float)rs[e]);
    }
    const half4_t o4 = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    *reinterpret_cast<half4_t*>(p.out + (size_t)m * p.ldo + n) = o4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float f = (float)o4[e];
      sx[e] += f;
      sq[e] += f * f;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ls[rg][4 * cq + e] = sx[e];
    lq[rg][4 * cq + e] = sq[e];
  }
  __syncthreads();
  if (tid < 64) {  // thread (channel, kind) adds the 32 row groups in order
    const int c = tid & 31;
    const float(*src)[33] = tid < 32 ? ls : lq;
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) acc += src[g][c];
    p.stats[((size_t)blockIdx.x * p.n_store + blockIdx.y * 32 + c) * 2 + (tid >> 5)] = acc;
  }
}

template <int WN, int WM, int TN, int TM, int NST = 2, int PF = 0, int BKK = 64, int PLAIN = 0>
int launch_glds(const GemmArgs& a0, hipStream_t s) {
  GemmArgs a = a0;
  constexpr int BN = WN * TN * 32, BM = WM * TM * 32;
  a.n_tiles = (a.N + BN - 1) / BN;
  a.m_tiles = (a.M + BM - 1) / BM;
  const long nblk = (long)a.n_tiles * a.m_tiles;
  if (nblk <= 0 || nblk > 0x7fffffffL) {
    mvoc_set_error("gemm: grid of %ld blocks", nblk);
    return -2;
  }
  // The epilogue of these kernels covers every activation / LayerNorm / GEGLU form at run time: ~25 k instructions behind a
  // 40-instruction K loop, executed once per block, far beyond the instruction cache (gemm8.hip, EPI, has the measurement).
  // The tile shapes that carry the denoising loop's small launches get the plain form as an own instantiation.
  constexpr bool HOT = (WN == 2 && WM == 2 && TN == 2 && TM == 2) || (WN == 1 && WM == 4 && TM == 1 && (TN == 2 || TN == 5));
  if (HOT && a.epi_lds && !a.ln_s && a.act == MVOC_ACT_NONE && a.split_k == 1) {
    if constexpr (HOT) {
      hipLaunchKernelGGL((gemm_glds_kernel<WN, WM, TN, TM, NST, PF, BKK, PLAIN, 1>), dim3((unsigned)nblk), dim3(WN * WM * 64), 0, s, a);
      return mvoc_check_launch("gemm_glds_kernel (plain epilogue)");
    }
  }
  hipLaunchKernelGGL((gemm_glds_kernel<WN, WM, TN, TM, NST, PF, BKK, PLAIN>), dim3((unsigned)(nblk * a.split_k)), dim3(WN * WM * 64), 0, s, a);
  if (a.split_k > 1) {
    if (a.stats) {  // (set by the caller only when splitk_reduce_sums_kernel's shape conditions hold)
      hipLaunchKernelGGL(splitk_reduce_sums_kernel, dim3((unsigned)(a.M / 256), (unsigned)(a.N / 32)), dim3(256), 0, s, a);
    } else {
      const long nthr = (long)a.M * (a.N / 4);
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, a);
    }
  }
  return mvoc_check_launch("gemm_glds_kernel");
}

}  // namespace

extern "C" size_t mvoc_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  if (m <= 0 || n <= 0 || k < 2048 || m > 8192) return 0;
  return (size_t)8 * (size_t)m * (size_t)n * sizeof(float);
}

namespace {
thread_local int g_sums_written = 0;
thread_local int g_rowmom_bx = 0;
const bool g_trace = getenv("MVOC_GEMM_TRACE") != nullptr;  // diagnostics: one line per launch with the dispatch decision
}
extern "C" int mvoc_gemm_chan_sums_written(void) { return g_sums_written; }
extern "C" int mvoc_gemm_row_moments_written(void) { return g_rowmom_bx; }

namespace {
// splitk_reduce_sums_kernel reads bias / row-add / residual and writes the output in 8-byte groups of four channels
bool sums_reduce_ok(const mvoc_gemm_desc* d) {
  return d->ldo % 4 == 0 && ((uintptr_t)d->out & 7) == 0 && (!d->resid || (d->ldr % 4 == 0 && ((uintptr_t)d->resid & 7) == 0)) &&
         (!d->rowadd || (d->ld_rowadd % 4 == 0 && ((uintptr_t)d->rowadd & 7) == 0));
}
}  // namespace

extern "C" int mvoc_gemm_f16(const mvoc_gemm_desc* d, void* stream) {
  g_sums_written = 0;
  g_rowmom_bx = 0;
  MVOC_REQUIRE(d && d->a && d->w && d->out, -1, "gemm: null operand");
  MVOC_REQUIRE(d->m > 0 && d->n > 0 && d->k > 0, -1, "gemm: empty problem m=%ld n=%ld k=%ld", (long)d->m, (long)d->n,
               (long)d->k);
  MVOC_REQUIRE(d->m < (1LL << 31) && d->n < (1 << 24) && d->k < (1 << 24), -2, "gemm: problem too large");
  MVOC_REQUIRE(d->n % 32 == 0 && d->k % BK == 0, -2, "gemm: n (%ld) must be a multiple of 32 and k (%ld) of %d",
               (long)d->n, (long)d->k, BK);
  MVOC_REQUIRE(d->cin > 0 && d->cin % 8 == 0 && d->c1 % 8 == 0 && d->c1 <= d->cin, -2,
               "gemm: cin (%d) and c1 (%d) must be multiples of 8", d->cin, d->c1);
  MVOC_REQUIRE(d->a2 || d->c1 == d->cin, -1, "gemm: c1 < cin needs a second source");
  MVOC_REQUIRE(d->a_mode >= 0 && d->a_mode <= 2, -1, "gemm: bad a_mode %d", d->a_mode);
  MVOC_REQUIRE(d->ldo % 4 == 0 && (d->resid == nullptr || d->ldr % 4 == 0), -2, "gemm: ldo/ldr must be multiples of 4");
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.a = (const half_t*)d->a; a.a2 = (const half_t*)d->a2; a.w = (const half_t*)d->w; a.out = (half_t*)d->out;
  a.bias = (const half_t*)d->bias; a.rowadd = (const half_t*)d->rowadd; a.resid = (const half_t*)d->resid;
  a.M = (int)d->m; a.N = (int)d->n; a.K = (int)d->k;
  a.n_store = d->n_store > 0 ? d->n_store : (int)d->n;
  a.ldo = d->ldo; a.ldr = d->ldr; a.ld_rowadd = d->ld_rowadd; a.rowadd_div = d->rowadd_div > 0 ? d->rowadd_div : 1;
  a.a_mode = d->a_mode; a.lda = d->lda; a.lda2 = d->lda2; a.c1 = d->c1; a.cin = d->cin;
  a.nimg = d->nimg; a.hout = d->hout; a.wout = d->wout; a.hsrc = d->hsrc; a.wsrc = d->wsrc;
  a.stride = d->stride > 0 ? d->stride : 1; a.upsample = d->upsample;
  a.pad = d->pad_mode == 1 ? 0 : 1;
  a.hup = d->upsample ? d->hup : d->hsrc; a.wup = d->upsample ? d->wup : d->wsrc;
  a.ups_sh = d->upsample ? (float)d->hsrc / (float)d->hup : 1.f;
  a.ups_sw = d->upsample ? (float)d->wsrc / (float)d->wup : 1.f;
  a.frames = d->frames; a.hw = d->hw; a.act = d->act;
  a.split_k = 1; a.k_per_split = (int)d->k; a.ws = nullptr;
  a.ln_s = (const float*)d->ln_rowsum; a.ln_c = (const float*)d->ln_bias; a.ln_eps = d->ln_eps;
  a.ln_stats = (const float*)d->ln_stats;
  {
    const int ns = d->act == MVOC_ACT_GEGLU ? (int)d->n / 2 : (d->n_store > 0 ? d->n_store : (int)d->n);
    a.epi_lds = d->ldo % 8 == 0 && ns % 8 == 0 && ((uintptr_t)d->out & 15) == 0 &&
                (d->resid == nullptr || (d->ldr % 8 == 0 && ((uintptr_t)d->resid & 15) == 0));
  }
  if (d->a_mode == MVOC_A_CONV3X3) {
    MVOC_REQUIRE(d->nimg > 0 && d->hout > 0 && d->wout > 0 && d->hsrc > 0 && d->wsrc > 0, -1, "gemm: conv dims");
    MVOC_REQUIRE((int64_t)d->nimg * d->hout * d->wout == d->m, -1, "gemm: conv m != nimg*hout*wout");
    MVOC_REQUIRE(d->upsample == 2 || d->k >= 9 * (int64_t)d->cin, -1, "gemm: conv k < 9*cin");
    MVOC_REQUIRE(d->pad_mode == 0 || (d->pad_mode == 1 && !d->upsample), -1, "gemm: pad_mode %d", d->pad_mode);
    if (d->upsample == 2) {
      // sub-pixel form of Upsample2D + conv (gemm_args.h: subpx): per output parity a 2 x 2 conv on the source image with the
      // caller's four summed kernels w [4][n][4 cin] -- 4 taps of MFMA work instead of 9.  Eight-phase tiles only.
      const int64_t spr = (int64_t)d->nimg * d->hsrc * d->wsrc;
      MVOC_REQUIRE(d->hup == 2 * d->hsrc && d->wup == 2 * d->wsrc && d->hout == d->hup && d->wout == d->wup && d->stride <= 1 &&
                       d->k == 4 * (int64_t)d->cin && spr % 256 == 0 && d->a2 == nullptr && d->resid == nullptr && d->rowadd == nullptr &&
                       d->act == MVOC_ACT_NONE && !d->ln_rowsum && d->split_k <= 1 && d->m >= 1024 && (d->tile == 0 || d->tile == 81),
                   -2, "gemm: upsample == 2 (sub-pixel) needs an exact 2x upsample, k == 4 cin, one source, nimg * hsrc * wsrc %% 256 == 0, "
                       "the plain epilogue, no split-K and m >= 1024");
      a.subpx = 1;
      a.sp_rows = (int)spr;
      a.hout = d->hsrc; a.wout = d->wsrc;  // the row grid of a phase is the SOURCE grid
      a.hup = d->hsrc; a.wup = d->wsrc;    // ... and so are the bounds the taps are tested against
    }
  } else if (d->a_mode == MVOC_A_TEMPORAL3) {
    MVOC_REQUIRE(d->frames > 0 && d->hw > 0 && d->m % ((int64_t)d->frames * d->hw) == 0, -1, "gemm: temporal dims");
    MVOC_REQUIRE(d->k == 3 * (int64_t)d->cin, -1, "gemm: temporal k != 3*cin");
  } else {
    MVOC_REQUIRE(d->k == d->cin, -1, "gemm: plain mode needs cin == k");
  }
  if (d->act == MVOC_ACT_GEGLU) {
    MVOC_REQUIRE(d->n % 64 == 0, -2, "gemm: GEGLU needs n %% 64 == 0");
    a.n_store = (int)d->n / 2;
  }
  hipStream_t s = (hipStream_t)stream;
  const double flops = 2.0 * (double)d->m * (double)d->n * (double)d->k;
  MvocProfScope prof(MVOC_FAM_GEMM, s, flops);
  const bool glds_ok = d->k % 64 == 0 && d->cin % 64 == 0 && d->c1 % 64 == 0 &&
                       (d->a_mode != MVOC_A_CONV3X3 || d->k == (d->upsample == 2 ? 4 : 9) * (int64_t)d->cin);
  if (d->ln_rowsum) {
    MVOC_REQUIRE(d->ln_bias && d->ln_stats && glds_ok && d->a_mode == MVOC_A_PLAIN && d->a2 == nullptr && d->split_k <= 1 &&
                     (d->tile == 0 || d->tile >= 11),
                 -2, "gemm: LayerNorm folding needs the rows' statistics (ln_stats, mvoc_row_stats_f16) and the plain single-source "
                     "direct-to-LDS path (k %% 64 == 0), no split-K");
  }
  int tile = d->tile;
  int model_sk = 0, g8_sk = 0;
  // mvoc_gemm_desc.concurrency: how many independent launches of this shape the caller runs at the same time on other streams (the
  // job's three source inversions: pipeline.invert_concurrent).  The dispatch below prices a grid's fill and decides on split-K as
  // if the grid were this many times larger: an under-filled batch-1 launch then keeps its K in one piece (no fp32 slabs, no
  // reduce pass) and lets the other clips' blocks take the idle CUs.
  const int conc = d->concurrency < 1 ? 1 : (d->concurrency > 8 ? 8 : d->concurrency);
  // ---- eight-phase tiles (gemm8.hip): 81 = 256 channels x 256 pixels per block, 82 = 320 x 256 ------------------------------
  const int64_t rows_a = d->a_mode == MVOC_A_CONV3X3 ? (int64_t)d->nimg * d->hsrc * d->wsrc : d->m;
  const int64_t lim = (int64_t)1 << 31;  // 32-bit MUBUF offsets, rows beyond the range read zeros
  const bool g8_ok = glds_ok && a.epi_lds && (a.n_store % 8 == 0) &&
                     rows_a * d->lda * 2 < lim && (d->a2 == nullptr || rows_a * d->lda2 * 2 < lim) &&
                     (int64_t)d->n * d->k * 2 < lim &&
                     (int64_t)d->m * d->ldo * 2 < lim && (!d->resid || (int64_t)d->m * d->ldr * 2 < lim) &&
                     // the epilogue's vector table (gemm8.hip): 16-byte loads of bias / LayerNorm vectors / row-add rows, at most
                     // 5 row-add rows per 256-row tile
                     ((uintptr_t)d->bias & 15) == 0 && ((uintptr_t)d->ln_rowsum & 15) == 0 && ((uintptr_t)d->ln_bias & 15) == 0 &&
                     (!d->rowadd || (((uintptr_t)d->rowadd & 15) == 0 && d->ld_rowadd % 8 == 0 && a.rowadd_div >= 64)) &&
                     // the folded-upsample instantiation carries the plain epilogue only (gemm8.hip: launch8)
                     (!d->upsample || (d->act == MVOC_ACT_NONE && !d->ln_rowsum));
  if (d->k_order) {
    MVOC_REQUIRE(d->k_order == 1 && d->a_mode != MVOC_A_PLAIN && !d->upsample && g8_ok && d->m >= 1024 && (tile == 0 || tile == 82) &&
                     d->k == (d->a_mode == MVOC_A_CONV3X3 ? 9 : 3) * (int64_t)d->cin && d->n % 320 == 0 && d->act == MVOC_ACT_NONE &&
                     !d->ln_rowsum && d->split_k <= 1 &&
                     (d->a_mode != MVOC_A_CONV3X3 || (a.stride == 1 && a.pad == 1 && d->hsrc == d->hout && d->wsrc == d->wout)),
                 -2, "gemm: k_order = 1 (chunk-major K) is a form of the 320-wide eight-phase tile: conv3x3 (stride 1, pad 1) / temporal3, "
                     "n %% 320 == 0, cin, c1 %% 64 == 0, no upsample, no activation, no split-K, m >= 1024, 16-byte addressable operands < 2 GB");
    a.korder = 1;
    tile = 82;
  }
  if (a.subpx) {
    MVOC_REQUIRE(g8_ok, -2, "gemm: the sub-pixel upsample conv runs on the eight-phase tiles only (k, cin %% 64 == 0, 16-byte addressable output)");
    tile = 81;
  }
  if (tile == 0 && g8_ok && d->m >= 1024 && d->k >= 256) {
    // Measured (tools/gemm_bench.py, B = 1 and B = 5 shape sets, profiles/r3/): the eight-phase tiles win wherever their grid
    // fills the chip; what decides between them and against the general tiles is quantisation -- channels wasted in the last
    // tile of a row and CUs idle in the last wave of blocks (one block per CU).  Per flop the 320-wide form runs at ~0.85 of the
    // 256-wide one (phase stamps: 3 430 against 2 300 ticks per K tile for 1.25 x the channels; re-reading Yh0 costs it the rest).
    // (Until the epilogue forms were split -- gemm8.hip, EPI -- the short-K projections, n and k <= 640, were faster on the
    // general tiles; with the 3 k-tick plain epilogue they are 30-45 % faster here.)
    auto eff = [&](int bx, double rate) {
      const int64_t nt = (d->n + bx - 1) / bx, blocks = ((d->m + 255) / 256) * nt * conc;
      return (double)d->n / (double)(nt * bx) * (double)blocks / (double)(((blocks + 255) / 256) * 256) * rate;
    };
    const double e81 = eff(256, 1.0);
    const double e82 = (d->act != MVOC_ACT_GEGLU && d->n % 320 == 0) ? eff(320, 0.85) : 0.0;
    if (e81 >= 0.55 || e82 >= 0.55) {
      tile = e82 > e81 ? 82 : 81;
    } else if (d->workspace && d->split_k == 0 && !d->ln_rowsum && d->act != MVOC_ACT_GEGLU && d->k >= 3840) {
      // deep K on a grid of 64..190 tiles (the 16x16 / 8x8 levels): K slices bring the grid to one block per CU; measured
      // 1.2-1.4x over the split-K form of the 128-wide tiles (M = 4096 / 5120, K = 3840 .. 23040)
      const int64_t blocks = ((d->m + 255) / 256) * ((d->n + 255) / 256) * conc;
      if (blocks >= 64 && blocks < 190) {
        for (int sk = (int)(256 / blocks) > 4 ? 4 : (int)(256 / blocks); sk >= 2; --sk)
          if (d->k % (64 * sk) == 0 && (size_t)sk * d->m * d->n * 4 <= d->workspace_bytes) {
            tile = 81;
            g8_sk = sk;
            break;
          }
      }
    }
  }
  if (tile == 0 && glds_ok) {
    // what the eight-phase tiles do not take: grids that fill less than ~55 % of the chip after quantisation, M < 1024, K < 256
    if (d->k <= 640 && d->m > 2048 && !(d->act == MVOC_ACT_GEGLU && d->k > 320 && d->m >= 16384)) {
      // five to ten K steps: the launch is prologue / epilogue bound, not MFMA bound -- K step 32 halves the LDS per
      // block so four (128x128) or three (160x128) blocks share a CU and hide each other's ramps: 10-20 % faster here
      tile = (d->act == MVOC_ACT_GEGLU || d->n % 128 == 0 || d->n > 320 || d->n % 160) ? 61 : 62;
    } else if (d->act == MVOC_ACT_GEGLU) {
      tile = 11;
    } else if (d->m <= 2048 && !(d->workspace && d->k >= 2048)) {
      tile = 13;  // few rows and no split-K: many small blocks (latency-bound regime, outside the model below)
    } else {
      // Pick the tile by a wave-quantisation cost model calibrated on MI355X (tools/gemm_bench.py, B = 1 and B = 5 shape
      // sets).  The K loop is bound by the L2 -> LDS fill rate (~70 GB/s per CU), so a tile's chip-wide rate grows with
      // its flop per staged byte: `rate` = TFLOP/s it sustains when the grid fills the chip.  A launch runs in "waves" of
      // `slots` resident blocks; a partly filled last wave costs 0.3 + 0.7 * fill of a full one.
      // (Round 4: the 8-wave tiles 14 / 15 / 65 / 66 / 67 and tile 63 left the build -- no row of any profile since the
      // eight-phase tiles took their launches; profiles/r3/gemm_per_shape_* keep their numbers.)
      static const struct { int tile, bn, bm, slots; float rate; } cand[] = {
          {12, 160, 128, 512, 950.f}, {11, 128, 128, 512, 890.f}, {64, 160, 256, 512, 1090.f}};
      const bool can_split = d->workspace && d->split_k == 0 && !d->ln_rowsum;
      double best = 0;
      for (const auto& c : cand) {
        if (d->n % c.bn) continue;
        if (c.bm == 256 && d->m < 65536) continue;  // measured: below 64 K rows the 256-row tile's tail costs more than the model says
        const long blocks = ((d->m + c.bm - 1) / c.bm) * ((d->n + c.bn - 1) / c.bn) * conc;
        int sk = 1;
        if (can_split && blocks < 384 && c.bm == 128)
          while (sk < 8 && blocks * sk < 512 && d->k % (64 * sk * 2) == 0 && d->k / (sk * 2) >= 512) sk *= 2;
        if (sk > 1 && (size_t)sk * d->m * d->n * 4 > d->workspace_bytes) sk = 1;
        const double w = (double)(blocks * sk) / c.slots;
        const double full = (double)(long)w, frac = w - full;
        const double waves = full + (frac > 0 ? 0.3 + 0.7 * frac : 0.0);
        double cost = waves * c.slots * 2.0 * c.bm * c.bn * (double)(d->k / sk) / (c.rate * 1e6);  // us
        if (sk > 1) cost += 5.0 + 2.0 * sk * (double)d->m * d->n * 4.0 / 3.0e6;                    // fp32 slabs + reduce pass
        if (tile == 0 || cost < best) { best = cost; tile = c.tile; model_sk = sk; }
      }
      if (tile == 0) tile = 13;  // n is a multiple of neither 160 nor 128
    }
  }
  if (tile == 0) {
    if (d->act == MVOC_ACT_GEGLU) tile = 1;
    else if (d->n % 160 == 0 && d->m >= 2048) tile = 2;
    else if (d->n % 128 == 0 && d->m >= 2048) tile = 1;
    else tile = 3;
  }
  if (tile >= 11 && d->act != MVOC_ACT_GEGLU && d->workspace && d->split_k != 1 && !d->ln_rowsum) {
    // split-K when the tile grid cannot fill the chip: slices of >= 512 deep, fp32 slabs in the caller's workspace
    const bool t8 = tile == 81 || tile == 82;
    const int bm = t8 || tile == 64 ? 256 : 128;
    const int bn = tile == 82 ? 320 : tile == 81 ? 256 : (tile % 10 == 2 || tile == 64) ? 160 : (tile % 10 == 3 ? 64 : 128);
    const long blocks = ((d->m + bm - 1) / bm) * ((d->n + bn - 1) / bn) * conc;
    int sk = d->split_k > 1 ? d->split_k : 1;
    if (model_sk > 0) {
      sk = model_sk;
    } else if (d->split_k == 0 && blocks < 384) {
      while (sk < 8 && blocks * sk < 512 && d->k % (64 * sk * 2) == 0 && d->k / (sk * 2) >= 512) sk *= 2;
    }
    if (sk > 1 && d->k % (64 * sk) == 0 && (size_t)sk * d->m * d->n * 4 <= d->workspace_bytes) {
      a.split_k = sk;
      a.k_per_split = (int)(d->k / sk);
      a.ws = (float*)d->workspace;
      if (!t8 && d->chan_sums && d->act == MVOC_ACT_NONE && a.M % 256 == 0 && a.N % 32 == 0 && a.n_store == a.N && sums_reduce_ok(d)) {
        a.stats = (float*)d->chan_sums;  // the general tiles' split-K reduction emits the GroupNorm statistics too
        g_sums_written = 1;
      }
    }
  }
  if (g_trace) {
    fprintf(stderr, "[mvoc gemm] mode %d m %ld n %ld k %ld cin %d act %d resid %d ln %d -> tile %d split_k %d\n", d->a_mode, (long)d->m,
            (long)d->n, (long)d->k, d->cin, d->act, d->resid != nullptr, d->ln_rowsum != nullptr, tile, (tile == 81 || tile == 82) ? (d->tile ? d->split_k : (g8_sk > 1 ? g8_sk : 1)) : a.split_k);
  }
  {
    if (tile == 81 || tile == 82) {
      MVOC_REQUIRE(g8_ok && !(tile == 82 && d->act == MVOC_ACT_GEGLU), -2,
                   "gemm: tiles 81 / 82 need k, cin, c1 %% 64 == 0, 16-byte addressable outputs, row statistics, operands < 2 GB "
                   "(82: no GEGLU)");
      a.split_k = 1; a.k_per_split = (int)d->k; a.ws = nullptr;
      if (g8_sk > 1) {  // the automatic choice for an under-filled grid
        a.split_k = g8_sk; a.k_per_split = (int)(d->k / g8_sk); a.ws = (float*)d->workspace;
      }
      if (d->tile != 0) {  // forced tile: honour a forced split too
        if (d->split_k > 1 && d->workspace && d->act != MVOC_ACT_GEGLU && !d->ln_rowsum && d->k % (64 * d->split_k) == 0 &&
            (size_t)d->split_k * d->m * d->n * 4 <= d->workspace_bytes) {
          a.split_k = d->split_k; a.k_per_split = (int)(d->k / d->split_k); a.ws = (float*)d->workspace;
        }
      }
      if (d->chan_sums && a.split_k == 1 && d->act == MVOC_ACT_NONE && a.M % 256 == 0 && !a.subpx) {  // statistics of the stored tile from the epilogue
        // (whole 256-row tiles only: stats_pass sums the tile as it stands in LDS, phantom rows >= M included)
        a.stats = (float*)d->chan_sums;
        g_sums_written = 1;
      }
      if (d->row_moments && a.split_k == 1 && d->act != MVOC_ACT_GEGLU && a.n_store == a.N && !a.subpx) {  // (sub-pixel form: tile rows are not output rows)  // LayerNorm statistics likewise
        MVOC_REQUIRE(d->row_moments_ld >= (a.N + 255) / 256, -2, "gemm: row_moments_ld %d < ceil(n / 256)", d->row_moments_ld);
        a.rowmom = (float*)d->row_moments;
        a.rowmom_ld = d->row_moments_ld;
      }
      int bx_used = 0;
      const int rc = mvoc_launch_gemm8(a, tile == 81 ? 256 : 320, s, &bx_used);
      if (rc == 0 && a.rowmom) g_rowmom_bx = bx_used;
      if (rc == 0 && a.split_k > 1) {
        if (d->chan_sums && d->act == MVOC_ACT_NONE && a.M % 256 == 0 && a.N % 32 == 0 && a.n_store == a.N && sums_reduce_ok(d)) {
          a.stats = (float*)d->chan_sums;
          g_sums_written = 1;
          hipLaunchKernelGGL(splitk_reduce_sums_kernel, dim3((unsigned)(a.M / 256), (unsigned)(a.N / 32)), dim3(256), 0, s, a);
          return mvoc_check_launch("splitk_reduce_sums_kernel");
        }
        const long nthr = (long)a.M * (a.N / 4);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, a);
        return mvoc_check_launch("splitk_reduce_kernel");
      }
      return rc;
    }
  }
  if (glds_ok && d->a_mode == MVOC_A_PLAIN && d->a2 == nullptr && !d->upsample) {
    switch (tile) {  // the tiles the selection above produces, in their single-source plain-linear form
      case 11: return launch_glds<2, 2, 2, 2, 2, 0, 64, 1>(a, s);
      case 12: if (d->act != MVOC_ACT_GEGLU) return launch_glds<1, 4, 5, 1, 2, 0, 64, 1>(a, s); break;
      case 13: return launch_glds<1, 4, 2, 1, 2, 0, 64, 1>(a, s);
      case 61: return launch_glds<2, 2, 2, 2, 2, 0, 32, 1>(a, s);
      case 62: if (d->act != MVOC_ACT_GEGLU) return launch_glds<1, 4, 5, 1, 2, 0, 32, 1>(a, s); break;
      default: break;
    }
  }
  switch (tile) {
    case 1: return launch<2, 2, 2, 2>(a, s);  // 128 x 128
    case 2:
      MVOC_REQUIRE(d->act != MVOC_ACT_GEGLU, -2, "gemm: tile 2 cannot do GEGLU");
      return launch<1, 4, 5, 1>(a, s);        // 160 x 128
    case 3:
      if (d->act == MVOC_ACT_GEGLU) return launch<1, 4, 2, 1>(a, s);
      return launch<2, 2, 1, 2>(a, s);        // 64 x 128
    case 4: return launch<1, 4, 2, 1>(a, s);  // 64 x 128, waves along m
    // direct-to-LDS variants (K step 64)
    case 11:
      MVOC_REQUIRE(glds_ok, -2, "gemm: glds tiles need k, cin, c1 %% 64 == 0");
      return launch_glds<2, 2, 2, 2>(a, s);  // 128 x 128
    case 12:
      MVOC_REQUIRE(glds_ok && d->act != MVOC_ACT_GEGLU, -2, "gemm: tile 12 needs k, cin, c1 %% 64 == 0 and no GEGLU");
      return launch_glds<1, 4, 5, 1>(a, s);  // 160 x 128
    case 13:
      MVOC_REQUIRE(glds_ok, -2, "gemm: glds tiles need k, cin, c1 %% 64 == 0");
      return launch_glds<1, 4, 2, 1>(a, s);  // 64 x 128
    // K step 32: half the LDS per block -> more resident blocks per CU
    case 61:
      MVOC_REQUIRE(glds_ok, -2, "gemm: tile 61 needs k, cin, c1 %% 64 == 0");
      return launch_glds<2, 2, 2, 2, 2, 0, 32>(a, s);
    case 62:
      MVOC_REQUIRE(glds_ok && d->act != MVOC_ACT_GEGLU, -2, "gemm: tile 62 needs k, cin, c1 %% 64 == 0, no GEGLU");
      return launch_glds<1, 4, 5, 1, 2, 0, 32>(a, s);
    // 64-row-per-wave register tile (fewer LDS reads per MFMA), K step 32 so that two blocks still fit a CU
    case 64:
      MVOC_REQUIRE(glds_ok && d->act != MVOC_ACT_GEGLU, -2, "gemm: tile 64 needs k, cin, c1 %% 64 == 0, no GEGLU");
      return launch_glds<1, 4, 5, 2, 2, 0, 32>(a, s);  // 160 x 256
    default: mvoc_set_error("gemm: unknown tile %d", tile); return -1;
  }
}
