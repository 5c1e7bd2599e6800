// GroupNorm(+SiLU) and LayerNorm on channels-last rows (HBM-bound; 16-byte loads along C, deterministic).
//
// GroupNorm: a "sample" is rows_per_sample consecutive rows (one image = H*W rows for the 4-D norms, one video =
// F*H*W rows for the 5-D norms inside TemporalConvLayer / TransformerTemporalModel).  Three launches:
//   gn_partial : grid (nchunk, nsample) -- each block reduces a slab of rows to per-group (count, mean, M2)
//   gn_final   : grid (nsample)         -- Chan-combines the slabs in fixed order -> (mean, rstd) per group
//   gn_apply   : grid (nchunk, nsample) -- y = silu?( (x-mean)*rstd*gamma + beta ), fp16-rounded like the
//                                          reference's group_norm -> silu op pair
// The input may be a channel concat of two tensors (decoder skip connections): the concat is only ever
// materialised as the *normalised* output.
#include "common.h"

namespace {

struct GnArgs {
  const half_t* x;
  const half_t* x2;
  const half_t* gamma;
  const half_t* beta;
  half_t* out;
  float* ws;  // [nsample][nchunk][G][3] partials, then [nsample][G][2] finals
  float* mom_out;  // when set, gn_final writes the sample's raw (count, mean, M2) here instead of (mean, rstd)
  // statistics from the producers' epilogues (mvoc_gemm_desc.chan_sums): per 256-row slab and channel {sum, sum of squares} of
  // the rows of x (x2); gn_partial is skipped and gn_final folds the channels of a group itself
  const float* sums;
  const float* sums2;
  int nslab;       // 256-row slabs per sample
  int inline_final;  // gn_apply finalises from the sums itself (few slabs, groups divide 256): no gn_final launch
  int nsample, R, c, c1, c2, G, cpg, silu;
  int nchunk, rows_per_chunk;
  int CW, RY, npass;
  int aCW, aRY, anpass;  // gn_apply's own thread geometry (no LDS limit on rows per block: all 256 threads get a channel chunk)
  float eps;
};

__device__ __forceinline__ const half_t* gn_src(const GnArgs& p, long row, int ch) {
  return ch < p.c1 ? p.x + row * p.c1 + ch : p.x2 + row * p.c2 + (ch - p.c1);
}

__global__ __launch_bounds__(256) void gn_partial(const GnArgs p) {
  __shared__ float lsum[2560], lsq[2560];
  const int tid = threadIdx.x;
  const int cx = tid % p.CW, ry = tid / p.CW;
  const int smp = blockIdx.y, chunk = blockIdx.x;
  const int r0 = chunk * p.rows_per_chunk;
  const int r1 = min(r0 + p.rows_per_chunk, p.R);
  const long rowbase = (long)smp * p.R;
  for (int pass = 0; pass < p.npass; ++pass) {
    const int cc = pass * p.CW + cx;
    const bool on = ry < p.RY && cc * 8 < p.c;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    if (on) {
      // four 16-byte loads in flight per thread (one was not enough to cover HBM latency: 1.7 TB/s); the accumulation
      // order per thread is unchanged, so the statistics are bit-identical to the rolled loop
      int rr = r0 + ry;
      for (; rr + 3 * p.RY < r1; rr += 4 * p.RY) {
        half8_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8_t*>(gn_src(p, rowbase + rr + u * p.RY, cc * 8));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = (float)v[u][e];
            s[e] += f;
            q[e] += f * f;
          }
      }
      for (; rr < r1; rr += p.RY) {
        const half8_t v = *reinterpret_cast<const half8_t*>(gn_src(p, rowbase + rr, cc * 8));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)v[e];
          s[e] += f;
          q[e] += f * f;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        lsum[ry * p.CW * 8 * p.npass + cc * 8 + e] = s[e];
        lsq[ry * p.CW * 8 * p.npass + cc * 8 + e] = q[e];
      }
    }
  }
  __syncthreads();
  if (tid < p.G) {
    float s = 0.f, q = 0.f;
    const int stride = p.CW * 8 * p.npass;
    for (int y = 0; y < p.RY; ++y)
      for (int ch = tid * p.cpg; ch < (tid + 1) * p.cpg; ++ch) {
        s += lsum[y * stride + ch];
        q += lsq[y * stride + ch];
      }
    const float n = (float)(r1 - r0) * p.cpg;
    const float mean = n > 0 ? s / n : 0.f;
    float* o = p.ws + (((long)smp * p.nchunk + chunk) * p.G + tid) * 3;
    o[0] = n;
    o[1] = mean;
    o[2] = fmaxf(q - s * mean, 0.f);
  }
}

// Chan-combine the slab partials of one sample (deterministic: fixed slab->lane assignment, fixed merge tree).
// 1024 threads: lane = tid / G walks slabs lane, lane+L, ...; the loads of 8 slabs are issued before any of them is
// consumed (the merge arithmetic is a dependent chain, the loads are not), then the lanes merge pairwise in LDS.
struct GnMoments { float n, mean, m2; };
__device__ __forceinline__ void gn_merge(GnMoments& a, float nb, float mb, float m2b) {
  if (nb <= 0.f) return;
  const float nt = a.n + nb, delta = mb - a.mean;
  a.mean += delta * (nb / nt);
  a.m2 += m2b + delta * delta * (a.n * nb / nt);
  a.n = nt;
}

// one (slab, group) of the producers' channel sums -> (n, mean, M2): channels g cpg .. of one source (a group never straddles
// the two sources: c1 % cpg == 0, checked by the host)
__device__ __forceinline__ void gn_fold_slab(const GnArgs& p, int smp, int g, int sl, float& n, float& mean, float& m2) {
  const int ch0 = g * p.cpg;
  const bool second = ch0 >= p.c1;
  const int cs = second ? p.c2 : p.c1;
  const float* in = (second ? p.sums2 : p.sums) + (((long)smp * p.nslab + sl) * cs + (second ? ch0 - p.c1 : ch0)) * 2;
  float sx = 0.f, sq = 0.f;
  int c = 0;
  if ((p.cpg & 1) == 0) {
    // two channels per 16-byte load (even cpg: the group's run starts 16-byte aligned), five loads in flight: as a rolled loop
    // of dependent 8-byte loads this fold was a chain of cpg L2 round trips per slab (gn_final 6 -> 9 us)
    for (; c + 10 <= p.cpg; c += 10) {
      float4 v[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) v[u] = *reinterpret_cast<const float4*>(in + 2 * c + 4 * u);
#pragma unroll
      for (int u = 0; u < 5; ++u) { sx += v[u].x; sq += v[u].y; sx += v[u].z; sq += v[u].w; }
    }
  }
  for (; c < p.cpg; ++c) {
    const float2 v = *reinterpret_cast<const float2*>(in + 2 * c);
    sx += v.x;
    sq += v.y;
  }
  n = 256.f * (float)p.cpg;
  mean = sx / n;
  m2 = fmaxf(sq - sx * mean, 0.f);
}

__global__ __launch_bounds__(1024) void gn_final(const GnArgs p) {
  __shared__ float ln[1024], lmean[1024], lm2[1024];
  // grid (nsample, gridDim.y): block y merges groups [y*GL, (y+1)*GL).  With few samples (the 5-D norms of a B = 1 step:
  // ONE sample, up to 1024 slabs) a single block was a serial chain of four dependent load rounds (11 us); four blocks
  // of 8 groups give every group 128 lanes -> one load round + a 7-level tree.
  const int GL = p.G / gridDim.y;
  const int tid = threadIdx.x, smp = blockIdx.x;
  int lanes = 1;
  while (lanes * 2 * GL <= 1024) lanes *= 2;  // power of two
  const int gl = tid % GL, lane = tid / GL;
  const int g = blockIdx.y * GL + gl;
  GnMoments acc = {0.f, 0.f, 0.f};
  if (p.sums) {
    if (lane < lanes) {
      for (int sl = lane; sl < p.nslab; sl += lanes) {
        float n, mean, m2;
        gn_fold_slab(p, smp, g, sl, n, mean, m2);
        gn_merge(acc, n, mean, m2);
      }
    }
  } else if (lane < lanes) {
    const float* base = p.ws + ((long)smp * p.nchunk * p.G + g) * 3;
    for (int c0 = lane; c0 < p.nchunk; c0 += lanes * 8) {
      float nb[8], mb[8], qb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + u * lanes;
        nb[u] = 0.f; mb[u] = 0.f; qb[u] = 0.f;
        if (c < p.nchunk) {
          const float* in = base + (long)c * p.G * 3;
          nb[u] = in[0]; mb[u] = in[1]; qb[u] = in[2];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) gn_merge(acc, nb[u], mb[u], qb[u]);
    }
  }
  ln[tid] = acc.n; lmean[tid] = acc.mean; lm2[tid] = acc.m2;
  __syncthreads();
  for (int half = lanes / 2; half >= 1; half /= 2) {
    if (lane < half) {
      const int o = (lane + half) * GL + gl;
      gn_merge(acc, ln[o], lmean[o], lm2[o]);
      ln[tid] = acc.n; lmean[tid] = acc.mean; lm2[tid] = acc.m2;
    }
    __syncthreads();
  }
  if (lane == 0) {
    if (p.mom_out) {
      float* mo = p.mom_out + ((long)smp * p.G + g) * 3;
      mo[0] = acc.n; mo[1] = acc.mean; mo[2] = acc.m2;
      return;
    }
    float* fin = p.ws + (long)p.nsample * p.nchunk * p.G * 3 + ((long)smp * p.G + g) * 2;
    fin[0] = acc.mean;
    fin[1] = rsqrtf(acc.m2 / acc.n + p.eps);
  }
}

// Statistics of a sample that is spread over several devices (pixel- or frame-sharded 5-D GroupNorm): every rank
// contributes one (count, mean, M2) triple per (sample, group); they are Chan-combined in rank order, so every rank
// derives bit-identical (mean, rstd).
__global__ __launch_bounds__(256) void gn_merge_parts(const float* __restrict__ parts, int nparts, int ng, float eps,
                                                      float* __restrict__ fin) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ng) return;
  GnMoments acc = {0.f, 0.f, 0.f};
  for (int r = 0; r < nparts; ++r) {
    const float* in = parts + ((long)r * ng + i) * 3;
    gn_merge(acc, in[0], in[1], in[2]);
  }
  fin[i * 2] = acc.mean;
  fin[i * 2 + 1] = rsqrtf(acc.m2 / acc.n + eps);
}

// NT: streaming (non-temporal) loads of the rows -- the rate of a copy on a tensor far larger than the caches (210 MB at the
// finest level of a batch-5 step: 94 -> 76 us, tools/dbg/gn_bench.py), 10-15 % SLOWER on tensors the Infinity Cache still holds
// from their producer: the host picks by size (mvoc_groupnorm_f16)
template <bool NT>
__global__ __launch_bounds__(256) void gn_apply(const GnArgs p) {
  const int tid = threadIdx.x;
  const int cx = tid % p.aCW, ry = tid / p.aCW;
  const int smp = blockIdx.y, chunk = blockIdx.x;
  const int r0 = chunk * p.rows_per_chunk;
  const int r1 = min(r0 + p.rows_per_chunk, p.R);
  const long rowbase = (long)smp * p.R;
  const float* fin = p.ws + (long)p.nsample * p.nchunk * p.G * 3 + (long)smp * p.G * 2;
  __shared__ float lfin[2 * 256 + 3 * 256];
  if (p.inline_final) {
    // few slabs per sample (the 4-D norms: 16 / 4 / 1): every block finalises its sample's statistics itself from the
    // producers' channel sums -- ~20 L2 loads per thread against a gn_final launch (6-9 us at B = 1).  Same arithmetic in every
    // block of the sample: identical statistics.  Thread t folds slabs (t / G) + k (256 / G) of group t % G, G threads merge.
    const int g = tid % p.G, part = tid / p.G, parts = 256 / p.G;
    GnMoments acc = {0.f, 0.f, 0.f};
    for (int sl = part; sl < p.nslab; sl += parts) {
      float n, mean, m2;
      gn_fold_slab(p, smp, g, sl, n, mean, m2);
      gn_merge(acc, n, mean, m2);
    }
    float* lm = lfin + 2 * 256;
    lm[tid * 3] = acc.n; lm[tid * 3 + 1] = acc.mean; lm[tid * 3 + 2] = acc.m2;
    __syncthreads();
    if (tid < p.G) {
      GnMoments t = {0.f, 0.f, 0.f};
      for (int k = 0; k < parts; ++k) gn_merge(t, lm[(k * p.G + tid) * 3], lm[(k * p.G + tid) * 3 + 1], lm[(k * p.G + tid) * 3 + 2]);
      lfin[tid * 2] = t.mean;
      lfin[tid * 2 + 1] = rsqrtf(t.m2 / t.n + p.eps);
    }
    __syncthreads();
    fin = lfin;
  }
  if (ry >= p.aRY) return;
  for (int pass = 0; pass < p.anpass; ++pass) {
    const int cc = pass * p.aCW + cx;
    if (cc * 8 >= p.c) continue;
    float sc[8], sh[8];
    const half8_t gm = *reinterpret_cast<const half8_t*>(p.gamma + cc * 8);
    const half8_t bt = *reinterpret_cast<const half8_t*>(p.beta + cc * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (cc * 8 + e) / p.cpg;
      const float mean = fin[g * 2], rstd = fin[g * 2 + 1];
      sc[e] = rstd * (float)gm[e];
      sh[e] = (float)bt[e] - mean * sc[e];
    }
    int rr = r0 + ry;
    for (; rr + 3 * p.aRY < r1; rr += 4 * p.aRY) {  // four loads in flight per thread
      half8_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const half8_t* src = reinterpret_cast<const half8_t*>(gn_src(p, rowbase + rr + u * p.aRY, cc * 8));
        v[u] = NT ? __builtin_nontemporal_load(src) : *src;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float y = r16((float)v[u][e] * sc[e] + sh[e]);
          if (p.silu) y = silu_f(y);
          o[e] = (half_t)y;
        }
        *reinterpret_cast<half8_t*>(p.out + (rowbase + rr + u * p.aRY) * p.c + cc * 8) = o;
      }
    }
    for (; rr < r1; rr += p.aRY) {
      const half8_t v = *reinterpret_cast<const half8_t*>(gn_src(p, rowbase + rr, cc * 8));
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = r16((float)v[e] * sc[e] + sh[e]);
        if (p.silu) y = silu_f(y);
        o[e] = (half_t)y;
      }
      *reinterpret_cast<half8_t*>(p.out + (rowbase + rr) * p.c + cc * 8) = o;
    }
  }
}


// GroupNorm folded into the LINEAR that reads it (GN -> proj_in of the two transformer kinds: pnp_utils.py:185-191, 433-438 --
// no activation in between): y = W (gamma (x - mean) rstd + beta) + b = (W gamma rstd) x + (b + W (beta - gamma mean rstd)).  Per
// SAMPLE (mean / rstd are per sample and group) this kernel writes the scaled weights W'_s, in mvoc_xs_linear_f16's packed
// fragment order, and the constants c'_s into the stream's last piece; the consumer reads the RAW rows with its sample's set
// (mvoc_xs_desc.wp_set_rows) and the normalised tensor -- a read and a write of the whole activation -- never exists.
// grid (n / 32 tiles, nsample), 256 threads; W [n][k] fp16 row-major, k % 16 == 0.
__global__ __launch_bounds__(256) void gn_fold_xs_kernel(const GnArgs p, const half_t* __restrict__ w, const half_t* __restrict__ bias,
                                                         int n, int k, half_t* __restrict__ wp) {
  const int tile = blockIdx.x, smp = blockIdx.y, tid = threadIdx.x;
  const int nk = k / 16, np = nk + 1;
  const float* fin = p.ws + (long)p.nsample * p.nchunk * p.G * 3 + (long)smp * p.G * 2;
  half_t* out = wp + ((long)smp * (n / 32) + tile) * np * 512;
  // scaled fragments: chunk q = (piece s, lane l) holds W'[32 tile + (l & 31)][16 s + 8 (l >> 5) .. + 7]
  for (int q = tid; q < nk * 64; q += 256) {
    const int s_ = q >> 6, l = q & 63;
    const int row = 32 * tile + (l & 31), col = 16 * s_ + 8 * (l >> 5);
    const half8_t wv = *reinterpret_cast<const half8_t*>(w + (long)row * k + col);
    const half8_t gv = *reinterpret_cast<const half8_t*>(p.gamma + col);
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)wv[e] * ((float)gv[e] * fin[2 * ((col + e) / p.cpg) + 1]));
    *reinterpret_cast<half8_t*>(out + (long)q * 8) = o;
  }
  // constants: 8 threads per row, each a fixed eighth of the columns, summed in lane order
  const int r = tid >> 3, part = tid & 7;
  const int row = 32 * tile + r;
  float acc = 0.f;
  for (int col = part * 8; col < k; col += 64) {
    const half8_t wv = *reinterpret_cast<const half8_t*>(w + (long)row * k + col);
    const half8_t gv = *reinterpret_cast<const half8_t*>(p.gamma + col);
    const half8_t bv = *reinterpret_cast<const half8_t*>(p.beta + col);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // the mean term from the weight AS ROUNDED above (the consumer computes sum fl16(W') x + c'): then y = sum W' (x - mean) + W beta
      // + b and the rounding error of W' scales with |x - mean|, not with |mean| (a group whose mean is many sigma off zero)
      const float* f = fin + 2 * ((col + e) / p.cpg);
      const float wr = (float)(half_t)((float)wv[e] * ((float)gv[e] * f[1]));
      acc += (float)wv[e] * (float)bv[e] - wr * f[0];
    }
  }
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  acc += __shfl_xor(acc, 4);
  if (part == 0) reinterpret_cast<float*>(out + (long)nk * 512)[r] = acc + (bias ? (float)bias[row] : 0.f);
}

// group-blocks per sample for gn_final: split the groups over 4 blocks when there are few samples and many slabs
int gn_final_blocks(const GnArgs& a) {
  const int n = a.sums ? a.nslab : a.nchunk;
  return (a.nsample < 16 && n > 64 && a.G % 4 == 0 && a.G >= 8) ? 4 : 1;
}

void gn_geometry(GnArgs& a) {
  const int cchunks = a.c / 8;
  a.CW = cchunks < 256 ? cchunks : 256;
  a.RY = 256 / a.CW;
  a.npass = (cchunks + a.CW - 1) / a.CW;
  // gn_apply walks the same geometry.  (Its own, with all 256 threads busy -- 1 280 channels as 80 chunks x 3 rows in two passes
  // instead of 160 x 1 with 96 idle threads -- was SLOWER: 37.1 against 31.8 us at 20 480 x 1 280, whole 2 560-byte rows per
  // thread-row beat two passes over 1 280-byte halves.)
  a.aCW = a.CW; a.aRY = a.RY; a.anpass = a.npass;
  int want = 2048 / (a.nsample > 0 ? a.nsample : 1);
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  int maxchunk = (a.R + a.RY * 4 - 1) / (a.RY * 4);  // at least 4 rows per thread-row
  if (maxchunk < 1) maxchunk = 1;
  a.nchunk = want < maxchunk ? want : maxchunk;
  a.rows_per_chunk = (a.R + a.nchunk - 1) / a.nchunk;
  a.nchunk = (a.R + a.rows_per_chunk - 1) / a.rows_per_chunk;
}

// ---- LayerNorm ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_kernel(const half_t* __restrict__ x, const half_t* __restrict__ gamma,
                                                 const half_t* __restrict__ beta, half_t* __restrict__ out, long rows,
                                                 int c, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int nch = c / 8;
  const half_t* xr = x + row * c;
  half8_t v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cc = lane + 64 * i;
    if (cc < nch) {
      v[i] = *reinterpret_cast<const half8_t*>(xr + cc * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)v[i][e];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / c;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cc = lane + 64 * i;
    if (cc < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)v[i][e] - mean;
        q += d * d;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / c + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cc = lane + 64 * i;
    if (cc < nch) {
      const half8_t gm = *reinterpret_cast<const half8_t*>(gamma + cc * 8);
      const half8_t bt = *reinterpret_cast<const half8_t*>(beta + cc * 8);
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)(((float)v[i][e] - mean) * rstd * (float)gm[e] + (float)bt[e]);
      *reinterpret_cast<half8_t*>(out + row * c + cc * 8) = o;
    }
  }
}

// per-row mean / rstd only (LayerNorm folded into the consumer GEMM): one wave per row, 16-byte loads
__global__ __launch_bounds__(256) void row_stats_kernel(const half_t* __restrict__ x, float* __restrict__ stats, long rows,
                                                        int c, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int nch = c / 8;
  const half_t* xr = x + row * c;
  // two passes (mean, then squared deviations) like F.layer_norm over the row held in registers: 4 x 16 bytes per lane, i.e.
  // c <= 2048 -- checked by the host, there is no wider form (include/mvoc_hip.h states the limit)
  half8_t v[4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cc = lane + 64 * i;
    if (cc < nch) {
      v[i] = *reinterpret_cast<const half8_t*>(xr + cc * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s1 += (float)v[i][e];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s1 += __shfl_xor(s1, o);
  const float mean = s1 / c;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cc = lane + 64 * i;
    if (cc < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dv = (float)v[i][e] - mean;
        s2 += dv * dv;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rsqrtf(s2 / c + eps);
  }
}

}  // namespace

// row softmax, in place (the score matrix of the VAE's single-head mid-block attention, head_dim 512): one wave per row,
// fp32 max / exp / sum, one rounding to fp16
namespace {
__global__ __launch_bounds__(256) void softmax_rows_kernel(half_t* __restrict__ x, long rows, int cols) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  half_t* xr = x + row * cols;
  const int nch = cols / 8;
  float mx = -3.0e38f;
  for (int c = lane; c < nch; c += 64) {
    const half8_t v = *reinterpret_cast<const half8_t*>(xr + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)v[e]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int c = lane; c < nch; c += 64) {
    const half8_t v = *reinterpret_cast<const half8_t*>(xr + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += __expf((float)v[e] - mx);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.0f / sum;
  for (int c = lane; c < nch; c += 64) {
    half8_t v = *reinterpret_cast<const half8_t*>(xr + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (half_t)(__expf((float)v[e] - mx) * inv);
    *reinterpret_cast<half8_t*>(xr + c * 8) = v;
  }
}
}  // namespace

extern "C" int mvoc_softmax_rows_f16(void* x, int64_t rows, int32_t cols, void* stream) {
  MVOC_REQUIRE(x && rows > 0 && cols > 0 && cols % 8 == 0 && ((uintptr_t)x & 15) == 0, -1, "softmax_rows: bad args (cols %% 8, 16-byte aligned)");
  hipStream_t s = (hipStream_t)stream;
  MvocProfScope prof(MVOC_FAM_MISC, s, 4.0 * rows * cols);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (half_t*)x, (long)rows, cols);
  return mvoc_check_launch("softmax_rows_kernel");
}

extern "C" int mvoc_row_stats_f16(const void* x, void* stats, int64_t rows, int32_t c, float eps, void* stream) {
  MVOC_REQUIRE(x && stats && rows > 0 && c >= 8 && c % 8 == 0 && c <= 2048, -1, "row_stats: bad args (c %% 8 == 0, c <= 2048)");
  hipStream_t s = (hipStream_t)stream;
  MvocProfScope prof(MVOC_FAM_LN, s, 2.0 * (double)rows * c);
  const long nblk = (rows + 3) / 4;
  MVOC_REQUIRE(nblk < 0x7fffffffL, -2, "row_stats: too many rows");
  hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)nblk), dim3(256), 0, s, (const half_t*)x, (float*)stats, (long)rows, c, eps);
  return mvoc_check_launch("row_stats_kernel");
}

namespace {
// {sum, sum of squares} of a row's n-tiles (a GEMM epilogue's row_moments) -> the row's {mean, rstd}: per tile {count, mean, M2},
// Chan-merged in tile order (the sum of squares of a tile is a sum of exact fp32 products; the subtraction happens per tile of
// <= 320 channels, the tiles meet through their means)
__global__ __launch_bounds__(256) void row_stats_from_moments_kernel(const float* __restrict__ mom, float* __restrict__ stats, long rows,
                                                                     int ld, int n, int tile_w, float eps) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const float2* in = reinterpret_cast<const float2*>(mom) + r * ld;
  float cnt = 0.f, mean = 0.f, m2 = 0.f;
  for (int t = 0; t * tile_w < n; ++t) {
    const float2 v = in[t];
    const float nb = (float)min(tile_w, n - t * tile_w);
    const float mb = v.x / nb;
    const float m2b = fmaxf(v.y - v.x * mb, 0.f);
    const float nt = cnt + nb, delta = mb - mean;
    mean += delta * (nb / nt);
    m2 += m2b + delta * delta * (cnt * nb / nt);
    cnt = nt;
  }
  reinterpret_cast<float2*>(stats)[r] = float2{mean, rsqrtf(m2 / cnt + eps)};
}
}  // namespace

extern "C" int mvoc_row_stats_from_moments_f32(const void* moments, int64_t rows, int32_t ld, int32_t n, int32_t tile_w, float eps,
                                               void* out_stats, void* stream) {
  MVOC_REQUIRE(moments && out_stats && rows > 0 && n > 0 && (tile_w == 256 || tile_w == 320) && ld >= (n + tile_w - 1) / tile_w, -1,
               "row_stats_from_moments: bad args (tile_w 256 | 320, ld >= ceil(n / tile_w))");
  hipStream_t s = (hipStream_t)stream;
  MvocProfScope prof(MVOC_FAM_LN, s, (double)rows * (8.0 * ((n + tile_w - 1) / tile_w) + 8.0));
  const long nblk = (rows + 255) / 256;
  MVOC_REQUIRE(nblk < 0x7fffffffL, -2, "row_stats_from_moments: too many rows");
  hipLaunchKernelGGL(row_stats_from_moments_kernel, dim3((unsigned)nblk), dim3(256), 0, s, (const float*)moments, (float*)out_stats,
                     (long)rows, ld, n, tile_w, eps);
  return mvoc_check_launch("row_stats_from_moments_kernel");
}

extern "C" size_t mvoc_groupnorm_workspace_bytes(int32_t nsample, int32_t rows_per_sample, int32_t c, int32_t groups) {
  GnArgs a;
  memset(&a, 0, sizeof(a));
  a.nsample = nsample; a.R = rows_per_sample; a.c = c; a.G = groups;
  if (nsample <= 0 || rows_per_sample <= 0 || c < 8 || groups <= 0) return 0;
  gn_geometry(a);
  return ((size_t)nsample * a.nchunk * groups * 3 + (size_t)nsample * groups * 2) * sizeof(float);
}

namespace {
// shared argument validation of the three GroupNorm entry points; fills a (geometry included)
int gn_setup(const mvoc_gn_desc* d, bool need_affine, GnArgs& a) {
  MVOC_REQUIRE(d && d->x && d->workspace && (!need_affine || (d->gamma && d->beta && d->out)), -1, "groupnorm: null operand");
  MVOC_REQUIRE(d->nsample > 0 && d->rows_per_sample > 0, -1, "groupnorm: empty problem");
  MVOC_REQUIRE(d->c % 8 == 0 && d->c1 % 8 == 0 && d->c <= 2560 && d->groups > 0 && d->groups <= 256 &&
                   d->c % d->groups == 0,
               -2, "groupnorm: unsupported channels %d (c1 %d) / groups %d", d->c, d->c1, d->groups);
  MVOC_REQUIRE(d->x2 || d->c1 == d->c, -1, "groupnorm: c1 < c needs a second source");
  MVOC_REQUIRE(d->nsample <= 65535, -2, "groupnorm: too many samples");
  memset(&a, 0, sizeof(a));
  a.x = (const half_t*)d->x; a.x2 = (const half_t*)d->x2; a.gamma = (const half_t*)d->gamma;
  a.beta = (const half_t*)d->beta; a.out = (half_t*)d->out; a.ws = (float*)d->workspace;
  a.nsample = d->nsample; a.R = d->rows_per_sample; a.c = d->c; a.c1 = d->c1; a.c2 = d->c - d->c1;
  a.G = d->groups; a.cpg = d->c / d->groups; a.silu = d->silu; a.eps = d->eps;
  gn_geometry(a);
  const size_t need = mvoc_groupnorm_workspace_bytes(d->nsample, d->rows_per_sample, d->c, d->groups);
  MVOC_REQUIRE(d->workspace_bytes >= need, -1, "groupnorm: workspace %zu < %zu bytes", d->workspace_bytes, need);
  return 0;
}
}  // namespace

extern "C" int mvoc_groupnorm_moments_f16(const mvoc_gn_desc* d, void* moments, void* stream) {
  GnArgs a;
  if (int rc = gn_setup(d, false, a)) return rc;
  MVOC_REQUIRE(moments, -1, "groupnorm_moments: null output");
  a.mom_out = (float*)moments;
  hipStream_t s = (hipStream_t)stream;
  MvocProfScope prof(MVOC_FAM_GN, s, 2.0 * (double)d->nsample * d->rows_per_sample * d->c);
  hipLaunchKernelGGL(gn_partial, dim3(a.nchunk, a.nsample), dim3(256), 0, s, a);
  hipLaunchKernelGGL(gn_final, dim3(a.nsample, gn_final_blocks(a)), dim3(1024), 0, s, a);
  return mvoc_check_launch("groupnorm_moments");
}

extern "C" int mvoc_groupnorm_apply_moments_f16(const mvoc_gn_desc* d, const void* parts, int32_t nparts, void* stream) {
  GnArgs a;
  if (int rc = gn_setup(d, true, a)) return rc;
  MVOC_REQUIRE(parts && nparts > 0, -1, "groupnorm_apply_moments: no statistics");
  hipStream_t s = (hipStream_t)stream;
  MvocProfScope prof(MVOC_FAM_GN, s, 2.0 * 2.0 * (double)d->nsample * d->rows_per_sample * d->c);
  const int ng = a.nsample * a.G;
  float* fin = a.ws + (long)a.nsample * a.nchunk * a.G * 3;
  hipLaunchKernelGGL(gn_merge_parts, dim3((ng + 255) / 256), dim3(256), 0, s, (const float*)parts, nparts, ng, a.eps, fin);
  hipLaunchKernelGGL(gn_apply<false>, dim3(a.nchunk, a.nsample), dim3(256), 0, s, a);
  return mvoc_check_launch("groupnorm_apply_moments");
}

extern "C" int mvoc_groupnorm_f16(const mvoc_gn_desc* d, void* stream) {
  GnArgs a;
  if (int rc = gn_setup(d, true, a)) return rc;
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: read x once for stats + once for apply is the implementation; compulsory = read + write
  MvocProfScope prof(MVOC_FAM_GN, s, 2.0 * 2.0 * (double)d->nsample * d->rows_per_sample * d->c);
  dim3 grid(a.nchunk, a.nsample);
  if (d->chan_sums) {
    MVOC_REQUIRE(d->rows_per_sample % 256 == 0 && (!d->x2 || d->chan_sums2) && a.c1 % a.cpg == 0, -2,
                 "groupnorm: chan_sums needs rows_per_sample %% 256 == 0, sums of both sources, and groups that do not straddle them");
    a.sums = (const float*)d->chan_sums;
    a.sums2 = (const float*)d->chan_sums2;
    a.nslab = d->rows_per_sample / 256;
    // every block of the sample re-reads the sample's sums (nslab x C x 8 bytes, out of L2): worth a launch less only while that
    // is small beside the rows a block streams -- 16 slabs x 1 280 channels = 164 KB per block against ~25 KB of rows was a loss
    // (tools/dbg/gn_bench.py: 47.9 us against 41.1 with its own statistics pass at B = 5)
    static const long inline_max = getenv("MVOC_GN_INLINE_BYTES") ? atol(getenv("MVOC_GN_INLINE_BYTES")) : 49152;
    a.inline_final = (long)a.nslab * a.c * 8 <= inline_max && a.nslab <= 16 && 256 % a.G == 0;
  } else {
    hipLaunchKernelGGL(gn_partial, grid, dim3(256), 0, s, a);
  }
  if (!a.inline_final) hipLaunchKernelGGL(gn_final, dim3(a.nsample, gn_final_blocks(a)), dim3(1024), 0, s, a);
  static const long nt_min = getenv("MVOC_GN_NT_BYTES") ? atol(getenv("MVOC_GN_NT_BYTES")) : (160L << 20);
  if ((long)d->nsample * d->rows_per_sample * d->c * 2 >= nt_min) hipLaunchKernelGGL(gn_apply<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(gn_apply<false>, grid, dim3(256), 0, s, a);
  return mvoc_check_launch("groupnorm");
}


extern "C" int mvoc_groupnorm_fold_xs_f16(const mvoc_gn_desc* d, const void* w, const void* bias, int32_t n, int32_t k, void* wp_sets,
                                          void* stream) {
  GnArgs a;
  if (int rc = gn_setup(d, false, a)) return rc;
  MVOC_REQUIRE(w && wp_sets && d->gamma && d->beta, -1, "groupnorm_fold_xs: null operand");
  MVOC_REQUIRE(d->x2 == nullptr && k == d->c && n > 0 && n % 32 == 0 && k % 16 == 0 && a.cpg >= 1, -2,
               "groupnorm_fold_xs: single source, k == c, n %% 32 == 0, k %% 16 == 0");
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: the statistics' read of the rows when no producer sums exist, the weight sets
  MvocProfScope prof(MVOC_FAM_GN, s, (d->chan_sums ? 0.0 : 2.0 * (double)d->nsample * d->rows_per_sample * d->c) +
                                         2.0 * 2.0 * (double)d->nsample * n * k);
  if (d->chan_sums) {
    MVOC_REQUIRE(d->rows_per_sample % 256 == 0, -2, "groupnorm_fold_xs: chan_sums needs rows_per_sample %% 256 == 0");
    a.sums = (const float*)d->chan_sums;
    a.nslab = d->rows_per_sample / 256;
  } else {
    hipLaunchKernelGGL(gn_partial, dim3(a.nchunk, a.nsample), dim3(256), 0, s, a);
  }
  hipLaunchKernelGGL(gn_final, dim3(a.nsample, gn_final_blocks(a)), dim3(1024), 0, s, a);
  hipLaunchKernelGGL(gn_fold_xs_kernel, dim3(n / 32, a.nsample), dim3(256), 0, s, a, (const half_t*)w, (const half_t*)bias, (int)n,
                     (int)k, (half_t*)wp_sets);
  return mvoc_check_launch("groupnorm_fold_xs");
}

extern "C" int mvoc_layernorm_f16(const void* x, const void* gamma, const void* beta, void* out, int64_t rows, int32_t c,
                                  float eps, void* stream) {
  MVOC_REQUIRE(x && gamma && beta && out, -1, "layernorm: null operand");
  MVOC_REQUIRE(rows > 0 && c >= 8 && c % 8 == 0 && c <= 2048, -2, "layernorm: unsupported c %d", c);
  hipStream_t s = (hipStream_t)stream;
  MvocProfScope prof(MVOC_FAM_LN, s, 2.0 * 2.0 * (double)rows * c);
  const long nblk = (rows + 3) / 4;
  MVOC_REQUIRE(nblk < 0x7fffffffL, -2, "layernorm: too many rows");
  hipLaunchKernelGGL(ln_kernel, dim3((unsigned)nblk), dim3(256), 0, s, (const half_t*)x, (const half_t*)gamma,
                     (const half_t*)beta, (half_t*)out, (long)rows, c, eps);
  return mvoc_check_launch("ln_kernel");
}
