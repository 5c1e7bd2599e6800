// Attention kernels for gfx950, head_dim 64, fp16 in/out, fp32 softmax + accumulate.
//
// flash_kernel  -- spatial self-attention (T up to 14400) and image/text cross-attention (145 keys):
//   block = 4 waves, each wave owns 32 query rows (Q fragments live in registers for the whole kernel);
//   K/V tiles of 64 keys are staged global -> registers -> LDS (loads of tile t+1 are in flight while tile t is
//   computed).  Scores are computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x16_f16 with K as the row operand),
//   so each lane holds the scores of ONE query (col = lane&31) -> row max / row sum are in-lane plus one
//   cross-half shuffle, and the fp16-packed P^T accumulator registers are directly the column operand of
//   O^T = V^T P^T.  V is staged ROW-major ([key][d], one ds_write_b128 per 16 bytes loaded) and its V^T fragments come
//   from gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group fetches 4 keys x 16 d and each lane receives
//   one d column): the accumulator layout's k order is two runs of 4 consecutive keys (32t + 16s + 4h + {0..3}, + 8), so
//   a fragment is two such reads.  (Writing V transposed with 2-byte stores cost 16 ds_write_b16 per lane and tile:
//   +4 % on the 4096-token self-attention, same box.)  K: pitch 144 B, conflict-free b128 reads; V: pitch 192 B, the 4
//   rows of a transposed read fall into 4 disjoint 64-byte bank windows.
//
// tattn_kernel  -- temporal self-attention: one wave per (sample, pixel, head); the sequence is the frame axis
//   (<= 32 frames), rows are strided by H*W*C in the canonical layout so no [B*HW, F, C] copy is ever made.
//   Same transposed-score scheme with a single 32x32 tile; V^T goes through a 4 KB wave-private LDS image.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
// flash kernel LDS pitches for head dim D: K rows of 2D bytes + 16 (an odd number of 16-byte units: conflict-free b128 reads);
// V rows row-major with a pitch == 64 or 192 (mod 256): the 4 rows of a transposed read hit 4 disjoint 64-byte bank windows
template <int D> struct FlashPitch { static constexpr int K = 2 * D + 16, V = D == 64 ? 192 : 320; };

struct AttnArgs {
  const half_t *q, *k, *v;
  half_t* out;
  long q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
  int nbatch, heads, tq, tk, kv_bdiv, causal;
  float scale_log2;
  long v2_off, o2_off;  // NV == 2: element offsets of the pair's second V / output (same Q and K)
  int qblocks;          // 128-query blocks per (batch entry, head)
};

__device__ __forceinline__ int vt_slot_group(int key) {  // 16-byte group (8 slots) of a key inside its 32-key tile
  return ((((key >> 5) * 2 + ((key >> 4) & 1)) * 2) + ((key >> 2) & 1));
}
__device__ __forceinline__ int vt_slot_elem(int key) { return ((key >> 3) & 1) * 4 + (key & 3); }

// The running maximum is DEFERRED: the accumulators are rescaled only when some query's maximum has grown by more than 2^FLASH_THR
// since the last rescale (wave-uniform test), so P = exp2(c (s - m)) may reach 2^8 instead of 1 -- exact in fp16's range, same
// relative rounding, fp32 sums.  With the plain "did any maximum grow" test 32 queries per wave rescaled on ~65 % of the 64 tiles
// of a 4096-key row (32 multiplies + an exponential per lane each time); with the threshold on the first tile or two only.
constexpr float FLASH_THR = 8.0f;

typedef float f32x2 __attribute__((ext_vector_type(2)));
// max of a value with the other half-wave's (lane ^ 32): v_permlane32_swap exchanges the upper half of one register with the lower
// half of another in the vector pipe (the ds_bpermute form of __shfl_xor is an LDS round trip on the per-tile critical path)
__device__ __forceinline__ float flash_xhalf_max(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
}

// NV = 2: TWO value tensors attend with the same Q and K (the PnP destination pair: pnp_utils.py:664-668 writes one blended
// q / k into both the unconditional and the conditional chunk, so softmax(q k^T) is computed once and multiplied into both V's;
// per tile 16 more MFMAs instead of a second S^T + softmax + PV pass).  Outputs are bit-identical to two NV = 1 launches.
template <int D, int NV>  // head dim: 64 (the UNet), 96 (CLIP ViT-H's 80, zero-padded by the projection weights)
// head dim 64, one V: 128 registers = four waves per SIMD (130 without the bound; +2 % at T = 4 096, +4-6 % at 1 024 / 256, same box)
#ifndef FLASH_MINW
#define FLASH_MINW 4
#endif
__global__ __launch_bounds__(256, (D == 64 && NV == 1) ? FLASH_MINW : 1) void flash_kernel(const AttnArgs p) {
  constexpr int KPITCH = FlashPitch<D>::K, VPITCH = FlashPitch<D>::V;
  constexpr int ND = D / 16, NT = D / 32, CPK = D / 8, NCH = 64 * CPK / 256;  // k steps, output tiles, 16-byte chunks per key / thread
  __shared__ __attribute__((aligned(16))) char smem[64 * KPITCH + NV * 64 * VPITCH];
  char* Ks = smem;
  char* Vs = smem + 64 * KPITCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: q0 below costs no vector register across the loop)
  const int r = lane & 31, h = lane >> 5;
  // 1-D grid, XCD-aware: the query blocks of one (batch entry, head) share its K / V (1 MB at 4 096 keys) and get consecutive
  // logical ids, i.e. one XCD's L2 (round-robin dispatch: a 3-D grid spread them over all eight and every XCD fetched the K / V
  // of every (entry, head) for itself)
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned qb_ = logical % (unsigned)p.qblocks, hb_ = logical / (unsigned)p.qblocks;
  const int head = (int)(hb_ % (unsigned)p.heads), b = (int)(hb_ / (unsigned)p.heads);
  const int q0 = (int)qb_ * 128 + wave * 32;

  const half_t* qp = p.q + (long)b * p.q_bs + head * D;
  const int qrow = min(q0 + r, p.tq - 1);
  half8_t qf[ND];
#pragma unroll
  for (int s = 0; s < ND; ++s) qf[s] = *reinterpret_cast<const half8_t*>(qp + (long)qrow * p.q_ts + 16 * s + 8 * h);

  // K / V rows of this (entry, head): a wave-uniform base (advanced per tile on the scalar unit) + a per-lane 32-bit byte offset
  // computed once.  Rows past the last key (ragged last tile) are CLAMPED to the last row instead of zero-filled: their scores are
  // masked to -inf below, so their probabilities are exactly 0 and 0 x (a finite V row) contributes exactly 0.
  const char* kb = reinterpret_cast<const char*>(p.k + (long)(b / p.kv_bdiv) * p.k_bs + head * D);
  const char* vb = reinterpret_cast<const char*>(p.v + (long)(b / p.kv_bdiv) * p.v_bs + head * D);
  const unsigned k_rb = (unsigned)p.k_ts * 2u, v_rb = (unsigned)p.v_ts * 2u;  // row pitches in bytes
  unsigned koff[NCH], voff[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * 256;
    koff[i] = (unsigned)(c / CPK) * k_rb + (unsigned)(c % CPK) * 16u;
    voff[i] = (unsigned)(c / CPK) * v_rb + (unsigned)(c % CPK) * 16u;
  }

  f32x16 ot[NV][NT];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int dt = 0; dt < NT; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) ot[v][dt][e] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;

  const int ntiles = (p.tk + 63) / 64;
  half8_t kreg[NCH], vreg[NV][NCH];
  auto gload = [&](int kt) {
    const char* kt_ = kb + (size_t)kt * 64u * k_rb;
    const char* vt_ = vb + (size_t)kt * 64u * v_rb;
    const bool ragged = __builtin_amdgcn_readfirstlane(kt * 64 + 64 > p.tk);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      unsigned ko = koff[i], vo = voff[i];
      if (ragged) {
        const int c = tid + i * 256;
        const unsigned row = (unsigned)min(c / CPK, p.tk - 1 - kt * 64);
        ko = row * k_rb + (unsigned)(c % CPK) * 16u;
        vo = row * v_rb + (unsigned)(c % CPK) * 16u;
      }
      kreg[i] = *reinterpret_cast<const half8_t*>(kt_ + ko);
#pragma unroll
      for (int v = 0; v < NV; ++v) vreg[v][i] = *reinterpret_cast<const half8_t*>(vt_ + (size_t)v * (size_t)p.v2_off * 2u + vo);
    }
  };
  gload(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();  // every wave is done reading the previous tile
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * 256;
      const int key = c / CPK, dc = c % CPK;
      *reinterpret_cast<half8_t*>(Ks + key * KPITCH + dc * 16) = kreg[i];
#pragma unroll
      for (int v = 0; v < NV; ++v)
        *reinterpret_cast<half8_t*>(Vs + v * 64 * VPITCH + key * VPITCH + dc * 16) = vreg[v][i];  // row-major; transposed on the read (below)
    }
    __syncthreads();
    if (kt + 1 < ntiles) gload(kt + 1);

    // S^T = K Q^T : st[t][reg] = S[query r][key 32t + 8(reg>>2) + 4h + (reg&3)]
    f32x16 st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) st[t][e] = 0.f;
#pragma unroll
      for (int s = 0; s < ND; ++s) {
        const half8_t kf = *reinterpret_cast<const half8_t*>(Ks + (32 * t + r) * KPITCH + (16 * s + 8 * h) * 2);
        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], st[t], 0, 0, 0);
      }
    }
    // online softmax on the raw scores; exp2(c*s - c*m) is one v_fma + one raw v_exp_f32 per element
    if (__builtin_amdgcn_readfirstlane((kt == ntiles - 1) && (p.tk & 63))) {  // ragged last tile: mask keys >= tk
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (kt * 64 + 32 * t + 8 * (e >> 2) + 4 * h + (e & 3) >= p.tk) st[t][e] = -INFINITY;
    }
    if (p.causal) {  // CLIP text tower: a query sees keys <= itself (key 0 is always visible, so the running maximum stays finite)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (kt * 64 + 32 * t + 8 * (e >> 2) + 4 * h + (e & 3) > q0 + r) st[t][e] = -INFINITY;
    }
    float mx = st[0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, st[t][e]);
    mx = flash_xhalf_max(mx);
    if (__any((mx - mrun) * p.scale_log2 > FLASH_THR)) {  // wave-uniform; always on the first tile (mrun = -inf)
      const float mnew = fmaxf(mrun, mx);
      const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * p.scale_log2);
      mrun = mnew;
      lrun *= alpha;
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int dt = 0; dt < NT; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) ot[v][dt][e] *= alpha;
    }
    // exponent arguments two at a time (v_pk_fma_f32: the same fp32 fma per element, half the instructions)
    const f32x2 sc2 = {p.scale_log2, p.scale_log2}, mc2 = {-mrun * p.scale_log2, -mrun * p.scale_log2};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const f32x2 a = __builtin_elementwise_fma(f32x2{st[t][e], st[t][e + 1]}, sc2, mc2);
        st[t][e] = __builtin_amdgcn_exp2f(a[0]);
        st[t][e + 1] = __builtin_amdgcn_exp2f(a[1]);
      }

    // O^T += V^T P^T : accumulator registers 8s..8s+7 of tile t are the column operand of k-step (t, s)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        half8_t pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (half_t)st[t][8 * s + j];
        // row sum from the fp16-rounded probabilities that enter P V (numerator and denominator see the same values): one
        // v_dot2_f32_f16 per pair, fp32 accumulation -- 16 instructions per tile where the fp32 adds were 32 (the kernel is bound
        // by its vector-instruction issue, not by the matrix pipe)
#pragma unroll
        for (int j = 0; j < 4; ++j) lrun = __builtin_amdgcn_fdot2(half2_t{pf[2 * j], pf[2 * j + 1]}, half2_t{(half_t)1.f, (half_t)1.f}, lrun, false);
        // V^T fragment by the hardware transpose read: each 16-lane group fetches a block of 4 keys x 16 d of the
        // row-major image (lane 4q+p supplies row q, columns 4p..4p+3) and lane i receives column i of the 4 rows.  The
        // accumulator's k order is two runs of 4 consecutive keys (32t + 16s + 4h + {0..3} and + 8): two reads.
        const int krow = 32 * t + 16 * s + 4 * h + ((lane & 15) >> 2);
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
          for (int dt = 0; dt < NT; ++dt) {
            const char* va = Vs + v * 64 * VPITCH + krow * VPITCH + (32 * dt + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
            const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)va);
            const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(va + 8 * VPITCH));
            half8_t vf;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              vf[e] = (half_t)lo[e];
              vf[4 + e] = (half_t)hi[e];
            }
            ot[v][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, ot[v][dt], 0, 0, 0);
          }
      }
  }
  const float ltot = lrun + __shfl_xor(lrun, 32);
  const float inv = 1.0f / ltot;
  // (the store addresses are formed from a lane index made opaque HERE: computed ahead of the loop, r and h cost registers across it
  // -- the ones that did not fit four waves per SIMD)
  int lane_e = tid;
  asm volatile("" : "+v"(lane_e));
  const int r_e = lane_e & 31, h_e = (lane_e >> 5) & 1;
  if (q0 + r_e < p.tq) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      half_t* op = p.out + v * p.o2_off + (long)b * p.o_bs + (long)(q0 + r_e) * p.o_ts + head * D;
#pragma unroll
      for (int dt = 0; dt < NT; ++dt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[v][dt][q * 4 + e] * inv);
          *reinterpret_cast<half4_t*>(op + 32 * dt + 8 * q + 4 * h_e) = o;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// flash3_kernel -- the same attention (head dim 64, no causal mask) SOFTWARE-PIPELINED inside each wave, for long key rows.
// tools/lab/overlap.hip: one wave that places ~7 vector instructions between consecutive MFMAs runs 16 MFMAs + 128 vector
// instructions in the time of the 16 MFMAs alone + 25 %; flash_kernel's wave runs them in PHASES (S^T = 8 MFMAs behind their
// fragment reads, then ~100 vector instructions of softmax with no MFMA in flight, then P V) and leaves the overlap to the other
// waves of the SIMD: ~2 900 ticks per tile and wave for 512 matrix cycles.  Here:
// * the tile loop is skewed by one tile: iteration t computes S^T of tile t + 1 (independent of everything else in the iteration)
//   WHILE it takes the softmax of tile t, and O^T += V^T P^T of tile t as the probability fragments appear; two score accumulators
//   (the loop is unrolled by two so they swap roles statically);
// * the instruction order is dictated to hipcc's scheduler (sched_group_barrier pipelines): one MFMA, a handful of vector /
//   transcendental instructions, the next MFMA ... -- its own order is all MFMAs first;
// * K / V tiles arrive by LDS-DMA (buffer_load ... lds, 1 KB pieces = 8 keys x 128 B, two K + two V pieces per wave and tile)
//   into a ring of NS stages, NS - 1 tiles ahead, behind a counted s_waitcnt vmcnt and ONE raw s_barrier per tile: no staging
//   registers, no ds_write pass, no address arithmetic in the loop (per-lane offsets are loop constants, the tile advance is the
//   instruction's scalar offset; rows past the last key are outside the buffer resource and read zeros).  Issued unconditionally,
//   also past the last tile (out of range: zeros, no traffic), so the wait count is one constant.
// * LDS images are unpadded (a DMA piece is contiguous): K rows 128 B with 16-byte chunk c of key r at c ^ ((r >> 1) & 7)
//   (conflict-free b128 operand reads, as in the GEMM kernels); V rows 128 B with the two 64-byte halves of key r swapped when
//   (r >> 1) & 1 -- the four keys of a transposing read then fall into four disjoint 64-byte bank windows.  Both swizzles are
//   applied to the DMA's per-lane source offset and again on the read.
// Same fragment maps, rounding points, summation order and deferred rescale as flash_kernel: outputs are bit-identical to it.
#define F3_BAR()                         \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
#define F3_FENCE() __builtin_amdgcn_sched_barrier(0)
// scheduling groups (IGroupLP): one MFMA, then V plain vector and T transcendental instructions; R LDS reads
#define F3_GRP(V, T)                                   \
  do {                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x002, V, 0); \
    __builtin_amdgcn_sched_group_barrier(0x400, T, 0); \
  } while (0)
#define F3_RD(R) __builtin_amdgcn_sched_group_barrier(0x100, R, 0)
#ifdef MVOC_F3_STAMPS  // diagnostic build (tools/dbg/f3_stamps.py): s_memtime accumulated per phase, block 0 wave 0
__device__ unsigned long long f3_dbg[8];
#define F3_STAMP(i)                                                                      \
  do {                                                                                   \
    unsigned long long t_;                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
    if (blockIdx.x == 0 && tid == 0) { f3_acc[i] += t_ - f3_last; }                      \
    f3_last = t_;                                                                        \
  } while (0)
#else
#define F3_STAMP(i)
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// One LDS-DMA piece (64 lanes x 16 B -> 1 KB at the wave-uniform LDS byte address `lds`), as an asm statement: hipcc makes every
// transposing LDS read (ds_read_b64_tr_b16 has no memory operand it can disambiguate) wait vmcnt(0) for the LDS-DMA it knows to be
// in flight -- the ring would drain on every tile (measured: 600 ns per tile).  Hidden here, the pieces are counted by the
// kernel's own s_waitcnt vmcnt(N) only.  No VGPR destination: register-safe; M0 is saved, set and restored inside the statement.
__device__ __forceinline__ void f3_dma(const u32x4 rsrc, unsigned voff, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds), "s"(rsrc) : "memory");
}
// buffer resource words for a raw buffer of `bytes` bytes at `base` (wave-uniform): range-checked, no swizzle
__device__ __forceinline__ u32x4 f3_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}
template <int N>
__device__ __forceinline__ void f3_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// waves per SIMD of the one-V form: 3 (168 registers, ring of three stages: 811-845 TFLOP/s at 4 096 keys) against 2 (ring of
// four: 772-798) -- profiles/r4/flash3_ab.txt; -DF3_W1=2 rebuilds the other for A/B (tools/dbg/attn_ab.sh)
#ifndef F3_W1
#define F3_W1 3
#endif
template <int NV>
__global__ __launch_bounds__(256, NV == 1 ? F3_W1 : 2) void flash3_kernel(const AttnArgs p) {
  constexpr int D = 64, NS = (NV == 1 && F3_W1 == 2) ? 4 : 3, TILE = 8192 * (1 + NV), PCS = 2 + 2 * NV;  // ring stages, bytes per stage, DMA pieces per wave and tile
  __shared__ __attribute__((aligned(1024))) char smem[NS * TILE];
#ifdef MVOC_F3_STAMPS
  unsigned long long f3_acc[6] = {0, 0, 0, 0, 0, 0}, f3_last = 0, f3_t0, f3_r0;
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
#ifdef MVOC_F3_STAMPS
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(f3_t0), "=s"(f3_r0)::"memory");
  f3_last = f3_t0;
#endif
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned qb_ = logical % (unsigned)p.qblocks, hb_ = logical / (unsigned)p.qblocks;
  const int head = (int)(hb_ % (unsigned)p.heads), b = (int)(hb_ / (unsigned)p.heads);
  const int q0 = (int)qb_ * 128 + wave * 32;
  const int nt = (p.tk + 63) >> 6;

  // ---- LDS-DMA sources: this wave stages keys 16 wave .. 16 wave + 15 of every tile (two pieces per tensor) -------------------
  const unsigned k_rb = (unsigned)p.k_ts * 2u, v_rb = (unsigned)p.v_ts * 2u;  // row pitches in bytes
  const half_t* kb = p.k + (long)(b / p.kv_bdiv) * p.k_bs + head * D;
  const half_t* vb = p.v + (long)(b / p.kv_bdiv) * p.v_bs + head * D;
  const u32x4 rs_k = f3_rsrc(kb, (unsigned)(p.tk - 1) * k_rb + 128u);
  const u32x4 rs_v0 = f3_rsrc(vb, (unsigned)(p.tk - 1) * v_rb + 128u);
  const u32x4 rs_v1 = f3_rsrc(vb + p.v2_off, (unsigned)(p.tk - 1) * v_rb + 128u);
  unsigned ksrc[2], vsrc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int key_l = 16 * wave + 8 * j + (lane >> 3), pc = lane & 7;
    ksrc[j] = (unsigned)key_l * k_rb + (unsigned)((pc ^ ((key_l >> 1) & 7)) * 16);
    vsrc[j] = (unsigned)key_l * v_rb + (unsigned)((pc ^ (((key_l >> 1) & 1) << 2)) * 16);
  }
  const unsigned ktile = 64u * k_rb, vtile = 64u * v_rb;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto issue = [&](int tile, int so) {  // tile -> stage at byte offset so (wave-uniform)
    const unsigned base = lds0 + (unsigned)so + (unsigned)wave * 2048u;
    // (the tile advance goes into the VECTOR offset: the range check that zero-fills keys >= tk must see it)
#pragma unroll
    for (int j = 0; j < 2; ++j) f3_dma(rs_k, ksrc[j] + (unsigned)tile * ktile, base + j * 1024);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int j = 0; j < 2; ++j) f3_dma(v ? rs_v1 : rs_v0, vsrc[j] + (unsigned)tile * vtile, base + 8192 * (1 + v) + j * 1024);
  };
  const half_t* qp = p.q + (long)b * p.q_bs + head * D;
  const int qrow = min(q0 + r, p.tq - 1);
  half8_t qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const half8_t*>(qp + (long)qrow * p.q_ts + 16 * s + 8 * h);
  // (the resource words come from v_readfirstlane: a vector-written SGPR needs five wait states before a buffer instruction reads
  // it as its descriptor, and hipcc pads nothing inside an asm statement)
  asm volatile("s_nop 4" ::: "memory");
  F3_FENCE();
  // tiles 0 .. NS - 2 go out (behind the Q loads: wherever hipcc places those, the counted wait below is on the safe side)
#pragma unroll
  for (int t = 0; t < NS - 1; ++t) issue(t, t * TILE);
  F3_FENCE();

  f32x16 ot[NV][2];
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) ot[v][dt][e] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;
  f32x16 sa[2], sb[2];
  f32x16 zero16;
#pragma unroll
  for (int e = 0; e < 16; ++e) zero16[e] = 0.f;

  // lane-constant parts of the fragment addresses inside a stage
  int kfo[4];  // K operand of k-step s: key row r (+ 32 tt), chunk (2 s + h) ^ swizzle
#pragma unroll
  for (int s = 0; s < 4; ++s) kfo[s] = r * 128 + (((2 * s + h) ^ ((r >> 1) & 7)) * 16);
  const int vq = (lane & 15) >> 2;  // key of this lane inside a 4-key transposed read
  int vfo[2];  // V^T fragment of output tile dt: key row 4 h + vq (+ 32 tt + 16 s, + 8), 64-byte half dt ^ swizzle
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) vfo[dt] = 8192 + (4 * h + vq) * 128 + (((32 * ((lane >> 4) & 1) + 8 * (lane & 3))) | ((dt ^ ((vq >> 1) & 1)) << 6));

  // ---- prologue: tile 0 landed, S^T of tile 0 --------------------------------------------------------------------------------
  f3_wait_vm<(NS - 2) * PCS>();
  F3_BAR();
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    sa[t] = zero16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const half8_t kf = *reinterpret_cast<const half8_t*>(smem + kfo[s] + 4096 * t);
      sa[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sa[t], 0, 0, 0);
    }
  }
  int o_v = 0, o_k = TILE, o_w = (NS - 1) * TILE;  // stages of tile t (V read), t + 1 (K read), t + NS - 1 (issued)

  // one tile: softmax + P V of the tile whose scores are in `cur`, S^T of the next tile into `nxt`
  auto body = [&](f32x16(&cur)[2], f32x16(&nxt)[2], const int t) {
    F3_STAMP(4);
    // tile t + 1 complete for this wave (tiles t + 2 .. t + NS - 2 may be in flight), then for every wave; the barrier also
    // retires every wave's reads of the stage about to be refilled (tile t - 1's)
    f3_wait_vm<(NS - 3) * PCS>();
    F3_BAR();
    F3_STAMP(0);
    issue(t + NS - 1, o_w);
    F3_FENCE();
    F3_STAMP(1);
    const char* Kn = smem + o_k;
    const char* Vc = smem + o_v;
    if (__builtin_amdgcn_readfirstlane((t == nt - 1) && (p.tk & 63))) {  // ragged last tile: mask keys >= tk
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (t * 64 + 32 * tt + 8 * (e >> 2) + 4 * h + (e & 3) >= p.tk) cur[tt][e] = -INFINITY;
    }
    F3_FENCE();
    // K fragments of tile t + 1 (on the last iteration: zeros or a stale stage, the scores are never used)
    half8_t kf[2][4];  // (k-steps 0, 1 now; 2, 3 in the second region: 16 registers less across the row maximum)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) kf[tt][s] = *reinterpret_cast<const half8_t*>(Kn + kfo[s] + 4096 * tt);
    half8_t pf[2];
    half8_t vf[3][NV][2];  // three fragment sets: the reads of k-step f + 1 are issued while P V of f - 1 is still to come
    auto vread = [&](const int f) {  // V^T fragments of k-step f = (tt, s) = (f >> 1, f & 1): two transposing reads each
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* va = Vc + 8192 * v + vfo[dt] + (32 * (f >> 1) + 16 * (f & 1)) * 128;
          const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)va);
          const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(va + 8 * 128));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            vf[f % 3][v][dt][e] = (half_t)lo[e];
            vf[f % 3][v][dt][4 + e] = (half_t)hi[e];
          }
        }
    };
    vread(0);
    F3_FENCE();
    auto smma = [&](const int i) {  // the i-th S^T MFMA of the next tile: (tt, s) = (i & 1, i >> 1)
      const int tt = i & 1, s_ = i >> 1;
      nxt[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[tt][s_], qf[s_], s_ == 0 ? zero16 : nxt[tt], 0, 0, 0);
    };
    auto pvmma = [&](const int f, const int dt) {  // O^T += V^T P^T, k-step f, output tile dt (both value tensors)
#pragma unroll
      for (int v = 0; v < NV; ++v) ot[v][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[f % 3][v][dt], pf[f & 1], ot[v][dt], 0, 0, 0);
    };
    // ---- row maximum of tile t with the first S^T MFMAs --------------------------------------------------------------------
    float mx = fmaxf(cur[0][0], cur[0][1]);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int e = tt ? 0 : 2; e < 16; e += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, cur[tt][e]), cur[tt][e + 1]);
    smma(0); smma(1); smma(2);
    if constexpr (NV == 2) smma(3);
    mx = flash_xhalf_max(mx);
    // instruction order of this region: one MFMA, then a few vector instructions (hipcc's own order is all MFMAs first)
    if constexpr (NV == 1) { F3_GRP(6, 0); F3_GRP(6, 0); F3_GRP(6, 0); }
    else { F3_GRP(4, 0); F3_GRP(5, 0); F3_GRP(4, 0); F3_GRP(5, 0); }
    F3_STAMP(2);
    if (__any((mx - mrun) * p.scale_log2 > FLASH_THR)) {  // wave-uniform; always on the first tile (mrun = -inf)
      const float mnew = fmaxf(mrun, mx);
      const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * p.scale_log2);
      mrun = mnew;
      lrun *= alpha;
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) ot[v][dt][e] *= alpha;
    }
    F3_FENCE();
    const f32x2 sc2 = {p.scale_log2, p.scale_log2}, mc2 = {-mrun * p.scale_log2, -mrun * p.scale_log2};
#pragma unroll
    for (int s = 2; s < 4; ++s)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) kf[tt][s] = *reinterpret_cast<const half8_t*>(Kn + kfo[s] + 4096 * tt);
    // ---- probabilities fragment by fragment (f = k-step (tt, s)): P V of fragment f - 1 and the rest of S^T run under the
    //      exponentials of fragment f ------------------------------------------------------------------------------------------
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int tt = f >> 1, s_ = f & 1;
      if (f < 3) vread(f + 1);
      half8_t& pn = pf[f & 1];
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const f32x2 a = __builtin_elementwise_fma(f32x2{cur[tt][8 * s_ + j], cur[tt][8 * s_ + j + 1]}, sc2, mc2);
        x[j] = __builtin_amdgcn_exp2f(a[0]);
        x[j + 1] = __builtin_amdgcn_exp2f(a[1]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) pn[j] = (half_t)x[j];
      // row sum from the fp16-rounded probabilities that enter P V (same order as flash_kernel: bit-identical sums)
#pragma unroll
      for (int j = 0; j < 4; ++j) lrun = __builtin_amdgcn_fdot2(half2_t{pn[2 * j], pn[2 * j + 1]}, half2_t{(half_t)1.f, (half_t)1.f}, lrun, false);
      if (f == 0) {
        smma(NV == 1 ? 3 : 4); smma(NV == 1 ? 4 : 5);
        if constexpr (NV == 2) { smma(6); smma(7); } else smma(5);
      } else {
        pvmma(f - 1, 0); pvmma(f - 1, 1);
        if constexpr (NV == 1) { if (f < 3) smma(5 + f); }
      }
    }
    pvmma(3, 0);
    pvmma(3, 1);
    // region order: the transposing reads of the next fragment, then MFMA : 5 vector : 2-3 transcendental, repeated
    if constexpr (NV == 1) {
      F3_RD(8); F3_GRP(5, 3); F3_GRP(5, 2); F3_GRP(5, 3);
      F3_RD(4); F3_GRP(5, 2); F3_GRP(5, 3); F3_GRP(5, 2);
      F3_RD(4); F3_GRP(5, 3); F3_GRP(5, 2); F3_GRP(5, 3);
      F3_GRP(5, 2); F3_GRP(5, 3); F3_GRP(5, 2); F3_GRP(5, 2);
    } else {
      F3_RD(12); F3_GRP(3, 2); F3_GRP(3, 2); F3_GRP(3, 2); F3_GRP(3, 2);
      F3_RD(8); F3_GRP(3, 2); F3_GRP(3, 2); F3_GRP(3, 2); F3_GRP(3, 2);
      F3_RD(8); F3_GRP(3, 2); F3_GRP(3, 2); F3_GRP(3, 2); F3_GRP(3, 2);
      F3_GRP(3, 2); F3_GRP(3, 2); F3_GRP(3, 2); F3_GRP(3, 2);
      F3_GRP(3, 0); F3_GRP(3, 0); F3_GRP(3, 0); F3_GRP(3, 0);
    }
    F3_FENCE();
    F3_STAMP(3);
    const int o_ = o_v; o_v = o_k;
    o_k = o_k + TILE == NS * TILE ? 0 : o_k + TILE;
    o_w = o_;
  };
#pragma unroll 1
  for (int t = 0;; t += 2) {
    body(sa, sb, t);
    if (t + 1 >= nt) break;
    body(sb, sa, t + 1);
    if (t + 2 >= nt) break;
  }
  f3_wait_vm<0>();  // the pieces issued past the last tile (nothing reads them; the LDS must outlive them)
#ifdef MVOC_F3_STAMPS
  if (blockIdx.x == 0 && tid == 0) {
    unsigned long long t1_, r1_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_), "=s"(r1_)::"memory");
    for (int i = 0; i < 5; ++i) f3_dbg[i] = f3_acc[i];
    f3_dbg[5] = t1_ - f3_t0;
    f3_dbg[7] = r1_ - f3_r0;
    f3_dbg[6] = (unsigned long long)nt;
  }
#endif

  const float ltot = lrun + __shfl_xor(lrun, 32);
  const float inv = 1.0f / ltot;
  if (q0 + r < p.tq) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      half_t* op = p.out + v * p.o2_off + (long)b * p.o_bs + (long)(q0 + r) * p.o_ts + head * D;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)(ot[v][dt][q * 4 + e] * inv);
          *reinterpret_cast<half4_t*>(op + 32 * dt + 8 * q + 4 * h) = o;
        }
    }
  }
}
#undef F3_BAR
#undef F3_FENCE
#undef F3_GRP
#undef F3_RD
#undef F3_STAMP

// ------------------------------------------------------------------------------------------------
struct TAttnArgs {
  const half_t *q, *k, *v;
  half_t* out;
  long q_bs, q_ps, q_ts, k_bs, k_ps, k_ts, v_bs, v_ps, v_ts, o_bs, o_ps, o_ts;
  int nsample, hw, heads, frames;
  int logf, logp, pgroups;  // frames == 1 << logf and 1 << logp pixels per 32-row tile when frames is 8 / 16 (else logp = 0); pixel groups per sample
  long ntask;
  float scale_log2;
};

constexpr int TVPITCH = 64;

__global__ __launch_bounds__(256) void tattn_kernel(const TAttnArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 64 * TVPITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  char* Vs = smem + wave * 64 * TVPITCH;
  // A 32-row tile holds 32 / F whole sequences when F is 8 or 16 (logp = log2 of the pixels per tile, set by the host; 0
  // otherwise): row r = (pixel r >> logf, frame r & (F - 1)), a row attends only to the rows of its own pixel -- every lane loads
  // and stores useful bytes (with one 16-frame sequence per tile half of each load instruction was masked off and the kernel
  // moved 3.2-3.5 TB/s).
  const int F = p.frames, logf = p.logf, P = 1 << p.logp;
  const long task = (long)blockIdx.x * 4 + wave;
  const bool active = task < p.ntask;  // whole wave uniform
  const long tk = active ? task : p.ntask - 1;
  const int head = (int)(tk % p.heads);
  const long bp = tk / p.heads;
  const long smp = bp / p.pgroups, px0 = (bp % p.pgroups) << p.logp;
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  // (row -> pixel, frame) of a tile row; rows past the last pixel / frame are dead
  auto row_px = [&](int row) { return (long)(P > 1 ? row >> logf : 0); };
  auto row_fr = [&](int row) { return P > 1 ? row & (F - 1) : row; };
  auto row_ok = [&](int row) { return (P > 1 || row < F) && px0 + row_px(row) < p.hw; };

  const half_t* qb = p.q + smp * p.q_bs + px0 * p.q_ps + head * 64;
  const half_t* kb = p.k + smp * p.k_bs + px0 * p.k_ps + head * 64;
  const half_t* vb = p.v + smp * p.v_bs + px0 * p.v_ps + head * 64;

  // V: 32 rows x 8 chunks = 256 chunks, 4 per lane, transposed into the wave-private LDS image
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    const int key = c >> 3, dc = c & 7;
    half8_t vr = zero8;
    if (row_ok(key)) vr = *reinterpret_cast<const half8_t*>(vb + row_px(key) * p.v_ps + (long)row_fr(key) * p.v_ts + dc * 8);
    const int g = (((key >> 4) & 1) * 2) + ((key >> 2) & 1), j = vt_slot_elem(key);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = dc * 8 + e;
      *reinterpret_cast<half_t*>(Vs + d * TVPITCH + ((g ^ ((d >> 3) & 3)) * 16) + j * 2) = vr[e];
    }
  }
  // S^T = K Q^T straight from global (row r, 16-byte pieces); dead rows are zero
  f32x16 st;
#pragma unroll
  for (int e = 0; e < 16; ++e) st[e] = 0.f;
  const bool rok = row_ok(r);
  const long koff = row_px(r) * p.k_ps + (long)row_fr(r) * p.k_ts, qoff = row_px(r) * p.q_ps + (long)row_fr(r) * p.q_ts;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    half8_t kf = zero8, qf = zero8;
    if (rok) {
      kf = *reinterpret_cast<const half8_t*>(kb + koff + 16 * s + 8 * h);
      qf = *reinterpret_cast<const half8_t*>(qb + qoff + 16 * s + 8 * h);
    }
    st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf, st, 0, 0, 0);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int key = 8 * (e >> 2) + 4 * h + (e & 3);
    // one tile, no running state: a key counts for this lane's query iff it is a live row of the same pixel
    const bool ok = P > 1 ? (key >> logf) == (r >> logf) : key < F;
    const float sv = ok ? st[e] * p.scale_log2 : -INFINITY;
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

  __syncthreads();  // V^T image written (wave-private, but orders the ds_write/ds_read for the compiler too)
  f32x16 ot[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) ot[dt][e] = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    half8_t pf;
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[j] = (half_t)st[8 * s + j];
    const int g = s * 2 + h;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int d = 32 * dt + r;
      const half8_t vf = *reinterpret_cast<const half8_t*>(Vs + d * TVPITCH + ((g ^ ((d >> 3) & 3)) * 16));
      ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, ot[dt], 0, 0, 0);
    }
  }
  if (active && rok) {
    half_t* op = p.out + smp * p.o_bs + (px0 + row_px(r)) * p.o_ps + (long)row_fr(r) * p.o_ts + head * 64;
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

}  // namespace

// load-time default of the kernel choice (diagnostics; read once, never written): MVOC_FLASH3 = 0 flash_kernel always, 1 flash3_kernel wherever
// it applies (head dim 64, no causal mask); unset: by key count.  Per call: mvoc_attn_desc.pipelined.
static const int g_f3_env = getenv("MVOC_FLASH3") ? atoi(getenv("MVOC_FLASH3")) : -1;

extern "C" int mvoc_flash_attn_f16(const mvoc_attn_desc* d, void* stream) {
  MVOC_REQUIRE(d && d->q && d->k && d->v && d->out, -1, "flash_attn: null operand");
  MVOC_REQUIRE(d->nbatch > 0 && d->heads > 0 && d->tq > 0 && d->tk > 0, -1, "flash_attn: empty problem");
  MVOC_REQUIRE(d->nbatch <= 65535 && d->heads <= 65535, -2, "flash_attn: grid too large");
  MVOC_REQUIRE(d->q_ts % 8 == 0 && d->k_ts % 8 == 0 && d->v_ts % 8 == 0 && d->o_ts % 4 == 0, -2,
               "flash_attn: row strides must keep 16-byte alignment");
  AttnArgs a;
  a.q = (const half_t*)d->q; a.k = (const half_t*)d->k; a.v = (const half_t*)d->v; a.out = (half_t*)d->out;
  a.q_bs = d->q_bs; a.q_ts = d->q_ts; a.k_bs = d->k_bs; a.k_ts = d->k_ts; a.v_bs = d->v_bs; a.v_ts = d->v_ts;
  a.o_bs = d->o_bs; a.o_ts = d->o_ts;
  a.nbatch = d->nbatch; a.heads = d->heads; a.tq = d->tq; a.tk = d->tk; a.kv_bdiv = d->kv_bdiv > 0 ? d->kv_bdiv : 1;
  const int hd = d->head_dim > 0 ? d->head_dim : 64;
  MVOC_REQUIRE(hd == 64 || hd == 96, -2, "flash_attn: head_dim (%d) must be 64 or 96", hd);
  MVOC_REQUIRE(!d->causal || d->tq == d->tk, -2, "flash_attn: the causal form is self-attention (tq == tk)");
  a.causal = d->causal;
  a.scale_log2 = (d->scale > 0.f ? d->scale : (hd == 64 ? 0.125f : 1.0f / sqrtf((float)hd))) * 1.4426950408889634f;
  hipStream_t s = (hipStream_t)stream;
  // (work = what the reference computes: the paired form stands for two attentions)
  MvocProfScope prof(MVOC_FAM_FLASH, s, (d->v2 ? 2.0 : 1.0) * 4.0 * d->nbatch * d->heads * (double)d->tq * d->tk * hd);
  a.qblocks = (d->tq + 127) / 128;
  const long nblk = (long)a.qblocks * d->heads * d->nbatch;
  MVOC_REQUIRE(nblk < 0x7fffffffL, -2, "flash_attn: grid too large");
  dim3 grid((unsigned)nblk);
  a.v2_off = a.o2_off = 0;
  // long key rows of the UNet's self-attention: the software-pipelined kernel (same results bit for bit; +1.5-4 % from 3 600 keys
  // up, -3 % at 1 024: same-box A/B, profiles/r4/flash3_ab.txt); mvoc_attn_desc.pipelined (per call) / MVOC_FLASH3=0 / 1 force one kernel
  MVOC_REQUIRE(d->pipelined >= 0 && d->pipelined <= 2, -1, "flash_attn: pipelined must be 0 (by key count), 1 (never) or 2 (wherever it applies)");
  const int f3_mode = d->pipelined ? d->pipelined - 1 : g_f3_env;
  const bool pipelined = hd == 64 && !d->causal && (f3_mode >= 0 ? f3_mode != 0 : d->tk >= 2048);
  if (d->v2) {
    MVOC_REQUIRE(d->out2 && hd == 64, -2, "flash_attn: the paired form needs out2 and head_dim 64");
    a.v2_off = (const half_t*)d->v2 - a.v;
    a.o2_off = (half_t*)d->out2 - a.out;
    MVOC_REQUIRE(a.v2_off % 8 == 0 && a.o2_off % 4 == 0, -2, "flash_attn: v2 / out2 must keep 16-byte alignment");
    if (pipelined) hipLaunchKernelGGL((flash3_kernel<2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((flash_kernel<64, 2>), grid, dim3(256), 0, s, a);
  } else if (hd == 64) {
    if (pipelined) hipLaunchKernelGGL((flash3_kernel<1>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((flash_kernel<64, 1>), grid, dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((flash_kernel<96, 1>), grid, dim3(256), 0, s, a);
  }
  return mvoc_check_launch("flash_kernel");
}

#ifdef MVOC_F3_STAMPS
extern "C" int mvoc_f3_stamps_read(unsigned long long* host8) {
  return hipMemcpyFromSymbol(host8, HIP_SYMBOL(f3_dbg), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int mvoc_temporal_attn_f16(const mvoc_tattn_desc* d, void* stream) {
  MVOC_REQUIRE(d && d->q && d->k && d->v && d->out, -1, "temporal_attn: null operand");
  MVOC_REQUIRE(d->nsample > 0 && d->hw > 0 && d->heads > 0, -1, "temporal_attn: empty problem");
  MVOC_REQUIRE(d->frames >= 1 && d->frames <= 32, -2, "temporal_attn: frames (%d) must be in [1, 32]", d->frames);
  MVOC_REQUIRE(d->q_ts % 8 == 0 && d->k_ts % 8 == 0 && d->v_ts % 8 == 0 && d->o_ts % 4 == 0 && d->q_ps % 8 == 0 &&
                   d->k_ps % 8 == 0 && d->v_ps % 8 == 0 && d->o_ps % 4 == 0,
               -2, "temporal_attn: strides must keep 16-byte alignment");
  TAttnArgs a;
  a.q = (const half_t*)d->q; a.k = (const half_t*)d->k; a.v = (const half_t*)d->v; a.out = (half_t*)d->out;
  a.q_bs = d->q_bs; a.q_ps = d->q_ps; a.q_ts = d->q_ts; a.k_bs = d->k_bs; a.k_ps = d->k_ps; a.k_ts = d->k_ts;
  a.v_bs = d->v_bs; a.v_ps = d->v_ps; a.v_ts = d->v_ts; a.o_bs = d->o_bs; a.o_ps = d->o_ps; a.o_ts = d->o_ts;
  a.nsample = d->nsample; a.hw = d->hw; a.heads = d->heads; a.frames = d->frames;
  a.logf = d->frames == 8 ? 3 : d->frames == 16 ? 4 : 5;
  a.logp = d->frames == 8 ? 2 : d->frames == 16 ? 1 : 0;
  a.pgroups = (d->hw + (1 << a.logp) - 1) >> a.logp;
  a.ntask = (long)d->nsample * a.pgroups * d->heads;
  a.scale_log2 = 0.125f * 1.4426950408889634f;
  const long nblk = (a.ntask + 3) / 4;
  MVOC_REQUIRE(nblk < 0x7fffffffL, -2, "temporal_attn: grid too large");
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: q, k, v read + out written once
  MvocProfScope prof(MVOC_FAM_TATTN, s, 4.0 * (double)d->nsample * d->hw * d->heads * d->frames * 64 * 2);
  hipLaunchKernelGGL(tattn_kernel, dim3((unsigned)nblk), dim3(256), 0, s, a);
  return mvoc_check_launch("tattn_kernel");
}
