// Eight-phase implicit GEMM for gfx950: the large launches of the network (linear / 3x3 conv / temporal conv with at
// least a chip's worth of 256-pixel tiles) run through this kernel.
//
//   out[m, n] = epilogue( sum_k A(m, k) * W[n, k] )      fp16 operands, fp32 accumulate (v_mfma_f32_16x16x32_f16)
//
// Same math, operand conventions, gathers (PLAIN / CONV3X3 / TEMPORAL3, two sources, stride, folded upsample) and the same
// fp16 rounding points in the epilogue as gemm.hip.  What differs is the structure around the MFMAs
// (cdna_hip_programming.md, "The 256^2 8-phase template", re-derived here for an implicit-GEMM gather):
//
// * One 8-wave block per CU computes 256 pixels x BX output channels (BX = 256 or 320); wave (wr, wc) owns channels
//   wr*BX/2 .. and pixels wc*64 ..: a 2 x 2 grid of QUADRANTS of (BX/4 channels) x (32 pixels), one quadrant per phase, 16 (20)
//   MFMAs of 16x16x32 each.  Weights are the MFMA row operand: a lane ends up with 4 consecutive channels of one pixel.
// * A K tile (64 deep) lives in LDS as four HALF-TILES in the order they are consumed: Yh0 (the pixel rows of every wave's
//   quadrant column 0), Xh0 (weight rows of quadrant row 0 of both wave groups), Yh1, Xh1.  Phase 1 reads Yh0 + Xh0 into
//   registers, phase 2 Yh1, phase 3 Xh1 (over Xh0's registers), phase 4 nothing (Y0's fragments are kept): a half-tile's LDS
//   slot is dead one or two phases after it was read, and every phase restages exactly one slot by LDS-DMA
//   (buffer_load ... lds, 1 KB per wave-instruction), THREE half-tiles ahead of the tile being multiplied.  The DMA stays in
//   flight across the raw s_barriers; the only wait is one counted s_waitcnt vmcnt per K tile (phase 4).
// * The two wave groups (wr = 0 / 1; one wave of each per SIMD) run the same program ONE BARRIER APART: while one group
//   issues its 16 MFMAs the other does its fragment reads and DMA issue.
// * LDS rows are 128 B (64 k), 16-byte chunk c of row r at position c ^ ((r >> 1) & 7): applied to the DMA's per-lane source
//   offset and to the fragment read (conflict-free for the 16x16x32 operand read: 16 rows x one chunk per 16-lane group).
// * LDS-DMA as MUBUF: one 32-bit byte offset per lane and staged row (recomputed only when the K position crosses a tap or
//   the second source), the K advance in the scalar offset, rows that must read zeros (padding taps, rows past M or N)
//   carry an offset beyond the resource's range -- the hardware range check returns zeros.  No per-issue vector ALU work.
// * Epilogue: arithmetic in the accumulator layout, fp16 result parked in a wave-private LDS tile, read back as 16-byte row
//   chunks, residual added, stored (same rounding points as gemm_epilogue_lds).
#include <stdlib.h>

#include <type_traits>

#include "gemm_args.h"

namespace {

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

// epilogue stores non-temporal: a block's 128 KB output tile leaves in a chip-wide burst while its CU idles (s_endpgm waits for the
// stores), and nobody re-reads it from this XCD's L2 (the consumer is another kernel, the tensor several times the L2): +7 % at
// K = 640, +2 % at K = 1280, +1 % on the job (profiles/r3/nt_store_ab.txt)
#ifndef G8_NT_STORE
#define G8_NT_STORE 1
#endif
constexpr unsigned G8_OOB = 0x80000000u;  // >= num_records of every resource: the load returns zeros
__device__ const uint4 g8_zero16 = {0u, 0u, 0u, 0u};

#define G8_BAR()                         \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

// n / d for n < 2^31 and a launch constant d as one multiply-high (g8_magic below): exact, see the derivation there
__device__ __forceinline__ unsigned g8_udiv(unsigned n, unsigned mg, int sh) { return sh < 0 ? n : __umulhi(n, mg) >> sh; }

template <int N>
__device__ __forceinline__ void g8_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// XT: 16-channel tiles per quadrant (4 -> 256 channels per block, 5 -> 320).
// RETAIN: keep Y0's fragments in registers for phase 4 (16 VGPRs); otherwise phase 4 re-reads them and the Yh0 half-tiles
//   rotate through THREE slots so that restaging tile t+2's Yh0 (phase 2 of tile t) never collides with that re-read.
// YP: the four activation row offsets live in registers and are rebuilt at tap / source changes; otherwise they are formed
//   at issue time from (row, tap mask) -- five VALU per piece, for the 320-wide tile whose register file is full.
// UPS: nearest-upsampled conv source (row not affine in the tap): own instantiation, the hot kernels carry no such code.
// AFF: the source row of a staged pixel is affine in its index (plain, temporal, and 3x3 stride-1 same-size convs:
//   row = m + (ky - 1) W + (kx - 1), validity in the tap masks): ONE row register serves the four staged rows of a lane.
#ifdef MVOC_G8_STAMPS  // diagnostic build (tools/lab): s_memtime at the phase boundaries of one block (0; a later round's: -DMVOC_G8_STAMP_BLOCK=n), waves 0 and 4
#ifndef MVOC_G8_STAMP_BLOCK
#define MVOC_G8_STAMP_BLOCK 0
#endif
__device__ unsigned long long g8_dbg[32];  // [0, 16): s_memtime (shader clock) stamps; [16, 32): s_memrealtime (100 MHz) beside them
#define G8_STAMP(i)                                                                                   \
  do {                                                                                                \
    if (blockIdx.x == MVOC_G8_STAMP_BLOCK && (wave == 0 || wave == 4) && lane == 0) {                 \
      unsigned long long t_, r_;                                                                      \
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory"); \
      g8_dbg[(wave >> 2) * 8 + (i)] = t_;                                                             \
      g8_dbg[16 + (wave >> 2) * 8 + (i)] = r_;                                                        \
    }                                                                                                 \
  } while (0)
#else
#define G8_STAMP(i) do { } while (0)
#endif

// EPI: the epilogue forms the UNet uses as their own instantiations -- 1 plain (bias / row-add / residual), 2 LayerNorm fold,
// 3 LayerNorm fold + GEGLU; 0 = everything, decided at run time.  The general epilogue is ~11 k instructions, executed once per
// block and mostly branched over: with the K loop the kernel does not fit the instruction cache, and a pass's arithmetic took
// 8-10 k ticks (phase stamps) -- as long as four K tiles -- for 64 values per lane; the plain form is 4 k instructions, 3 k ticks.
// KO: K runs (64-channel chunk, tap, 64) instead of (tap, channel) -- mvoc_gemm_desc.k_order = 1 (conv / temporal conv with host-repacked
//   weights): the tap changes with EVERY K tile, so a block's nine reads of a pixel row's 128-byte slab follow each other within nine
//   K tiles (~50 KB of other rows in between instead of a whole tap's cin x 256 rows) and hit the XCD's L2.  An instantiation of its
//   own, of the 320-wide plain-epilogue form only: that form builds its activation offsets at issue time anyway (!YP), so the tap
//   switch costs it nothing, while the 256-wide form pays 7-18 % for rebuilding its four offset registers per K tile
//   (profiles/r6/gemm_per_shape_pmc_korder.txt) -- and as a RUN-TIME branch in y_advance the switch cost every instantiation 1.5 %
//   (the K loop's code placement moved; profiles/r6/gemm8_round6_experiments.txt).
template <int XT, bool RETAIN, bool YP, bool UPS, bool AFF, int EPI, bool KO = false>
__global__ __launch_bounds__(512) void gemm8_kernel(const GemmArgs p) {
  constexpr int XQ = XT * 16;            // channels per quadrant (and per wave group per X half-tile)
  constexpr int XH = 2 * XQ;             // rows of an X half-tile
  constexpr int BX = 4 * XQ;             // channels per block
  constexpr int XB = XH * 128, YB = 128 * 128;
  constexpr int NY0 = RETAIN ? 2 : 3;    // Yh0 slots
  constexpr int KB = 2 * XB + YB;        // bytes of one K-tile buffer {Xh0, Yh1, Xh1}
  constexpr int OX0 = 0, OY1 = XB, OX1 = XB + YB;
  constexpr int OY0 = 2 * KB;            // Yh0 slots behind the two buffers
  constexpr int SMEM = 2 * KB + NY0 * YB;
  constexpr int XPC = XH / 8;            // 1 KB pieces per X half-tile: 16 / 20
  constexpr int PXM = (XPC + 7) / 8;     // per wave, at most
  // Epilogue vector table (bias fp16 [BX] | ln_s fp32 [BX] | ln_c fp32 [BX] | row-add fp16 [RA_ROWS][BX]; one 16-byte chunk per
  // thread, filled by ONE LDS-DMA per wave): behind the ring where the LDS has room (256-wide: requested in the prologue, landed
  // long before it is read), otherwise inside the dead ring behind the eight waves' epilogue tiles (320-wide: requested after
  // the K loop).
  constexpr int PITCH = XQ * 2 + 16;     // epilogue: 144 / 176 B per pixel row of a wave's fp16 tile
  constexpr int RA_ROWS = 5;             // rowadd_div >= 64 (host check): a 256-row tile touches at most 5 row-add rows
  constexpr int NB = BX / 8, NL = BX / 4, TOT = NB + 2 * NL + RA_ROWS * NB;  // 16-byte chunks of the table
  constexpr bool TBL_EARLY = SMEM + 8192 <= 163840;
  constexpr int T_B = TBL_EARLY ? SMEM : 8 * 64 * PITCH, T_LS = T_B + NB * 16, T_LC = T_LS + NL * 16, T_RA = T_LC + NL * 16;
  static_assert(TOT <= 512 && T_B + 8192 <= (TBL_EARLY ? SMEM + 8192 : SMEM), "epilogue table");
  static_assert(SMEM + (TBL_EARLY ? 8192 : 0) <= 163840, "LDS budget");
  static_assert(YP || !UPS, "the upsample form keeps its offsets in registers");
  static_assert(!(AFF && UPS), "an upsampled source is not affine");
  __shared__ __attribute__((aligned(1024))) char smem[SMEM + (TBL_EARLY ? 8192 : 0)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  G8_STAMP(0);
  const unsigned logical0 = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned logical = g8_udiv(logical0, p.mg_sk, p.sh_sk);
  const int slice = (int)(logical0 - logical * (unsigned)p.split_k);
  // Tile order inside an XCD's contiguous range of logical ids (xcd_remap): bands of `band` m-tiles, inside a band m fastest.
  // The 32 blocks an XCD runs at a time then form a band x (32 / band) patch of the tile grid and share band activation panels
  // and 32 / band weight panels through the XCD's L2, instead of 32 / n_tiles m-tiles x ALL n-tiles (n fastest: every m-tile of
  // a 20-n-tile GEGLU launch re-streamed all 6.5 MB of weights past the 4 MB L2 -- 715 MB fetched for 105 + 6.5 MB of operands,
  // profiles/r4/pmc_gemm_traffic.json).  band = 1 is the old order.
  unsigned mtile, ntile;
  if (p.band > 1) {
    const unsigned per = (unsigned)p.band * (unsigned)p.n_tiles;
    const unsigned b_ = g8_udiv(logical, p.mg_band, p.sh_band);
    const unsigned idx = logical - b_ * per;
    const unsigned rows = min((unsigned)p.band, (unsigned)p.m_tiles - b_ * (unsigned)p.band);
    ntile = idx / rows;  // (scalar; rows <= band)
    mtile = b_ * (unsigned)p.band + (idx - ntile * rows);
  } else {
    mtile = g8_udiv(logical, p.mg_nt, p.sh_nt);
    ntile = logical - mtile * (unsigned)p.n_tiles;
  }
  if (p.tmap_t > 0) {
    // Temporal conv (rows [video][frame][pixel]): a tile's three taps are the same 256 pixels of frames f - 1, f, f + 1.  In row
    // order those are hw / 256 tiles apart -- at 64 x 64 sixteen tiles, half a round of an XCD's 32 resident blocks, 2.6 MB of
    // other rows in between -- and every activation line came from beyond L2 three times (profiles/r4/pmc_gemm_traffic.json: the
    // 320-wide form fetched 490 MB per launch for 84 MB of rows).  Frame-fastest: the 32 tiles an XCD runs at a time are 2 (8, 32)
    // patches x all 16 frames, so a line is fetched once and its other two readers hit L2.
    const unsigned per = (unsigned)p.tmap_t * (unsigned)p.frames;
    const unsigned vid = g8_udiv(mtile, p.mg_tm, p.sh_tm);
    const unsigned r_ = mtile - vid * per;
    const unsigned patch = g8_udiv(r_, p.mg_fr, p.sh_fr);
    mtile = vid * per + (r_ - patch * (unsigned)p.frames) * (unsigned)p.tmap_t + patch;
  }
  const int n0 = (int)ntile * BX;
  const int m0 = (int)mtile * 256;
  const int kbeg = slice * p.k_per_split;
  const int nk = p.k_per_split / 64;
  // X pieces of this wave per half-tile (wave-uniform): 2, or 3 / 2 for BX = 320
  const int npx = XPC % 8 == 0 ? XPC / 8 : (XPC - wave + 7) / 8;

  // ---- LDS-DMA lane geometry: piece j covers local rows 8 j .. 8 j + 7; lane = (row in piece, chunk position) -------------
  const int lrow = lane >> 3, pos = lane & 7;
  const int cch = (pos ^ ((4 * wave + (lrow >> 1)) & 7)) * 8;  // source chunk (halfs): swizzle of row 8 (wave + 8 q) + lrow
  // weight side: one byte offset per staged row of X half 0 (half 1 = + XQ rows); the K advance lives in the scalar
  // offset; rows >= N fall outside the resource (num_records = N K 2 bytes) and read zeros
  // sub-pixel upsample conv (UPS instantiation only): the tile's output parity; its weights are phase `sp_ph` of four kernels
  int sp_ph = 0;
  if constexpr (UPS) {
    if (p.subpx) sp_ph = m0 / p.sp_rows;  // (scalar; sp_rows is a multiple of 256: a tile never straddles two phases)
  }
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (size_t)sp_ph * p.N * p.K), 0,
                                                                          (int)((size_t)p.N * p.K * 2), 0x00020000);
  unsigned xoff[PXM];
#pragma unroll
  for (int q = 0; q < PXM; ++q) {
    const int lr = 8 * (wave + 8 * q) + lrow;
    const int row = n0 + (lr / XQ) * (2 * XQ) + (lr % XQ);
    xoff[q] = (unsigned)(((size_t)row * p.K + cch) * 2);
  }
  const unsigned xh1 = (unsigned)XQ * (unsigned)p.K * 2u;  // byte distance of half 1's rows
  int xso0 = kbeg * 2, xso1 = kbeg * 2;  // scalar byte offsets of the two X streams
  const int wbase = wave * 1024;
#define G8_LDS(off) ((__attribute__((address_space(3))) void*)(smem + (off)))
  // (half 1's offset is formed at issue time by an opaque add: loop-invariant, hipcc would otherwise keep PXM more registers)
  auto x_off = [&](int h, int q) -> unsigned {
    if (!h || YP) return xoff[q] + (h ? xh1 : 0u);
    unsigned r;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "s"(xh1), "v"(xoff[q]));
    return r;
  };
#define G8_ISSUE_X(h, base, so)                                                                                         \
  do {                                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < PXM; ++q) if (q < npx)                                                        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, G8_LDS((base) + wbase + q * 8192), 16, (int)x_off(h, q), so, 0, 0); \
    so += 128;                                                                                                          \
  } while (0)
  // The weight halves of K tile 0 go out HERE: the gather geometry below (two integer divisions and nine tap tests per staged
  // row) is ~900 instructions -- with it in front, the first DMA left the wave 3.5 k ticks after kernel entry (phase stamps),
  // one and a half K tiles of nothing.  vmcnt arithmetic unchanged: all of tile 0 is still issued before any of tile 1.
  G8_ISSUE_X(0, OX0, xso0); G8_ISSUE_X(1, OX1, xso1);
  __builtin_amdgcn_sched_barrier(0);
  // chunk `tid` of the epilogue table: absent operands, channels past N / n_store and rows past the tile's last row-add row read
  // the zero constant (r16(x + 0) == x)
  const bool t_ln = EPI == 0 ? p.ln_s != nullptr : EPI >= 2;
  const unsigned ra_row0 = p.rowadd ? g8_udiv((unsigned)m0, p.mg_ra, p.sh_ra) : 0u;
  auto request_table = [&]() {
    // (the two LayerNorm vectors' base pointers pinned in scalar registers: selecting between the kernel arguments per lane, hipcc
    // fetched the chosen POINTER from the kernarg segment with a vector load and waited vmcnt(0) for it -- behind the LDS-DMA of
    // K tile 1 in flight)
    unsigned long long a_ls = (unsigned long long)p.ln_s, a_lc = (unsigned long long)p.ln_c;
    asm volatile("" : "+s"(a_ls), "+s"(a_lc));
    const void* src = &g8_zero16;
    if (tid < NB) {
      const int n = n0 + tid * 8;
      if (p.bias && !t_ln && n < p.N) src = p.bias + n;
    } else if (tid < NB + 2 * NL) {
      const int c = tid - NB, v = c >= NL;
      const int n = n0 + (c - v * NL) * 4;
      if (EPI != 1 && t_ln && n < p.N) src = reinterpret_cast<const float*>(v ? a_lc : a_ls) + n;
    } else if (tid < TOT) {
      const int c = tid - NB - 2 * NL;
      const int r = c / NB, n = n0 + (c - r * NB) * 8;
      const unsigned ra_rows = p.rowadd ? g8_udiv((unsigned)min(m0 + 255, p.M - 1), p.mg_ra, p.sh_ra) - ra_row0 + 1u : 0u;
      if (EPI != 2 && EPI != 3 && (unsigned)r < ra_rows && n < p.n_store) src = p.rowadd + (size_t)(ra_row0 + r) * p.ld_rowadd + n;
    }
    if (wave * 64 < TOT)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, G8_LDS(T_B + wbase), 16, 0, 0);
  };
  __builtin_amdgcn_sched_barrier(0);
  // activation side: per staged row a source-row index for tap (0,0) and a mask of the taps inside the image (two 16-bit
  // masks per register)
  int rowoff[AFF ? 1 : 2][AFF ? 1 : 2];
  unsigned vmask[2];  // [h]: q = 0 in bits 0..15, q = 1 in bits 16..31
  unsigned yoff[YP ? 2 : 1][YP ? 2 : 1];
  auto row_m = [&](int h, int q) {
    const int lr = 8 * (wave + 8 * q) + lrow;
    return m0 + (lr >> 5) * 64 + h * 32 + (lr & 31);
  };
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    vmask[h] = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int m = row_m(h, q);
      const bool live = m < p.M;
      const int mm = live ? m : 0;
      if constexpr (!AFF) rowoff[h][q] = mm;
      if constexpr (YP) yoff[h][q] = G8_OOB;
      unsigned mk = live ? 1u : 0u;
      if (p.a_mode == MVOC_A_CONV3X3) {
        // (sub-pixel form: rows are phase-local source pixels, hout x wout = the source grid, taps 2 x 2 starting one pixel up /
        // left for parity 0 and at the pixel for parity 1)
        const bool sp = UPS && p.subpx;
        const int ml = sp ? mm - sp_ph * p.sp_rows : mm;
        const int ks = sp ? 2 : 3;
        const int img = (int)g8_udiv((unsigned)ml, p.mg_hwout, p.sh_hwout);
        const int rem = ml - img * (p.hout * p.wout);
        const int oy = (int)g8_udiv((unsigned)rem, p.mg_wout, p.sh_wout);
        const int y0 = oy * p.stride - p.pad + (sp ? sp_ph >> 1 : 0), x0 = (rem - oy * p.wout) * p.stride - p.pad + (sp ? sp_ph & 1 : 0);
        if constexpr (!AFF) rowoff[h][q] = (img * p.hsrc + y0) * p.wsrc + x0;
        // tap (ky, kx) is inside the image iff row y0 + ky and column x0 + kx are: ks column bits, replicated per live row
        unsigned cb = 0;
        mk = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t) cb |= (t < ks && (unsigned)(x0 + t) < (unsigned)p.wup) ? 1u << t : 0u;
#pragma unroll
        for (int t = 0; t < 3; ++t) mk |= (t < ks && (unsigned)(y0 + t) < (unsigned)p.hup) ? cb << (ks * t) : 0u;
        if (!live) mk = 0;
      } else if (p.a_mode == MVOC_A_TEMPORAL3) {
        const unsigned vf = g8_udiv((unsigned)mm, p.mg_hw, p.sh_hw);  // frame index over all videos
        const int f = (int)(vf - g8_udiv(vf, p.mg_fr, p.sh_fr) * (unsigned)p.frames);
        mk = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t)
          if (live && f + t - 1 >= 0 && f + t - 1 < p.frames) mk |= 1u << t;
      }
      vmask[h] |= mk << (16 * q);
    }
  }
  if constexpr (AFF) rowoff[0][0] = row_m(0, 0) - (p.a_mode == MVOC_A_CONV3X3 ? p.wsrc + 1 : 0);
  // K position of the NEXT activation tile to issue (both Y halves of a tile are issued from the same position): all scalar
  const int ntaps = p.a_mode == MVOC_A_CONV3X3 ? 9 : 3;  // (KO only)
  int ytap, ych0;
  if constexpr (KO) {
    const int ck = kbeg / (ntaps * 64);
    ytap = (kbeg - ck * ntaps * 64) >> 6;
    ych0 = ck * 64;
  } else {
    ytap = kbeg / p.cin;
    ych0 = kbeg - ytap * p.cin;
  }
  int yso = 0, yld2 = 0, ytaprows = 0;
  bool yre = true;
  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)G8_OOB, 0x00020000);

  auto y_offset = [&](int h, int q) -> unsigned {
    const bool ok = (vmask[h] >> (16 * q + ytap)) & 1u;
    unsigned srow = (unsigned)((AFF ? rowoff[0][0] + h * 32 + q * 128 : rowoff[AFF ? 0 : h][AFF ? 0 : q]) + ytaprows);
    if (UPS && !p.subpx) {  // nearest-upsampled source: the row is not affine in the tap (three launches per forward)
      const int ky = ytap / 3, kx = ytap - ky * 3;
      const int m = min(row_m(h, q), p.M - 1);
      const int hwout = p.hout * p.wout;
      const int img = m / hwout;
      const int rem = m - img * hwout;
      const int oy = rem / p.wout;
      const int iy = min((int)floorf((oy * p.stride - p.pad + ky) * p.ups_sh), p.hsrc - 1);
      const int ix = min((int)floorf(((rem - oy * p.wout) * p.stride - p.pad + kx) * p.ups_sw), p.wsrc - 1);
      srow = (unsigned)((img * p.hsrc + iy) * p.wsrc + ix);
    }
    const unsigned off = (unsigned)__umul24(srow, (unsigned)yld2) + (unsigned)(cch * 2);  // rows and pitches are < 2^24
    return ok ? off : G8_OOB;
  };
  // new tap or new source (wave-uniform event, every cin / 64 or c1 / 64 K tiles)
  auto y_prepare = [&]() {
    if (!yre) return;
    const bool second = ych0 >= p.c1;
    yld2 = (second ? p.lda2 : p.lda) * 2;
    ytaprows = 0;
    if (p.a_mode == MVOC_A_CONV3X3) {
      if (UPS && p.subpx) ytaprows = (ytap >> 1) * p.wsrc + (ytap & 1);
      else ytaprows = (ytap / 3) * p.wsrc + (ytap % 3);
    } else if (p.a_mode == MVOC_A_TEMPORAL3) ytaprows = (ytap - 1) * p.hw;
    yso = (second ? ych0 - p.c1 : ych0) * 2;
    rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(second ? p.a2 : p.a), 0, (int)G8_OOB, 0x00020000);
    if constexpr (YP) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) yoff[h][q] = y_offset(h, q);
    }
  };
  auto y_advance = [&]() {
    if constexpr (KO) {
      yre = true;
      if (++ytap == ntaps) { ytap = 0; ych0 += 64; }
      return;
    }
    yso += 128;
    ych0 += 64;
    yre = ych0 == p.c1;
    if (ych0 >= p.cin) { ych0 = 0; ++ytap; yre = true; }
  };

#define G8_ISSUE_Y(h, base)                                                                                             \
  do {                                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, G8_LDS((base) + wbase + q * 8192), 16,                           \
                                                 (int)(YP ? yoff[YP ? h : 0][YP ? q : 0] : y_offset(h, q)), yso, 0, 0);                   \
  } while (0)

  // ---- fragment reads: lane reads row (l & 15) of a 16-row tile, chunk 4 s + (l >> 4) -----------------------------------
  const int fr = lane & 15, fg = lane >> 4;
  const int sw = fr >> 1;
  const char* xr0 = smem + (wr * XQ + fr) * 128 + (fg ^ sw) * 16;        // k-step 0 / 1 of this lane's X rows
  const char* xr1 = smem + (wr * XQ + fr) * 128 + ((4 + fg) ^ sw) * 16;
  const char* yr0 = smem + (wc * 32 + fr) * 128 + (fg ^ sw) * 16;
  const char* yr1 = smem + (wc * 32 + fr) * 128 + ((4 + fg) ^ sw) * 16;
#define G8_LDP(ptr, off) (*reinterpret_cast<const half8_t*>((ptr) + (off)))

  half8_t xf[XT][2], yf0[2][2], yf1s[RETAIN ? 2 : 1][2][2];
  auto& yf1 = RETAIN ? yf1s[RETAIN ? 1 : 0] : yf0;  // !RETAIN: one fragment set serves Y0 and Y1 in turn
  (void)yf1s;
  f32x4 acc[2][2][XT][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < XT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define G8_RDX(off)                                             \
  _Pragma("unroll") for (int i = 0; i < XT; ++i) {              \
    xf[i][0] = G8_LDP(xr0, (off) + i * 2048);                   \
    xf[i][1] = G8_LDP(xr1, (off) + i * 2048);                   \
  }
#define G8_RDY(dst, off)                                        \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {               \
    dst[j][0] = G8_LDP(yr0, (off) + j * 2048);                  \
    dst[j][1] = G8_LDP(yr1, (off) + j * 2048);                  \
  }
  // XT = 5: the accumulate is pinned IN PLACE by an asm statement (D = C): the register file is full (160 accumulators + 56
  // fragment registers) and hipcc's out-of-place MFMA form cost ~110 v_mov_b64 accumulator copies per two K tiles.  Hazards
  // hipcc no longer sees: an MFMA's D feeding the next MFMA as C needs no wait state; every other reader of an accumulator
  // (the epilogue) sits behind the loop's last s_barrier AND an explicit s_nop pair; the A / B operands come from LDS reads that
  // hipcc waits for (they are register inputs of the statement).
  auto mma = [&](f32x4& c, const half8_t& a, const half8_t& b) {
    if constexpr (XT == 5) {
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    } else {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
  };
#define G8_MMA(A, B, YF)                                                                                              \
  do {                                                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                                    \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int i = 0; i < XT; ++i)                     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) mma(acc[A][B][i][j], xf[i][s], YF[j][s]);                       \
    __builtin_amdgcn_s_setprio(0);                                                                                    \
  } while (0)

  // in flight behind the per-tile wait: {Yh0, Xh0, Yh1} of the tile after next = 2 + npx + 2 LDS-DMA of this wave
  auto wait_tile = [&]() {
    if (XT == 5 && npx == 3) g8_wait_vm<7>(); else g8_wait_vm<6>();
  };

  // ---- prologue: K tile 0 landed, {Yh0, Xh0, Yh1} of tile 1 in flight ------------------------------------------------------
  y_prepare();
  G8_ISSUE_Y(0, OY0); G8_ISSUE_Y(1, OY1);
  y_advance();
  if (nk > 1) {
    y_prepare();
    G8_ISSUE_Y(0, OY0 + YB); G8_ISSUE_X(0, KB + OX0, xso0); G8_ISSUE_Y(1, KB + OY1);
    y_advance();
    wait_tile();
  } else {
    g8_wait_vm<0>();
  }
  // 256-wide: the table is requested HERE, behind the prologue's wait (in front of it the first K tile waited for the table's
  // cold lines too: + 900 ticks of prologue, phase stamps); every later counted wait retires it -- it is older than the pieces
  // they leave in flight -- and the epilogue starts with vmcnt(0) + a barrier anyway
  if constexpr (TBL_EARLY) request_table();
  G8_BAR();
  G8_STAMP(1);
  if (wr == 1) G8_BAR();  // group 1 falls one barrier behind

  // Yh0 slots: RETAIN -> the tile's parity (static); otherwise three rotating byte offsets (scalar)
  int y0a = 0, y0b = YB, y0c = 2 * YB;  // slot of tile t, t+1, t+2 (mod 3)

  // One K tile = 4 phases; P = parity of the tile (static: every LDS offset is an immediate).  Phase j restages one
  // half-tile: Xh1 of tile t+1 (the other buffer), then Yh0, Xh0, Yh1 of tile t+2 (RETAIN: Yh0's slot was read in phase 1 and
  // those reads were waited for before phase 1's first barrier; Xh0 / Yh1 are two phases old; !RETAIN: Yh0 goes to the
  // third slot).
#define G8_KTILE(P, t)                                                                                 \
  do {                                                                                                 \
    /* ---- phase 1: (X0, Y0) ---- */                                                                  \
    if constexpr (RETAIN) { G8_RDY(yf0, OY0 + (P) * YB); } else { G8_RDY(yf0, OY0 + y0a); }            \
    asm volatile("" ::: "memory");                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    G8_RDX((P) * KB + OX0);                                                                            \
    asm volatile("" ::: "memory");                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    if ((t) + 1 < nk) G8_ISSUE_X(1, (1 - (P)) * KB + OX1, xso1);                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * XT) : "memory"); /* the four Y reads are back */    \
    G8_BAR();                                                                                          \
    G8_MMA(0, 0, yf0);                                                                                 \
    G8_BAR();                                                                                          \
    /* ---- phase 2: (X0, Y1) ---- */                                                                  \
    G8_RDY(yf1, (P) * KB + OY1);                                                                       \
    asm volatile("" ::: "memory");                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    if ((t) + 2 < nk) {                                                                                \
      y_prepare();                                                                                     \
      if constexpr (RETAIN) { G8_ISSUE_Y(0, OY0 + (P) * YB); } else { G8_ISSUE_Y(0, OY0 + y0c); }      \
    }                                                                                                  \
    G8_BAR();                                                                                          \
    G8_MMA(0, 1, yf1);                                                                                 \
    G8_BAR();                                                                                          \
    /* ---- phase 3: (X1, Y1) ---- */                                                                  \
    G8_RDX((P) * KB + OX1);                                                                            \
    asm volatile("" ::: "memory");                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    if ((t) + 2 < nk) G8_ISSUE_X(0, (P) * KB + OX0, xso0);                                             \
    G8_BAR();                                                                                          \
    G8_MMA(1, 1, yf1);                                                                                 \
    G8_BAR();                                                                                          \
    /* ---- phase 4: (X1, Y0) ---- */                                                                  \
    if constexpr (!RETAIN) {                                                                           \
      G8_RDY(yf0, OY0 + y0a);                                                                          \
      asm volatile("" ::: "memory");                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                               \
    }                                                                                                  \
    if ((t) + 2 < nk) {                                                                                \
      G8_ISSUE_Y(1, (P) * KB + OY1);                                                                   \
      y_advance();                                                                                     \
      wait_tile();                                                                                     \
    } else if ((t) + 2 == nk) {                                                                        \
      g8_wait_vm<0>();                                                                                 \
    }                                                                                                  \
    G8_BAR();                                                                                          \
    G8_MMA(1, 0, yf0);                                                                                 \
    G8_BAR();                                                                                          \
    if constexpr (!RETAIN) { const int r_ = y0a; y0a = y0b; y0b = y0c; y0c = r_; }                     \
  } while (0)

  int t = 0;
  // (build.py compiles this file with -falign-loops=64: the loop's time moved by 1.2 % -- 2 446 -> 2 476 ticks per K tile -- with
  // the placement an unrelated edit of the prologue happened to give its first instruction)
#pragma unroll 1
  for (; t + 1 < nk; t += 2) {
    G8_KTILE(0, t);
    G8_KTILE(1, t + 1);
  }
  if (t < nk) G8_KTILE(0, t);
  if (wr == 0) G8_BAR();  // group 0's balancing barrier: every wave is out of the K loop, the ring is free
  // XT = 5: hipcc does not know the asm statements above are MFMAs and inserts no MFMA -> VALU wait states; whatever reads an
  // accumulator first (the slab stores, the epilogue arithmetic) must not depend on what happens to sit in between
  if constexpr (XT == 5) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  G8_STAMP(2);

  const int g = fg;  // lane owns channels nq + 4 g .. + 3 of pixel (lane & 15) per 16 x 16 tile
  if (p.split_k > 1) {  // raw fp32 partials; bias / activation / residual happen in the reduce pass
    float* slab = p.ws + (size_t)slice * p.M * p.N;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < XT; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int n = n0 + wr * XH + a * XQ + i * 16 + 4 * g;
            const int m = m0 + wc * 64 + b * 32 + j * 16 + fr;
            if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(slab + (size_t)m * p.N + n) = acc[a][b][i][j];
          }
    return;
  }

  // ---- epilogue ----------------------------------------------------------------------------------------------------------
  // Two passes (a = 0, 1: the wave's two channel halves), each: arithmetic in the accumulator layout -> fp16 tile in a
  // wave-private LDS region -> read back as 16-byte row chunks, residual added, stored.  One block per CU: every memory round
  // trip in here is paid with all MFMAs idle, so
  // * the per-channel vectors of the block (bias, LayerNorm-fold row sums / constants, the time-embedding row-add rows of the
  //   tile's <= RA_ROWS samples) are fetched ONCE per block into an LDS table (request_table above) and read from LDS where they
  //   are used (round 3 fetched them per pass into 40-50 registers: two more L2 round trips per block, and no room to issue
  //   anything else early);
  // * the residual chunks of pass 0 are requested before pass 0's arithmetic, those of pass 1 too where the registers allow
  //   (256-wide), at the latest before pass 0's stores: vmcnt retires in issue order, a load issued behind the 8 / 10 stores of pass 0
  //   waits for their acknowledgement under the chip-wide store burst (phase stamps, 320-wide: pass 1 20 k ticks against 12 k).
  const bool geglu = EPI == 0 ? p.act == MVOC_ACT_GEGLU : EPI == 3;
  const bool use_ln = EPI == 0 ? p.ln_s != nullptr : EPI >= 2;
  constexpr bool GG = XT == 4 && (EPI == 0 || EPI == 3);  // instantiations that carry the GEGLU arithmetic
  constexpr bool LNF = EPI != 1;                          // ... the LayerNorm-fold arithmetic
  constexpr bool RADD = EPI != 2 && EPI != 3;             // ... a row-add (the host sends LayerNorm fold + row-add to the general form)
  char* epi = smem + wave * (64 * PITCH);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  if constexpr (!TBL_EARLY) request_table();
  // pixel row of lane fr in the four (b, j) tiles: LayerNorm statistics, row-add row inside the table
  float ln_mu[2][2], ln_rs[2][2];
  int raoff[2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + wc * 64 + b * 32 + j * 16 + fr;
      const int ms = m < p.M ? m : p.M - 1;
      ln_mu[b][j] = 0.f; ln_rs[b][j] = 1.f;
      if (LNF && use_ln) { ln_mu[b][j] = p.ln_stats[2 * (size_t)ms]; ln_rs[b][j] = p.ln_stats[2 * (size_t)ms + 1]; }
      raoff[b][j] = RADD && p.rowadd ? (int)(g8_udiv((unsigned)ms, p.mg_ra, p.sh_ra) - ra_row0) * (BX * 2) : 0;
    }
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)((size_t)p.M * p.ldo * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_r =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.resid ? p.resid : p.out), 0, p.resid ? (int)((size_t)p.M * p.ldr * 2) : 0, 0x00020000);
  const int nchunk = geglu ? 4 : XT * 2;
  auto nq_of = [&](int a) { return n0 + wr * XH + a * XQ; };  // first packed weight row of pass a

  // arithmetic of pass A in the accumulator layout (same fp16 rounding points as gemm_epilogue_lds), fp16 tile -> LDS
  auto arith = [&](auto A_) {
    constexpr int a = decltype(A_)::value;  // (the accumulators must be indexed statically)
    const int cb = wr * XH + a * XQ + 4 * g;  // this lane's first channel of tile i = 0, relative to n0
    if (geglu) {
      if constexpr (GG) {
        // packed rows: blocks of 64 = 32 value rows then 32 gate rows -> tiles i = 0, 1 are values, i + 2 their gates
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int c = cb + i * 16;
          const half4_t bh = *reinterpret_cast<const half4_t*>(smem + T_B + c * 2), bg = *reinterpret_cast<const half4_t*>(smem + T_B + (c + 32) * 2);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(smem + T_LS + c * 4), sg = *reinterpret_cast<const f32x4*>(smem + T_LS + (c + 32) * 4);
          const f32x4 ch = *reinterpret_cast<const f32x4*>(smem + T_LC + c * 4), cg = *reinterpret_cast<const f32x4*>(smem + T_LC + (c + 32) * 4);
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              half4_t o;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float av = acc[a][b][i][j][e], ag = acc[a][b][i + 2][j][e];
                const float hv = r16(use_ln ? ln_rs[b][j] * (av - ln_mu[b][j] * sh[e]) + ch[e] : av + (float)bh[e]);
                const float gv = r16(use_ln ? ln_rs[b][j] * (ag - ln_mu[b][j] * sg[e]) + cg[e] : ag + (float)bg[e]);
                o[e] = (half_t)(hv * r16(gelu_fast_f(gv)));
              }
              *reinterpret_cast<half4_t*>(epi + (b * 32 + j * 16 + fr) * PITCH + (i * 16 + 4 * g) * 2) = o;
            }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < XT; ++i) {
        const int c = cb + i * 16;
        const half4_t b4 = *reinterpret_cast<const half4_t*>(smem + T_B + c * 2);
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, c4 = s4;
        if constexpr (LNF) { s4 = *reinterpret_cast<const f32x4*>(smem + T_LS + c * 4); c4 = *reinterpret_cast<const f32x4*>(smem + T_LC + c * 4); }
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float x[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
              x[e] = r16(LNF && use_ln ? ln_rs[b][j] * (acc[a][b][i][j][e] - ln_mu[b][j] * s4[e]) + c4[e] : acc[a][b][i][j][e] + (float)b4[e]);
            if constexpr (RADD) {
              const half4_t t4 = *reinterpret_cast<const half4_t*>(smem + T_RA + raoff[b][j] + c * 2);  // zeros without a row-add: r16(x + 0) == x
#pragma unroll
              for (int e = 0; e < 4; ++e) x[e] = r16(x[e] + (float)t4[e]);
            }
            if (EPI == 0 && p.act == MVOC_ACT_SILU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) x[e] = r16(silu_f(x[e]));
            } else if (EPI == 0 && p.act == MVOC_ACT_GELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) x[e] = r16(gelu_fast_f(x[e]));
            }
            half4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)x[e];
            *reinterpret_cast<half4_t*>(epi + (b * 32 + j * 16 + fr) * PITCH + (i * 16 + 4 * g) * 2) = o;
          }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's tile is in LDS (a wave's DS ops execute in order)
  };
  // Read back as rows: 16-byte chunks, consecutive lanes on consecutive chunks of a pixel row; residual added, stored.
  // Branch-free through range-checked buffer instructions (rows >= M fall outside the resource, columns >= n_store get an
  // out-of-range offset): one 32-bit offset per chunk instead of a 64-bit address and an exec mask, so that ALL residual
  // chunks of a pass are in flight together (two loads in flight per lane left the chip-wide output burst at 2.4 TB/s with
  // every MFMA idle -- phase stamps: 2 x 22 k ticks per block against 2.3 k per K tile).
  auto chunk_off = [&](int a, int ln, int it, int ld) -> unsigned {
    const int nbase = geglu ? nq_of(a) / 2 : nq_of(a);
    const int idx = ln + 64 * it;
    const int px = idx / nchunk, c = idx - px * nchunk;
    int m = m0 + wc * 64 + px;
    const int n = nbase + c * 8;
    bool ok = it < nchunk && n < p.n_store;
    if constexpr (UPS) {
      if (p.subpx) {  // source pixel (img, i, j) of phase (pa, pb) -> output pixel (2 i + pa, 2 j + pb) of the 2 x larger image
        ok = ok && m < p.M;
        const int ml = m - sp_ph * p.sp_rows;
        const int img = (int)g8_udiv((unsigned)ml, p.mg_hwout, p.sh_hwout);
        const int rem = ml - img * (p.hout * p.wout);
        const int i_ = (int)g8_udiv((unsigned)rem, p.mg_wout, p.sh_wout);
        m = (img * 2 * p.hout + 2 * i_ + (sp_ph >> 1)) * (2 * p.wout) + 2 * (rem - i_ * p.wout) + (sp_ph & 1);
      }
    }
    return ok ? (unsigned)(m * ld + n) * 2u : G8_OOB;
  };
  auto load_resid = [&](int a, u32x4 (&rr)[XT * 2], int i0, int i1) {
#pragma unroll
    for (int it = 0; it < XT * 2; ++it)
      if (it >= i0 && it < i1) rr[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, chunk_off(a, lane, it, p.ldr), 0, 0);  // zeros without a residual
  };
  // 320-wide: 160 accumulators are live until pass 0's arithmetic is done, and 7 of pass 1's 10 chunks are what fits beside
  // pass 0's readback without a spill (8: 2 registers spilled, 10: 12)
  constexpr int R1_EARLY = EPI == 2 ? 5 : 7;  // (the LayerNorm-fold form also holds the rows' statistics)
  auto readback_store = [&](int a, const u32x4 (&rr)[XT * 2]) {
    // (the index arithmetic is redone from an opaque copy of the lane id: shared with the loads, hipcc keeps four values per
    // chunk alive across the wait and spills them)
    int lane2 = lane;
    asm volatile("" : "+v"(lane2));
#pragma unroll
    for (int it = 0; it < XT * 2; ++it) {
      const int idx = lane2 + 64 * it;
      const int px = idx / nchunk, c = idx - px * nchunk;
      half8_t v = *reinterpret_cast<const half8_t*>(epi + (it < nchunk ? px * PITCH + c * 16 : 0));
      const half8_t r8 = __builtin_bit_cast(half8_t, rr[it]);
      if (p.resid) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)r8[e]);
      }
      // non-temporal only where a store instruction writes whole 128-byte lines (256-wide: 64 channels per pass).  The 320-wide
      // tile's passes are 160 bytes per pixel row: every line but the first and last of a row is written in two pieces by two
      // passes / waves, and as streaming stores those pieces do not merge in L2 -- phase stamps: readback + stores 20-30 k ticks
      // per pass against 4-5 k with plain stores (launch 230 -> 218 us at K = 960, 288 -> 262 with a residual); the 256-wide
      // tile with plain stores loses 15 % instead (its stores then evict operand lines: 2 734 -> 3 242 ticks per K tile)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_o, chunk_off(a, lane2, it, p.ldo), 0, (G8_NT_STORE && XT == 4) ? 2 : 0);
      if ((p.stats || p.rowmom) && it < nchunk) *reinterpret_cast<half8_t*>(epi + px * PITCH + c * 16) = v;  // the STORED values, for stats_pass / rowmom_pass
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next pass overwrites the tile
  };
  // Per-channel sum and sum of squares of the values just stored (what the consumer's GroupNorm reads: pnp_utils.py:909-910,
  // 953-965, 1048-1051 normalise exactly these tensors), over this wave's 64 pixels, on the idle matrix pipe: with X^T fragments
  // (channel on the lane, 8 pixels per lane: transposing LDS reads of the tile) as BOTH operands, D = X^T X is the Gram matrix of
  // a 16-channel block -- its diagonal the sums of squares, fp32-exact products -- and ones x X the column sums.  4 (5) x 2 x 2
  // MFMAs per pass instead of a separate pass over the tensor (gn_partial: 3.4 % of GPU time in round 3).
  constexpr int S_OFF = TBL_EARLY ? 8 * 64 * PITCH : T_B + 8192;  // fp32 [wc][BX][2] behind the tiles / the table
  static_assert(S_OFF + 4 * BX * 8 <= SMEM, "statistics area");
  auto stats_pass = [&](int a) {
    const half8_t ones = {(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
    const int kg = lane >> 4, n = lane & 15;
    float* so = reinterpret_cast<float*>(smem + S_OFF) + (wc * BX + wr * XH + a * XQ) * 2;
#pragma unroll
    for (int cb = 0; cb < XT; ++cb) {
      f32x4 dg = {0.f, 0.f, 0.f, 0.f}, dsum = dg;
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        // 16-lane group kg: pixels 32 ps + 8 kg + {0..3} and + {4..7}; lane 4 q + pc supplies row q, columns 4 pc .. 4 pc + 3
        const char* va = epi + (32 * ps + 8 * kg + (n >> 2)) * PITCH + (cb * 16 + 4 * (n & 3)) * 2;
        const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)va);
        const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(va + 4 * PITCH));
        half8_t f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[e] = (half_t)lo[e]; f[4 + e] = (half_t)hi[e]; }
        dg = __builtin_amdgcn_mfma_f32_16x16x32_f16(f, f, dg, 0, 0, 0);
        dsum = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, f, dsum, 0, 0, 0);
      }
      // D[m][n] sits in lane (n, m >> 2), register m & 3: the diagonal element of channel n in the lane whose group is n >> 2
      const float sq = (n & 3) == 0 ? dg[0] : (n & 3) == 1 ? dg[1] : (n & 3) == 2 ? dg[2] : dg[3];
      if (kg == (n >> 2)) *reinterpret_cast<float2*>(so + (cb * 16 + n) * 2) = float2{dsum[0], sq};
    }
  };
  // Per-ROW sum and sum of squares of the values just stored, over this wave's XQ channels of the pass (what the consumer's
  // LayerNorm reads: pnp_utils.py:250-257, 296, 322), the same way: with ROW-major fragments of the tile (pixel on the lane, 8
  // channels per lane: plain 16-byte LDS reads) as both operands, D = X X^T is the Gram matrix of a 16-pixel block -- its diagonal
  // the rows' sums of squares -- and ones x X^T the row sums.  The four (channel half, pass) quarters of a row meet in LDS and
  // leave as ONE {sum, sum of squares} per row and n-tile; the consumer's statistics come from merging a row's n-tiles
  // (mvoc_row_stats_from_moments_f32) instead of a pass over the tensor (row_stats_kernel: 1.4 % of GPU time).
  constexpr int R_OFF = S_OFF + 4 * BX * 8;  // fp32 [quarter][256 rows][2]
  static_assert(R_OFF + 4 * 256 * 8 <= SMEM, "row-moment area");
  auto rowmom_pass = [&](int a) {
    const half8_t ones = {(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
    const half8_t zero8 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    const int kg = lane >> 4, n = lane & 15;
    float* ro = reinterpret_cast<float*>(smem + R_OFF) + ((wr * 2 + a) * 256 + wc * 64) * 2;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      f32x4 dg = {0.f, 0.f, 0.f, 0.f}, dsum = dg;
#pragma unroll
      for (int ks = 0; ks * 32 < XQ; ++ks) {
        // lane (pixel n of the block, channel group kg): channels 32 ks + 8 kg .. + 7 (320-wide: the third step holds 16 channels)
        half8_t f = zero8;
        if (ks * 32 + kg * 8 < XQ) f = *reinterpret_cast<const half8_t*>(epi + (rb * 16 + n) * PITCH + (ks * 32 + kg * 8) * 2);
        dg = __builtin_amdgcn_mfma_f32_16x16x32_f16(f, f, dg, 0, 0, 0);
        dsum = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, f, dsum, 0, 0, 0);
      }
      // D[m][n] sits in lane (n, m >> 2), register m & 3: the diagonal element of pixel n in the lane whose group is n >> 2
      const float sq = (n & 3) == 0 ? dg[0] : (n & 3) == 1 ? dg[1] : (n & 3) == 2 ? dg[2] : dg[3];
      if (kg == (n >> 2)) *reinterpret_cast<float2*>(ro + (rb * 16 + n) * 2) = float2{dsum[0], sq};
    }
  };
  {
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    u32x4 r0[XT * 2], r1[XT * 2];
    g8_wait_vm<0>();  // the table has landed (256-wide: long ago; this wave's piece -- the barrier makes it every wave's)
    G8_BAR();
    load_resid(0, r0, 0, XT * 2);  // pass 0's residual chunks fly under pass 0's arithmetic in both forms
    if constexpr (XT == 4) load_resid(1, r1, 0, XT * 2);
    __builtin_amdgcn_sched_barrier(0);
    arith(I0{});
    G8_STAMP(6);
    __builtin_amdgcn_sched_barrier(0);  // (320-wide: 160 live accumulators until here)
    if constexpr (XT != 4) load_resid(1, r1, 0, R1_EARLY);
    __builtin_amdgcn_sched_barrier(0);
    readback_store(0, r0);
    G8_STAMP(3);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (XT != 4) load_resid(1, r1, R1_EARLY, XT * 2);
    __builtin_amdgcn_sched_barrier(0);
    if (p.stats) stats_pass(0);
    if (p.rowmom) rowmom_pass(0);
    __builtin_amdgcn_sched_barrier(0);
    arith(I1{});
    __builtin_amdgcn_sched_barrier(0);
    readback_store(1, r1);
    G8_STAMP(4);
    if (p.stats || p.rowmom) {
      if (p.stats) stats_pass(1);
      if (p.rowmom) rowmom_pass(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      G8_BAR();
      if (p.stats) {
        // the four pixel quarters of the tile, summed in a fixed order: [m tile][n_store][2] fp32, coalesced
        for (int i = tid; i < BX * 2; i += 512) {
          const int ch = i >> 1;
          if (n0 + ch < p.n_store) {
            const float* si = reinterpret_cast<const float*>(smem + S_OFF) + i;
            p.stats[((size_t)(m0 >> 8) * p.n_store + n0) * 2 + i] = ((si[0] + si[BX * 2]) + si[2 * BX * 2]) + si[3 * BX * 2];
          }
        }
      }
      if (p.rowmom) {
        // the four channel quarters of a row, summed in a fixed order: [row][n tile][2] fp32
        for (int i = tid; i < 256 * 2; i += 512) {
          const int row = m0 + (i >> 1);
          if (row < p.M) {
            const float* ri = reinterpret_cast<const float*>(smem + R_OFF) + i;
            p.rowmom[((size_t)row * p.rowmom_ld + ntile) * 2 + (i & 1)] = ((ri[0] + ri[512]) + ri[1024]) + ri[1536];
          }
        }
      }
    }
  }
#ifdef MVOC_G8_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G8_STAMP(5);
#endif
}

// floor(n / d) == umulhi(n, mg) >> sh for every n < 2^31: with l = ceil(log2 d), P = 31 + l and mg = ceil(2^P / d) (< 2^32 as
// d > 2^(l-1), = 2^31 for a power of two), the error e = mg d - 2^P lies in [0, d), so n e < 2^31 2^l = 2^P and
// floor(n mg / 2^P) = floor(n / d) (Granlund & Montgomery); umulhi drops 32 of the P bits, sh = l - 1.
void g8_magic(unsigned d, unsigned& mg, int& sh) {
  if (d <= 1) { mg = 0; sh = -1; return; }
  int l = 0;
  while ((1ull << l) < d) ++l;
  mg = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
  sh = l - 1;
}

template <int XT, bool RETAIN, bool YP, bool UPS, bool AFF, bool KO = false>
int launch8(const GemmArgs& a0, hipStream_t s) {
  const int epi = a0.act == MVOC_ACT_NONE ? (a0.ln_s ? (a0.rowadd ? 0 : 2) : 1) : (a0.act == MVOC_ACT_GEGLU && a0.ln_s && XT == 4) ? 3 : 0;
  GemmArgs a = a0;
  constexpr int BX = XT * 64;
  a.n_tiles = (a.N + BX - 1) / BX;
  a.m_tiles = (a.M + 255) / 256;
  g8_magic((unsigned)a.n_tiles, a.mg_nt, a.sh_nt);
  {
    static const int force = getenv("MVOC_G8_BAND") ? atoi(getenv("MVOC_G8_BAND")) : 0;  // diagnostics: 1 = the n-fastest order
    a.band = force ? force : (a.n_tiles >= 8 ? 4 : a.n_tiles >= 4 ? 2 : 1);
    if (a.band > a.m_tiles) a.band = 1;
    g8_magic((unsigned)(a.band * a.n_tiles), a.mg_band, a.sh_band);
  }
  {
    static const bool off = getenv("MVOC_G8_TMAP") && atoi(getenv("MVOC_G8_TMAP")) == 0;  // diagnostics: row order
    a.tmap_t = 0;
    if (!off && a.a_mode == MVOC_A_TEMPORAL3 && a.frames > 1 && a.hw % 256 == 0 && a.M % (a.frames * a.hw) == 0 && a.hw / 256 > 1)
      a.tmap_t = a.hw / 256;
    g8_magic((unsigned)(a.tmap_t > 0 ? a.tmap_t * a.frames : 1), a.mg_tm, a.sh_tm);
  }
  g8_magic((unsigned)a.split_k, a.mg_sk, a.sh_sk);
  g8_magic((unsigned)(a.hout * a.wout), a.mg_hwout, a.sh_hwout);
  g8_magic((unsigned)a.wout, a.mg_wout, a.sh_wout);
  g8_magic((unsigned)a.hw, a.mg_hw, a.sh_hw);
  g8_magic((unsigned)a.frames, a.mg_fr, a.sh_fr);
  g8_magic((unsigned)a.rowadd_div, a.mg_ra, a.sh_ra);
  const long nblk = (long)a.n_tiles * a.m_tiles * a.split_k;
  if (nblk <= 0 || nblk > 0x7fffffffL) {
    mvoc_set_error("gemm8: grid of %ld blocks", nblk);
    return -2;
  }
  if constexpr (KO) {
    if (epi != 1) return -9;  // (conv / temporal conv: bias, row-add, residual)
    hipLaunchKernelGGL((gemm8_kernel<XT, RETAIN, YP, UPS, AFF, 1, true>), dim3((unsigned)nblk), dim3(512), 0, s, a);
  } else if constexpr (UPS) {
    // the folded-upsample form exists with the plain epilogue only (Upsample2D's conv: bias): its other epilogue forms spilled
    // registers once the sub-pixel gather joined it, and no caller has them (gemm.hip sends such a request to the general tiles)
    if (epi != 1) return -9;
    hipLaunchKernelGGL((gemm8_kernel<XT, RETAIN, YP, UPS, AFF, 1>), dim3((unsigned)nblk), dim3(512), 0, s, a);
  } else {
    if (epi == 1) hipLaunchKernelGGL((gemm8_kernel<XT, RETAIN, YP, UPS, AFF, 1>), dim3((unsigned)nblk), dim3(512), 0, s, a);
    else if (epi == 2) hipLaunchKernelGGL((gemm8_kernel<XT, RETAIN, YP, UPS, AFF, 2>), dim3((unsigned)nblk), dim3(512), 0, s, a);
    else if (epi == 3) hipLaunchKernelGGL((gemm8_kernel<XT, RETAIN, YP, UPS, AFF, 3>), dim3((unsigned)nblk), dim3(512), 0, s, a);
    else if constexpr (XT == 4) hipLaunchKernelGGL((gemm8_kernel<XT, RETAIN, YP, UPS, AFF, 0>), dim3((unsigned)nblk), dim3(512), 0, s, a);
    else return -9;  // (unreachable: mvoc_launch_gemm8 sends the general epilogue of a 320-wide request to the 256-wide tile)
  }
  return mvoc_check_launch("gemm8_kernel");
}

}  // namespace

// bx = 256 or 320 output channels per block.  Preconditions (checked by the caller, gemm.hip): k, cin, c1 multiples of 64,
// conv k == 9 cin, 16-byte addressable outputs (epi_lds), row statistics precomputed when a LayerNorm is folded in, GEGLU
// only with bx = 256, every operand spanning < 2 GB (32-bit MUBUF offsets).
int mvoc_launch_gemm8(const GemmArgs& a, int bx, hipStream_t s, int* bx_used) {
  int dummy;
  int& used = bx_used ? *bx_used : dummy;
  used = 256;
  if (a.korder) {  // chunk-major weights are read by ONE instantiation (below); any other route would multiply them in tap-major order
    const bool aff_ = a.a_mode == MVOC_A_TEMPORAL3 || (a.a_mode == MVOC_A_CONV3X3 && a.stride == 1 && a.pad == 1 && a.hsrc == a.hout && a.wsrc == a.wout);
    if (!(bx == 320 && aff_ && !a.upsample && a.act == MVOC_ACT_NONE && !a.ln_s)) {
      mvoc_set_error("gemm8: k_order = 1 needs the 320-wide tile, an affine conv / temporal gather and the plain epilogue");
      return -2;
    }
  }
  if (a.upsample) {  // (rare: the three upsampler convs of a forward) one form serves every width
    if (bx == 256 || bx == 320) return launch8<4, true, true, true, false>(a, s);
  } else {
    const bool affine = a.a_mode != MVOC_A_CONV3X3 || (a.stride == 1 && a.pad == 1 && a.hsrc == a.hout && a.wsrc == a.wout);
    if (bx == 256 || (bx == 320 && !affine)) return launch8<4, true, true, false, false>(a, s);
    if (bx == 320) {
      // the 320-wide tile exists for the epilogue forms without activation; anything else (never on a 320-wide shape of this
      // model) takes the 256-wide tile, which holds every form
      if (a.act != MVOC_ACT_NONE || (a.ln_s && a.rowadd)) return launch8<4, true, true, false, false>(a, s);
      used = 320;
      if (a.korder) {
        // chunk-major K (gemm_args.h: korder): the 320-wide plain-epilogue form only -- gemm.hip sends nothing else here
        if (a.ln_s) return -9;
        return launch8<5, false, false, false, true, true>(a, s);
      }
      return launch8<5, false, false, false, true>(a, s);
    }
  }
  mvoc_set_error("gemm8: unsupported tile width %d", bx);
  return -1;
}

#ifdef MVOC_G8_STAMPS
extern "C" int mvoc_g8_stamps_read(unsigned long long* host32) {
  return hipMemcpyFromSymbol(host32, HIP_SYMBOL(g8_dbg), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1;
}
#endif
