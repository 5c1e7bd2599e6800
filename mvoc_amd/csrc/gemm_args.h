// Argument block shared by the implicit-GEMM kernel families (gemm.hip: the general tiles; gemm8.hip: the eight-phase
// 256-pixel tiles).
#pragma once
#include "common.h"

struct GemmArgs {
  const half_t* a;
  const half_t* a2;
  const half_t* w;
  half_t* out;
  const half_t* bias;
  const half_t* rowadd;
  const half_t* resid;
  int M, N, K, n_store;
  int ldo, ldr, ld_rowadd, rowadd_div;
  int a_mode, lda, lda2, c1, cin;
  int nimg, hout, wout, hsrc, wsrc, stride, upsample, hup, wup;
  int pad;  // conv: leading (top / left) zero padding: 1, or 0 for the bottom/right-only padding of the VAE's downsamplers
  float ups_sh, ups_sw;
  int frames, hw;
  int act;
  int n_tiles, m_tiles;
  const float* ln_s;   // LayerNorm folded into this GEMM: row sums of the gamma-scaled weights (fp32 [N]) or NULL
  const float* ln_c;   //   beta @ W^T (+ bias), fp32 [N]
  const float* ln_stats;  // optional precomputed {mean, rstd} per row (fp32 [M][2]); NULL: accumulated in the K loop
  float ln_eps;
  int split_k, k_per_split;  // split-K: grid covers n_tiles*m_tiles*split_k; slice s accumulates k in [s*kps, (s+1)*kps)
  float* ws;                 // fp32 partial slabs [split_k][M][N] (deterministic: summed in slice order by splitk_reduce)
  // gemm8: divisions by launch constants as multiply-high + shift (g8_magic; sh < 0: divisor 1) -- the block's tile
  // coordinates and the (image, y, x) / frame of every staged row took ~20 integer divisions (~40 instructions each) in the prologue
  unsigned mg_nt, mg_sk, mg_hwout, mg_wout, mg_hw, mg_fr, mg_ra, mg_band;
  int sh_nt, sh_sk, sh_hwout, sh_wout, sh_hw, sh_fr, sh_ra, sh_band;
  int band;  // gemm8 tile order: m-tiles per band (gemm8.hip)
  // gemm8, temporal conv: m-tiles walk the FRAMES of one 256-pixel patch before the next patch (tmap_t = patches per frame, 0 =
  // off): the three taps of a tile are then the tiles its XCD runs beside it, not 16 tiles away (gemm8.hip)
  int tmap_t, sh_tm;
  unsigned mg_tm;
  float* stats;  // gemm8, optional: per 256-row tile and output channel {sum, sum of squares} of the stored values, fp32 [m_tiles][n_store][2]
  float* rowmom;  // gemm8, optional: per output row and n-tile {sum, sum of squares} of the stored values, fp32 [M][rowmom_ld][2]
  int rowmom_ld;
  // gemm8, Upsample2D + conv in SUB-PIXEL form (mvoc_gemm_desc.upsample == 2): the 3 x 3 conv on a nearest-2x-upsampled image is, per
  // output parity (a, b), a 2 x 2 conv on the SOURCE image with summed weights -- 4 taps instead of 9.  Rows [ph sp_rows, (ph + 1)
  // sp_rows) are the source pixels (img, i, j) of phase ph = 2 a + b; their outputs go to pixel (2 i + a, 2 j + b); w holds the four
  // phase kernels [4][N][4 cin]
  int subpx, sp_rows;
  int korder;   // gemm8, conv / temporal: K runs (64-channel chunk, tap, 64) instead of (tap, channel): mvoc_gemm_desc.k_order
  int epi_lds;  // outputs / residual are 16-byte addressable per 8-channel chunk: LDS-transposed epilogue
};


// eight-phase kernel (gemm8.hip): bx = 256 or 320 output channels per block, 256 pixels per block
int mvoc_launch_gemm8(const GemmArgs& a, int bx, hipStream_t s, int* bx_used = nullptr);  // *bx_used: the tile width that ran
