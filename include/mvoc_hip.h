/*
 * libmvoc_hip.so -- C ABI of the MI355X (gfx950) kernels behind MVOC's denoising hot path.
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes, is asynchronous on the HIP
 * stream passed as `void* stream` (a hipStream_t), never allocates or frees device memory, never
 * synchronises the device and keeps no pointer past return (graph-capture safe).  Return value: 0 on
 * success, -1 invalid argument, -2 unsupported shape, -3 HIP error; text via mvoc_last_error()
 * (thread-local).  The caller (PyTorch) owns all buffers.  fp16 = IEEE binary16.
 *
 * Canonical activation layout in HBM: channels-last "tokens" [B, F, H*W, C] (row-major, C contiguous),
 * i.e. a [rows = B*F*H*W, C] matrix.  Spatial ops see it as [B*F images][H*W][C]; temporal ops walk
 * the frame axis with row stride H*W.  The reference's NCHW <-> token <-> frame-major permute copies
 * (pnp_utils.py:185-189, 207-213, 434-437, 502-506) do not exist here.
 *
 * Each entry names the reference interface it replaces (paths relative to the reference checkout).
 */
#ifndef MVOC_HIP_H
#define MVOC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVOC_VERSION 100

int mvoc_version(void);
const char* mvoc_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Implicit-GEMM on MFMA (fp16 in, fp32 accumulate, fp16 out):  out[m, n] = epi( sum_k A(m,k) * W[n,k] )
 * Replaces every F.linear / F.conv2d(3x3, 1x1) / F.conv3d((3,1,1)) on the path:
 *   F.linear  pnp_utils.py:191,206,438,505,604-612,692,759-767,870; pipeline_i2vgen_xl.py:188,193,231
 *   conv2d    pnp_utils.py:939,968,1013 (ResnetBlock2D), :1112 (conv_out); pipeline_i2vgen_xl.py:284
 *   conv3d    pnp_utils.py:1048-1051 (TemporalConvLayer)
 * ------------------------------------------------------------------------------------------- */
enum { MVOC_A_PLAIN = 0, MVOC_A_CONV3X3 = 1, MVOC_A_TEMPORAL3 = 2 };
enum { MVOC_ACT_NONE = 0, MVOC_ACT_GEGLU = 1, MVOC_ACT_SILU = 2, MVOC_ACT_GELU = 3 };

typedef struct mvoc_gemm_desc {
  /* operands */
  const void* a;        /* activation source 1 (fp16) */
  const void* a2;       /* optional source 2 for a channel-concat input (cat([x, skip], dim=1)); NULL if none */
  const void* w;        /* weights [n_pad][k] fp16, k contiguous; for conv k = tap*cin + c (tap-major; k_order = 1: chunk-major, below) */
  void* out;            /* [m][ldo] fp16 */
  const void* bias;     /* [n] fp16 or NULL (GEGLU: packed like w rows) */
  const void* rowadd;   /* optional [m / rowadd_div][ld_rowadd] fp16 added per output row (time embedding) */
  const void* resid;    /* optional residual [m][ldr] fp16 added after bias/act */
  int64_t m, n, k;      /* n = rows of w (multiple of 32); k multiple of 32 */
  int32_t n_store;      /* columns actually written (<= n; GEGLU: n/2) */
  int32_t ldo, ldr, ld_rowadd, rowadd_div;
  int32_t a_mode;       /* MVOC_A_* */
  int32_t lda, lda2;    /* row strides (elements) of a / a2 */
  int32_t c1;           /* channels taken from `a` per tap (rest from a2); == cin when a2 == NULL */
  int32_t cin;          /* conv / temporal: channels per tap (c1 + c2); plain: == k */
  /* conv3x3 (pad 1): output pixel grid [nimg][hout][wout]; source image [hsrc][wsrc]; if upsample != 0 the
   * conv runs on the nearest-upsampled [hup][wup] view of the source (Upsample2D folded into the gather).
   * upsample == 2: the SUB-PIXEL form of an exact 2x upsample (hup == 2 hsrc, wup == 2 wsrc): the three taps of a row of the 3 x 3
   * kernel fall on only two source rows (columns likewise), so per output parity (a, b) the conv is a 2 x 2 conv on the source
   * image -- 4 taps of matrix work instead of 9.  `w` then holds the four parity kernels [4 = 2 a + b][n][dy][dx][cin] (k = 4 cin,
   * tap (dy, dx) of parity a: dy = 0 <- ky 0, dy = 1 <- ky 1 + ky 2 for a = 0; dy = 0 <- ky 0 + ky 1, dy = 1 <- ky 2 for a = 1; columns
   * likewise: mvoc_amd.unet.pack_conv3x3_subpixel) and the source pixel (i, j) of parity (a, b) is output pixel (2 i + a, 2 j + b).
   * Plain epilogue (bias) only, one source, nimg hsrc wsrc a multiple of 256, no split-K, no chan_sums. */
  int32_t nimg, hout, wout, hsrc, wsrc, stride, upsample, hup, wup;
  /* temporal3 (pad 1 over frames): rows are [nvid][frames][hw] */
  int32_t frames, hw;
  int32_t act;          /* MVOC_ACT_* */
  int32_t tile;         /* 0 = auto; otherwise force a tile config id (tuning) */
  int32_t split_k;      /* 0 = auto, 1 = off, n = force n K-slices (needs workspace; direct-to-LDS tiles only) */
  void* workspace;      /* optional fp32 scratch for split-K partial slabs: split_k * m * n * 4 bytes */
  size_t workspace_bytes;
  /* LayerNorm folded into the GEMM (F.layer_norm feeding F.linear at pnp_utils.py:250/296/322 -> :604-612, :335):
   * `a` holds the RAW rows, `w` holds gamma-scaled weights W' = W * gamma; the epilogue forms
   * rstd*(acc - mean*ln_rowsum[n]) + ln_bias[n] with the rows' {mean, rstd} from `ln_stats` (REQUIRED in this mode).
   * ln_rowsum = sum_k W'[n,k], ln_bias = beta @ W^T + bias, both fp32 [n]; `bias` is ignored in this mode.  NULL = off. */
  const void* ln_rowsum;
  const void* ln_bias;
  float ln_eps;
  int32_t pad_mode;     /* conv3x3: 0 = zero padding 1 on every side (F.conv2d(padding=1)); 1 = one row / column of zeros at the
                           BOTTOM / RIGHT only (F.pad(x, (0,1,0,1)) + conv2d(padding=0, stride=2): the downsamplers of the VAE
                           encoder, diffusers Downsample2D(padding=0)) */
  const void* ln_stats; /* {mean, rstd} per row, fp32 [m][2], from mvoc_row_stats_f16 (two-pass variance, one read of the rows, shared
                           by every n-tile) or mvoc_row_stats_from_moments_f32.  REQUIRED with ln_rowsum: the GEMM kernels take no
                           row statistics themselves. */
  void* chan_sums;      /* optional REQUEST, fp32 [m / 256][n_store][2] (honoured only when m is a multiple of 256): per 256-row slab
                           and output channel, the sum and the
                           sum of squares of the values this call stores (fp16-rounded, residual included) -- the first pass of the
                           GroupNorm that reads `out` next (F.group_norm at pnp_utils.py:909-910, 953-965, 1048-1051, 185-188),
                           taken from the producer's epilogue instead of a separate read of the tensor.  Written only when the
                           launch runs on the eight-phase tiles without split-K and without activation: ask
                           mvoc_gemm_chan_sums_written() after the call; when it answers 0 the buffer is untouched and the
                           consumer computes its own statistics (mvoc_groupnorm_f16 without chan_sums).  These are ONE-PASS moments per
                           slab (fp32-exact products summed in fp32); the consumers form M2 = sumsq - sum * mean per slab and channel
                           group and Chan-merge the slabs, so cancellation is bounded by one slab's |mean| / sigma (as row_moments
                           below: 2e-3 relative on rstd at |mean| / sigma = 150, far from this network's activations). */
  void* row_moments;    /* optional REQUEST, fp32 [m][row_moments_ld][2]: per output ROW and n-tile of the launch, the sum and the sum of
                           squares of the values this call stores over the tile's channels -- the statistics of the LayerNorm that
                           reads `out` next (F.layer_norm at pnp_utils.py:250-257, 296, 322), taken from the producer's epilogue
                           instead of a pass over the tensor (mvoc_row_stats_f16).  Written only by the eight-phase tiles without
                           split-K and with n_store == n: ask mvoc_gemm_row_moments_written() after the call -- 0: untouched;
                           otherwise the tile width w (256 or 320): entry [r][t] covers channels [t w, min(n, (t + 1) w)), and
                           mvoc_row_stats_from_moments_f32 turns the entries of a row into its {mean, rstd}. */
  int32_t row_moments_ld; /* entries per row of row_moments: >= ceil(n / 256) */
  int32_t concurrency;  /* scheduling hint, 0 / 1 = the launch has the chip to itself: the caller runs this many independent launches of
                           this shape at the same time on other streams (the job's per-object inversions, inverse.py:136-190, as
                           concurrent loops), so an under-filled grid need not be split over K to fill the chip.  Clamped to 8.  It
                           changes the tile / split-K choice, hence the summation order of a launch (results differ from the unhinted
                           launch by fp16 rounding of another order, never by more); per call, no process-wide state. */
  int32_t k_order;      /* conv3x3 / temporal3 only.  0: `w` is tap-major, k = tap * cin + c (above).  1: CHANNEL-CHUNK-major with the taps
                           innermost, k = (c / 64) * (ntaps * 64) + tap * 64 + c % 64 (ntaps = 9 / 3; mvoc_amd.ops.chunk_major_weights):
                           the nine (three) taps of one 64-channel slab of the source are consumed back to back, so the re-reads a block
                           makes of its own pixel rows (each source line is read once per tap: F.conv2d at pnp_utils.py:939, 968, conv3d at
                           :1042-1057) are ~50 KB apart instead of a whole tap's cin * 256 rows and hit the XCD's L2.  Same products, same
                           fp32 sums in another order (exact on integer operands).  A form of the 320-WIDE eight-phase tile (it runs on that
                           tile whatever `tile` says): n a multiple of 320, cin, c1 multiples of 64, conv stride 1 / pad 1 on a
                           same-size source or temporal3, no upsample, no activation, no LayerNorm fold, no split-K, m >= 1024, 16-byte
                           addressable operands < 2 GB: anything else is an error (-2), not a fall-back -- the caller holds the
                           tap-major weights for those launches. */
} mvoc_gemm_desc;

int mvoc_gemm_f16(const mvoc_gemm_desc* d, void* stream);
/* 1 when this thread's most recent mvoc_gemm_f16 call wrote its descriptor's chan_sums */
int mvoc_gemm_chan_sums_written(void);
/* tile width (256 / 320) when this thread's most recent mvoc_gemm_f16 call wrote its descriptor's row_moments, else 0 */
int mvoc_gemm_row_moments_written(void);
/* {mean, rstd} per row (fp32 [rows][2], the layout of mvoc_row_stats_f16 / ln_stats) from a producer's row_moments: the n-tiles'
 * {sum, sum of squares} become {count, mean, M2} per tile and are Chan-merged in tile order. */
int mvoc_row_stats_from_moments_f32(const void* moments, int64_t rows, int32_t ld, int32_t n, int32_t tile_w, float eps,
                                    void* out_stats, void* stream);
/* scratch for the deterministic split-K form of a launch: the most the auto policy uses is 8 slices of m*n fp32
 * partials; 0 when the policy would never split this shape (m > 8192 or k < 2048).  Passing no workspace is always valid:
 * the launch then runs unsplit. */
size_t mvoc_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k);

/* ---------------------------------------------------------------------------------------------
 * Attention, head_dim 64, softmax(Q K^T / 8) V, no mask (F.scaled_dot_product_attention at
 * pnp_utils.py:684-686, 862-864 and the stock AttnProcessor2_0 sites).
 * q/k/v/out element (b, t, h, d) lives at base + b*bs + t*ts + h*head_dim + d  (elements).
 * kv batch index = b / kv_bdiv (cross-attention context shared by all frames of a sample).
 * ------------------------------------------------------------------------------------------- */
typedef struct mvoc_attn_desc {
  const void *q, *k, *v;
  void* out;
  int64_t q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
  int32_t nbatch, heads, tq, tk, kv_bdiv;
  /* the three below default (0) to the UNet's form: head_dim 64, no mask, scale 1/8.  head_dim 96 with scale 1/sqrt(80)
   * serves CLIP ViT-H's 80-wide heads (projection weights zero-padded per head); causal: key <= query (CLIP text tower,
   * pipeline_i2vgen_xl.py:552-737 encode_prompt -> CLIPTextModel) */
  int32_t head_dim, causal;
  float scale;
  /* paired form (head_dim 64): a second value tensor / output with the layout of v / out that attends with the SAME q and k --
   * the PnP destination chunks after Q/K injection (pnp_utils.py:664-668 assigns one blended q / k to both the unconditional and
   * the conditional chunk; only their v differ).  NULL: plain attention.  Results equal two plain calls bit for bit. */
  const void* v2;
  void* out2;
  /* kernel choice of THIS call (head_dim 64, no causal mask): 0 by key count (the software-pipelined kernel from 2 048 keys up), 1 the
   * phase kernel always, 2 the pipelined kernel wherever it applies.  Speed only -- both kernels return the same bits (the tests
   * compare them).  Replaces round 4's process-wide setter (flash_pipelined): the library keeps no mutable state between calls
   * (MVOC_FLASH3=0|1 in the environment changes the default of 0 at load time, for diagnostics). */
  int32_t pipelined;
} mvoc_attn_desc;

/* spatial self-attention and image/text cross-attention (flash-style, K/V tiles LDS-staged) */
int mvoc_flash_attn_f16(const mvoc_attn_desc* d, void* stream);
/* temporal self-attention: one sequence per (sample, pixel), tq == tk == frames <= 32; "t" walks frames
 * (ts = H*W*C for the canonical layout), "b" walks sample*pixel via (b / hw)*bs + (b % hw)*ps */
typedef struct mvoc_tattn_desc {
  const void *q, *k, *v;
  void* out;
  int64_t q_bs, q_ps, q_ts, k_bs, k_ps, k_ts, v_bs, v_ps, v_ts, o_bs, o_ps, o_ts;
  int32_t nsample, hw, heads, frames;
} mvoc_tattn_desc;
int mvoc_temporal_attn_f16(const mvoc_tattn_desc* d, void* stream);

/* Activation-stationary linear for K in {64, 128, 320} (the 320-channel level's projections: F.linear at pnp_utils.py:191, 206,
 * 335, 438, 505, 604-612, 692): out[m][n] = epi(x[m][:] . W[n][:] + c[n]).  A wave keeps 32 consecutive rows of x in registers;
 * the weights AND the per-channel constants stream through LDS from a packed copy (mvoc_amd.unet.pack_xs_weights):
 *   wp [n/32 tiles][k/16 + 1 pieces][lane < 64][8 fp16]
 *      piece s < k/16: element = W[32 tile + (lane & 31)][16 s + 8 (lane >> 5) + j]  (MFMA fragment order)
 *      piece k/16    : its first 128 bytes are the tile's 32 constants c[32 tile + i] as fp32 (bias, or beta @ W^T + bias)
 * normalize != 0 folds a LayerNorm in front (rows normalised in registers; W must be gamma-scaled, c = beta @ W^T + bias).
 * act as for mvoc_gemm_f16 (GEGLU: weight rows in (value, gate) blocks of 32, n/2 outputs, no residual).  x is [m][k]
 * contiguous; out / resid rows 16-byte addressable per 8 channels; n_store may only trim the last 32-channel tile. */
typedef struct mvoc_xs_desc {
  const void* x;
  const void* wp;
  const void* resid;
  void* out;
  int64_t m;
  int32_t n, k, n_store, ldo, ldr, act, normalize;
  float ln_eps;
  int64_t wp_set_rows;  /* 0: one weight stream for every row.  > 0: `wp` holds m / wp_set_rows consecutive streams, set j for rows
                           [j wp_set_rows, (j + 1) wp_set_rows) -- the per-sample weights of a folded GroupNorm
                           (mvoc_groupnorm_fold_xs_f16); a multiple of 256 */
} mvoc_xs_desc;
int mvoc_xs_linear_f16(const mvoc_xs_desc* d, void* stream);


/* Fused front half of a temporal self-attention at the finest level: LayerNorm -> to_q/to_k/to_v -> attention over the frame
 * axis, Q/K/V never written to HBM (TransformerTemporalModel's attn1 / attn2: pnp_utils.py:170-220, 222-346, 720-887; replaces
 * mvoc_row_stats_f16 + the fused-QKV mvoc_gemm_f16 + mvoc_temporal_attn_f16 where no PnP Q/K injection is scheduled).
 *   x   [nsample*frames*hw][c] raw rows of the canonical layout (c = heads*64 in {64, 128, 320}, frames in {8, 16, 32})
 *   wp  the gamma-scaled [3c][c] projection W' = cat(Wq, Wk, Wv) * gamma re-ordered in MFMA fragment order:
 *       [head][tile: q0 k0 q1 k1 v0 v1][k16 step s < c/16][lane < 64][8 fp16], element = W'[row0 + (lane & 31)][16 s + 8 (lane >> 5) + j]
 *       with row0 = 64 head + {0, c, 32, c + 32, 2c, 2c + 32}[tile]   (mvoc_amd.unet.pack_tfused_weights)
 *   ln_rowsum / ln_bias fp32 [3c] as for the folded GEMM; out [rows][c] = heads concatenated, before to_out */
typedef struct mvoc_tfused_desc {
  const void* x;
  const void* wp;
  const void* ln_rowsum;
  const void* ln_bias;
  void* out;
  int32_t nsample, frames, hw, c, heads;
  float ln_eps;
} mvoc_tfused_desc;
int mvoc_temporal_qkv_attn_f16(const mvoc_tfused_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm (+SiLU) on channels-last rows; F.group_norm + F.silu at pnp_utils.py:909-910, 953-965 (4-D,
 * statistics per image), :188 and TemporalConvLayer (5-D, statistics per video), :430, and
 * pipeline_i2vgen_xl.py:351-352.  Input may be a 2-source channel concat.  Deterministic (no atomics).
 *   workspace floats: nsample * (nchunk*3 + 2) * groups  -> mvoc_groupnorm_workspace_bytes
 * ------------------------------------------------------------------------------------------- */
typedef struct mvoc_gn_desc {
  const void* x;        /* [nsample*rows_per_sample][c1] fp16 */
  const void* x2;       /* optional second source [..][c - c1] */
  const void* gamma;    /* [c] fp16 */
  const void* beta;     /* [c] fp16 */
  void* out;            /* [nsample*rows_per_sample][c] fp16 */
  void* workspace;
  size_t workspace_bytes;
  int32_t nsample, rows_per_sample, c, c1, groups, silu;
  float eps;
  /* optional: the statistics pass taken from the producers of x (and x2): what mvoc_gemm_f16 wrote into its descriptor's
   * chan_sums when it stored these rows -- fp32 [nsample * rows_per_sample / 256][c1][2] (and [..][c - c1][2]).  Needs
   * rows_per_sample % 256 == 0 and c1 % (c / groups) == 0; both sources' sums when x2 is given.  NULL: the statistics are
   * read from x. */
  const void* chan_sums;
  const void* chan_sums2;
} mvoc_gn_desc;
size_t mvoc_groupnorm_workspace_bytes(int32_t nsample, int32_t rows_per_sample, int32_t c, int32_t groups);
int mvoc_groupnorm_f16(const mvoc_gn_desc* d, void* stream);
/* GroupNorm folded into the linear that reads it (GN -> proj_in, no activation in between: pnp_utils.py:185-191, 433-438):
 * takes the statistics of d's rows as mvoc_groupnorm_f16 does (d->x, nsample, rows_per_sample, c, groups, eps, gamma, beta,
 * workspace, chan_sums; single source; d->out unused) and writes, per SAMPLE, the weight stream of
 * mvoc_xs_linear_f16 for W'_s = W gamma rstd_s and c'_s = bias + W (beta - gamma mean_s rstd_s): wp_sets
 * [nsample][n / 32][k / 16 + 1][512] fp16.  The consumer then runs on the RAW rows with wp_set_rows = rows_per_sample; the
 * normalised tensor is never written.  w [n][k] fp16 row-major (k == c), bias [n] fp16 or NULL. */
int mvoc_groupnorm_fold_xs_f16(const mvoc_gn_desc* d, const void* w, const void* bias, int32_t n, int32_t k, void* wp_sets,
                               void* stream);

/* The same GroupNorm in two halves, for a sample whose rows are spread over several GPUs (frame-axis shard of one
 * long clip, SURVEY 8e / BASELINE configs[3]: the 5-D norms at pnp_utils.py:185-188 and inside TemporalConvLayer,
 * pnp_utils.py:1048-1051, take their statistics over ALL frames and pixels):
 *   moments : this rank's rows -> moments [nsample][groups][3] fp32 {count, mean, M2}   (gamma/beta/out unused)
 *   exchange: the caller all-gathers the triples (RCCL; 12*nsample*groups bytes per rank) -> parts [nparts][nsample][groups][3]
 *   apply   : Chan-combines the parts in index order (every rank gets bit-identical mean/rstd) and normalises this rank's rows.
 * With nparts == 1 the pair reproduces mvoc_groupnorm_f16 bit for bit. */
int mvoc_groupnorm_moments_f16(const mvoc_gn_desc* d, void* moments, void* stream);
int mvoc_groupnorm_apply_moments_f16(const mvoc_gn_desc* d, const void* parts, int32_t nparts, void* stream);

/* per-row {mean, rstd} (fp32 [rows][2]) of a [rows, c] fp16 matrix: the statistics half of F.layer_norm for GEMMs that
 * fold the normalisation into their epilogue (mvoc_gemm_desc.ln_*) */
/* c % 8 == 0, c <= 2048 (a row lives in one wave's registers: 4 x 16 bytes per lane) */
int mvoc_row_stats_f16(const void* x, void* stats, int64_t rows, int32_t c, float eps, void* stream);

/* LayerNorm over the last dim (F.layer_norm at pnp_utils.py:250,296,322), eps 1e-5, affine */
int mvoc_layernorm_f16(const void* x, const void* gamma, const void* beta, void* out, int64_t rows, int32_t c,
                       float eps, void* stream);

/* ---------------------------------------------------------------------------------------------
 * PnP masked blend + scatter.  For every (frame f, pixel p, channel c):
 *     inj = base;  for j in objects:  inj = inj*(1-m_j) + obj_j*m_j     (three fp16-rounded ops, in this order)
 *     chunk[n-2] = chunk[n-1] = inj
 * with chunks positional [bg, obj_1..obj_k, uncond, cond] (k <= 4; the reference hard-codes k = 2, n = 5:
 * pnp_utils.py:592,747,784,972,1061,1115), base = chunk n-1 (or chunk 0 when
 * base_chunk0 != 0), m_j = mask_j[f, nearest(p)] (fp16 values: exact {0,1} for bool masks, k/255 for soft masks).
 * BIT-EXACT against the reference arithmetic (not a select: -0.0 / inf / NaN propagate as in x*(1-m)+y*m).
 *   tokens: element (chunk, f, p, c) at x + chunk*chunk_stride + f*f_stride + p*p_stride + c   (c contiguous)
 *           -> spatial Q/K [5F, HW, C]   (pnp_utils.py:624-672):  chunk_stride=F*HW*C, f_stride=HW*C, p_stride=C
 *           -> temporal Q/K [5HW, F, C]  (pnp_utils.py:778-850):  chunk_stride=HW*F*C, f_stride=C,    p_stride=F*C
 *           -> canonical [5, F, HW, C] features (resnet / temporal-conv / conv_out injection)
 *   nchw:   x [5F, C, H, W] (pnp_utils.py:970-1004, 1059-1082, 1114-1146): p contiguous
 * masks: [nobj][F][mh][mw] fp16, nearest-resized to (H, W) like F.interpolate(mode='nearest').
 * Up to two tensors (Q and K) are processed by one launch (x2 may be NULL).
 * ------------------------------------------------------------------------------------------- */
typedef struct mvoc_pnp_desc {
  void* x;
  void* x2;
  const void* masks;
  int64_t chunk_stride, f_stride, p_stride;
  int32_t nobj, frames, height, width, channels, mask_h, mask_w, base_chunk0;
  int32_t ndst;  /* trailing destination chunks: 0 or 2 = [uncond, cond] (the reference's layout), 1 = [cond] only: the
                    classifier-free-guidance-off batch [bg, obj_1..obj_k, cond] (n = k + 2) */
} mvoc_pnp_desc;
int mvoc_pnp_blend_scatter_tokens(const mvoc_pnp_desc* d, void* stream);
int mvoc_pnp_blend_scatter_nchw(const mvoc_pnp_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Loop glue on [B,4,F,h,w] fp16 latents, BIT-EXACT vs the reference's eager fp16 op chain.
 *   cfg + DDIM / inverse-DDIM update (pipeline_i2vgen_xl.py:1187-1199, 1713-1731, 1967-1984;
 *   diffusers DDIMScheduler.step / DDIMInverseScheduler.step, v_prediction, eta = 0):
 *     v   = uncond + g*(cond-uncond)            (skipped when uncond == NULL: v = cond)
 *     x0  = sa*x - sb*v ; eps = sa*v + sb*x ; out = sp*x0 + sq*eps
 *   coef = {sa, sb, sp, sq, g} fp32 on the DEVICE (so a captured graph can be replayed per step)
 * ------------------------------------------------------------------------------------------- */
int mvoc_ddim_step_f16(const void* x, const void* v_uncond, const void* v_cond, const float* coef_dev, void* out,
                       int64_t n, void* stream);
/* latent noise fusion (pipeline_i2vgen_xl.py:1644-1663); objs/masks: nobj device pointers packed in arrays
 * of contiguous [n] fp16 tensors: obj[j] at objs + j*n, mask[j] at masks + j*n */
int mvoc_latent_fusion_f16(const void* latents, const void* bg, const void* objs, const void* masks, void* out,
                           int32_t nobj, int64_t n, double mix_ratio, int32_t obj_random_noise_fusion, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stem / small ops (pipeline_i2vgen_xl.py:166-290, 351-357).
 * ------------------------------------------------------------------------------------------- */
/* sinusoidal Timesteps(dim, flip_sin_to_cos=True, shift 0): out[b] = [cos | sin] rounded to fp16; t fp32 on device */
int mvoc_timestep_embedding_f16(const float* t_dev, int32_t nb, int32_t dim, void* out, void* stream);
/* y = silu(x) / gelu(x) elementwise, fp16 */
int mvoc_act_f16(const void* x, void* out, int64_t n, int32_t act, void* stream);
/* out = a + b (fp16, rounded once) */
int mvoc_add_f16(const void* a, const void* b, void* out, int64_t n, void* stream);
/* direct 3x3 conv (pad 1) on channels-last images for tiny channel counts; w [cout][3][3][cin] fp16 */
int mvoc_conv3x3_small_f16(const void* x, const void* w, const void* bias, void* out, int32_t nimg, int32_t h,
                           int32_t wd, int32_t cin, int32_t cout, int32_t stride, int32_t silu, void* stream);
/* AdaptiveAvgPool2d on channels-last images */
int mvoc_adaptive_avgpool_f16(const void* x, void* out, int32_t nimg, int32_t h, int32_t w, int32_t c, int32_t oh,
                              int32_t ow, void* stream);
/* [B,C,F,h,w] (reference boundary layout) -> channels-last [B,F,h*w,ldo] at channel offset coff */
int mvoc_ncfhw_to_tokens_f16(const void* x, void* out, int32_t b, int32_t c, int32_t f, int32_t hw, int32_t ldo,
                             int32_t coff, void* stream);
/* channels-last [B,F,h*w,ld] (first c channels) -> [B,C,F,h,w] */
int mvoc_tokens_to_ncfhw_f16(const void* x, void* out, int32_t b, int32_t c, int32_t f, int32_t hw, int32_t ld,
                             void* stream);
/* I2VGenXLTransformerTemporalEncoder (dim 4, 2 heads x 4) fused: LN -> self-attn over frames (+res) -> FF(gelu)
 * (+res) on x [B,F,hw,4] channels-last, written into out [B,F,hw,ldo] at channel offset coff.
 * params: fp16 blob {ln_g[4], ln_b[4], wq[8][4], wk[8][4], wv[8][4], wo[4][8], bo[4], w1[16][4], b1[16], w2[4][16], b2[4]} */
int mvoc_temporal_encoder4_f16(const void* x, const void* params, void* out, int32_t b, int32_t f, int32_t hw,
                               int32_t ldo, int32_t coff, void* stream);

/* ---------------------------------------------------------------------------------------------
 * VAE encode / decode either side of the loops (pipeline_i2vgen_xl.py:771-791 decode_latents, :860-890
 * prepare_image_latents, :893-920 encode_vae_video -> diffusers AutoencoderKL).  The convolutions, GroupNorms and the
 * mid-block attention's projections run through the entries above (mvoc_gemm_f16 incl. pad_mode = 1 for the encoder's
 * Downsample2D(padding=0), mvoc_groupnorm_f16); what they need in addition:
 * ------------------------------------------------------------------------------------------- */
/* 1x1 conv over a handful of channels on channels-last rows: out[r][o] = sum_c x[r][c] w[o][c] + bias[o]
 * (AutoencoderKL.quant_conv 8 -> 8, post_quant_conv 4 -> 4); cin, cout <= 64 */
int mvoc_conv1x1_small_f16(const void* x, const void* w, const void* bias, void* out, int64_t rows, int32_t cin,
                           int32_t cout, void* stream);
/* in-place softmax over the rows of a contiguous [rows][cols] fp16 matrix (fp32 math): the score matrix of the VAE's
 * single-head attention (head_dim 512; Attention(residual_connection=True) in UNetMidBlock2D), cols % 8 == 0 */
int mvoc_softmax_rows_f16(void* x, int64_t rows, int32_t cols, void* stream);
/* vae.encode(x).latent_dist.sample() (diffusers DiagonalGaussianDistribution): out = mean + exp(0.5*clamp(logvar,-30,20)) * noise,
 * each eager fp16 op rounded; and python-float * fp16 tensor (latents * scaling_factor, 1/scaling_factor * latents:
 * pipeline_i2vgen_xl.py:772, 869, 911) with the reference's fp32-then-fp16 double rounding */
int mvoc_gaussian_sample_f16(const void* mean, const void* logvar, const void* noise, void* out, int64_t n, void* stream);
int mvoc_scale_f16(const void* x, void* out, int64_t n, double scale, void* stream);
/* [n][c][hw] images / latents (reference NCHW) <-> channels-last rows [n*hw][c] (ld >= c on the way back) */
int mvoc_image_to_tokens_f16(const void* x, void* out, int32_t n, int32_t c, int32_t hw, void* stream);
int mvoc_tokens_to_image_f16(const void* x, void* out, int32_t n, int32_t c, int32_t hw, int32_t ld, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Mask preprocessing on the device (reference i2vgen-xl/utils.py:92-154): after PNG decode + convert("L") on the
 * host, PIL's `mask.resize((W//8, H//8))` (BICUBIC, the 8-bit fixed-point path: 22 fractional bits, horizontal pass rounded to
 * uint8, then vertical) reproduced bit for bit from host-built coefficient tables (per output index {xmin, n} and n int32
 * weights, row pitch ksize), then float = v/255 in fp16 and bool = v > 10 (cv.threshold(.., 10, 255) / 255 -> bool).
 *   in [n][H][W] u8, tmp [n][H][w] u8, out [n][h][w] u8;  float_mask fp16 / bool_mask u8 of the same element count
 * ------------------------------------------------------------------------------------------- */
int mvoc_mask_resize_u8(const void* in, void* tmp, void* out, int32_t n, int32_t H, int32_t W, int32_t h, int32_t w,
                        const int32_t* bounds_h, const int32_t* kk_h, int32_t ksize_h, const int32_t* bounds_v,
                        const int32_t* kk_v, int32_t ksize_v, void* stream);
int mvoc_mask_finish(const void* v, void* float_mask, void* bool_mask, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Conditioning prep: the embedding stages of the CLIP towers the reference runs through transformers
 * (pipeline_i2vgen_xl.py:739-769 `_encode_image` -> CLIPVisionModelWithProjection, :552-737 `encode_prompt` -> CLIPTextModel);
 * the transformer stacks themselves are mvoc_gemm_f16 / mvoc_layernorm_f16 / mvoc_flash_attn_f16 (head_dim 96 | causal).
 *   clip_patches: NCHW fp16 pixels [nimg,3,size,size] -> im2col rows [nimg*(size/patch)^2, kpad] of the patch-embedding conv
 *                 (k = (c, py, px) like Conv2d's weight.view(N, -1); columns >= 3*patch^2 are zero)
 *   clip_embed  : out[row] = src(row) + pos[row % t]; src = table[ids[row]] (text: token + position embedding), or with
 *                 ids == NULL the class embedding at t == 0 and patch row (row/t)*(t-1) + (row%t) - 1 otherwise (vision)
 * ------------------------------------------------------------------------------------------- */
int mvoc_clip_patches_f16(const void* pixels, void* out, int32_t nimg, int32_t size, int32_t patch, int32_t kpad, void* stream);
int mvoc_clip_embed_f16(const void* table, const int32_t* ids, const void* cls, const void* pos, void* out, int64_t rows,
                        int32_t t, int32_t c, void* stream);

/* ---------------------------------------------------------------------------------------------
 * RCCL exchanges of the frame-axis shard (SURVEY 8b `allgather_frames`, 8e / BASELINE configs[3]: one long clip over the GPUs
 * of a node; the reference has no collective -- this is new work north_star asks for).  One process per GPU; rank 0 makes a
 * 128-byte id (mvoc_comm_unique_id), the host distributes it by any means, every rank calls mvoc_comm_init.  The exchanges
 * are asynchronous on `stream` and capturable; RCCL is resolved at run time (the copy PyTorch-ROCm already loaded, else
 * librccl.so.1), so the library loads without it and these entries then return -3.
 *   allgather: rank r's `bytes_per_rank` bytes land at recv + r*bytes_per_rank on every rank (temporal attention's K/V form)
 *   alltoall : send + j*bytes_per_peer goes to rank j, recv + i*bytes_per_peer came from rank i (frame shard <-> pixel shard)
 * ------------------------------------------------------------------------------------------- */
/* packing around the exchanges: out rows contiguous over (i0,i1,i2,i3) < dims4, source row = sum_k i_k * strides4[k]
 * (rows of c fp16, c % 8 == 0) -- frame shard [B,F/N,HW,C] <-> per-peer blocks <-> pixel shard [B,F,HW/N,C] */
int mvoc_permute_rows_f16(const void* x, void* out, const int64_t* dims4, const int64_t* strides4, int32_t c, void* stream);
int mvoc_comm_unique_id(void* id128);
int mvoc_comm_init(const void* id128, int32_t rank, int32_t world, void** comm);
int mvoc_comm_destroy(void* comm);
int mvoc_allgather_frames(void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);
int mvoc_alltoall_frames(void* comm, const void* send, void* recv, size_t bytes_per_peer, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-kernel-family timing with HIP events on the launch stream (bench.py roofline leg).
 * ------------------------------------------------------------------------------------------- */
enum { MVOC_FAM_GEMM = 0, MVOC_FAM_FLASH = 1, MVOC_FAM_TATTN = 2, MVOC_FAM_GN = 3, MVOC_FAM_LN = 4, MVOC_FAM_PNP = 5,
       MVOC_FAM_MISC = 6, MVOC_FAM_TFUSED = 7, MVOC_FAM_COUNT = 8 };
/* work unit per family: flops for GEMM / FLASH / TFUSED (MFMA-bound), bytes for the others (HBM-bound) */
int mvoc_prof_enable(int on);                  /* 1: bracket every launch with hipEvents (not capture-safe) */
int mvoc_prof_collect(double* ms_per_family, int64_t* launches_per_family, double* flops_or_bytes_per_family);
int mvoc_prof_reset(void);
/* enqueue a kernel that busy-waits `us` microseconds (lets the host run ahead so event brackets see no launch gaps) */
int mvoc_delay_us(int64_t us, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVOC_HIP_H */
