// PnP masked blend + scatter, latent noise fusion and the fused CFG + DDIM step.  All HBM-bound elementwise
// work; every arithmetic op is rounded to fp16 exactly where the reference's eager fp16 op chain rounds, so
// the results are BIT-EXACT against the reference arithmetic (fp32 compute + one rounding per op equals a
// correctly rounded fp16 op: 24 >= 2*11+2 significand bits).
//
// Traffic per injection site (spatial Q/K at the 320-channel level, B=5, F=16, 64x64): read base + 2 objects,
// write 2 destination chunks, for Q and K = 10 x 41.9 MB + masks (1 fp16 per (frame, pixel), never expanded to
// [F,H,W,C]) = 419.6 MB -- the reference moves several times that through rearrange/repeat copies.
#include "common.h"

namespace {

struct PnpArgs {
  half_t* x[2];
  const half_t* masks;
  long chunk_stride, f_stride, p_stride;
  int nobj, frames, height, width, channels, mask_h, mask_w, base_chunk0;
  int ndst;      // trailing destination chunks: 2 = [uncond, cond] (the reference's CFG layout), 1 = [cond] (CFG off)
  float sy, sx;  // nearest-resize scales mask_h/height, mask_w/width (as F.interpolate computes them)
  long total;    // work items per tensor
};

__device__ __forceinline__ float blend16(float inj, float obj, float m) {
  const float om = r16(1.0f - m);
  return r16(r16(inj * om) + r16(obj * m));
}

__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
  return min((int)floorf(dst * scale), in_size - 1);
}

// item = (f, p, c8): 8 consecutive channels of one (frame, pixel)
__global__ __launch_bounds__(256) void pnp_tokens_kernel(const PnpArgs p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  half_t* x = p.x[blockIdx.y];
  const int c8n = p.channels >> 3;
  const int c8 = (int)(idx % c8n);
  const long fp = idx / c8n;
  const int hw = p.height * p.width;
  const int f = (int)(fp / hw), px = (int)(fp % hw);
  const int py = px / p.width, pxx = px - py * p.width;
  const int my = nearest_src(py, p.sy, p.mask_h), mx = nearest_src(pxx, p.sx, p.mask_w);
  const long off = (long)f * p.f_stride + (long)px * p.p_stride + c8 * 8;
  const int nchunk = p.nobj + 1 + p.ndst;
  const int basec = p.base_chunk0 ? 0 : nchunk - 1;
  const half8_t bv = *reinterpret_cast<const half8_t*>(x + basec * p.chunk_stride + off);
  float inj[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) inj[e] = (float)bv[e];
  for (int j = 0; j < p.nobj; ++j) {
    const float m = (float)p.masks[(((long)j * p.frames + f) * p.mask_h + my) * p.mask_w + mx];
    const half8_t ov = *reinterpret_cast<const half8_t*>(x + (j + 1) * p.chunk_stride + off);
#pragma unroll
    for (int e = 0; e < 8; ++e) inj[e] = blend16(inj[e], (float)ov[e], m);
  }
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)inj[e];
  if (p.ndst == 2) *reinterpret_cast<half8_t*>(x + (nchunk - 2) * p.chunk_stride + off) = o;
  *reinterpret_cast<half8_t*>(x + (nchunk - 1) * p.chunk_stride + off) = o;
}

// NCHW: x [(nobj+3)*F, C, H, W]; item = (f, c, p8) with VEC consecutive pixels
template <int VEC>
__global__ __launch_bounds__(256) void pnp_nchw_kernel(const PnpArgs p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  half_t* x = p.x[blockIdx.y];
  const int hw = p.height * p.width;
  const int pvn = hw / VEC;
  const int pv = (int)(idx % pvn);
  const long fc = idx / pvn;
  const int f = (int)(fc / p.channels);
  const long off = fc * hw + (long)pv * VEC;  // (f*C + c)*HW + p
  const long chunk = (long)p.frames * p.channels * hw;
  const int nchunk = p.nobj + 1 + p.ndst;
  const int basec = p.base_chunk0 ? 0 : nchunk - 1;
  float inj[VEC];
  half_t tmp[VEC];
  if constexpr (VEC == 8) {
    *reinterpret_cast<half8_t*>(tmp) = *reinterpret_cast<const half8_t*>(x + basec * chunk + off);
  } else {
    tmp[0] = x[basec * chunk + off];
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) inj[e] = (float)tmp[e];
  for (int j = 0; j < p.nobj; ++j) {
    if constexpr (VEC == 8) {
      *reinterpret_cast<half8_t*>(tmp) = *reinterpret_cast<const half8_t*>(x + (j + 1) * chunk + off);
    } else {
      tmp[0] = x[(j + 1) * chunk + off];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int px = pv * VEC + e;
      const int py = px / p.width, pxx = px - py * p.width;
      const int my = nearest_src(py, p.sy, p.mask_h), mx = nearest_src(pxx, p.sx, p.mask_w);
      const float m = (float)p.masks[(((long)j * p.frames + f) * p.mask_h + my) * p.mask_w + mx];
      inj[e] = blend16(inj[e], (float)tmp[e], m);
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) tmp[e] = (half_t)inj[e];
  if constexpr (VEC == 8) {
    if (p.ndst == 2) *reinterpret_cast<half8_t*>(x + (nchunk - 2) * chunk + off) = *reinterpret_cast<half8_t*>(tmp);
    *reinterpret_cast<half8_t*>(x + (nchunk - 1) * chunk + off) = *reinterpret_cast<half8_t*>(tmp);
  } else {
    if (p.ndst == 2) x[(nchunk - 2) * chunk + off] = tmp[0];
    x[(nchunk - 1) * chunk + off] = tmp[0];
  }
}

int fill_args(const mvoc_pnp_desc* d, PnpArgs& a) {
  MVOC_REQUIRE(d && d->x && d->masks, -1, "pnp: null operand");
  MVOC_REQUIRE(d->nobj >= 1 && d->nobj <= 4, -2, "pnp: nobj %d not in [1,4]", d->nobj);
  MVOC_REQUIRE(d->frames > 0 && d->height > 0 && d->width > 0 && d->channels > 0 && d->mask_h > 0 && d->mask_w > 0, -1,
               "pnp: bad dims");
  a.x[0] = (half_t*)d->x;
  a.x[1] = (half_t*)d->x2;
  a.masks = (const half_t*)d->masks;
  a.chunk_stride = d->chunk_stride; a.f_stride = d->f_stride; a.p_stride = d->p_stride;
  a.nobj = d->nobj; a.frames = d->frames; a.height = d->height; a.width = d->width; a.channels = d->channels;
  a.mask_h = d->mask_h; a.mask_w = d->mask_w; a.base_chunk0 = d->base_chunk0;
  MVOC_REQUIRE(d->ndst >= 0 && d->ndst <= 2, -1, "pnp: ndst %d not in {0 (= 2), 1, 2}", d->ndst);
  a.ndst = d->ndst == 0 ? 2 : d->ndst;
  a.sy = (float)d->mask_h / (float)d->height;
  a.sx = (float)d->mask_w / (float)d->width;
  return 0;
}

// ---- loop glue -----------------------------------------------------------------------------------
// fp32-scalar x fp16-tensor products of the reference's eager chain (python float / 0-dim fp32 tensor times a
// half tensor) are formed in fp32, ROUNDED TO fp32, then rounded to fp16 -- two roundings.  Left to itself hipcc
// contracts `(half)(s * (float)h)` into v_fma_mixlo_f16, which rounds the exact 35-bit product once and differs
// by 1 ulp at ~2^-9 of the elements (observed on MI355X).  The empty asm pins the fp32 product in a VGPR.
__device__ __forceinline__ float smul16(float scalar, float x) {
  float p = scalar * x;
  asm volatile("" : "+v"(p));
  return r16(p);
}

__global__ __launch_bounds__(256) void ddim_step_kernel(const half_t* __restrict__ x, const half_t* __restrict__ vu,
                                                        const half_t* __restrict__ vc, const float* __restrict__ coef,
                                                        half_t* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float sa = coef[0], sb = coef[1], sp = coef[2], sq = coef[3], g = coef[4];
  const float xs = (float)x[i];
  float v = (float)vc[i];
  if (vu) {
    const float u = (float)vu[i];
    v = r16(u + smul16(g, r16(v - u)));
  }
  const float x0 = r16(smul16(sa, xs) - smul16(sb, v));
  const float eps = r16(smul16(sa, v) + smul16(sb, xs));
  const float dir = smul16(sq, eps);
  out[i] = (half_t)(smul16(sp, x0) + dir);
}

__global__ __launch_bounds__(256) void fusion_kernel(const half_t* __restrict__ lat, const half_t* __restrict__ bg,
                                                     const half_t* __restrict__ objs, const half_t* __restrict__ masks,
                                                     half_t* __restrict__ out, int nobj, long n, float mix, float omix,
                                                     int rnf) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float l = r16(smul16(mix, (float)lat[i]) + smul16(omix, (float)bg[i]));
  for (int j = 0; j < nobj; ++j) {
    const float m = (float)masks[(long)j * n + i];
    const float inv_obj = r16((float)objs[(long)j * n + i] * m);
    const float background = r16(l * r16(1.0f - m));
    float fusion = inv_obj;
    if (rnf) fusion = r16(smul16(mix, r16(l * m)) + smul16(omix, inv_obj));
    l = r16(background + fusion);
  }
  out[i] = (half_t)l;
}

}  // namespace

extern "C" int mvoc_pnp_blend_scatter_tokens(const mvoc_pnp_desc* d, void* stream) {
  PnpArgs a;
  if (int rc = fill_args(d, a)) return rc;
  MVOC_REQUIRE(d->channels % 8 == 0 && d->chunk_stride % 8 == 0 && d->f_stride % 8 == 0 && d->p_stride % 8 == 0, -2,
               "pnp tokens: channels/strides must be multiples of 8");
  a.total = (long)d->frames * d->height * d->width * (d->channels / 8);
  const long nblk = (a.total + 255) / 256;
  MVOC_REQUIRE(nblk < 0x7fffffffL, -2, "pnp tokens: grid too large");
  const int ntens = d->x2 ? 2 : 1;
  hipStream_t s = (hipStream_t)stream;
  const double elems = (double)d->frames * d->height * d->width * d->channels;
  MvocProfScope prof(MVOC_FAM_PNP, s, ntens * (elems * 2.0 * (d->nobj + 1 + a.ndst) + 2.0 * d->nobj * d->frames * d->height * d->width));
  hipLaunchKernelGGL(pnp_tokens_kernel, dim3((unsigned)nblk, ntens), dim3(256), 0, s, a);
  return mvoc_check_launch("pnp_tokens_kernel");
}

extern "C" int mvoc_pnp_blend_scatter_nchw(const mvoc_pnp_desc* d, void* stream) {
  PnpArgs a;
  if (int rc = fill_args(d, a)) return rc;
  const long hw = (long)d->height * d->width;
  const bool vec = hw % 8 == 0;
  a.total = (long)d->frames * d->channels * (vec ? hw / 8 : hw);
  const long nblk = (a.total + 255) / 256;
  MVOC_REQUIRE(nblk < 0x7fffffffL, -2, "pnp nchw: grid too large");
  const int ntens = d->x2 ? 2 : 1;
  hipStream_t s = (hipStream_t)stream;
  const double elems = (double)d->frames * hw * d->channels;
  MvocProfScope prof(MVOC_FAM_PNP, s, ntens * (elems * 2.0 * (d->nobj + 1 + a.ndst) + 2.0 * d->nobj * d->frames * hw));
  if (vec)
    hipLaunchKernelGGL(pnp_nchw_kernel<8>, dim3((unsigned)nblk, ntens), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(pnp_nchw_kernel<1>, dim3((unsigned)nblk, ntens), dim3(256), 0, s, a);
  return mvoc_check_launch("pnp_nchw_kernel");
}

extern "C" int mvoc_ddim_step_f16(const void* x, const void* v_uncond, const void* v_cond, const float* coef_dev,
                                  void* out, int64_t n, void* stream) {
  MVOC_REQUIRE(x && v_cond && coef_dev && out && n > 0, -1, "ddim_step: null operand / empty");
  hipStream_t s = (hipStream_t)stream;
  MvocProfScope prof(MVOC_FAM_MISC, s, 2.0 * n * (v_uncond ? 4 : 3));
  hipLaunchKernelGGL(ddim_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const half_t*)x,
                     (const half_t*)v_uncond, (const half_t*)v_cond, coef_dev, (half_t*)out, (long)n);
  return mvoc_check_launch("ddim_step_kernel");
}

extern "C" int mvoc_latent_fusion_f16(const void* latents, const void* bg, const void* objs, const void* masks, void* out,
                                      int32_t nobj, int64_t n, double mix_ratio, int32_t obj_random_noise_fusion,
                                      void* stream) {
  MVOC_REQUIRE(latents && bg && objs && masks && out && n > 0 && nobj >= 0, -1, "latent_fusion: null operand / empty");
  hipStream_t s = (hipStream_t)stream;
  MvocProfScope prof(MVOC_FAM_MISC, s, 2.0 * n * (3 + 2 * nobj));
  // python-float factors of the reference: mix_ratio and (1.0 - mix_ratio) are formed in double, used as fp32
  const float mix = (float)mix_ratio, omix = (float)(1.0 - mix_ratio);
  hipLaunchKernelGGL(fusion_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const half_t*)latents,
                     (const half_t*)bg, (const half_t*)objs, (const half_t*)masks, (half_t*)out, nobj, (long)n, mix, omix,
                     obj_random_noise_fusion);
  return mvoc_check_launch("fusion_kernel");
}
