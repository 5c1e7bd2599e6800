"""I2VGen-XL 3D UNet on MI355X: the product forward of MVOC's denoising hot path.

Mirrors the module tree of diffusers' ``I2VGenXLUNet`` (attribute paths, state_dict keys) so that the
reference's hook protocol keeps working -- ``unet.up_blocks[i].attentions[j].transformer_blocks[0].attn1.processor``
carries ``t`` / ``mask`` / ``injection_schedule`` / ``inject_background`` exactly as ``pnp_utils.register_*``
and ``register_time_all`` set them (reference ``i2vgen-xl/pnp_utils.py:48-166, 706-715, 889-897, 1031-1037,
1099-1105, 1157-1159``) -- but every node is a plain Python object holding packed fp16 weights, and every
computation is a libmvoc_hip kernel (``mvoc_amd.ops``).  There is no torch.nn / MIOpen / rocBLAS on this path.

Layout: activations are channels-last rows ``[B*F*H*W, C]`` for the whole network: spatial
transformers, temporal transformers, 3x3 convs and temporal convs all consume it directly, so none of the
reference's permute/reshape copies exist.  ``torch.cat([x, skip], 1)`` is never materialised (two-source
gathers), Upsample2D is folded into the following conv's gather, q/k/v projections are one fused GEMM whose
output the attention kernels read through strides, GEGLU / bias / time-embedding add / residual adds live in
GEMM epilogues.

Forward entry points follow the reference:
  * ``forward``      -- stock ``I2VGenXLUNet.forward`` call protocol (``pipeline_i2vgen_xl.py:1173-1182, 1952-1961``)
  * ``forward_ext``  -- ``I2VGenXLUnetExtension.forward`` (``pipeline_i2vgen_xl.py:109-362``), adds
                        ``image_latents_first`` and ``multi_frame_guidance``
"""
import math
import os
from collections import OrderedDict

import torch

from . import ops
from .ops import ACT_GEGLU, ACT_NONE, ACT_SILU
from .unet_spec import UNetConfig, param_shapes

H16 = torch.float16


def _pad_rows(w, mult=32):
    n = w.shape[0]
    if n % mult == 0:
        return w.contiguous()
    out = torch.zeros((n + mult - n % mult,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    out[:n] = w
    return out


def _pad_cols(w, mult=32):
    k = w.shape[1]
    if k % mult == 0:
        return w.contiguous()
    out = torch.zeros((w.shape[0], k + mult - k % mult), dtype=w.dtype, device=w.device)
    out[:, :k] = w
    return out


def pack_conv3x3(w):
    """[Cout, Cin, 3, 3] -> [Cout_pad32, Kpad32] with k = (ky*3+kx)*Cin + c"""
    co, ci = w.shape[0], w.shape[1]
    return _pad_rows(_pad_cols(w.permute(0, 2, 3, 1).reshape(co, 9 * ci)))


def pack_conv3x3_subpixel(w):
    """[Cout, Cin, 3, 3] -> the four parity kernels of `nearest-2x upsample + conv3x3(pad 1)` as 2 x 2 convs on the source image
    (include/mvoc_hip.h: upsample == 2): [4 * Cout_pad][2][2][Cin] -> [4 * Cout_pad, 4 * Cin], phase 2 a + b major.  Row parity a = 0:
    the kernel rows (ky 0 | ky 1 + ky 2) fall on source rows (i - 1 | i); a = 1: (ky 0 + ky 1 | ky 2) on (i | i + 1); columns likewise.
    Sums in fp32, rounded to fp16 once."""
    w32 = w.float()
    rows = {0: [(0,), (1, 2)], 1: [(0, 1), (2,)]}
    out = []
    for a in (0, 1):
        for b in (0, 1):
            k = torch.stack([torch.stack([sum(w32[:, :, ky, kx] for ky in rows[a][dy] for kx in rows[b][dx]) for dx in (0, 1)], -1)
                             for dy in (0, 1)], -2)                       # [Cout, Cin, dy, dx]
            out.append(_pad_rows(k.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(H16)))  # tap-major like pack_conv3x3
    return torch.cat(out).contiguous()


def pack_conv3x3_small(w):
    """[Cout, Cin, 3, 3] -> [Cout, 3, 3, Cin] for the direct small-channel conv"""
    return w.permute(0, 2, 3, 1).contiguous()


def pack_tconv(w):
    """[Cout, Cin, 3, 1, 1] -> [Cout, 3*Cin] with k = tap*Cin + c"""
    co, ci = w.shape[0], w.shape[1]
    return w.reshape(co, ci, 3).permute(0, 2, 1).reshape(co, 3 * ci).contiguous()


def pack_geglu(w, b):
    """GEGLU proj [2*inner, C]: interleave value/gate rows in blocks of 32 so a wave's tile pair holds both."""
    inner = w.shape[0] // 2
    idx = torch.arange(2 * inner, device=w.device)
    blk, s, i = idx // 64, (idx // 32) % 2, idx % 32
    src = s * inner + blk * 32 + i
    return w[src].contiguous(), b[src].contiguous()


def pack_xs_weights(w, consts=None):
    """[N, K] (N % 32 == 0, K % 16 == 0) + per-channel constants [N] -> the stream mvoc_xs_linear_f16 reads:
    [tile][K/16 + 1 pieces][lane 64][8]; piece s < K/16 in MFMA fragment order (element = W[32 tile + (lane & 31)][16 s +
    8 (lane >> 5) + j]), the last piece starts with the tile's 32 constants as fp32"""
    n, k = w.shape
    nk = k // 16
    out = torch.zeros((n // 32, nk + 1, 512), dtype=H16, device=w.device)
    out[:, :nk] = w.view(n // 32, 32, nk, 2, 8).permute(0, 2, 3, 1, 4).reshape(n // 32, nk, 512)
    if consts is not None:
        out[:, nk, :64] = consts.to(torch.float32).reshape(n // 32, 32).contiguous().view(H16)
    return out


def pack_tfused_weights(wqkv, heads):
    """[3C, C] gamma-scaled QKV weights -> MFMA fragment order for mvoc_temporal_qkv_attn_f16:
    [head][tile q0 k0 q1 k1 v0 v1][k16 step][lane][8], element = W[row0 + (lane & 31)][16 s + 8 (lane >> 5) + j]"""
    c = wqkv.shape[1]
    assert wqkv.shape[0] == 3 * c and c == heads * 64
    rows = []
    for hd in range(heads):
        for off in (0, c, 32, c + 32, 2 * c, 2 * c + 32):
            row0 = hd * 64 + off
            rows.append(wqkv[row0:row0 + 32])
    w = torch.stack(rows)                                   # [heads*6, 32 rows, C]
    w = w.view(heads * 6, 32, c // 16, 2, 8)                # row r, k16 step s, half h, 8 elements
    return w.permute(0, 2, 3, 1, 4).contiguous()            # [tile, s, h, r, 8]: lane = 32 h + r


class Hookable:
    """carrier of the reference's per-site hook state"""

    def __init__(self):
        self.t = None
        self.mask = None
        self.injection_schedule = None
        self.inject_background = False

    def injecting(self):
        s = self.injection_schedule
        if s is None or self.t is None:
            return False
        if self.t == 1000:
            return True
        if isinstance(s, torch.Tensor):
            return bool((s == self.t).any().item()) if s.numel() else False
        return self.t in s


class Processor(Hookable):
    pass


class Linear:
    def __init__(self, w, b=None):
        self.n = w.shape[0]
        self.w = _pad_rows(w.to(H16))
        self.b = None
        if b is not None:
            self.b = torch.zeros(self.w.shape[0], dtype=H16, device=w.device)
            self.b[:self.n] = b.to(H16)
        self.ln = None
        self.wp = None      # fragment-packed copies for the activation-stationary kernel (made on first use)
        self.wp_ln = None

    XS_MIN_ROWS = 4096  # below this the launch is latency-bound either way; the tiled GEMM has split-K for deep K

    def _xs_ok(self, x, kw):
        """the activation-stationary kernel takes this call: K in registers (64 / 128 / 320), single contiguous source, plain
        epilogue (bias | folded LayerNorm constant, activation, residual), many rows"""
        k = self.w.shape[1]
        if (Linear.resid_tiled_rows and kw.get("resid") is not None and x.shape[0] >= Linear.resid_tiled_rows and self.w.shape[0] <= 320):
            # to_out / proj_out + residual at batch 5: since the 320-wide eight-phase tile stores plain and prefetches its residual
            # (round 4) it is 10 % faster than the activation-stationary kernel on this HBM-bound launch (tools/xs_bench.py, round 5:
            # 161-167 us against 182-183 at 327 680 rows; equal at 65 536)
            return False
        return (Linear.use_xs and k in ops.XS_K and x.shape[0] >= self.XS_MIN_ROWS and
                x.dim() == 2 and x.shape[1] == k and x.is_contiguous() and
                not (set(kw) - {"act", "resid", "out"}) and (kw.get("resid") is None or kw.get("act", ACT_NONE) != ACT_GEGLU) and
                (self.n % 8 == 0 if kw.get("act", ACT_NONE) != ACT_GEGLU else True) and
                (kw.get("out") is None or kw["out"].stride(0) % 8 == 0) and (kw.get("resid") is None or kw["resid"].stride(0) % 8 == 0))

    use_xs = os.environ.get("MVOC_XS", "1") != "0"  # MVOC_XS=0: A/B against the tiled GEMM (diagnostics)
    resid_tiled_rows = int(os.environ.get("MVOC_XS_RESID_TILED_ROWS", "131072"))  # 0: the residual projections stay on xslin (A/B)

    def __call__(self, x, sums=False, rowmom=False, **kw):
        """``sums`` / ``rowmom``: ask the GEMM for the GroupNorm / LayerNorm statistics of its output (ops._gemm; the
        activation-stationary kernel has none -- its LayerNorm consumers normalise in registers)"""
        if self._xs_ok(x, kw):
            if self.wp is None:
                self.wp = pack_xs_weights(self.w, self.b)
            return ops.xs_linear(x, self.wp, self.w.shape[0], n_store=self.n, **kw)
        return ops.linear(x, self.w, self.b, n_store=self.n, sums=sums, rowmom=rowmom, **kw)

    USE_GN_FOLD = os.environ.get("MVOC_GN_FOLD", "1") != "0"  # MVOC_GN_FOLD=0: A/B against GroupNorm + linear

    def call_gn(self, eng, x, norm, *, nsample, rows_per_sample, groups, eps, rowmom=False, video_norm=False):
        """GroupNorm(x) -> this linear (no activation in between: GN -> proj_in, pnp_utils.py:185-191, 433-438).  Where the
        activation-stationary kernel takes the call (K = 320 and many rows: the finest level) the norm is FOLDED into
        per-sample weights (ops.groupnorm_fold_xs) and the linear reads the raw rows: the normalised tensor -- a read and a
        write of the whole activation -- never exists.  Elsewhere: the GroupNorm kernels, then the linear.
        ``rowmom`` (row moments of the output for the LayerNorm behind it) is honoured on the unfolded path only: the
        activation-stationary kernel emits none, and its consumers take their row statistics from ``row_stats`` then."""
        if (Linear.USE_GN_FOLD and eng.shard is None and self._xs_ok(x, {}) and rows_per_sample % 256 == 0 and self.w.shape[0] == self.n
                and x.shape[0] == nsample * rows_per_sample and nsample <= 4096):
            wp = ops.groupnorm_fold_xs(x, *norm, self.w, self.b, nsample=nsample, rows_per_sample=rows_per_sample, groups=groups, eps=eps)
            return ops.xs_linear(x, wp, self.w.shape[0], n_store=self.n, set_rows=rows_per_sample)
        if video_norm:  # statistics over the frames too: a frame-sharded clip merges the ranks' moments (eng.groupnorm5d)
            h = eng.groupnorm5d(x, norm, nsample=nsample, rows_per_sample=rows_per_sample, groups=groups, eps=eps, silu=False)
        else:
            h = ops.groupnorm(x, *norm, nsample=nsample, rows_per_sample=rows_per_sample, groups=groups, eps=eps, silu=False)
        return self(h, rowmom=rowmom)

    def fold_layernorm(self, gamma, beta, eps=1e-5):
        """LayerNorm(x) @ W^T + b  ==  rstd * (x @ (W*gamma)^T - mean * rowsum(W*gamma)) + (beta @ W^T + b): the GEMM
        reads the raw rows, accumulates their mean / rstd from the tiles it stages anyway (csrc/gemm.hip) and the
        LayerNorm kernel, its output tensor and one HBM round trip disappear."""
        if self.w.shape[1] % 64:
            return self  # the folded path needs the direct-to-LDS GEMM (K % 64 == 0); keep the explicit LayerNorm
        w32 = self.w.float()
        wg = (w32 * gamma.float()[None, :]).to(H16)
        c = w32 @ beta.float()
        if self.b is not None:
            c = c + self.b.float()
        self.w_ln = wg.contiguous()
        self.ln = (wg.float().sum(dim=1).contiguous(), c.contiguous(), float(eps))
        return self

    def call_ln(self, x, norm, **kw):
        """x: raw rows; norm = (gamma, beta) of the LayerNorm that precedes this linear"""
        if self.ln is None:
            return self(ops.layernorm(x, *norm), **kw)
        if self._xs_ok(x, kw):
            # rows normalised in registers, gamma on the weights, beta @ W^T + bias as the per-channel constant: no statistics
            # pass, no LayerNorm tensor, no row-sum correction
            if self.wp_ln is None:
                self.wp_ln = pack_xs_weights(self.w_ln, self.ln[1])
            return ops.xs_linear(x, self.wp_ln, self.w_ln.shape[0], normalize=True, eps=self.ln[2], n_store=self.n, **kw)
        stats = ops.row_stats_of(x, self.ln[2])  # the producer's row moments, else one read of the rows; every n-tile shares it
        return ops.linear(x, self.w_ln, None, n_store=self.n, ln=self.ln + (stats,), **kw)


class Attention:
    def __init__(self, sd, prefix, heads, cross):
        self.heads = heads
        self.cross = cross
        self.processor = Processor()
        g = lambda k: sd[prefix + k].to(H16)
        if cross:
            self.to_q = Linear(g(".to_q.weight"))
            self.to_kv = Linear(torch.cat([g(".to_k.weight"), g(".to_v.weight")], 0))
        else:
            self.to_qkv = Linear(torch.cat([g(".to_q.weight"), g(".to_k.weight"), g(".to_v.weight")], 0))
        self.to_out = Linear(g(".to_out.0.weight"), g(".to_out.0.bias"))
        self.inner = g(".to_q.weight").shape[0]
        self.tfused = None  # fragment-packed folded QKV weights of a temporal self-attention (TransformerTemporalModel)

    def pack_tfused(self):
        """after the LayerNorm has been folded into to_qkv: the weights in the fused temporal kernel's order"""
        q = self.to_qkv
        if not self.cross and q.ln is not None and self.inner in ops.TFUSED_CHANNELS and q.w_ln.shape == (3 * self.inner, self.inner):
            self.tfused = pack_tfused_weights(q.w_ln, self.heads)


class BasicTransformerBlock:
    def __init__(self, sd, prefix, dim, heads, cross):
        g = lambda k: sd[prefix + k].to(H16).contiguous()
        self.norm1 = (g(".norm1.weight"), g(".norm1.bias"))
        self.norm2 = (g(".norm2.weight"), g(".norm2.bias"))
        self.norm3 = (g(".norm3.weight"), g(".norm3.bias"))
        self.attn1 = Attention(sd, prefix + ".attn1", heads, False)
        self.attn2 = Attention(sd, prefix + ".attn2", heads, cross)
        w, b = pack_geglu(g(".ff.net.0.proj.weight"), g(".ff.net.0.proj.bias"))
        self.ff1 = Linear(w, b).fold_layernorm(*self.norm3)
        self.ff2 = Linear(g(".ff.net.2.weight"), g(".ff.net.2.bias"))
        self.dim = dim
        self.attn1.to_qkv.fold_layernorm(*self.norm1)
        (self.attn2.to_q if cross else self.attn2.to_qkv).fold_layernorm(*self.norm2)


class _TransformerBase:
    def __init__(self, sd, prefix, cin, heads, groups, cross):
        g = lambda k: sd[prefix + k].to(H16).contiguous()
        self.norm = (g(".norm.weight"), g(".norm.bias"))
        self.groups = groups
        self.proj_in = Linear(g(".proj_in.weight"), g(".proj_in.bias"))
        self.proj_out = Linear(g(".proj_out.weight"), g(".proj_out.bias"))
        inner = self.proj_in.n
        self.transformer_blocks = [BasicTransformerBlock(sd, prefix + ".transformer_blocks.0", inner, heads, cross)]
        self.heads = heads


class Transformer2DModel(_TransformerBase):
    """``pnp_utils.py:387-548`` (+ ``:222-346``, ``:565-704``): GN -> proj_in -> [LN, self-attn(+PnP Q/K injection),
    LN, cross-attn, LN, GEGLU ff] -> proj_out + residual."""

    def __init__(self, sd, prefix, cin, heads, groups):
        super().__init__(sd, prefix, cin, heads, groups, True)

    def forward(self, eng, x, geo, ctx):
        B, F, H, W = geo
        hw, nimg = H * W, B * F
        blk = self.transformer_blocks[0]
        h = self.proj_in.call_gn(eng, x, self.norm, nsample=nimg, rows_per_sample=hw, groups=self.groups, eps=1e-6,
                                 rowmom=True)  # (norm1 reads it next)
        c = blk.dim
        # self-attention over the H*W tokens of each image
        qkv = blk.attn1.to_qkv.call_ln(h, blk.norm1)
        q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
        proc = blk.attn1.processor
        ndst = 0
        if proc.injecting() and not eng._pruned:
            ndst = eng.check_pnp_batch(B, proc.mask)
            masks = eng.device_masks(proc.mask)[1]  # bool masks as {0,1} fp16
            ld = qkv.stride(0)
            ops.pnp_blend_tokens(q, masks, x2=k, frames=F, height=H, width=W, channels=c, chunk_stride=F * hw * ld,
                                 f_stride=hw * ld, p_stride=ld, base_chunk0=proc.inject_background, ndst=ndst)
        if ndst == 2 and eng.pair_destinations:
            # the injection has just written ONE blended q / k into both destination chunks (pnp_utils.py:664-668): their
            # softmax(q k^T) is the same matrix -- computed once, multiplied into the two chunks' own v (bit-identical outputs)
            a = torch.empty((nimg * hw, c), dtype=H16, device=x.device)
            ns, rows = (B - 2) * F, F * hw
            s0, s1, s2 = slice(0, ns * hw), slice(ns * hw, ns * hw + rows), slice(ns * hw + rows, ns * hw + 2 * rows)
            ops.flash_attn(q[s0], k[s0], v[s0], nbatch=ns, heads=self.heads, tq=hw, tk=hw, out=a[s0])
            ops.flash_attn(q[s1], k[s1], v[s1], nbatch=F, heads=self.heads, tq=hw, tk=hw, out=a[s1], v2=v[s2], out2=a[s2])
        else:
            a = ops.flash_attn(q, k, v, nbatch=nimg, heads=self.heads, tq=hw, tk=hw)
        h = blk.attn1.to_out(a, resid=h, rowmom=True)
        if eng._expand_site is self:
            # shared_prefix_chunks: the unconditional chunk's rows so far ARE the conditional chunk's -- from here on (their
            # contexts differ) both exist
            h, x = eng.expand_last_chunk(h, F * hw), eng.expand_last_chunk(x, F * hw)
            B, nimg = B + 1, (B + 1) * F
        # cross-attention to the 77 text + 64 image-latent + 4 CLIP-image tokens
        q2 = blk.attn2.to_q.call_ln(h, blk.norm2)
        kv = ctx.kv.get(self)
        if kv is None:
            kv = blk.attn2.to_kv(ctx.tokens)
            if ctx.keep:
                ctx.kv[self] = kv
        a = ops.flash_attn(q2, kv[:, :c], kv[:, c:], nbatch=nimg, heads=self.heads, tq=hw, tk=ctx.length,
                           kv_bdiv=ctx.frames_per_ctx)
        h = blk.attn2.to_out(a, resid=h, rowmom=True)
        f1 = blk.ff1.call_ln(h, blk.norm3, act=ACT_GEGLU)
        h = blk.ff2(f1, resid=h)
        return self.proj_out(h, resid=x, sums=True)  # (the next module opens with a GroupNorm of this tensor)


class TransformerTemporalModel(_TransformerBase):
    """``pnp_utils.py:170-220`` (+ ``:720-887``): 5-D GroupNorm (statistics over frames too), then a transformer
    block whose two attentions both run over the frame axis of each pixel."""

    use_fused = True  # LN -> QKV -> frame attention in one kernel where the shape allows it (C in {64,128,320}, F in {8,16,32})

    def __init__(self, sd, prefix, cin, heads, groups):
        super().__init__(sd, prefix, cin, heads, groups, False)
        blk = self.transformer_blocks[0]
        if blk.dim == heads * 64:
            blk.attn1.pack_tfused()
            blk.attn2.pack_tfused()

    def forward(self, eng, x, geo):
        return eng.temporal_section(x, geo, self._section)

    def _section(self, eng, x, geo, full_hw):
        """geo: (B, all F frames, H, W) of the rows in x -- the whole frames on one GPU, or this rank's pixel slab
        (H = 1, W = slab length, ``full_hw`` = the real feature size) when the clip is frame-sharded"""
        B, F, H, W = geo
        hw = H * W
        blk = self.transformer_blocks[0]
        h = self.proj_in.call_gn(eng, x, self.norm, nsample=B, rows_per_sample=F * hw, groups=self.groups, eps=1e-6, rowmom=True,
                                 video_norm=True)
        c = blk.dim
        for attn, norm in ((blk.attn1, blk.norm1), (blk.attn2, blk.norm2)):
            proc = attn.processor
            inject = attn is blk.attn1 and proc.injecting() and not eng._pruned
            if self.use_fused and attn.tfused is not None and not inject and F in ops.TFUSED_FRAMES and h.is_contiguous():
                # Q/K/V never leave the chip: LayerNorm, projection and the frame attention of 32/F pixels per wave in one kernel
                a = ops.temporal_qkv_attn(h, attn.tfused, attn.to_qkv.ln, nsample=B, frames=F, hw=hw, heads=self.heads)
                h = attn.to_out(a, resid=h, rowmom=True)
                continue
            qkv = attn.to_qkv.call_ln(h, norm)
            q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
            if inject:
                ndst = eng.check_pnp_batch(B, proc.mask)
                masks = eng.section_masks(proc.mask, 0, full_hw)  # soft float masks, channel 0
                ld = qkv.stride(0)
                ops.pnp_blend_tokens(q, masks, x2=k, frames=F, height=H, width=W, channels=c, chunk_stride=F * hw * ld,
                                     f_stride=hw * ld, p_stride=ld, base_chunk0=proc.inject_background, ndst=ndst)
                if eng._tail_site is self:  # (prune_source_tail) the last reader of the source chunks was this blend
                    r0 = (B - ndst) * F * hw
                    q, k, v, h, x = q[r0:], k[r0:], v[r0:], h[r0:], x[r0:]
                    B = ndst
            a = ops.temporal_attn(q, k, v, nsample=B, frames=F, hw=hw, heads=self.heads)
            h = attn.to_out(a, resid=h, rowmom=True)
        f1 = blk.ff1.call_ln(h, blk.norm3, act=ACT_GEGLU)
        h = blk.ff2(f1, resid=h)
        return self.proj_out(h, resid=x, sums=True)  # (the next module opens with a GroupNorm of this tensor)


class ResnetBlock2D(Hookable):
    """``pnp_utils.py:902-1020``.  The feature injection sits between conv2 and the shortcut add."""

    def __init__(self, sd, prefix, groups):
        super().__init__()
        g = lambda k: sd[prefix + k].to(H16).contiguous()
        self.groups = groups
        self.norm1 = (g(".norm1.weight"), g(".norm1.bias"))
        self.norm2 = (g(".norm2.weight"), g(".norm2.bias"))
        self.conv1_w, self.conv1_b = pack_conv3x3(g(".conv1.weight")), g(".conv1.bias")
        self.conv2_w, self.conv2_b = pack_conv3x3(g(".conv2.weight")), g(".conv2.bias")
        self.time_emb_proj = Linear(g(".time_emb_proj.weight"), g(".time_emb_proj.bias"))
        self.tslice = None  # (column offset, width) in the engine's batched time projection
        self.cout = self.conv1_b.shape[0]
        self.conv_shortcut = None
        if prefix + ".conv_shortcut.weight" in sd:
            w = g(".conv_shortcut.weight")
            self.conv_shortcut = Linear(w.reshape(w.shape[0], w.shape[1]), g(".conv_shortcut.bias"))

    def forward(self, eng, x, skip, temb_act, geo):
        B, F, H, W = geo
        hw, nimg = H * W, B * F
        h = ops.groupnorm(x, *self.norm1, x2=skip, nsample=nimg, rows_per_sample=hw, groups=self.groups, eps=1e-5,
                          silu=True)
        if eng._tall is not None and self.tslice is not None:
            assert eng._tall.shape[0] == temb_act.shape[0], "batched time projection of another call"
            tproj = eng._tall[:, self.tslice[0]:self.tslice[0] + self.tslice[1]]  # [B, Cout] columns of the batched projection
        else:
            tproj = self.time_emb_proj(temb_act)  # [B, Cout]; identical for all frames of a sample
        h, _, _ = ops.conv3x3(h, self.conv1_w, self.conv1_b, nimg=nimg, h=H, wd=W, rowadd=tproj, rowadd_div=F * hw,
                              n_store=self.cout, sums=True)  # norm2's statistics from this conv's epilogue
        h = ops.groupnorm(h, *self.norm2, nsample=nimg, rows_per_sample=hw, groups=self.groups, eps=1e-5, silu=True)
        if self.injecting() and not eng._pruned:
            h, _, _ = ops.conv3x3(h, self.conv2_w, self.conv2_b, nimg=nimg, h=H, wd=W, n_store=self.cout)
            eng.inject_features(h, self.mask, geo, self.cout)
            if self.conv_shortcut is not None:
                return self.conv_shortcut(x, x2=skip, resid=h, sums=True)
            return ops.add(x, h)
        if self.conv_shortcut is not None:
            sc = self.conv_shortcut(x, x2=skip)
        else:
            sc = x
        out, _, _ = ops.conv3x3(h, self.conv2_w, self.conv2_b, nimg=nimg, h=H, wd=W, resid=sc, n_store=self.cout,
                                sums=True)  # the temporal conv stack that follows opens with a GroupNorm
        return out


class TemporalConvLayer(Hookable):
    """``pnp_utils.py:1042-1088``: 4 x (GN over the whole video, SiLU, Conv3d(3,1,1)) + identity; injection after."""

    def __init__(self, sd, prefix, groups):
        super().__init__()
        g = lambda k: sd[prefix + k].to(H16).contiguous()
        self.groups = groups
        self.stages = []
        for i, ci in ((1, 2), (2, 3), (3, 3), (4, 3)):
            self.stages.append(((g(f".conv{i}.0.weight"), g(f".conv{i}.0.bias")),
                                pack_tconv(g(f".conv{i}.{ci}.weight")), g(f".conv{i}.{ci}.bias")))

    def forward(self, eng, x, geo):
        return eng.temporal_section(x, geo, self._section)

    def _section(self, eng, x, geo, full_hw):
        B, F, H, W = geo
        hw = H * W
        h = x
        for i, (norm, w, b) in enumerate(self.stages):
            h = eng.groupnorm5d(h, norm, nsample=B, rows_per_sample=F * hw, groups=self.groups, eps=1e-5, silu=True)
            h = ops.tconv3(h, w, b, nvid=B, frames=F, hw=hw, resid=x if i == 3 else None, sums=True)
        if self.injecting() and not eng._pruned:
            eng.inject_features(h, self.mask, geo, h.shape[1], full_hw=full_hw)
        return h


class Upsample2D:
    """``F.interpolate(scale 2, nearest)`` + conv3x3 (diffusers Upsample2D; ``pnp_utils.py:919-934`` checks for it): the upsample is
    never materialised -- folded into the conv's gather, and for the exact 2x case into the conv's WEIGHTS (sub-pixel form: per
    output parity a 2 x 2 conv on the source image, 4 / 9 of the matrix work)"""
    use_subpixel = os.environ.get("MVOC_SUBPIXEL", "1") != "0"  # MVOC_SUBPIXEL=0: A/B against the 9-tap gather

    def __init__(self, sd, prefix):
        w = sd[prefix + ".conv.weight"].to(H16)
        self.w = pack_conv3x3(w)
        self.w_sub = pack_conv3x3_subpixel(w) if w.shape[1] % 64 == 0 else None
        self.b = sd[prefix + ".conv.bias"].to(H16).contiguous()

    def forward(self, x, geo, size=None):
        B, F, H, W = geo
        up = size if size is not None else (2 * H, 2 * W)
        out, ho, wo = ops.conv3x3(x, self.w, self.b, nimg=B * F, h=H, wd=W, upsample_to=tuple(up), n_store=self.b.shape[0], sums=True,
                                  w_subpixel=self.w_sub if Upsample2D.use_subpixel else None)
        return out, (B, F, ho, wo)


class Downsample2D:
    def __init__(self, sd, prefix):
        self.w = pack_conv3x3(sd[prefix + ".conv.weight"].to(H16))
        self.b = sd[prefix + ".conv.bias"].to(H16).contiguous()

    def forward(self, x, geo):
        B, F, H, W = geo
        out, ho, wo = ops.conv3x3(x, self.w, self.b, nimg=B * F, h=H, wd=W, stride=2, n_store=self.b.shape[0], sums=True)
        return out, (B, F, ho, wo)


class Block:
    def __init__(self):
        self.resnets, self.temp_convs, self.attentions, self.temp_attentions = [], [], [], []
        self.downsamplers = None
        self.upsamplers = None
        self.has_cross_attention = False


class ConvOut(Hookable):
    def __init__(self, w, b):
        super().__init__()
        self.w = pack_conv3x3(w.to(H16))
        self.b = torch.zeros(self.w.shape[0], dtype=H16, device=w.device)
        self.b[:b.shape[0]] = b.to(H16)
        self.cout = b.shape[0]


class Context:
    def __init__(self, tokens, length, frames_per_ctx):
        self.tokens, self.length, self.frames_per_ctx = tokens, length, frames_per_ctx
        self.kv = {}       # Transformer2DModel -> cross-attention K|V projection of `tokens` (filled when `keep`)
        self.keep = False


class Conditioning:
    """Everything of a forward that depends only on the conditioning inputs, not on the noisy sample or the timestep
    (``I2VGenXLUNet.prepare_conditioning``): context tokens (``pipeline_i2vgen_xl.py:204-260``), every spatial
    transformer's cross-attention K/V of them, and the image-latent half of the stem (``:262-284``)."""

    def __init__(self, ctx, stem8, geometry, frames, key):
        self.ctx, self.stem8, self.geometry, self.frames, self.key = ctx, stem8, geometry, frames, key

    def copy_from(self, other):
        """refresh this (graph-captured) conditioning in place from a freshly prepared one of the same shapes"""
        if other.key != self.key or set(other.ctx.kv) != set(self.ctx.kv):
            raise RuntimeError(f"Conditioning.copy_from: prepared for {other.key}, this one for {self.key}")
        self.ctx.tokens.copy_(other.ctx.tokens)
        self.stem8.copy_(other.stem8)
        for k, v in self.ctx.kv.items():
            v.copy_(other.ctx.kv[k])
        return self


class I2VGenXLUNet:
    """MI355X engine with the reference UNet's call protocol."""

    def __init__(self, config=None, device="cuda:0"):
        self.config = UNetConfig.from_any(config) if config is not None else UNetConfig()
        self.device = torch.device(device)
        if self.device.type == "cuda" and torch.cuda.is_available():
            # one process per GPU: the library launches on the CURRENT device's current stream, so the engine's device
            # becomes the process's current device (launch.pick_device hands rank i its own GPU)
            torch.cuda.set_device(self.device)
        self.dtype = H16
        self.num_upsamplers = len(self.config.block_out_channels) - 1
        self._mask_cache = (None, None, None)
        self._section_mask_cache = {}
        self._loaded = False
        self.shard = None  # mvoc_amd.frame_shard.FrameShard: frame-axis shard of one long clip (set_frame_shard)
        # At a timestep where conv_out injects (pnp_utils.py:1114-1146) the destination chunks' OUTPUT is a blend of the
        # source chunks' outputs, and every hook only ever writes destination chunks: nothing the network computes for
        # [uncond, cond] at such a step reaches the result.  prune_dead_chunks runs those steps on the source chunks only
        # ([bg, objects..]: batch 3 of 5) and lets the conv_out blend fill the rest -- same values, 40 % less work on the
        # demo's first 5 of 50 composition steps.
        self.prune_dead_chunks = True
        self._pruned = False
        # The SOURCE chunks [bg, obj_1..obj_n] of a composition batch exist to feed the injection sites: no hook, and nothing in
        # the loop around the UNet (pipeline_i2vgen_xl.py:1713-1728 reads the two destination chunks only), ever reads their
        # OUTPUT.  Behind the last site that still takes their q / k -- up_blocks[3].temp_attentions[2].attn1 on a Q/K-only step
        # (pnp_utils.py:720-887; 45 of the demo's 50 steps) -- their rows are dead: with prune_source_tail the rest of that
        # temporal transformer, conv_norm_out and conv_out run on the destination chunks only and the source chunks of the
        # returned tensor are zeros.  Off by default (forward_ext stays faithful chunk by chunk); the composition loop of
        # pipeline.py, which never reads those chunks, turns it on.
        self.prune_source_tail = False
        self._tail_site = None
        # Classifier-free guidance feeds the SAME latent, image latents, fps and timestep to the unconditional and the conditional
        # chunk (pipeline_i2vgen_xl.py:1676-1690); they differ in the prompt / CLIP-image embeddings, which enter through the
        # spatial transformers' cross-attention only.  Up to the first cross-attention (conv_in, transformer_in, the first
        # resnet + temporal conv, GroupNorm / proj_in / self-attention of down_blocks[0].attentions[0]) the two chunks' rows are
        # therefore identical: with shared_prefix_chunks = 2 (set by a caller that KNOWS the last two chunks' inputs to be
        # equal: the composition loop of pipeline.py checks it once per state) that prefix runs on B - 1 chunks and the last
        # chunk's rows are copied where the chunks part.  0 = off: every chunk computed (the default).
        self.shared_prefix_chunks = 0
        self._expand_site = None
        # Q/K-injection sites: the two destination chunks attend with identical q and k (the hook assigns one blend to both);
        # their attention probabilities are computed once (ops.flash_attn v2 / out2).  False: five independent passes (A/B, tests)
        self.pair_destinations = True

    def set_frame_shard(self, shard):
        """Frame-shard every forward over the ranks of ``shard`` (``mvoc_amd.frame_shard``): each rank receives the FULL
        inputs, computes its F/world frames (temporal sections pixel-sharded, see that module) and returns the FULL
        output (one all-gather of the 4-channel prediction), so the loops around the UNet stay unchanged."""
        self.shard = shard
        self.shard_generation = getattr(self, "shard_generation", 0) + 1  # cache keys: id() of a freed shard can be reused
        self._mask_cache = (None, None, None)
        self._section_mask_cache = {}
        return self

    # ---- weights --------------------------------------------------------------------------------
    def expected_shapes(self):
        return param_shapes(self.config)

    def load_state_dict(self, sd):
        """sd: diffusers ``I2VGenXLUNet`` state_dict (any float dtype, any device)."""
        exp = self.expected_shapes()
        missing = [k for k in exp if k not in sd]
        if missing:
            raise KeyError(f"state_dict is missing {len(missing)} keys, e.g. {missing[:3]}")
        for k, shp in exp.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{k}: expected shape {shp}, got {tuple(sd[k].shape)}")
        sd = {k: sd[k].detach().to(self.device, H16) for k in exp}
        self._build(sd)
        return self

    def init_random(self, seed=8888):
        """Synthetic weights of the exact architecture, generated on the device (no checkpoint in this
        environment): U(+-1/sqrt(fan_in)) matrices, norm gains ~1, residual-branch output layers damped."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        sd = OrderedDict()
        damp = ("proj_out.weight", "conv2.weight", "conv4.3.weight", "to_out.0.weight", "ff.net.2.weight")
        for k, shp in self.expected_shapes().items():
            if len(shp) >= 2:
                fan_in = 1
                for s in shp[1:]:
                    fan_in *= s
                t = (torch.rand(shp, generator=g, device=self.device) * 2 - 1) * (1.0 / math.sqrt(fan_in))
                if k.endswith(damp):
                    t *= 0.5
            elif "norm" in k or any(f".conv{i}.0." in k for i in (1, 2, 3, 4)):
                r = torch.rand(shp, generator=g, device=self.device) * 2 - 1
                t = 1.0 + 0.1 * r if k.endswith("weight") else 0.1 * r
            else:
                t = 0.05 * (torch.rand(shp, generator=g, device=self.device) * 2 - 1)
            sd[k] = t.to(H16)
        self._build(sd)
        return self

    def _build(self, sd):
        cfg = self.config
        boc, hd, g = cfg.block_out_channels, cfg.attention_head_dim, cfg.norm_num_groups
        f16 = lambda k: sd[k].to(H16).contiguous()
        # stem
        self.conv_in = ConvOut(sd["conv_in.weight"], sd["conv_in.bias"])  # Hookable (register_time_all sets t/mask on it)
        self.transformer_in = TransformerTemporalModel(sd, "transformer_in", boc[0], cfg.transformer_in_heads, g)
        self.proj_in_convs = [(pack_conv3x3_small(f16(f"image_latents_proj_in.{i}.weight")),
                               f16(f"image_latents_proj_in.{i}.bias")) for i in (0, 2, 4)]
        e = "image_latents_temporal_encoder"
        blob = [f16(f"{e}.norm1.weight"), f16(f"{e}.norm1.bias"), f16(f"{e}.attn1.to_q.weight"), f16(f"{e}.attn1.to_k.weight"),
                f16(f"{e}.attn1.to_v.weight"), f16(f"{e}.attn1.to_out.0.weight"), f16(f"{e}.attn1.to_out.0.bias"),
                f16(f"{e}.ff.net.0.proj.weight"), f16(f"{e}.ff.net.0.proj.bias"), f16(f"{e}.ff.net.2.weight"),
                f16(f"{e}.ff.net.2.bias")]
        self.enc4_params = torch.cat([t.reshape(-1) for t in blob]).contiguous()
        assert self.enc4_params.numel() == 4 + 4 + 32 * 4 + 4 + 64 + 16 + 64 + 4
        self.ctx_convs = [(pack_conv3x3_small(f16(f"image_latents_context_embedding.{i}.weight")),
                           f16(f"image_latents_context_embedding.{i}.bias")) for i in (0, 3, 5)]
        self.time_embedding = (Linear(f16("time_embedding.linear_1.weight"), f16("time_embedding.linear_1.bias")),
                               Linear(f16("time_embedding.linear_2.weight"), f16("time_embedding.linear_2.bias")))
        self.fps_embedding = (Linear(f16("fps_embedding.0.weight"), f16("fps_embedding.0.bias")),
                              Linear(f16("fps_embedding.2.weight"), f16("fps_embedding.2.bias")))
        self.context_embedding = (Linear(f16("context_embedding.0.weight"), f16("context_embedding.0.bias")),
                                  Linear(f16("context_embedding.2.weight"), f16("context_embedding.2.bias")))
        # blocks
        self.down_blocks = []
        for i, t in enumerate(cfg.down_block_types):
            b = Block()
            p = f"down_blocks.{i}"
            for j in range(cfg.layers_per_block):
                b.resnets.append(ResnetBlock2D(sd, f"{p}.resnets.{j}", g))
                b.temp_convs.append(TemporalConvLayer(sd, f"{p}.temp_convs.{j}", g))
                if t == "CrossAttnDownBlock3D":
                    b.has_cross_attention = True
                    b.attentions.append(Transformer2DModel(sd, f"{p}.attentions.{j}", boc[i], boc[i] // hd, g))
                    b.temp_attentions.append(TransformerTemporalModel(sd, f"{p}.temp_attentions.{j}", boc[i], boc[i] // hd, g))
            if i != len(boc) - 1:
                b.downsamplers = [Downsample2D(sd, f"{p}.downsamplers.0")]
            self.down_blocks.append(b)
        m = Block()
        m.has_cross_attention = True
        for j in range(2):
            m.resnets.append(ResnetBlock2D(sd, f"mid_block.resnets.{j}", g))
            m.temp_convs.append(TemporalConvLayer(sd, f"mid_block.temp_convs.{j}", g))
        m.attentions.append(Transformer2DModel(sd, "mid_block.attentions.0", boc[-1], boc[-1] // hd, g))
        m.temp_attentions.append(TransformerTemporalModel(sd, "mid_block.temp_attentions.0", boc[-1], boc[-1] // hd, g))
        self.mid_block = m
        self.up_blocks = []
        rev = list(reversed(boc))
        for i, t in enumerate(cfg.up_block_types):
            b = Block()
            p = f"up_blocks.{i}"
            for j in range(cfg.layers_per_block + 1):
                b.resnets.append(ResnetBlock2D(sd, f"{p}.resnets.{j}", g))
                b.temp_convs.append(TemporalConvLayer(sd, f"{p}.temp_convs.{j}", g))
                if t == "CrossAttnUpBlock3D":
                    b.has_cross_attention = True
                    b.attentions.append(Transformer2DModel(sd, f"{p}.attentions.{j}", rev[i], rev[i] // hd, g))
                    b.temp_attentions.append(TransformerTemporalModel(sd, f"{p}.temp_attentions.{j}", rev[i], rev[i] // hd, g))
            if i != len(boc) - 1:
                b.upsamplers = [Upsample2D(sd, f"{p}.upsamplers.0")]
            self.up_blocks.append(b)
        self.conv_norm_out = (f16("conv_norm_out.weight"), f16("conv_norm_out.bias"))
        self.conv_out = ConvOut(sd["conv_out.weight"], sd["conv_out.bias"])
        # every resnet's time_emb_proj (pnp_utils.py:924-931: a [B, 1280] x [1280, Cout] linear per resnet, 26 launches of
        # M = B rows per forward at 2 % MFMA utilisation) as ONE linear over the concatenated output channels, once per step
        rns = [rn for blk in list(self.down_blocks) + [self.mid_block] + list(self.up_blocks) for rn in blk.resnets]
        off = 0
        for rn in rns:
            rn.tslice = (off, rn.cout)
            off += rn.cout
        self.time_proj_all = Linear(torch.cat([rn.time_emb_proj.w[:rn.cout] for rn in rns]),
                                    torch.cat([rn.time_emb_proj.b[:rn.cout] for rn in rns]))
        self._tall = None
        self._loaded = True

    # ---- PnP helpers ------------------------------------------------------------------------------
    @staticmethod
    def check_pnp_batch(B, mask_list):
        """the hooks address chunks positionally [bg, obj_1..obj_n, uncond, cond] (pnp_utils.py:592 hard-codes 5); with
        classifier-free guidance off the batch is [bg, obj_1..obj_n, cond].  Returns the number of
        trailing destination chunks (2 or 1)."""
        if mask_list is None or B - len(mask_list) - 1 not in (1, 2):
            raise RuntimeError(f"PnP injection is active but the UNet batch is {B}, expected n_objects+3 = "
                               f"{None if mask_list is None else len(mask_list) + 3} ([bg, objects.., uncond, cond]) or "
                               "n_objects+2 with guidance off; "
                               "clear the hook state (register_time_all(pipe, None, None)) before non-composition calls")
        return B - len(mask_list) - 1

    def device_masks(self, mask_list):
        """list of (float [1,4,F,h,w], bool [1,4,F,h,w]) pairs (``register_time_all``'s ``mask``) ->
        (soft fp16 [nobj,F,h,w] from channel 0, hard {0,1} fp16 [nobj,F,h,w]); cached per mask list object."""
        return self._all_frame_masks(mask_list)[2:]

    def _all_frame_masks(self, mask_list):
        """(soft, hard) over ALL frames, then the slices of this rank's frames (the same tensors without a frame shard).
        The cache holds a reference to the source tensors and their versions: a freed-and-reallocated mask set (same
        addresses from the caching allocator) or an in-place edit of ``obj_masks_tensors`` cannot hit a stale entry."""
        key = self.mask_key(mask_list)
        if self._mask_cache[0] != key:
            soft = torch.stack([m[0].reshape(-1, *m[0].shape[-3:])[0] for m in mask_list]).to(self.device, H16).contiguous()
            hard = torch.stack([m[1].reshape(-1, *m[1].shape[-3:])[0] for m in mask_list]).to(self.device, H16).contiguous()
            lsoft, lhard = soft, hard
            if self.shard is not None:
                f0, f1 = self.shard.frame_range(soft.shape[1])
                lsoft, lhard = soft[:, f0:f1].contiguous(), hard[:, f0:f1].contiguous()
            keep = [(m[0], m[1]) for m in mask_list]  # ids stay unique for as long as the entry lives
            self._mask_cache = (key, (soft, hard, lsoft, lhard), keep)
            self._section_mask_cache = {}
        return self._mask_cache[1]

    @staticmethod
    def mask_key(mask_list):
        return tuple((id(m[0]), m[0]._version, id(m[1]), m[1]._version) for m in mask_list)

    def hook_sites(self):
        """every node whose ``injecting()`` decision shapes a forward (the sites ``pnp_utils.register_*`` may address)"""
        sites = [self.conv_out]
        for blk in list(self.down_blocks) + [self.mid_block] + list(self.up_blocks):
            sites += list(blk.resnets) + list(blk.temp_convs)
            for tr in list(blk.attentions) + list(blk.temp_attentions):
                sites.append(tr.transformer_blocks[0].attn1.processor)
        return sites

    def injection_flags(self):
        """the per-site injecting() bits of the CURRENT hook state: what a captured iteration bakes in"""
        return tuple(bool(s.injecting()) for s in self.hook_sites())

    def section_masks(self, mask_list, kind, full_hw):
        """masks for a temporal section (kind 0 = soft, 1 = hard): all frames.  Frame-sharded, the section sees a slab of
        pixels of every frame: the masks are nearest-resized to the feature size (``F.interpolate`` as at
        ``pnp_utils.py:807``; the kernel's in-line resize is the same index rule), flattened and cut to the slab."""
        full = self._all_frame_masks(mask_list)[kind]
        if self.shard is None:
            return full
        key = (kind, tuple(full_hw))
        if key not in self._section_mask_cache:
            m = full
            if tuple(m.shape[2:]) != tuple(full_hw):
                m = torch.nn.functional.interpolate(m, size=tuple(full_hw), mode="nearest")  # [nobj, F, h, w]: F rides as channels
            p0, p1 = self.shard.pixel_range(full_hw[0] * full_hw[1])
            self._section_mask_cache[key] = m.reshape(m.shape[0], m.shape[1], 1, -1)[..., p0:p1].contiguous()
        return self._section_mask_cache[key]

    def inject_features(self, h, mask_list, geo, channels, full_hw=None):
        """feature injection (``pnp_utils.py:970-1004, 1059-1082, 1114-1146``): base = chunk 0, bool mask, no resize.
        ``full_hw`` is given by temporal sections (see ``section_masks``); elsewhere the rows are whole local frames."""
        B, F, H, W = geo
        ndst = self.check_pnp_batch(B, mask_list)
        hard = self._all_frame_masks(mask_list)[1]
        fh, fw = full_hw if full_hw is not None else (H, W)
        if hard.shape[2] != fh or hard.shape[3] != fw:
            raise RuntimeError(f"feature injection needs masks at the feature resolution {(fh, fw)}, got "
                               f"{tuple(hard.shape[2:])} (reference: pnp_utils.py:994-1000 has no resize)")
        hard = self.section_masks(mask_list, 1, full_hw) if full_hw is not None else self.device_masks(mask_list)[1]
        ld = h.stride(0)
        ops.pnp_blend_tokens(h, hard, frames=F, height=H, width=W, channels=channels, chunk_stride=F * H * W * ld,
                             f_stride=H * W * ld, p_stride=ld, base_chunk0=True, ndst=ndst)
        if getattr(h, "chan_sums", None) is not None:
            h.chan_sums = None  # rewritten in place: the producer's GroupNorm statistics no longer describe these rows

    @staticmethod
    def expand_last_chunk(t, rows):
        """[B chunks of `rows` rows] -> B + 1 chunks, the last one repeated (shared_prefix_chunks); a row copy drops any producer
        statistics riding on the tensor object, which described B chunks"""
        out = torch.empty((t.shape[0] + rows, t.shape[1]), dtype=t.dtype, device=t.device)
        out[:t.shape[0]].copy_(t)
        out[t.shape[0]:].copy_(t[t.shape[0] - rows:])
        return out

    # ---- frame-axis shard plumbing --------------------------------------------------------------------
    def temporal_section(self, x, geo, section):
        """run ``section(eng, rows, geo, full_hw)`` on rows that hold ALL frames: as they are on one GPU, or exchanged
        to the pixel-sharded layout and back when the clip is frame-sharded"""
        sh = self.shard
        B, F, H, W = geo
        if sh is None:
            return section(self, x, geo, (H, W))
        hw = H * W
        sh.check(F * sh.world, hw)
        xp = sh.to_pixel_shard(x, B, F, hw)
        yp = section(self, xp, (B, F * sh.world, 1, hw // sh.world), (H, W))
        return sh.to_frame_shard(yp, B, F, hw)

    def groupnorm5d(self, x, norm, *, nsample, rows_per_sample, groups, eps, silu):
        """GroupNorm whose statistics span the whole video (``pnp_utils.py:185-188, 1048-1051``)"""
        if self.shard is None:
            return ops.groupnorm(x, *norm, nsample=nsample, rows_per_sample=rows_per_sample, groups=groups, eps=eps, silu=silu)
        mom = ops.groupnorm_moments(x, nsample=nsample, rows_per_sample=rows_per_sample, groups=groups)
        parts = self.shard.all_gather(mom)
        return ops.groupnorm_apply_moments(x, parts, *norm, nsample=nsample, rows_per_sample=rows_per_sample, groups=groups,
                                           eps=eps, silu=silu)

    # ---- forward -----------------------------------------------------------------------------------
    def _embeddings(self, timestep, fps, B):
        if torch.is_tensor(timestep):
            t = timestep.to(self.device, torch.float32).reshape(-1)
        else:
            t = torch.tensor([float(timestep)], dtype=torch.float32, device=self.device)
        t = t.expand(B).contiguous()
        f = fps.to(self.device, torch.float32).reshape(-1).expand(B).contiguous()
        boc0 = self.config.block_out_channels[0]
        te = ops.timestep_embedding(t, boc0)
        te = self.time_embedding[1](self.time_embedding[0](te, act=ACT_SILU))
        fe = ops.timestep_embedding(f, boc0)
        fe = self.fps_embedding[1](self.fps_embedding[0](fe, act=ACT_SILU))
        emb = ops.add(te, fe)
        return ops.act(emb, ACT_SILU)  # every resnet applies SiLU before its time_emb_proj

    def _context(self, image_latents, image_embeddings, encoder_hidden_states, F, multi_frame_guidance):
        """77 text + 64 image-latent + 4 CLIP-image tokens per sample (per frame with multi_frame_guidance).
        Without multi-frame guidance the reference recomputes the same context F times (``pipeline_i2vgen_xl.py:211``
        loops over frames reading frame 0); here it is computed once and shared through kv_bdiv."""
        cfg = self.config
        B = image_latents.shape[0]
        h, w = image_latents.shape[-2:]
        nf = F if multi_frame_guidance else 1
        lat = image_latents[:, :, :nf].to(self.device, H16)  # [B,4,nf,h,w]  (slices only: graph-capture safe)
        tok = torch.empty((B * nf * h * w, 4), dtype=H16, device=self.device)
        ops.ncfhw_to_tokens(lat, tok)
        (w0, b0), (w1, b1), (w2, b2) = self.ctx_convs
        x, _, _ = ops.conv3x3_small(tok, w0, b0, nimg=B * nf, h=h, wd=w, cin=4, cout=w0.shape[0], silu=True)
        P = cfg.context_pool
        x = ops.adaptive_avgpool(x, nimg=B * nf, h=h, w=w, c=w0.shape[0], oh=P, ow=P)
        x, h1, w1_ = ops.conv3x3_small(x, w1, b1, nimg=B * nf, h=P, wd=P, cin=w0.shape[0], cout=w1.shape[0], stride=2, silu=True)
        x, h2, w2_ = ops.conv3x3_small(x, w2, b2, nimg=B * nf, h=h1, wd=w1_, cin=w1.shape[0], cout=w2.shape[0], stride=2)
        nlat = h2 * w2_
        ie = image_embeddings.to(self.device, H16)
        if ie.dim() == 2:
            ie = ie[:, None]
        ie = ie[:, :nf].reshape(B * nf, -1).contiguous()
        it = self.context_embedding[1](self.context_embedding[0](ie, act=ACT_SILU))  # [B*nf, 4*ctx]
        ntext = encoder_hidden_states.shape[1]
        L = ntext + nlat + cfg.in_channels
        ctx = torch.empty((B, nf, L, cfg.cross_attention_dim), dtype=H16, device=self.device)
        ctx[:, :, :ntext] = encoder_hidden_states.to(self.device, H16)[:, None]
        ctx[:, :, ntext:ntext + nlat] = x.view(B, nf, nlat, -1)
        ctx[:, :, ntext + nlat:] = it.view(B, nf, cfg.in_channels, -1)
        return Context(ctx.view(B * nf * L, -1), L, 1 if multi_frame_guidance else F)

    def forward(self, sample, timestep, fps, image_latents, image_embeddings=None, encoder_hidden_states=None,
                cross_attention_kwargs=None, return_dict=True, conditioning=None, **_):
        out = self._forward(sample, timestep, fps, image_latents, image_latents, image_embeddings, encoder_hidden_states, False,
                            conditioning)
        return (out,)

    __call__ = forward

    def forward_ext(self, sample, timestep, fps, image_latents_first, image_latents, image_embeddings=None,
                    encoder_hidden_states=None, timestep_cond=None, cross_attention_kwargs=None,
                    multi_frame_guidance=False, return_dict=True, conditioning=None):
        out = self._forward(sample, timestep, fps, image_latents_first, image_latents, image_embeddings,
                            encoder_hidden_states, multi_frame_guidance, conditioning)
        return (out,)

    # ---- loop-invariant part of a forward ------------------------------------------------------------------
    def _conditioning(self, shape, image_latents_first, image_latents, image_embeddings, encoder_hidden_states,
                      multi_frame_guidance, keep=False):
        """context tokens + the image-latent half of the stem for a sample of ``shape`` [B,4,F,h,w]"""
        B, C, F, H, W = shape
        hw = H * W
        sh = self.shard
        f0, f1 = (0, F) if sh is None else sh.frame_range(F)
        if sh is not None:
            sh.check(F, hw)
        if sh is not None and multi_frame_guidance:  # per-frame context: this rank's frames only
            ie = image_embeddings if image_embeddings.dim() == 2 else image_embeddings[:, f0:f1]
            ctx = self._context(image_latents[:, :, f0:f1], ie, encoder_hidden_states, f1 - f0, True)
        else:  # one context per sample, built from frame 0 of the full clip
            ctx = self._context(image_latents, image_embeddings, encoder_hidden_states, f1 - f0, multi_frame_guidance)
        ctx.keep = keep
        # stem, image-latent half: image_latents_proj_in -> frame-axis encoder -> channels 4..7 of the conv_in input
        il = torch.empty((B * F * hw, 4), dtype=H16, device=self.device)
        ops.ncfhw_to_tokens(image_latents_first.to(self.device, H16), il)
        for i, (w_, b_) in enumerate(self.proj_in_convs):
            il, _, _ = ops.conv3x3_small(il, w_, b_, nimg=B * F, h=H, wd=W, cin=w_.shape[3], cout=w_.shape[0], silu=i < 2)
        stem8 = torch.empty((B * F * hw, 8), dtype=H16, device=self.device)  # channels 0..3: the sample, per step
        ops.temporal_encoder4(il, self.enc4_params, stem8, b=B, f=F, hw=hw, coff=4)
        return Conditioning(ctx, stem8, (B, F, H, W), (f0, f1), None)

    def prepare_conditioning(self, sample_shape, fps, image_latents_first, image_latents, image_embeddings=None,
                             encoder_hidden_states=None, multi_frame_guidance=False):
        """Hoist the loop-invariant work of a denoising loop out of its iterations: returns a
        ``Conditioning`` to pass as ``conditioning=`` to ``forward`` / ``forward_ext`` for as long as these inputs (and
        the frame shard) stay the same.  Per step this removes the context convs / embeddings, the 16 cross-attention K/V
        projections of the context and the image-latent stem (``fps`` is part of the time embedding and stays per step)."""
        if not self._loaded:
            raise RuntimeError("I2VGenXLUNet: load_state_dict() or init_random() first")
        cond = self._conditioning(tuple(sample_shape), image_latents_first, image_latents, image_embeddings,
                                  encoder_hidden_states, multi_frame_guidance, keep=True)
        for tr in self.spatial_transformers():
            cond.ctx.kv[tr] = tr.transformer_blocks[0].attn2.to_kv(cond.ctx.tokens)
        cond.key = (tuple(sample_shape), bool(multi_frame_guidance), id(self.shard))
        return cond

    def _forward_source_chunks(self, sample, timestep, fps, image_latents_first, image_latents, image_embeddings,
                               encoder_hidden_states, multi_frame_guidance, conditioning):
        """a conv_out-injection step (see ``prune_dead_chunks``): the network on the source chunks [bg, obj_1..obj_n] only,
        then the reference's conv_out blend (``pnp_utils.py:1114-1146``) writes the destination chunks from them"""
        B, C, F, H, W = sample.shape
        co = self.conv_out
        ndst = self.check_pnp_batch(B, co.mask)
        ns = B - ndst
        cut = lambda t, per=1: None if t is None else (t if (not torch.is_tensor(t)) or t.dim() == 0 or t.shape[0] != B * per
                                                       else t[:ns * per])
        cond = None
        if conditioning is not None:
            if conditioning.key != ((B, C, F, H, W), bool(multi_frame_guidance), id(self.shard)):
                raise RuntimeError(f"conditioning was prepared for {conditioning.key}, this call is "
                                   f"{((B, C, F, H, W), bool(multi_frame_guidance), id(self.shard))}")
            c0 = conditioning.ctx
            per = c0.tokens.shape[0] // B
            ctx = Context(c0.tokens[:ns * per], c0.length, c0.frames_per_ctx)
            ctx.kv = {k: v[:ns * per] for k, v in c0.kv.items()}
            cond = Conditioning(ctx, conditioning.stem8[:ns * F * H * W], (ns, F, H, W), conditioning.frames,
                                ((ns, C, F, H, W), bool(multi_frame_guidance), id(self.shard)))
        self._pruned = True
        try:
            src = self._forward(sample[:ns], cut(timestep), cut(fps), cut(image_latents_first), cut(image_latents),
                                cut(image_embeddings), cut(encoder_hidden_states), multi_frame_guidance, cond)
        finally:
            self._pruned = False
        nchw = torch.empty((B * F, C, H, W), dtype=H16, device=self.device)
        nchw[:ns * F] = src.permute(0, 2, 1, 3, 4).reshape(ns * F, C, H, W)
        ops.pnp_blend_nchw(nchw, self.device_masks(co.mask)[1], frames=F, base_chunk0=True, ndst=ndst)
        return nchw.reshape(B, F, C, H, W).permute(0, 2, 1, 3, 4).contiguous()

    def spatial_transformers(self):
        for blk in list(self.down_blocks) + [self.mid_block] + list(self.up_blocks):
            yield from blk.attentions

    @torch.no_grad()
    def _forward(self, *args, **kw):
        try:
            return self._forward_impl(*args, **kw)
        finally:
            self._tall = None  # the batched time projection belongs to this call only (a resnet run on its own recomputes it)

    def _forward_impl(self, sample, timestep, fps, image_latents_first, image_latents, image_embeddings, encoder_hidden_states,
                      multi_frame_guidance, conditioning=None):
        if not self._loaded:
            raise RuntimeError("I2VGenXLUNet: load_state_dict() or init_random() first")
        cfg = self.config
        sample = sample.to(self.device, H16)
        B, C, F, H, W = sample.shape
        hw = H * W
        co = self.conv_out
        if self.prune_dead_chunks and not self._pruned and self.shard is None and co.injecting():
            return self._forward_source_chunks(sample, timestep, fps, image_latents_first, image_latents, image_embeddings,
                                               encoder_hidden_states, multi_frame_guidance, conditioning)
        self._tail_site = None
        if self.prune_source_tail and not self._pruned and self.shard is None and not co.injecting():
            last = self.up_blocks[-1].temp_attentions[-1] if self.up_blocks[-1].has_cross_attention else None
            if last is not None and last.transformer_blocks[0].attn1.processor.injecting():
                self._tail_site = last
        B_full = B
        up_factor = 2 ** self.num_upsamplers
        forward_upsample_size = any(s % up_factor != 0 for s in (H, W))
        temb_act = self._embeddings(timestep, fps, B)
        self._tall = self.time_proj_all(temb_act)
        sh = self.shard
        if conditioning is None:
            conditioning = self._conditioning((B, C, F, H, W), image_latents_first, image_latents, image_embeddings,
                                              encoder_hidden_states, multi_frame_guidance)
        elif conditioning.key != ((B, C, F, H, W), bool(multi_frame_guidance), id(sh)):
            raise RuntimeError(f"conditioning was prepared for {conditioning.key}, this call is "
                               f"{((B, C, F, H, W), bool(multi_frame_guidance), id(sh))}")
        ctx = conditioning.ctx
        f0, f1 = conditioning.frames

        # stem: [sample | encoded image latents] -> conv_in -> transformer_in
        x8 = conditioning.stem8.clone() if conditioning.key is not None else conditioning.stem8  # cached: keep the template
        ops.ncfhw_to_tokens(sample, x8, coff=0)
        if sh is not None:
            # the 4-channel stem (three small convs + the frame-axis encoder) is computed for the whole clip on every
            # rank -- 0.01 % of the step; from conv_in on, the rank holds only its own frames
            x8 = x8.view(B, F, hw, 8)[:, f0:f1].reshape(-1, 8)
            F = f1 - f0
        geo = (B, F, H, W)
        self._expand_site = None
        if (self.shared_prefix_chunks == 2 and B >= 2 and sh is None and not self._pruned and
                self.down_blocks[0].has_cross_attention):
            # the last two chunks are equal up to the first cross-attention: the prefix on B - 1 chunks
            self._expand_site = self.down_blocks[0].attentions[0]
            geo = (B - 1, F, H, W)
            x8 = x8[:(B - 1) * F * hw]
        x, _, _ = ops.conv3x3(x8, self.conv_in.w, self.conv_in.b, nimg=geo[0] * F, h=H, wd=W, n_store=self.conv_in.cout)
        x = self.transformer_in.forward(self, x, geo)

        skips = [(x, geo)]
        if self._expand_site is not None:
            skips = [(self.expand_last_chunk(x, F * hw), (B, F, H, W))]  # (the decoder's last resnet reads it with every chunk)
        for blk in self.down_blocks:
            for j, rn in enumerate(blk.resnets):
                x = rn.forward(self, x, None, temb_act, geo)
                x = blk.temp_convs[j].forward(self, x, geo)
                if blk.has_cross_attention:
                    x = blk.attentions[j].forward(self, x, geo, ctx)
                    if blk.attentions[j] is self._expand_site:
                        geo = (B, F, H, W)
                        self._expand_site = None
                    x = blk.temp_attentions[j].forward(self, x, geo)
                skips.append((x, geo))
            if blk.downsamplers is not None:
                x, geo = blk.downsamplers[0].forward(x, geo)
                skips.append((x, geo))
        m = self.mid_block
        x = m.resnets[0].forward(self, x, None, temb_act, geo)
        x = m.temp_convs[0].forward(self, x, geo)
        x = m.attentions[0].forward(self, x, geo, ctx)
        x = m.temp_attentions[0].forward(self, x, geo)
        x = m.resnets[1].forward(self, x, None, temb_act, geo)
        x = m.temp_convs[1].forward(self, x, geo)

        for i, blk in enumerate(self.up_blocks):
            n = len(blk.resnets)
            res, skips = skips[-n:], skips[:-n]
            upsample_size = None
            if i != len(self.up_blocks) - 1 and forward_upsample_size:
                upsample_size = skips[-1][1][2:]
            for j, rn in enumerate(blk.resnets):
                skip, sgeo = res[-1]
                res = res[:-1]
                assert sgeo == geo, (sgeo, geo)
                x = rn.forward(self, x, skip, temb_act, geo)
                x = blk.temp_convs[j].forward(self, x, geo)
                if blk.has_cross_attention:
                    x = blk.attentions[j].forward(self, x, geo, ctx)
                    x = blk.temp_attentions[j].forward(self, x, geo)
                    if blk.temp_attentions[j] is self._tail_site:  # the destination chunks' rows only from here on
                        B = x.shape[0] // (F * hw)
                        geo = (B, F, H, W)
            if blk.upsamplers is not None:
                x, geo = blk.upsamplers[0].forward(x, geo, upsample_size)

        h = ops.groupnorm(x, *self.conv_norm_out, nsample=B * F, rows_per_sample=hw, groups=cfg.norm_num_groups, eps=1e-5,
                          silu=True)
        y, _, _ = ops.conv3x3(h, co.w, co.b, nimg=B * F, h=H, wd=W, n_store=co.cout)
        if co.injecting() and not self._pruned:
            ndst = self.check_pnp_batch(B, co.mask)
            # conv_out writes cout (4) channels into a [rows, 4] buffer: the token kernel needs channels % 8 == 0,
            # so this tiny tensor goes through the NCHW form of the kernel on the boundary layout instead
            out = ops.tokens_to_ncfhw(y, B, co.cout, F, H, W)  # [B,C,F,h,w]
            nchw = out.permute(0, 2, 1, 3, 4).reshape(B * F, co.cout, H, W).contiguous()
            ops.pnp_blend_nchw(nchw, self.device_masks(co.mask)[1], frames=F, base_chunk0=True, ndst=ndst)
            out = nchw.reshape(B, F, co.cout, H, W).permute(0, 2, 1, 3, 4).contiguous()
        else:
            out = ops.tokens_to_ncfhw(y, B, co.cout, F, H, W)
        if B != B_full:  # prune_source_tail: the source chunks' (never read) outputs are zeros
            full = torch.zeros((B_full,) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
            full[B_full - B:] = out
            out = full
        self._tail_site = self._expand_site = None
        return out if sh is None else sh.gather_frames(out, dim=2)
