"""Thin tensor-level wrappers over the C ABI (include/mvoc_hip.h).

PyTorch supplies device memory and the current HIP stream; every computation is a libmvoc_hip kernel.
All activations are fp16, channels-last 2-D ``[rows, C]`` tensors (rows = B*F*H*W).
"""
import ctypes as C
import os
import threading

import torch

from . import _ffi
from ._ffi import (A_CONV3X3, A_PLAIN, A_TEMPORAL3, ACT_GEGLU, ACT_GELU, ACT_NONE, ACT_SILU, AttnDesc, GemmDesc, GnDesc,
                   PnpDesc, TAttnDesc, TFusedDesc, XsDesc, check, lib)

__all__ = ["linear", "conv3x3", "tconv3", "flash_attn", "temporal_attn", "groupnorm", "groupnorm_moments", "groupnorm_apply_moments", "layernorm", "pnp_blend_tokens",
           "pnp_blend_nchw", "ddim_step", "latent_fusion", "timestep_embedding", "act", "add", "conv3x3_small",
           "adaptive_avgpool", "ncfhw_to_tokens", "tokens_to_ncfhw", "temporal_encoder4", "conv1x1_small", "softmax_rows", "ACT_NONE", "ACT_GEGLU",
           "ACT_SILU", "ACT_GELU"]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, dtype=torch.float16):
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f"mvoc_amd.ops: `{name}` must be a device tensor (this package has no CPU path)")
    if t.dtype != dtype:
        raise RuntimeError(f"mvoc_amd.ops: `{name}` must be {dtype}, got {t.dtype}")
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def _rowmajor(t, name):
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError(f"mvoc_amd.ops: `{name}` must be a 2-D row-major view, got {tuple(t.shape)} / {t.stride()}")
    return t.stride(0)


class _Conc(threading.local):
    n = 1


_CONCURRENCY = _Conc()
USE_ROW_MOMENTS = os.environ.get("MVOC_ROW_MOMENTS", "1") != "0"  # LayerNorm statistics from the producing GEMM's epilogue


def _gemm(d: GemmDesc, dev=None, out=None, sums=False, rowmom=False):
    # problems with few output tiles and a deep K get a scratch for deterministic split-K (see gemm.hip)
    if dev is not None and d.act != ACT_GEGLU and d.split_k != 1:
        nbytes = lib.mvoc_gemm_workspace_bytes(d.m, d.n, d.k)
        if nbytes:
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
    cs = None
    if out is not None:  # statistics ride on the tensor OBJECT: whatever an earlier producer hung on a caller-supplied `out` is stale now
        for attr in ("chan_sums", "row_moments"):
            if hasattr(out, attr):
                delattr(out, attr)
    d.concurrency = _CONCURRENCY.n
    if sums and out is not None and d.m % 256 == 0 and d.act == ACT_NONE and out.is_contiguous() and out.shape[1] == d.n_store:
        # REQUEST for the producer-epilogue GroupNorm statistics (include/mvoc_hip.h: chan_sums); honoured by the eight-phase tiles
        cs = torch.empty((d.m // 256, out.shape[1], 2), dtype=torch.float32, device=out.device)
        d.chan_sums = cs.data_ptr()
    rm = None
    if rowmom and USE_ROW_MOMENTS and out is not None and d.act != ACT_GEGLU and out.is_contiguous() and out.shape[1] == d.n:
        # REQUEST for the producer-epilogue LayerNorm statistics (include/mvoc_hip.h: row_moments); honoured by the eight-phase tiles
        ld = (d.n + 255) // 256
        rm = torch.empty((d.m, ld, 2), dtype=torch.float32, device=out.device)
        d.row_moments, d.row_moments_ld = rm.data_ptr(), ld
    check(lib.mvoc_gemm_f16(C.byref(d), _stream()), "gemm")
    if cs is not None and lib.mvoc_gemm_chan_sums_written():
        out.chan_sums = cs  # rides on the tensor OBJECT: a view / slice / copy of it carries no statistics
    if rm is not None:
        w_ = lib.mvoc_gemm_row_moments_written()
        if w_:
            out.row_moments = (rm, w_)  # likewise: (fp32 [m][ld][2], tile width)


class gemm_concurrency:
    """``with gemm_concurrency(n):`` -- the GEMM launches issued (or CAPTURED) inside by THIS thread run beside n - 1 independent
    launches of the same shape on other streams; passed to every call as ``mvoc_gemm_desc.concurrency`` (include/mvoc_hip.h).
    Host-side and thread-local: the library itself keeps no such state."""

    def __init__(self, n):
        self.n = max(1, int(n))

    def __enter__(self):
        self.old, _CONCURRENCY.n = _CONCURRENCY.n, self.n
        return self

    def __exit__(self, *exc):
        _CONCURRENCY.n = self.old


def chan_sums_of(x, rows_per_sample):
    """the producer's per-slab channel sums of the rows in ``x`` if the GEMM that wrote it emitted them and they fit the norm"""
    cs = getattr(x, "chan_sums", None)
    if cs is None or rows_per_sample % 256 or cs.shape[0] * 256 != x.shape[0] or cs.shape[1] != x.shape[1]:
        return None
    return cs


def _fill_common(d, x, x2, w, out, bias, rowadd, rowadd_div, resid, act, n_store, tile):
    d.a, d.a2, d.w, d.out = _ptr(x), _ptr(x2), _ptr(w), _ptr(out)
    d.bias, d.rowadd, d.resid = _ptr(bias), _ptr(rowadd), _ptr(resid)
    d.n = w.shape[0]
    d.k = w.shape[1]
    d.n_store = n_store
    d.ldo = out.stride(0)
    d.ldr = resid.stride(0) if resid is not None else 0
    d.ld_rowadd = rowadd.stride(0) if rowadd is not None else 0
    d.rowadd_div = rowadd_div
    d.lda = x.stride(0)
    d.lda2 = x2.stride(0) if x2 is not None else 0
    d.act = act
    d.tile = tile


def _out_cols(w, n_store, act):
    if act == ACT_GEGLU:
        return w.shape[0] // 2
    return n_store if n_store else w.shape[0]


def linear(x, w, bias=None, *, x2=None, act=ACT_NONE, resid=None, n_store=0, out=None, rowadd=None, rowadd_div=1, tile=0,
           split_k=0, ln=None, sums=False, rowmom=False):
    """out[M, n] = act(x @ w.T + bias) (+ resid).  ``x2``: second source of a channel concat ([x | x2] @ w.T).
    ``ln=(rowsum fp32 [n], lnbias fp32 [n], eps)``: LayerNorm(x) folded into the GEMM (w must be gamma-scaled).
    ``rowmom``: ask for the row statistics of the OUTPUT (the LayerNorm that reads it next: ``row_stats_of``)."""
    _chk(x, "x"), _chk(w, "w"), _chk(bias, "bias"), _chk(x2, "x2"), _chk(resid, "resid"), _chk(rowadd, "rowadd")
    _rowmajor(x, "x")
    m = x.shape[0]
    k1 = x.shape[1]
    k2 = x2.shape[1] if x2 is not None else 0
    if k1 + k2 != w.shape[1]:
        raise RuntimeError(f"linear: K mismatch {k1}+{k2} vs weight {tuple(w.shape)}")
    cols = _out_cols(w, n_store, act)
    if out is None:
        out = torch.empty((m, cols), dtype=torch.float16, device=x.device)
    d = GemmDesc()
    _fill_common(d, x, x2, w, out, bias, rowadd, rowadd_div, resid, act, cols if act != ACT_GEGLU else 0, tile)
    d.m = m
    d.a_mode = A_PLAIN
    d.c1 = k1
    d.cin = k1 + k2
    d.split_k = split_k
    if ln is not None:
        rowsum, lnbias, eps = ln[:3]
        _chk(rowsum, "ln rowsum", torch.float32), _chk(lnbias, "ln bias", torch.float32)
        d.ln_rowsum, d.ln_bias, d.ln_eps, d.split_k = rowsum.data_ptr(), lnbias.data_ptr(), eps, 1
        stats = ln[3] if len(ln) > 3 and ln[3] is not None else row_stats_of(x, eps)  # (the library has no in-kernel statistics any more)
        d.ln_stats = _chk(stats, "ln stats", torch.float32).data_ptr()
    _gemm(d, x.device, out, sums, rowmom)
    return out


# ---- chunk-major K order of the conv / temporal-conv weights (include/mvoc_hip.h: mvoc_gemm_desc.k_order) -----------------------------------
K_ORDER_CHUNK = os.environ.get("MVOC_KORDER", "1") != "0"
_CHUNK_CACHE = {}


def chunk_major_weights(w, ntaps):
    """tap-major conv weights [N, ntaps * cin] (k = tap * cin + c; unet.pack_conv3x3 / pack_tconv) -> chunk-major
    [N, (cin / 64) * ntaps * 64] (k = (c / 64) * ntaps * 64 + tap * 64 + c % 64).  Built once per weight tensor and kept (the weights
    of a loaded network are persistent; the cache holds a reference to the source so its address cannot be reused)."""
    key = (w.data_ptr(), tuple(w.shape), ntaps)
    hit = _CHUNK_CACHE.get(key)
    if hit is not None and hit[0] is w:
        return hit[1]
    n, k = w.shape
    cin = k // ntaps
    if k != ntaps * cin or cin % 64:
        raise RuntimeError(f"chunk_major_weights: k = {k} is not {ntaps} taps of a multiple of 64 channels")
    wc = w.view(n, ntaps, cin // 64, 64).permute(0, 2, 1, 3).reshape(n, k).contiguous()
    _CHUNK_CACHE[key] = (w, wc)
    return wc


def _g8_choice(m, n, conc, geglu=False):
    """gemm.hip's choice for an un-forced launch (the `eff` model there): 82 / 81 when the 320- / 256-wide eight-phase grid fills
    >= 55 % of the chip after quantisation and prices better, else 0 (another tile family takes the launch)"""
    def eff(bx, rate):
        nt = (n + bx - 1) // bx
        blocks = ((m + 255) // 256) * nt * conc
        return n / (nt * bx) * blocks / (((blocks + 255) // 256) * 256) * rate
    e81 = eff(256, 1.0)
    e82 = eff(320, 0.85) if (not geglu and n % 320 == 0) else 0.0
    if e81 < 0.55 and e82 < 0.55:
        return 0
    return 82 if e82 > e81 else 81


def _chunk_ok(d, x, x2, w, out, bias, resid, rowadd, ntaps, rows_a):
    """mirror of gemm.hip's g8_ok for a conv / temporal launch + the grid-fill rule: a k_order = 1 request the library cannot place on
    the eight-phase tiles is an error there (the caller holds the tap-major weights), so it is only made where it will be taken"""
    lim = 1 << 31
    al = lambda t: t is None or t.data_ptr() % 16 == 0
    return (K_ORDER_CHUNK and d.cin % 64 == 0 and d.c1 % 64 == 0 and d.k == ntaps * d.cin and d.m >= 1024 and d.tile in (0, 81, 82) and not d.upsample
            and d.split_k <= 1 and d.pad_mode == 0
            and out.stride(0) % 8 == 0 and d.n_store % 8 == 0 and al(out) and al(bias) and al(resid) and al(rowadd)
            and (resid is None or resid.stride(0) % 8 == 0) and (rowadd is None or (rowadd.stride(0) % 8 == 0 and d.rowadd_div >= 64))
            and rows_a * x.stride(0) * 2 < lim and (x2 is None or rows_a * x2.stride(0) * 2 < lim) and w.shape[0] * d.k * 2 < lim
            and d.m * out.stride(0) * 2 < lim and (resid is None or d.m * resid.stride(0) * 2 < lim)
            # Measured (profiles/r6/gemm_per_shape_pmc_korder.txt): FETCH falls 3-7 x on every conv (L2 hit rate 52 -> 85-92 %) and the
            # held clock rises; the 320-wide tile, whose activation offsets are formed at issue time anyway, gains (temporal K = 960:
            # - 7 %); the 256-wide tile paid 7-18 % for rebuilding its four offset registers EVERY K tile -- so the library has the
            # form for the 320-wide tile on an affine gather only (gemm8.hip: KO)
            and w.shape[0] % 320 == 0 and (d.tile == 82 or (d.tile == 0 and _g8_choice(d.m, w.shape[0], _CONCURRENCY.n) == 82))
            and (d.a_mode != A_CONV3X3 or (d.stride == 1 and d.hsrc == d.hout and d.wsrc == d.wout)))


SUBPIXEL_MIN_TILES = 200  # the sub-pixel form of Upsample2D + conv needs a grid that fills the chip; under it the 9-tap split-K form
#                           stays (8 -> 16 at batch 1).  Tests set 0 to put every upsampler of a small forward on the sub-pixel form.


def conv3x3(x, w, bias, *, nimg, h, wd, x2=None, stride=1, upsample_to=None, rowadd=None, rowadd_div=1, resid=None,
            n_store=0, out=None, tile=0, split_k=0, pad_mode=0, sums=False, w_subpixel=None):
    """3x3 conv, pad 1, on channels-last images x [nimg*h*wd, C1] (+ x2 [.., C2]); w [N, Kpad>=9*(C1+C2)] tap-major.
    ``upsample_to=(H2, W2)`` folds a nearest upsample of the source into the gather (Upsample2D).
    ``pad_mode=1``: zeros at the bottom / right only (``F.pad(x, (0,1,0,1))`` + ``conv2d(padding=0)``: VAE downsamplers).
    ``w_subpixel`` (``unet.pack_conv3x3_subpixel(w)``): with an exact 2x ``upsample_to`` the conv runs in its sub-pixel form -- four
    2 x 2 parity kernels on the source image, 4 taps of matrix work instead of 9 (include/mvoc_hip.h: upsample == 2)."""
    _chk(x, "x"), _chk(w, "w"), _chk(bias, "bias"), _chk(x2, "x2"), _chk(resid, "resid"), _chk(rowadd, "rowadd")
    _rowmajor(x, "x")
    c1 = x.shape[1]
    cin = c1 + (x2.shape[1] if x2 is not None else 0)
    hup, wup = (upsample_to if upsample_to is not None else (h, wd))
    pt = 2 if pad_mode == 0 else 1  # total padding per axis
    ho = (hup + pt - 3) // stride + 1
    wo = (wup + pt - 3) // stride + 1
    m = nimg * ho * wo
    cols = _out_cols(w, n_store, ACT_NONE)
    if out is None:
        out = torch.empty((m, cols), dtype=torch.float16, device=x.device)
    d = GemmDesc()
    _fill_common(d, x, x2, w, out, bias, rowadd, rowadd_div, resid, ACT_NONE, cols, tile)
    d.m = m
    d.a_mode = A_CONV3X3
    d.c1, d.cin = c1, cin
    d.nimg, d.hout, d.wout, d.hsrc, d.wsrc, d.stride = nimg, ho, wo, h, wd, stride
    d.upsample = 1 if upsample_to is not None else 0
    d.hup, d.wup = hup, wup
    d.split_k = split_k
    d.pad_mode = pad_mode
    lim = 1 << 31  # the eight-phase tiles address every operand through 32-bit MUBUF offsets (gemm.hip: g8_ok -- mirrored here,
    #               because a sub-pixel request the library cannot place on them is an error there, not a fall-back)
    if (w_subpixel is not None and upsample_to is not None and (hup, wup) == (2 * h, 2 * wd) and stride == 1 and x2 is None and
            resid is None and rowadd is None and pad_mode == 0 and (nimg * h * wd) % 256 == 0 and m >= 1024 and c1 % 64 == 0 and
            tile in (0, 81) and out.stride(0) % 8 == 0 and cols % 8 == 0 and
            w_subpixel.is_contiguous() and out.data_ptr() % 16 == 0 and (bias is None or bias.data_ptr() % 16 == 0) and
            nimg * h * wd * x.stride(0) * 2 < lim and m * out.stride(0) * 2 < lim and w.shape[0] * 4 * cin * 2 < lim and
            (tile == 81 or (m // 256) * ((w.shape[0] + 255) // 256) * _CONCURRENCY.n >= SUBPIXEL_MIN_TILES)):  # (an under-filled grid keeps the split-K form)
        _chk(w_subpixel, "w_subpixel")
        if tuple(w_subpixel.shape) != (4 * w.shape[0], 4 * cin):
            raise RuntimeError("conv3x3: w_subpixel must be pack_conv3x3_subpixel() of the same kernel")
        d.upsample, d.w, d.k, d.split_k = 2, w_subpixel.data_ptr(), 4 * cin, 1
        sums = False
    elif _chunk_ok(d, x, x2, w, out, bias, resid, rowadd, 9, nimg * h * wd):
        d.w, d.k_order, d.split_k = chunk_major_weights(w, 9).data_ptr(), 1, 1
    _gemm(d, x.device, out, sums)
    return out, ho, wo


def tconv3(x, w, bias, *, nvid, frames, hw, resid=None, out=None, tile=0, split_k=0, sums=False):
    """Conv3d (3,1,1), pad (1,0,0), on x [nvid*frames*hw, C]; w [N, 3*C] tap-major."""
    _chk(x, "x"), _chk(w, "w"), _chk(bias, "bias"), _chk(resid, "resid")
    _rowmajor(x, "x")
    m = nvid * frames * hw
    if out is None:
        out = torch.empty((m, w.shape[0]), dtype=torch.float16, device=x.device)
    d = GemmDesc()
    _fill_common(d, x, None, w, out, bias, None, 1, resid, ACT_NONE, w.shape[0], tile)
    d.m = m
    d.a_mode = A_TEMPORAL3
    d.c1 = d.cin = x.shape[1]
    d.frames, d.hw = frames, hw
    d.split_k = split_k
    if _chunk_ok(d, x, None, w, out, bias, resid, None, 3, m):
        d.w, d.k_order, d.split_k = chunk_major_weights(w, 3).data_ptr(), 1, 1
    _gemm(d, x.device, out, sums)
    return out


class _FlashMode(threading.local):
    field = 0


_FLASH_MODE = _FlashMode()


def flash_pipelined(mode):
    """kernel choice of the flash_attn calls THIS thread makes from now on (passed per call as ``mvoc_attn_desc.pipelined``; the library
    itself keeps no such state): -1 by key count (default), 0 the phase kernel always, 1 the software-pipelined kernel wherever it
    applies; same bits either way"""
    _FLASH_MODE.field = {-1: 0, 0: 1, 1: 2}[int(mode)]


def flash_attn(q, k, v, *, nbatch, heads, tq, tk, kv_bdiv=1, out=None, head_dim=64, causal=False, scale=0.0, v2=None, out2=None):
    """softmax(q k^T * scale) v.  q [nbatch*tq, >=heads*head_dim] / k, v [(nbatch/kv_bdiv)*tk, ..] row-major views.
    head_dim 64 (scale 1/8) is the UNet's; 96 + an explicit scale + ``causal`` serve the CLIP towers (clip.py).
    ``v2`` / ``out2`` (views laid out like ``v`` / ``out``): a second value tensor attending with the same q and k -- the PnP
    destination pair (include/mvoc_hip.h); the result equals two plain calls bit for bit."""
    _chk(q, "q"), _chk(k, "k"), _chk(v, "v"), _chk(v2, "v2"), _chk(out2, "out2")
    if out is None:
        out = torch.empty((nbatch * tq, heads * head_dim), dtype=torch.float16, device=q.device)
    d = AttnDesc()
    d.q, d.k, d.v, d.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    if v2 is not None:
        if out2 is None or _rowmajor(v2, "v2") != _rowmajor(v, "v") or _rowmajor(out2, "out2") != _rowmajor(out, "out"):
            raise RuntimeError("flash_attn: the paired form needs out2, and v2 / out2 laid out like v / out")
        d.v2, d.out2 = v2.data_ptr(), out2.data_ptr()
    d.q_ts, d.k_ts, d.v_ts, d.o_ts = _rowmajor(q, "q"), _rowmajor(k, "k"), _rowmajor(v, "v"), _rowmajor(out, "out")
    d.q_bs, d.k_bs, d.v_bs, d.o_bs = tq * d.q_ts, tk * d.k_ts, tk * d.v_ts, tq * d.o_ts
    d.nbatch, d.heads, d.tq, d.tk, d.kv_bdiv = nbatch, heads, tq, tk, kv_bdiv
    d.head_dim, d.causal, d.scale = head_dim, int(bool(causal)), scale
    d.pipelined = _FLASH_MODE.field
    check(lib.mvoc_flash_attn_f16(C.byref(d), _stream()), "flash_attn")
    return out


def temporal_attn(q, k, v, *, nsample, frames, hw, heads, out=None):
    """Attention over the frame axis of canonical [nsample, frames, hw, C] rows (q/k/v may be column slices)."""
    _chk(q, "q"), _chk(k, "k"), _chk(v, "v")
    if out is None:
        out = torch.empty((nsample * frames * hw, heads * 64), dtype=torch.float16, device=q.device)
    d = TAttnDesc()
    d.q, d.k, d.v, d.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    for name, t in (("q", q), ("k", k), ("v", v), ("o", out)):
        ld = _rowmajor(t, name)
        setattr(d, name + "_ps", ld)
        setattr(d, name + "_ts", hw * ld)
        setattr(d, name + "_bs", frames * hw * ld)
    d.nsample, d.hw, d.heads, d.frames = nsample, hw, heads, frames
    check(lib.mvoc_temporal_attn_f16(C.byref(d), _stream()), "temporal_attn")
    return out


XS_K = (64, 128, 320)   # activation-stationary kernels: the rows' K channels live in registers


def xs_linear(x, wp, n, *, normalize=False, eps=1e-5, act=ACT_NONE, resid=None, n_store=0, out=None, set_rows=0):
    """activation-stationary linear (include/mvoc_hip.h: mvoc_xs_linear_f16): x contiguous [m, k]; wp = unet.pack_xs_weights(W
    [n, k], constants [n]) (fragment-ordered weights + the per-channel constants); ``normalize``: LayerNorm folded (rows
    normalised in registers; W gamma-scaled, constants = beta @ W^T + bias)"""
    _chk(x, "x"), _chk(wp, "wp"), _chk(resid, "resid")
    m, k = x.shape
    nsets = m // set_rows if set_rows else 1
    if not x.is_contiguous() or k % 16 or n % 32 or wp.numel() != nsets * (n // 32) * (k // 16 + 1) * 512 or (set_rows and m % set_rows):
        raise RuntimeError("xs_linear: x must be contiguous [m, k] and wp the pack_xs_weights() image of an [n, k] matrix "
                           "(one image per set_rows rows when set_rows is given)")
    cols = n // 2 if act == ACT_GEGLU else (n_store if n_store else n)
    if out is None:
        out = torch.empty((m, cols), dtype=torch.float16, device=x.device)
    d = XsDesc()
    d.x, d.wp, d.resid, d.out = x.data_ptr(), wp.data_ptr(), _ptr(resid), out.data_ptr()
    d.m, d.n, d.k, d.n_store, d.ldo = m, n, k, cols, _rowmajor(out, "out")
    d.ldr = _rowmajor(resid, "resid") if resid is not None else 0
    d.act, d.normalize, d.ln_eps, d.wp_set_rows = act, int(bool(normalize)), eps, set_rows
    check(lib.mvoc_xs_linear_f16(C.byref(d), _stream()), "xs_linear")
    return out


TFUSED_CHANNELS = (64, 128, 320)
TFUSED_FRAMES = (8, 16, 32)


def temporal_qkv_attn(x, wp, ln, *, nsample, frames, hw, heads, out=None):
    """LayerNorm -> QKV -> attention over the frame axis in one kernel (see include/mvoc_hip.h): x raw rows
    [nsample*frames*hw, c], wp the fragment-packed gamma-scaled QKV weights, ln = (rowsum fp32 [3c], bias fp32 [3c], eps)."""
    _chk(x, "x"), _chk(wp, "wp"), _chk(ln[0], "ln rowsum", torch.float32), _chk(ln[1], "ln bias", torch.float32)
    c = heads * 64
    if not x.is_contiguous() or tuple(x.shape) != (nsample * frames * hw, c):
        raise RuntimeError(f"temporal_qkv_attn: x must be contiguous [{nsample * frames * hw}, {c}], got {tuple(x.shape)}")
    if wp.numel() != 3 * c * c or ln[0].numel() != 3 * c or ln[1].numel() != 3 * c:
        raise RuntimeError("temporal_qkv_attn: packed weights / LayerNorm vectors do not match c")
    if out is None:
        out = torch.empty_like(x)
    d = TFusedDesc()
    d.x, d.wp, d.ln_rowsum, d.ln_bias, d.out = x.data_ptr(), wp.data_ptr(), ln[0].data_ptr(), ln[1].data_ptr(), out.data_ptr()
    d.nsample, d.frames, d.hw, d.c, d.heads, d.ln_eps = nsample, frames, hw, c, heads, ln[2]
    check(lib.mvoc_temporal_qkv_attn_f16(C.byref(d), _stream()), "temporal_qkv_attn")
    return out


def _gn_desc(x, gamma, beta, x2, out, nsample, rows_per_sample, groups, eps, silu):
    _chk(x, "x"), _chk(gamma, "gamma"), _chk(beta, "beta"), _chk(x2, "x2")
    c1 = x.shape[1]
    c = c1 + (x2.shape[1] if x2 is not None else 0)
    if not x.is_contiguous() or (x2 is not None and not x2.is_contiguous()):
        raise RuntimeError("groupnorm: inputs must be contiguous")
    wsb = lib.mvoc_groupnorm_workspace_bytes(nsample, rows_per_sample, c, groups)
    ws = torch.empty((max(wsb, 4) + 3) // 4, dtype=torch.float32, device=x.device)
    d = GnDesc()
    d.x, d.x2, d.gamma, d.beta, d.out = x.data_ptr(), _ptr(x2), _ptr(gamma), _ptr(beta), _ptr(out)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    d.nsample, d.rows_per_sample, d.c, d.c1, d.groups, d.silu, d.eps = nsample, rows_per_sample, c, c1, groups, int(silu), eps
    return d, ws


USE_CHAN_SUMS = os.environ.get("MVOC_CHAN_SUMS", "1") != "0"  # MVOC_CHAN_SUMS=0: every GroupNorm reads its own statistics (A/B)


def groupnorm(x, gamma, beta, *, nsample, rows_per_sample, groups, eps, silu, x2=None, out=None):
    """GroupNorm over ``rows_per_sample`` rows x (C/groups) channels per sample, optional fused SiLU; [x | x2] concat.
    When the GEMMs that produced x (and x2) left their per-slab channel sums on the tensors (``_gemm``), the statistics pass
    over the input is skipped."""
    c = x.shape[1] + (x2.shape[1] if x2 is not None else 0)
    if out is None:
        out = torch.empty((nsample * rows_per_sample, c), dtype=torch.float16, device=x.device)
    d, ws = _gn_desc(x, gamma, beta, x2, out, nsample, rows_per_sample, groups, eps, silu)
    cs = chan_sums_of(x, rows_per_sample) if USE_CHAN_SUMS else None
    cs2 = chan_sums_of(x2, rows_per_sample) if (x2 is not None and cs is not None) else None
    if cs is not None and (x2 is None or cs2 is not None) and x.shape[1] % (c // groups) == 0:
        d.chan_sums, d.chan_sums2 = cs.data_ptr(), _ptr(cs2)
    check(lib.mvoc_groupnorm_f16(C.byref(d), _stream()), "groupnorm")
    return out


def groupnorm_fold_xs(x, gamma, beta, w, bias, *, nsample, rows_per_sample, groups, eps):
    """GroupNorm(x) folded into the linear ``w`` [n, k] (+ ``bias``) that reads it (include/mvoc_hip.h:
    mvoc_groupnorm_fold_xs_f16): the per-sample weight streams for ``xs_linear(x_raw, wp, n, set_rows=rows_per_sample)``; the
    statistics come from the producer's channel sums when x carries them"""
    _chk(w, "w"), _chk(bias, "bias")
    n, k = w.shape
    if k != x.shape[1] or n % 32 or k % 16 or not w.is_contiguous():
        raise RuntimeError("groupnorm_fold_xs: w must be contiguous [n % 32 == 0, k == channels of x]")
    d, ws = _gn_desc(x, gamma, beta, None, None, nsample, rows_per_sample, groups, eps, False)
    cs = chan_sums_of(x, rows_per_sample) if USE_CHAN_SUMS else None
    if cs is not None:
        d.chan_sums = cs.data_ptr()
    wp = torch.empty((nsample, n // 32, k // 16 + 1, 512), dtype=torch.float16, device=x.device)
    check(lib.mvoc_groupnorm_fold_xs_f16(C.byref(d), w.data_ptr(), _ptr(bias), n, k, wp.data_ptr(), _stream()), "groupnorm_fold_xs")
    return wp


def groupnorm_moments(x, *, nsample, rows_per_sample, groups):
    """This rank's {count, mean, M2} per (sample, group): fp32 [nsample, groups, 3] (first half of a sharded GroupNorm)."""
    mom = torch.empty((nsample, groups, 3), dtype=torch.float32, device=x.device)
    d, ws = _gn_desc(x, None, None, None, None, nsample, rows_per_sample, groups, 0.0, False)
    check(lib.mvoc_groupnorm_moments_f16(C.byref(d), mom.data_ptr(), _stream()), "groupnorm_moments")
    return mom


def groupnorm_apply_moments(x, parts, gamma, beta, *, nsample, rows_per_sample, groups, eps, silu, out=None):
    """Second half: ``parts`` fp32 [nparts, nsample, groups, 3] (all ranks' moments in rank order) -> normalised rows."""
    _chk(parts, "parts", torch.float32)
    if parts.dim() != 4 or tuple(parts.shape[1:]) != (nsample, groups, 3) or not parts.is_contiguous():
        raise RuntimeError(f"groupnorm_apply_moments: parts must be contiguous [nparts, {nsample}, {groups}, 3]")
    if out is None:
        out = torch.empty((nsample * rows_per_sample, x.shape[1]), dtype=torch.float16, device=x.device)
    d, ws = _gn_desc(x, gamma, beta, None, out, nsample, rows_per_sample, groups, eps, silu)
    check(lib.mvoc_groupnorm_apply_moments_f16(C.byref(d), parts.data_ptr(), parts.shape[0], _stream()), "groupnorm_apply_moments")
    return out


def row_stats(x, eps=1e-5):
    """{mean, rstd} per row, fp32 [rows, 2]"""
    _chk(x, "x")
    if not x.is_contiguous():
        raise RuntimeError("row_stats: x must be contiguous")
    out = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device)
    check(lib.mvoc_row_stats_f16(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], eps, _stream()), "row_stats")
    return out


def row_stats_of(x, eps=1e-5):
    """{mean, rstd} per row of x: from the row moments the producing GEMM left on the tensor (``_gemm``) when there are any -- a few
    bytes per row instead of a pass over the tensor -- else ``row_stats``"""
    rm = getattr(x, "row_moments", None)
    if rm is None or not USE_ROW_MOMENTS or rm[0].shape[0] != x.shape[0]:
        return row_stats(x, eps)
    mom, tile_w = rm
    out = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device)
    check(lib.mvoc_row_stats_from_moments_f32(mom.data_ptr(), x.shape[0], mom.shape[1], x.shape[1], tile_w, eps, out.data_ptr(), _stream()),
          "row_stats_from_moments")
    return out


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    _chk(x, "x"), _chk(gamma, "gamma"), _chk(beta, "beta")
    if not x.is_contiguous():
        raise RuntimeError("layernorm: x must be contiguous")
    if out is None:
        out = torch.empty_like(x)
    check(lib.mvoc_layernorm_f16(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                 eps, _stream()), "layernorm")
    return out


def _pnp_desc(x, x2, masks, chunk_stride, f_stride, p_stride, frames, height, width, channels, base_chunk0, ndst=2):
    _chk(x, "x"), _chk(x2, "x2"), _chk(masks, "masks")
    if masks.dim() != 4 or not masks.is_contiguous() or masks.shape[1] != frames:
        raise RuntimeError(f"pnp: masks must be contiguous [nobj, F, mh, mw] fp16, got {tuple(masks.shape)}")
    d = PnpDesc()
    d.x, d.x2, d.masks = x.data_ptr(), _ptr(x2), masks.data_ptr()
    d.chunk_stride, d.f_stride, d.p_stride = chunk_stride, f_stride, p_stride
    d.nobj, d.frames, d.height, d.width, d.channels = masks.shape[0], frames, height, width, channels
    d.mask_h, d.mask_w, d.base_chunk0, d.ndst = masks.shape[2], masks.shape[3], int(base_chunk0), int(ndst)
    return d


def pnp_blend_tokens(x, masks, *, frames, height, width, channels, chunk_stride, f_stride, p_stride, x2=None,
                     base_chunk0=False, ndst=2):
    """In-place masked blend + scatter on channel-contiguous data (see include/mvoc_hip.h).  ``ndst``: trailing
    destination chunks (2 = [uncond, cond], 1 = [cond] with CFG off)."""
    d = _pnp_desc(x, x2, masks, chunk_stride, f_stride, p_stride, frames, height, width, channels, base_chunk0, ndst)
    check(lib.mvoc_pnp_blend_scatter_tokens(C.byref(d), _stream()), "pnp_blend_scatter_tokens")
    return x


def pnp_blend_nchw(x, masks, *, frames, x2=None, base_chunk0=True, ndst=2):
    """In-place on x [(nobj+1+ndst)*F, C, H, W] (reference feature-map layout)."""
    if x.dim() != 4 or not x.is_contiguous():
        raise RuntimeError("pnp_blend_nchw: x must be contiguous [N, C, H, W]")
    d = _pnp_desc(x, x2, masks, 0, 0, 0, frames, x.shape[2], x.shape[3], x.shape[1], base_chunk0, ndst)
    check(lib.mvoc_pnp_blend_scatter_nchw(C.byref(d), _stream()), "pnp_blend_scatter_nchw")
    return x


def ddim_step(x, v_cond, coef_dev, v_uncond=None, out=None):
    """coef_dev: fp32 device tensor {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev), guidance_scale}."""
    _chk(x, "x"), _chk(v_cond, "v_cond"), _chk(v_uncond, "v_uncond"), _chk(coef_dev, "coef", torch.float32)
    if out is None:
        out = torch.empty_like(x)
    for t in (x, v_cond, v_uncond, out):
        if t is not None and not t.is_contiguous():
            raise RuntimeError("ddim_step: tensors must be contiguous")
    check(lib.mvoc_ddim_step_f16(x.data_ptr(), _ptr(v_uncond), v_cond.data_ptr(), coef_dev.data_ptr(), out.data_ptr(),
                                 x.numel(), _stream()), "ddim_step")
    return out


def latent_fusion(latents, bg, objs, masks, mix_ratio, obj_random_noise_fusion=False, out=None):
    """objs / masks: contiguous [nobj, *latents.shape] fp16."""
    _chk(latents, "latents"), _chk(bg, "bg"), _chk(objs, "objs"), _chk(masks, "masks")
    if out is None:
        out = torch.empty_like(latents)
    n = latents.numel()
    if objs.numel() != masks.numel() or objs.numel() % n:
        raise RuntimeError("latent_fusion: objs/masks must be [nobj, ...latents.shape]")
    check(lib.mvoc_latent_fusion_f16(latents.data_ptr(), bg.data_ptr(), objs.data_ptr(), masks.data_ptr(), out.data_ptr(),
                                     objs.numel() // n, n, float(mix_ratio), int(obj_random_noise_fusion), _stream()),
          "latent_fusion")
    return out


def timestep_embedding(t_dev, dim):
    _chk(t_dev, "t", torch.float32)
    out = torch.empty((t_dev.numel(), dim), dtype=torch.float16, device=t_dev.device)
    check(lib.mvoc_timestep_embedding_f16(t_dev.data_ptr(), t_dev.numel(), dim, out.data_ptr(), _stream()), "timestep_embedding")
    return out


def act(x, kind):
    _chk(x, "x")
    out = torch.empty_like(x)
    check(lib.mvoc_act_f16(x.data_ptr(), out.data_ptr(), x.numel(), kind, _stream()), "act")
    return out


def add(a, b):
    _chk(a, "a"), _chk(b, "b")
    out = torch.empty_like(a)
    check(lib.mvoc_add_f16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "add")
    return out


def conv3x3_small(x, w, bias, *, nimg, h, wd, cin, cout, stride=1, silu=False):
    _chk(x, "x"), _chk(w, "w"), _chk(bias, "bias")
    ho, wo = (h + 2 - 3) // stride + 1, (wd + 2 - 3) // stride + 1
    out = torch.empty((nimg * ho * wo, cout), dtype=torch.float16, device=x.device)
    check(lib.mvoc_conv3x3_small_f16(x.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), nimg, h, wd, cin, cout, stride,
                                     int(silu), _stream()), "conv3x3_small")
    return out, ho, wo


def adaptive_avgpool(x, *, nimg, h, w, c, oh, ow):
    _chk(x, "x")
    out = torch.empty((nimg * oh * ow, c), dtype=torch.float16, device=x.device)
    check(lib.mvoc_adaptive_avgpool_f16(x.data_ptr(), out.data_ptr(), nimg, h, w, c, oh, ow, _stream()), "adaptive_avgpool")
    return out


def ncfhw_to_tokens(x, out, coff=0):
    """x [B,C,F,h,w] -> out [B*F*h*w, ld] channels coff..coff+C"""
    _chk(x, "x"), _chk(out, "out")
    b, c, f, h, w = x.shape
    x = x.contiguous()  # (a copy, if one is made, is held until the launch is enqueued)
    check(lib.mvoc_ncfhw_to_tokens_f16(x.data_ptr(), out.data_ptr(), b, c, f, h * w, out.stride(0), coff,
                                       _stream()), "ncfhw_to_tokens")
    return out


def tokens_to_ncfhw(x, b, c, f, h, w):
    _chk(x, "x")
    out = torch.empty((b, c, f, h, w), dtype=torch.float16, device=x.device)
    check(lib.mvoc_tokens_to_ncfhw_f16(x.data_ptr(), out.data_ptr(), b, c, f, h * w, x.stride(0), _stream()), "tokens_to_ncfhw")
    return out


def temporal_encoder4(x, params, out, *, b, f, hw, coff):
    _chk(x, "x"), _chk(params, "params"), _chk(out, "out")
    check(lib.mvoc_temporal_encoder4_f16(x.data_ptr(), params.data_ptr(), out.data_ptr(), b, f, hw, out.stride(0), coff,
                                         _stream()), "temporal_encoder4")
    return out


def conv1x1_small(x, w, bias):
    """out[r, o] = sum_c x[r, c] * w[o, c] + bias[o] for a handful of channels (the VAE's quant / post_quant 1x1 convs)"""
    _chk(x, "x"), _chk(w, "w"), _chk(bias, "bias")
    if not x.is_contiguous() or x.shape[1] != w.shape[1]:
        raise RuntimeError("conv1x1_small: x must be contiguous [rows, cin] with cin == w.shape[1]")
    out = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float16, device=x.device)
    w = w.contiguous()
    check(lib.mvoc_conv1x1_small_f16(x.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), x.shape[0], x.shape[1],
                                     w.shape[0], _stream()), "conv1x1_small")
    return out


def image_to_tokens(x):
    """[n, c, h, w] fp16 -> channels-last rows [n*h*w, c]"""
    _chk(x, "x")
    n, c, h, w = x.shape
    out = torch.empty((n * h * w, c), dtype=torch.float16, device=x.device)
    x = x.contiguous()
    check(lib.mvoc_image_to_tokens_f16(x.data_ptr(), out.data_ptr(), n, c, h * w, _stream()), "image_to_tokens")
    return out


def tokens_to_image(x, n, c, h, w):
    """channels-last rows [n*h*w, ld] (first c channels) -> [n, c, h, w]"""
    _chk(x, "x")
    out = torch.empty((n, c, h, w), dtype=torch.float16, device=x.device)
    check(lib.mvoc_tokens_to_image_f16(x.data_ptr(), out.data_ptr(), n, c, h * w, _rowmajor(x, "x"), _stream()), "tokens_to_image")
    return out


def permute_rows(x, shape4, perm):
    """x: contiguous [prod(shape4), c] rows viewed as [*shape4, c]; returns the rows of ``x.view(*shape4, c).permute(*perm, 4)``
    made contiguous ([prod, c]) -- the pack / unpack copies around the frame-shard exchanges as one HIP kernel"""
    _chk(x, "x")
    if not x.is_contiguous() or x.dim() != 2:
        raise RuntimeError("permute_rows: x must be contiguous [rows, c]")
    st = [shape4[1] * shape4[2] * shape4[3], shape4[2] * shape4[3], shape4[3], 1]
    dims = (C.c_int64 * 4)(*[shape4[p] for p in perm])
    strides = (C.c_int64 * 4)(*[st[p] for p in perm])
    out = torch.empty_like(x)
    check(lib.mvoc_permute_rows_f16(x.data_ptr(), out.data_ptr(), dims, strides, x.shape[1], _stream()), "permute_rows")
    return out


def softmax_rows(x):
    """in-place softmax over the last dim of a contiguous [rows, cols] fp16 matrix (fp32 math, one rounding)"""
    _chk(x, "x")
    if x.dim() != 2 or not x.is_contiguous():
        raise RuntimeError("softmax_rows: x must be contiguous [rows, cols]")
    check(lib.mvoc_softmax_rows_f16(x.data_ptr(), x.shape[0], x.shape[1], _stream()), "softmax_rows")
    return x


def prof_enable(on):
    lib.mvoc_prof_enable(int(on))


def prof_reset():
    lib.mvoc_prof_reset()


def prof_collect():
    n = len(_ffi.FAMILIES)
    ms = (C.c_double * n)()
    cnt = (C.c_int64 * n)()
    work = (C.c_double * n)()
    check(lib.mvoc_prof_collect(ms, cnt, work), "prof_collect")
    return {fam: {"ms": ms[i], "launches": cnt[i], "work": work[i]} for i, fam in enumerate(_ffi.FAMILIES)}


def delay_us(us):
    check(lib.mvoc_delay_us(int(us), _stream()), "delay_us")
