"""GPU parity of every libmvoc_hip op (through the C ABI) against the CPU oracle / the PyTorch CPU primitive the
reference's diffusers module calls.  Integer/bit-level work (PnP injection, DDIM step, latent fusion) must be
BIT-EXACT; floating-point kernels are held to the tolerance written in each test."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from mvoc_amd import ops as _ops
    return _ops


def dev(t):
    return t.to("cuda", torch.float16).contiguous()


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16)


# ---- GEMM family -----------------------------------------------------------------------------------
@pytest.mark.parametrize("tile", [1, 2, 3, 4, 11, 12, 13, 61, 62, 64, 81, 82])
@pytest.mark.parametrize("m", [128, 333, 2048])
def test_linear_exact_integers(ops, tile, m):
    """integer-valued operands: every product and partial sum is exact, so the MFMA operand / accumulator lane
    maps (asymmetric data) are checked bit for bit"""
    g = torch.Generator().manual_seed(tile * 1000 + m)
    n, k = 320, (320 if tile > 10 else 96)  # 5 K steps of 64 (10 of 32): ring prologue, steady state and drain
    x = torch.randint(-3, 4, (m, k), generator=g).float()
    w = torch.randint(-3, 4, (n, k), generator=g).float()
    b = torch.randint(-8, 9, (n,), generator=g).float()
    out = ops.linear(dev(x), dev(w), dev(b), tile=tile)
    ref = x @ w.t() + b
    assert torch.equal(out.float().cpu(), ref)


@pytest.mark.parametrize("n,k", [(320, 320), (640, 1280), (512, 320), (960, 64), (64, 1024)])
def test_linear_random(ops, n, k):
    g = torch.Generator().manual_seed(n + k)
    m = 777
    x = torch.randn(m, k, generator=g).half()
    w = (torch.randn(n, k, generator=g) / math.sqrt(k)).half()
    b = torch.randn(n, generator=g).half()
    r = torch.randn(m, n, generator=g).half()
    out = ops.linear(dev(x), dev(w), dev(b), resid=dev(r))
    ref = (x.float() @ w.float().t() + b.float()).half().float() + r.float()
    assert rel_l2(out, ref) < 1e-3
    assert (out.float().cpu() - ref).abs().max() < 2e-2


@pytest.mark.parametrize("split_k", [2, 4, 0])
def test_split_k(ops, split_k):
    """deterministic split-K (fp32 slabs summed in slice order): conv with temb row-add + residual, temporal conv, linear"""
    from mvoc_amd.unet import pack_conv3x3, pack_tconv
    g = torch.Generator().manual_seed(50 + split_k)
    n, c, cout, h, w, fr = 4, 256, 128, 8, 8, 2
    x = torch.randn(n, c, h, w, generator=g).half()
    wt = (torch.randn(cout, c, 3, 3, generator=g) / 48).half()
    b = torch.randn(cout, generator=g).half()
    temb = torch.randn(n // fr, cout, generator=g).half()
    res = torch.randn(n, cout, h, w, generator=g).half()
    ref = F.conv2d(x.float(), wt.float(), b.float(), padding=1) + temb.float().repeat_interleave(fr, 0)[:, :, None, None] + res.float()
    out, _, _ = ops.conv3x3(dev(_nhwc(x)), pack_conv3x3(dev(wt)), dev(b), nimg=n, h=h, wd=w, rowadd=dev(temb), rowadd_div=fr * h * w,
                            resid=dev(_nhwc(res)), n_store=cout, split_k=split_k)
    assert rel_l2(_from_rows(out, n, h, w), ref) < 1.5e-3
    out2, _, _ = ops.conv3x3(dev(_nhwc(x)), pack_conv3x3(dev(wt)), dev(b), nimg=n, h=h, wd=w, rowadd=dev(temb), rowadd_div=fr * h * w,
                             resid=dev(_nhwc(res)), n_store=cout, split_k=split_k)
    assert torch.equal(out, out2)  # run-to-run reproducible
    m, k, nn = 300, 2048, 192
    a = torch.randn(m, k, generator=g).half()
    wl = (torch.randn(nn, k, generator=g) / 45).half()
    o = ops.linear(dev(a), dev(wl), None, split_k=split_k, act=ops.ACT_SILU)
    assert rel_l2(o, F.silu((a.float() @ wl.float().t()).half().float())) < 2e-3
    nb, cc, frames, hw = 1, 768, 4, 16
    xt = torch.randn(nb, cc, frames, hw, 1, generator=g).half()
    wt3 = (torch.randn(cc, cc, 3, 1, 1, generator=g) / 48).half()
    rows = xt[..., 0].permute(0, 2, 3, 1).reshape(nb * frames * hw, cc)
    reft = F.conv3d(xt.float(), wt3.float(), None, padding=(1, 0, 0))
    ot = ops.tconv3(dev(rows), pack_tconv(dev(wt3)), None, nvid=nb, frames=frames, hw=hw, split_k=split_k)
    assert rel_l2(ot.reshape(nb, frames, hw, cc).permute(0, 3, 1, 2)[..., None], reft) < 1.5e-3


@pytest.mark.parametrize("c,n,geglu", [(320, 960, False), (64, 192, False), (1280, 1280, False), (128, 1024, True), (640, 5120, True)])
def test_linear_layernorm_fold(ops, c, n, geglu):
    """LayerNorm folded into the GEMM (row statistics accumulated in the K loop) vs F.layer_norm -> F.linear"""
    from mvoc_amd.unet import Linear, pack_geglu
    g = torch.Generator().manual_seed(c + n)
    m = 1000
    x = (torch.randn(m, c, generator=g) * 1.7 + 0.6).half()
    x[3] = x[3] * 8 + 5  # a row with a large mean: the mean*rowsum cancellation must hold up
    w = (torch.randn(n, c, generator=g) / math.sqrt(c)).half()
    b = torch.randn(n, generator=g).half()
    gm, bt = (1 + 0.3 * torch.randn(c, generator=g)).half(), (0.3 * torch.randn(c, generator=g)).half()
    y = F.layer_norm(x.float(), (c,), gm.float(), bt.float(), 1e-5).half().float() @ w.float().t() + b.float()
    if geglu:
        hh, gg = y.half().float().chunk(2, dim=-1)
        ref = hh * F.gelu(gg)
        wp, bp = pack_geglu(dev(w), dev(b))
        lin = Linear(wp, bp).fold_layernorm(dev(gm), dev(bt))
        out = lin.call_ln(dev(x), (dev(gm), dev(bt)), act=ops.ACT_GEGLU)
    else:
        ref = y
        lin = Linear(dev(w), dev(b)).fold_layernorm(dev(gm), dev(bt))
        out = lin.call_ln(dev(x), (dev(gm), dev(bt)))
    assert lin.ln is not None
    assert rel_l2(out, ref) < 2e-3
    assert (out.float().cpu() - ref).abs().max() < 3e-2 * ref.abs().max()
    # without precomputed statistics ops.linear takes them itself (mvoc_row_stats_f16): same result
    out2 = ops.linear(dev(x), lin.w_ln, None, n_store=lin.n, ln=lin.ln, act=ops.ACT_GEGLU if geglu else ops.ACT_NONE)
    assert rel_l2(out2, out) < 1e-3


def test_linear_concat_and_acts(ops):
    g = torch.Generator().manual_seed(5)
    m, k1, k2, n = 300, 128, 64, 192
    x1, x2 = torch.randn(m, k1, generator=g).half(), torch.randn(m, k2, generator=g).half()
    w = (torch.randn(n, k1 + k2, generator=g) / 14).half()
    b = torch.randn(n, generator=g).half()
    ref = torch.cat([x1, x2], 1).float() @ w.float().t() + b.float()
    out = ops.linear(dev(x1), dev(w), dev(b), x2=dev(x2))
    assert rel_l2(out, ref) < 1e-3
    out = ops.linear(dev(x1), dev(w), dev(b), x2=dev(x2), act=ops.ACT_SILU)
    assert rel_l2(out, F.silu(ref.half().float())) < 2e-3


def test_linear_geglu(ops):
    from mvoc_amd.unet import pack_geglu
    g = torch.Generator().manual_seed(6)
    m, c, inner = 257, 128, 512
    x = torch.randn(m, c, generator=g).half()
    w = (torch.randn(2 * inner, c, generator=g) / math.sqrt(c)).half()
    b = torch.randn(2 * inner, generator=g).half()
    y = x.float() @ w.float().t() + b.float()
    hh, gg = y.half().float().chunk(2, dim=-1)
    ref = hh * F.gelu(gg).half().float()
    wp, bp = pack_geglu(dev(w), dev(b))
    out = ops.linear(dev(x), wp, bp, act=ops.ACT_GEGLU)
    assert out.shape == (m, inner)
    assert rel_l2(out, ref) < 2e-3


def _nhwc(x):  # [n,c,h,w] -> rows
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c)


def _from_rows(r, n, h, w):
    return r.reshape(n, h, w, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("tile", [0, 11, 13, 61, 81, 82])
@pytest.mark.parametrize("cin,cout,h,w,stride", [(64, 64, 8, 8, 1), (32, 96, 7, 9, 1), (64, 128, 9, 6, 2), (8, 64, 8, 8, 1)])
def test_conv3x3(ops, cin, cout, h, w, stride, tile):
    if tile and cin % 64:
        pytest.skip("direct-to-LDS tiles need cin % 64 == 0")
    from mvoc_amd.unet import pack_conv3x3
    g = torch.Generator().manual_seed(cin + cout + h)
    n = 5
    x = torch.randn(n, cin, h, w, generator=g).half()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half()
    b = torch.randn(cout, generator=g).half()
    ref = F.conv2d(x.float(), wt.float(), b.float(), stride=stride, padding=1)
    out, ho, wo = ops.conv3x3(dev(_nhwc(x)), pack_conv3x3(dev(wt)), dev(b), nimg=n, h=h, wd=w, stride=stride, n_store=cout,
                              tile=tile)
    assert (ho, wo) == tuple(ref.shape[2:])
    assert rel_l2(_from_rows(out, n, ho, wo), ref) < 1.5e-3


@pytest.mark.parametrize("tile", [0, 11, 13, 61, 81, 82])
def test_conv3x3_concat_glds(ops, tile):
    """two-source gather (decoder skip concat) with both channel counts multiples of 64"""
    from mvoc_amd.unet import pack_conv3x3
    g = torch.Generator().manual_seed(15)
    n, c1, c2, cout, h, w = 4, 128, 64, 64, 7, 6
    x1, x2 = torch.randn(n, c1, h, w, generator=g).half(), torch.randn(n, c2, h, w, generator=g).half()
    wt = (torch.randn(cout, c1 + c2, 3, 3, generator=g) / 40).half()
    b = torch.randn(cout, generator=g).half()
    ref = F.conv2d(torch.cat([x1, x2], 1).float(), wt.float(), b.float(), padding=1)
    out, _, _ = ops.conv3x3(dev(_nhwc(x1)), pack_conv3x3(dev(wt)), dev(b), nimg=n, h=h, wd=w, x2=dev(_nhwc(x2)), n_store=cout,
                            tile=tile)
    assert rel_l2(_from_rows(out, n, h, w), ref) < 1.5e-3


def test_conv3x3_concat_temb_resid(ops):
    from mvoc_amd.unet import pack_conv3x3
    g = torch.Generator().manual_seed(11)
    n, c1, c2, cout, h, w, fr = 6, 64, 32, 64, 6, 5, 3
    x1, x2 = torch.randn(n, c1, h, w, generator=g).half(), torch.randn(n, c2, h, w, generator=g).half()
    wt = (torch.randn(cout, c1 + c2, 3, 3, generator=g) / 29).half()
    b = torch.randn(cout, generator=g).half()
    temb = torch.randn(n // fr, cout, generator=g).half()  # one row per sample (fr frames each)
    res = torch.randn(n, cout, h, w, generator=g).half()
    ref = F.conv2d(torch.cat([x1, x2], 1).float(), wt.float(), b.float(), padding=1)
    ref = ref + temb.float().repeat_interleave(fr, 0)[:, :, None, None] + res.float()
    out, _, _ = ops.conv3x3(dev(_nhwc(x1)), pack_conv3x3(dev(wt)), dev(b), nimg=n, h=h, wd=w, x2=dev(_nhwc(x2)),
                            rowadd=dev(temb), rowadd_div=fr * h * w, resid=dev(_nhwc(res)), n_store=cout)
    assert rel_l2(_from_rows(out, n, h, w), ref) < 1.5e-3


@pytest.mark.parametrize("tile", [0, 13, 81, 82])
@pytest.mark.parametrize("size", [None, (11, 7)])
def test_conv3x3_upsample(ops, size, tile):
    from mvoc_amd.unet import pack_conv3x3
    g = torch.Generator().manual_seed(12)
    n, c, h, w = 3, 64, 6, 4
    x = torch.randn(n, c, h, w, generator=g).half()
    wt = (torch.randn(c, c, 3, 3, generator=g) / 24).half()
    b = torch.randn(c, generator=g).half()
    up = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if size is None else F.interpolate(x.float(), size=size, mode="nearest")
    ref = F.conv2d(up, wt.float(), b.float(), padding=1)
    out, ho, wo = ops.conv3x3(dev(_nhwc(x)), pack_conv3x3(dev(wt)), dev(b), nimg=n, h=h, wd=w,
                              upsample_to=size or (2 * h, 2 * w), n_store=c, tile=tile)
    assert (ho, wo) == tuple(ref.shape[2:])
    assert rel_l2(_from_rows(out, n, ho, wo), ref) < 1.5e-3


@pytest.mark.parametrize("n,cin,cout,h,w", [(16, 64, 64, 8, 8), (4, 128, 320, 16, 16), (5, 64, 96, 16, 32)])
def test_upsample_conv_subpixel_form(ops, n, cin, cout, h, w):
    """Upsample2D + conv in sub-pixel form (mvoc_gemm_desc.upsample == 2: four 2 x 2 parity kernels on the source image instead of
    the 9-tap gather on the upsampled one): exact on integer operands (the summed kernels stay fp16-exact), against
    F.interpolate(nearest, 2x) + conv2d, and equal to the 9-tap form of the same call"""
    from mvoc_amd.unet import pack_conv3x3, pack_conv3x3_subpixel
    g = torch.Generator().manual_seed(n * 100 + cin + h)
    x = _ints(g, (n, cin, h, w), -2, 2)
    wt = _ints(g, (cout, cin, 3, 3))
    wt[torch.rand(wt.shape, generator=g) < 0.4] = 0
    b = _ints(g, (cout,), -4, 4)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt, b, padding=1)
    assert ref.abs().max() < 2048
    xs, wd, bd = dev(_nhwc(x)), dev(wt), dev(b)
    sub, ho, wo = ops.conv3x3(xs, pack_conv3x3(wd), bd, nimg=n, h=h, wd=w, upsample_to=(2 * h, 2 * w), n_store=cout, tile=81,
                              w_subpixel=pack_conv3x3_subpixel(wd))
    nine, _, _ = ops.conv3x3(xs, pack_conv3x3(wd), bd, nimg=n, h=h, wd=w, upsample_to=(2 * h, 2 * w), n_store=cout, tile=81)
    assert (ho, wo) == (2 * h, 2 * w)
    assert torch.equal(_from_rows(nine, n, ho, wo).float().cpu(), ref)
    bad = (_from_rows(sub, n, ho, wo).float().cpu() != ref)
    assert not bad.any(), f"{int(bad.sum())} wrong outputs, first at {bad.nonzero()[0].tolist()}"
    # random operands: the parity kernels are fp16 roundings of fp32 sums -- one more rounding than the 9-tap form
    xr = torch.randn(n, cin, h, w, generator=g).half()
    wr = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).half()
    refr = F.conv2d(F.interpolate(xr.float(), scale_factor=2.0, mode="nearest"), wr.float(), b, padding=1)
    subr, _, _ = ops.conv3x3(dev(_nhwc(xr)), pack_conv3x3(dev(wr)), bd, nimg=n, h=h, wd=w, upsample_to=(2 * h, 2 * w), n_store=cout,
                             tile=81, w_subpixel=pack_conv3x3_subpixel(dev(wr)))
    assert rel_l2(_from_rows(subr, n, ho, wo), refr) < 1.5e-3


@pytest.mark.parametrize("tile", [0, 11, 13, 61, 81, 82])
@pytest.mark.parametrize("frames", [1, 3, 16])
def test_tconv3(ops, frames, tile):
    from mvoc_amd.unet import pack_tconv
    g = torch.Generator().manual_seed(13 + frames)
    nb, c, hw = 2, 64, 12
    x = torch.randn(nb, c, frames, hw, 1, generator=g).half()
    wt = (torch.randn(c, c, 3, 1, 1, generator=g) / 14).half()
    b = torch.randn(c, generator=g).half()
    ref = F.conv3d(x.float(), wt.float(), b.float(), padding=(1, 0, 0)) + x.float()
    rows = x[..., 0].permute(0, 2, 3, 1).reshape(nb * frames * hw, c)
    out = ops.tconv3(dev(rows), pack_tconv(dev(wt)), dev(b), nvid=nb, frames=frames, hw=hw, resid=dev(rows), tile=tile)
    got = out.reshape(nb, frames, hw, c).permute(0, 3, 1, 2)[..., None]
    assert rel_l2(got, ref) < 1.5e-3


# ---- production shapes x production tiles --------------------------------------------------------------
# The tiles the dispatcher selects at the network's real shapes (csrc/gemm.hip: 256-row 8-wave tiles, K-step-32 tiles,
# 160-wide tiles), forced one by one, in every A-gather mode, at production K (cin 640 / 1280+640 / 3x640) with the
# time-embedding row add and the residual.  Operands are small integers: every product and every fp32 partial sum is
# exact and |result| < 2048 is exact in fp16, so the comparison with torch's CPU conv is BIT-EXACT -- any indexing slip in
# a tile (tap order, source switch, swizzle, tail rows) shows as a wrong integer.
PROD_TILES = [0, 11, 12, 13, 61, 62, 64, 81, 82]  # (round 4 pruned 14, 15, 63, 65, 66, 67: no profile row used them)
_prod_cache = {}


def _ints(g, shape, lo=-1, hi=1):
    return torch.randint(lo, hi + 1, shape, generator=g).float()


def _prod_case(name):
    """(inputs, exact reference) per case, computed once and shared by every tile"""
    if name in _prod_cache:
        return _prod_cache[name]
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    if name == "conv640_320":      # up_blocks[3].resnets[0].conv1 shape class at 32x32: M = 16384, K = 5760
        n, c1, c2, cout, h, w, fr = 16, 640, 0, 320, 32, 32, 16
    elif name == "conv1280+640_640":  # decoder skip concat (two sources), M = 4096, K = 17280
        n, c1, c2, cout, h, w, fr = 16, 1280, 640, 640, 16, 16, 16
    else:
        raise KeyError(name)
    x1 = _ints(g, (n, c1, h, w))
    x2 = _ints(g, (n, c2, h, w)) if c2 else None
    wt = _ints(g, (cout, c1 + c2, 3, 3))
    wt[torch.rand(wt.shape, generator=g) < 0.5] = 0  # keep |sum| well inside the fp16-exact range
    b = _ints(g, (cout,), -4, 4)
    temb = _ints(g, (n // fr, cout), -4, 4)
    res = _ints(g, (n, cout, h, w), -4, 4)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = F.conv2d(xin, wt, b, padding=1) + temb.repeat_interleave(fr, 0)[:, :, None, None] + res
    assert ref.abs().max() < 2048
    _prod_cache[name] = (dict(n=n, h=h, w=w, fr=fr, cout=cout, x1=dev(_nhwc(x1)), x2=None if x2 is None else dev(_nhwc(x2)),
                              wt=dev(wt), b=dev(b), temb=dev(temb), res=dev(_nhwc(res))), _nhwc(ref).contiguous())
    return _prod_cache[name]


@pytest.mark.parametrize("tile", PROD_TILES)
@pytest.mark.parametrize("case", ["conv640_320", "conv1280+640_640"])
def test_conv3x3_production_tiles_exact(ops, case, tile):
    from mvoc_amd.unet import pack_conv3x3
    c, ref = _prod_case(case)
    out, _, _ = ops.conv3x3(c["x1"], pack_conv3x3(c["wt"]), c["b"], nimg=c["n"], h=c["h"], wd=c["w"], x2=c["x2"], rowadd=c["temb"],
                            rowadd_div=c["fr"] * c["h"] * c["w"], resid=c["res"], n_store=c["cout"], tile=tile, split_k=1)
    assert torch.equal(out.float().cpu(), ref), f"{case} tile {tile}: {(out.float().cpu() != ref).sum().item()} wrong outputs"


class _chunk_major:
    """``with _chunk_major(ops) as seen:`` -- conv / temporal launches take the chunk-major K order (mvoc_gemm_desc.k_order = 1) where
    ops gates it in; ``seen`` collects the k_order of every GEMM launch so that a test cannot pass on the tap-major path"""

    def __init__(self, ops):
        self.ops = ops

    def __enter__(self):
        o = self.ops
        self.saved = (o._gemm, o.K_ORDER_CHUNK)
        seen = []
        real = o._gemm

        def spy(d, *a, **kw):
            seen.append(int(d.k_order))
            return real(d, *a, **kw)

        o._gemm, o.K_ORDER_CHUNK = spy, True
        return seen

    def __exit__(self, *exc):
        self.ops._gemm, self.ops.K_ORDER_CHUNK = self.saved


@pytest.mark.parametrize("tile", [0, 82])
@pytest.mark.parametrize("case", ["conv640_320", "conv1280+640_640"])
def test_conv3x3_chunk_major_k_order_exact(ops, case, tile):
    """K = (64-channel chunk, tap, 64) with host-repacked weights (include/mvoc_hip.h: k_order): the same products summed in another
    order -- exact on integer operands; one and two sources (the chunk index runs over the concatenated channel axis), row-add,
    residual, image borders inside tiles"""
    from mvoc_amd.unet import pack_conv3x3
    c, ref = _prod_case(case)
    # (tile 0 = the library's own choice: the test shapes are a 16th of the production batch, so the grid-fill rule that gates the
    # chunk-major form is met through the concurrency hint, as eight such launches side by side would)
    with _chunk_major(ops) as seen, ops.gemm_concurrency(8 if tile == 0 else 1):
        out, _, _ = ops.conv3x3(c["x1"], pack_conv3x3(c["wt"]), c["b"], nimg=c["n"], h=c["h"], wd=c["w"], x2=c["x2"], rowadd=c["temb"],
                                rowadd_div=c["fr"] * c["h"] * c["w"], resid=c["res"], n_store=c["cout"], tile=tile)
    assert seen == [1], seen
    assert torch.equal(out.float().cpu(), ref), f"{case} tile {tile}: {(out.float().cpu() != ref).sum().item()} wrong outputs"


@pytest.mark.parametrize("stride,hw,want", [(2, 32, 0), (1, 24, 1)])
def test_conv3x3_chunk_major_gate_and_ragged_rows(ops, stride, hw, want):
    """the chunk-major form exists for the affine gather of the 320-wide tile: a stride-2 conv (Downsample2D) keeps the tap-major
    weights (want 0), a stride-1 conv whose M is not a multiple of the 256-pixel tile takes the form (want 1); both exact"""
    from mvoc_amd.unet import pack_conv3x3
    g = torch.Generator().manual_seed(7 + stride)
    n, cin, cout = 5, 128, 320
    x = _ints(g, (n, cin, hw, hw))
    wt = _ints(g, (cout, cin, 3, 3))
    wt[torch.rand(wt.shape, generator=g) < 0.5] = 0
    b = _ints(g, (cout,), -4, 4)
    ref = F.conv2d(x, wt, b, padding=1, stride=stride)
    assert ref.abs().max() < 2048
    with _chunk_major(ops) as seen:
        out, ho, wo = ops.conv3x3(dev(_nhwc(x)), pack_conv3x3(dev(wt)), dev(b), nimg=n, h=hw, wd=hw, stride=stride, n_store=cout, tile=82)
    assert seen == [want] and (ho, wo) == tuple(ref.shape[2:])
    assert torch.equal(out.float().cpu(), _nhwc(ref))


@pytest.mark.parametrize("tile", [0, 82])
def test_tconv3_chunk_major_k_order_exact(ops, tile):
    from mvoc_amd.unet import pack_tconv
    test_tconv3_production_tiles_exact(ops, 81)  # (fills the case cache)
    c, ref = _prod_cache["tconv640"]
    with _chunk_major(ops) as seen, ops.gemm_concurrency(8 if tile == 0 else 1):
        out = ops.tconv3(c["rows"], pack_tconv(c["wt"]), c["b"], nvid=c["nb"], frames=c["frames"], hw=c["hw"], resid=c["rows"], tile=tile)
    assert seen == [1], seen
    assert torch.equal(out.float().cpu(), ref), f"tile {tile}: {(out.float().cpu() != ref).sum().item()} wrong outputs"


def test_chunk_major_request_outside_the_eight_phase_tiles_is_an_error(ops):
    """k_order = 1 is a form of the 320-wide eight-phase tile: a request it cannot take fails loudly (the library never reads
    chunk-major weights with a tap-major kernel)"""
    from mvoc_amd._ffi import GemmDesc, lib
    import ctypes as C
    x = torch.zeros(512, 64, dtype=torch.float16, device="cuda")
    w = torch.zeros(64, 576, dtype=torch.float16, device="cuda")
    out = torch.empty(512, 64, dtype=torch.float16, device="cuda")
    d = GemmDesc()
    d.a, d.w, d.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.m, d.n, d.k, d.n_store, d.ldo, d.lda, d.c1, d.cin = 512, 64, 576, 64, 64, 64, 64, 64
    d.a_mode, d.nimg, d.hout, d.wout, d.hsrc, d.wsrc, d.stride, d.k_order = 1, 2, 16, 16, 16, 16, 1, 1
    st = torch.cuda.current_stream().cuda_stream
    assert lib.mvoc_gemm_f16(C.byref(d), st) == -2 and b"k_order" in lib.mvoc_last_error()  # m < 1024, n = 64
    x = torch.zeros(2048, 64, dtype=torch.float16, device="cuda")
    w = torch.zeros(320, 576, dtype=torch.float16, device="cuda")
    out = torch.empty(2048, 320, dtype=torch.float16, device="cuda")
    d.a, d.w, d.out, d.m, d.n, d.n_store, d.ldo, d.nimg = x.data_ptr(), w.data_ptr(), out.data_ptr(), 2048, 320, 320, 320, 8
    assert lib.mvoc_gemm_f16(C.byref(d), st) == 0                                           # the form's own shape class
    d.tile = 81
    assert lib.mvoc_gemm_f16(C.byref(d), st) == -2 and b"k_order" in lib.mvoc_last_error()  # a forced 256-wide tile
    d.tile, d.act = 0, 2
    assert lib.mvoc_gemm_f16(C.byref(d), st) == -2                                          # an activation
    torch.cuda.synchronize()


@pytest.mark.parametrize("tile", [81, 82])
@pytest.mark.parametrize("split_k", [2, 4])
def test_g8_split_k_exact(ops, split_k, tile):
    """eight-phase kernel with K slices (fp32 slabs + the reduce pass)"""
    from mvoc_amd.unet import pack_conv3x3
    c, ref = _prod_case("conv1280+640_640")
    out, _, _ = ops.conv3x3(c["x1"], pack_conv3x3(c["wt"]), c["b"], nimg=c["n"], h=c["h"], wd=c["w"], x2=c["x2"], rowadd=c["temb"],
                            rowadd_div=c["fr"] * c["h"] * c["w"], resid=c["res"], n_store=c["cout"], split_k=split_k, tile=tile)
    assert torch.equal(out.float().cpu(), ref)


@pytest.mark.parametrize("split_k", [0, 2, 4, 8])
def test_conv3x3_production_split_k_exact(ops, split_k):
    """the deep-K, few-rows convs of the coarse levels take the split-K path (fp32 slabs + reduce pass)"""
    from mvoc_amd.unet import pack_conv3x3
    c, ref = _prod_case("conv1280+640_640")
    out, _, _ = ops.conv3x3(c["x1"], pack_conv3x3(c["wt"]), c["b"], nimg=c["n"], h=c["h"], wd=c["w"], x2=c["x2"], rowadd=c["temb"],
                            rowadd_div=c["fr"] * c["h"] * c["w"], resid=c["res"], n_store=c["cout"], split_k=split_k)
    assert torch.equal(out.float().cpu(), ref)


@pytest.mark.parametrize("tile", PROD_TILES)
def test_tconv3_production_tiles_exact(ops, tile):
    """(3,1,1) temporal conv at C = 640, 16 frames (frame-boundary zero padding on the first / last frame), + residual"""
    from mvoc_amd.unet import pack_tconv
    key = "tconv640"
    if key not in _prod_cache:
        g = torch.Generator().manual_seed(640)
        nb, c, frames, hw = 2, 640, 16, 256
        x = _ints(g, (nb, c, frames, hw, 1))
        wt = _ints(g, (c, c, 3, 1, 1))
        wt[torch.rand(wt.shape, generator=g) < 0.5] = 0
        b = _ints(g, (c,), -4, 4)
        ref = F.conv3d(x, wt, b, padding=(1, 0, 0)) + x
        assert ref.abs().max() < 2048
        rows = x[..., 0].permute(0, 2, 3, 1).reshape(nb * frames * hw, c)
        _prod_cache[key] = (dict(rows=dev(rows), wt=dev(wt), b=dev(b), nb=nb, frames=frames, hw=hw, c=c),
                            ref[..., 0].permute(0, 2, 3, 1).reshape(nb * frames * hw, c).contiguous())
    c, ref = _prod_cache[key]
    out = ops.tconv3(c["rows"], pack_tconv(c["wt"]), c["b"], nvid=c["nb"], frames=c["frames"], hw=c["hw"], resid=c["rows"], tile=tile,
                     split_k=1)
    assert torch.equal(out.float().cpu(), ref), f"tile {tile}: {(out.float().cpu() != ref).sum().item()} wrong outputs"


@pytest.mark.parametrize("tile", [0, 81, 82])
def test_tconv3_frame_fastest_tile_order_exact(ops, tile):
    """temporal conv whose frames span several 256-pixel tiles (hw = 1024: four patches per frame): the eight-phase kernel walks
    the FRAMES of a patch before the next patch (gemm8.hip: tmap_t) -- same function, exact on integers, and the producer's
    channel sums still land in row order ([m / 256] slabs)"""
    from mvoc_amd.unet import pack_tconv
    g = torch.Generator().manual_seed(1024)
    nb, c, frames, hw = 2, 320, 16, 1024
    x = _ints(g, (nb, c, frames, hw, 1))
    wt = _ints(g, (c, c, 3, 1, 1))
    wt[torch.rand(wt.shape, generator=g) < 0.5] = 0
    b = _ints(g, (c,), -4, 4)
    ref = F.conv3d(x, wt, b, padding=(1, 0, 0)) + x
    assert ref.abs().max() < 2048
    rows = dev(x[..., 0].permute(0, 2, 3, 1).reshape(nb * frames * hw, c))
    ref = ref[..., 0].permute(0, 2, 3, 1).reshape(nb * frames * hw, c).contiguous()
    out = ops.tconv3(rows, pack_tconv(dev(wt)), dev(b), nvid=nb, frames=frames, hw=hw, resid=rows, tile=tile, split_k=1, sums=True)
    assert torch.equal(out.float().cpu(), ref), f"tile {tile}: {(out.float().cpu() != ref).sum().item()} wrong outputs"
    cs = getattr(out, "chan_sums", None)
    assert cs is not None
    o = ref.reshape(-1, 256, c)
    assert torch.allclose(cs[..., 0].cpu(), o.sum(1), rtol=1e-5, atol=1e-3) and torch.allclose(cs[..., 1].cpu(), (o * o).sum(1), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("tile", PROD_TILES)
@pytest.mark.parametrize("m,n,k", [(65536, 320, 320), (16384, 960, 320), (20480, 1280, 1280)])
def test_linear_production_tiles_exact(ops, m, n, k, tile):
    """the L0 / L2 projections at their real row counts (all auto-dispatch thresholds crossed), bias + residual"""
    key = ("lin", m, n, k)
    if key not in _prod_cache:
        g = torch.Generator().manual_seed(m + n + k)
        x, w = _ints(g, (m, k)), _ints(g, (n, k))
        w[torch.rand(w.shape, generator=g) < 0.5] = 0
        b, r = _ints(g, (n,), -4, 4), _ints(g, (m, n), -4, 4)
        _prod_cache[key] = (dev(x), dev(w), dev(b), dev(r), x @ w.t() + b + r)
    x, w, b, r, ref = _prod_cache[key]
    out = ops.linear(x, w, b, resid=r, tile=tile)
    assert torch.equal(out.float().cpu(), ref), f"tile {tile}"


@pytest.mark.parametrize("tile", [81, 82])
@pytest.mark.parametrize("m,n,ns,k", [(1000, 352, 328, 640), (2309, 672, 672, 576), (4099, 1312, 1288, 512)])
def test_g8_readback_ragged_rows_and_column_views(ops, m, n, ns, k, tile):
    """the eight-phase tiles' branch-free epilogue readback (range-checked buffer loads / stores): rows that end inside a block,
    a last column tile that is mostly outside n, fewer columns stored than computed (n_store < n), and `out` / `resid` that are
    column slices of wider tensors (ldo, ldr > n_store) -- nothing may be written outside the slice; exact on integer operands"""
    g = torch.Generator().manual_seed(m + n + k + tile)
    x, w = _ints(g, (m, k)), _ints(g, (n, k))
    w[torch.rand(w.shape, generator=g) < 0.5] = 0
    b = _ints(g, (n,), -4, 4)
    rwide = _ints(g, (m, ns + 24), -4, 4)
    ref = (x @ w.t() + b)[:, :ns] + rwide[:, 8:8 + ns]
    owide = torch.full((m + 3, ns + 40), 7.0).half().cuda()      # sentinel everywhere, 3 extra rows below
    rdev = dev(rwide)
    ops.linear(dev(x), dev(w), dev(b), resid=rdev[:, 8:8 + ns], out=owide[:m, 16:16 + ns], n_store=ns, tile=tile, split_k=1)
    torch.cuda.synchronize()
    got = owide.float().cpu()
    assert torch.equal(got[:m, 16:16 + ns], ref), f"tile {tile}: {(got[:m, 16:16 + ns] != ref).sum().item()} wrong outputs"
    untouched = got.clone()
    untouched[:m, 16:16 + ns] = 7.0
    assert bool((untouched == 7.0).all()), "the epilogue wrote outside the output slice"


@pytest.mark.parametrize("tile", [0, 11, 61, 81])
@pytest.mark.parametrize("m,c,inner", [(16384, 320, 1280), (4096, 1280, 5120)])
def test_geglu_layernorm_fold_production(ops, m, c, inner, tile):
    """GEGLU feed-forward entry with the LayerNorm folded in, at C = 320 (L0) and C = 1280 (L2) and production rows, on the
    GEGLU-capable tiles (even TN); tolerance as test_linear_layernorm_fold"""
    from mvoc_amd.unet import Linear, pack_geglu
    key = ("geglu", m, c, inner)
    if key not in _prod_cache:
        g = torch.Generator().manual_seed(m + c)
        x = (torch.randn(m, c, generator=g) * 1.3 + 0.4).half()
        w = (torch.randn(2 * inner, c, generator=g) / math.sqrt(c)).half()
        b = torch.randn(2 * inner, generator=g).half()
        gm, bt = (1 + 0.3 * torch.randn(c, generator=g)).half(), (0.3 * torch.randn(c, generator=g)).half()
        y = F.layer_norm(x.float(), (c,), gm.float(), bt.float(), 1e-5).half().float() @ w.float().t() + b.float()
        hh, gg = y.half().float().chunk(2, dim=-1)
        wp, bp = pack_geglu(dev(w), dev(b))
        lin = Linear(wp, bp).fold_layernorm(dev(gm), dev(bt))
        _prod_cache[key] = (dev(x), lin, hh * F.gelu(gg))
    x, lin, ref = _prod_cache[key]
    stats = ops.row_stats(x, lin.ln[2])
    out = ops.linear(x, lin.w_ln, None, n_store=lin.n, ln=lin.ln + (stats,), act=ops.ACT_GEGLU, tile=tile)
    assert rel_l2(out, ref) < 2e-3
    assert (out.float().cpu() - ref).abs().max() < 3e-2 * ref.abs().max()


# ---- attention ----------------------------------------------------------------------------------------
@pytest.fixture(params=[0, 1], ids=["flash_kernel", "flash3_kernel"])
def fmode(request, ops):
    """every flash-attention test runs through BOTH kernels behind mvoc_flash_attn_f16 (attention.hip: the phase kernel and the
    software-pipelined LDS-DMA kernel; by default the key count picks one, and most test shapes are below the switch point).
    head_dim 96 / causal calls ignore the switch."""
    ops.flash_pipelined(request.param)
    yield request.param
    ops.flash_pipelined(-1)


@pytest.mark.parametrize("tq,tk,heads,kv_bdiv", [(256, 256, 2, 1), (100, 100, 1, 1), (200, 145, 2, 3), (64, 64, 5, 1), (130, 77, 1, 1)])
def test_flash_attn(ops, fmode, tq, tk, heads, kv_bdiv):
    g = torch.Generator().manual_seed(tq + tk)
    nb = 6
    c = heads * 64
    qkv = torch.randn(nb * tq, 3 * c, generator=g).half()  # q read through a strided view like the fused qkv GEMM output
    k = torch.randn((nb // kv_bdiv) * tk, c, generator=g).half()
    v = torch.randn((nb // kv_bdiv) * tk, c, generator=g).half()
    dq = dev(qkv)
    out = ops.flash_attn(dq[:, c:2 * c], dev(k), dev(v), nbatch=nb, heads=heads, tq=tq, tk=tk, kv_bdiv=kv_bdiv)
    q4 = qkv[:, c:2 * c].float().reshape(nb, tq, heads, 64).transpose(1, 2)
    k4 = k.float().reshape(nb // kv_bdiv, tk, heads, 64).transpose(1, 2).repeat_interleave(kv_bdiv, 0)
    v4 = v.float().reshape(nb // kv_bdiv, tk, heads, 64).transpose(1, 2).repeat_interleave(kv_bdiv, 0)
    ref = F.scaled_dot_product_attention(q4, k4, v4).transpose(1, 2).reshape(nb * tq, c)
    assert rel_l2(out, ref) < 2e-3
    assert (out.float().cpu() - ref).abs().max() < 1e-2


@pytest.mark.parametrize("tq,tk,heads,kv_bdiv,nb", [(1024, 1024, 5, 1, 3), (300, 512, 2, 2, 4), (256, 256, 1, 1, 2), (4096, 4096, 1, 1, 1),
                                                     (920, 960, 2, 1, 2), (2100, 192, 1, 1, 1)])
def test_flash_attn_eight_wave(ops, fmode, tq, tk, heads, kv_bdiv, nb):
    """many KV tiles (3 / 4 / 8 / 15 / 16 / 64), query counts that leave waves of the last block without rows, shared K/V across
    batch entries, strided q/k/v views of a fused QKV buffer (written for the eight-wave laboratory kernel of round 3,
    tools/lab/flash2_kernel.hip.inc; kept as coverage of the production kernel at the network's real token counts)"""
    g = torch.Generator().manual_seed(tq + tk + heads)
    c = heads * 64
    qkv = torch.randn(nb * tq, 3 * c, generator=g).half()
    kv = torch.randn((nb // kv_bdiv) * tk, 2 * c, generator=g).half()
    dq, dkv = dev(qkv), dev(kv)
    out = ops.flash_attn(dq[:, c:2 * c], dkv[:, :c], dkv[:, c:], nbatch=nb, heads=heads, tq=tq, tk=tk, kv_bdiv=kv_bdiv)
    q4 = qkv[:, c:2 * c].float().reshape(nb, tq, heads, 64).transpose(1, 2)
    k4 = kv[:, :c].float().reshape(nb // kv_bdiv, tk, heads, 64).transpose(1, 2).repeat_interleave(kv_bdiv, 0)
    v4 = kv[:, c:].float().reshape(nb // kv_bdiv, tk, heads, 64).transpose(1, 2).repeat_interleave(kv_bdiv, 0)
    ref = F.scaled_dot_product_attention(q4, k4, v4).transpose(1, 2).reshape(nb * tq, c)
    assert rel_l2(out, ref) < 2e-3
    assert (out.float().cpu() - ref).abs().max() < 1e-2


def test_flash_attn_eight_wave_large_scores(ops, fmode):
    """spiked keys in different KV tiles force running-max jumps (the online-softmax rescale branch); an exact-integer value
    matrix makes a wrong V fragment map (transposed LDS read) show as a wrong integer"""
    g = torch.Generator().manual_seed(5)
    t = 512
    q = torch.randn(t, 64, generator=g).half()
    k = torch.randn(t, 64, generator=g).half()
    v = torch.randn(t, 64, generator=g).half()
    k[70] = q[5] * 4
    k[200] = q[5] * 8
    k[300] = q[9] * 10
    k[450] = q[300] * 9
    out = ops.flash_attn(dev(q), dev(k), dev(v), nbatch=1, heads=1, tq=t, tk=t)
    ref = F.scaled_dot_product_attention(q.float()[None, None], k.float()[None, None], v.float()[None, None])[0, 0]
    assert (out.float().cpu() - ref).abs().max() < 1e-2
    # one-hot attention: key j matches query j overwhelmingly -> out[j] == v[j] exactly (integers), for every (key, d) position
    eye = torch.zeros(t, 64)
    code = torch.arange(t)
    for bit in range(9):
        eye[:, bit] = ((code >> bit) & 1).float() * 2 - 1
    qe = (eye * 24).half()
    vi = torch.randint(-64, 65, (t, 64), generator=g).half()
    out = ops.flash_attn(dev(qe), dev(qe), dev(vi), nbatch=1, heads=1, tq=t, tk=t)
    assert torch.equal(out.float().cpu(), vi.float())


def test_flash_attn_large_scores(ops, fmode):
    """spiked keys force big running-max jumps between KV tiles (online-softmax rescale path)"""
    g = torch.Generator().manual_seed(3)
    nb, t, heads = 1, 320, 1
    q = torch.randn(nb * t, 64, generator=g).half()
    k = torch.randn(nb * t, 64, generator=g).half()
    v = torch.randn(nb * t, 64, generator=g).half()
    k[70] = q[5] * 4
    k[200] = q[5] * 8
    k[300] = q[9] * 10
    out = ops.flash_attn(dev(q), dev(k), dev(v), nbatch=nb, heads=heads, tq=t, tk=t)
    ref = F.scaled_dot_product_attention(q.float()[None, None], k.float()[None, None], v.float()[None, None])[0, 0]
    assert (out.float().cpu() - ref).abs().max() < 1e-2


def test_flash_attn_deferred_max(ops, fmode):
    """the running maximum is rescaled only when it grew by more than 2^8 (attention.hip: FLASH_THR): keys that lift a row's
    maximum by LESS than the threshold in late tiles leave P > 1 against the stale maximum -- checked against an fp64 softmax
    for growth just below, at and above the threshold, in the first and in later tiles"""
    g = torch.Generator().manual_seed(11)
    t = 640
    q = torch.randn(t, 64, generator=g).half()
    k = (0.25 * torch.randn(t, 64, generator=g)).half()
    v = torch.randn(t, 64, generator=g).half()
    n2 = (q[7].float() ** 2).sum()
    for key, lift in ((3, 2.0), (130, 5.0), (200, 7.5), (330, 8.0), (460, 8.5), (590, 15.0)):  # score = lift / log2(e) in natural units
        k[key] = (q[7].float() * (lift / 1.4426950408889634 * 8.0 / n2)).half()
    k[100] = (q[40].float() * (6.0 / 1.4426950408889634 * 8.0 / (q[40].float() ** 2).sum())).half()
    out = ops.flash_attn(dev(q), dev(k), dev(v), nbatch=1, heads=1, tq=t, tk=t)
    p64 = torch.softmax(q.double() @ k.double().t() / 8.0, dim=-1)
    ref = (p64 @ v.double()).float()
    assert rel_l2(out, ref) < 2e-3
    assert (out.float().cpu() - ref).abs().max() < 1e-2


@pytest.mark.parametrize("t,heads,nb", [(256, 2, 3), (1000, 5, 2), (4096, 1, 1)])
def test_flash_attn_pair_equals_two_calls(ops, fmode, t, heads, nb):
    """PnP destination pair (pnp_utils.py:664-668: one blended q / k for the unconditional and the conditional chunk): the paired
    launch (v2 / out2) returns, bit for bit, what two plain launches with the same q / k return"""
    g = torch.Generator().manual_seed(t + heads)
    c = heads * 64
    qkv = dev(torch.randn(2 * nb * t, 3 * c, generator=g).half())  # [pair member, batch, token] rows of a fused QKV buffer
    q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
    half = nb * t
    a = ops.flash_attn(q[:half], k[:half], v[:half], nbatch=nb, heads=heads, tq=t, tk=t)
    b = ops.flash_attn(q[:half], k[:half], v[half:], nbatch=nb, heads=heads, tq=t, tk=t)
    out = torch.zeros(2 * half, c, dtype=torch.float16, device=q.device)
    ops.flash_attn(q[:half], k[:half], v[:half], nbatch=nb, heads=heads, tq=t, tk=t, out=out[:half], v2=v[half:], out2=out[half:])
    assert torch.equal(out[:half], a) and torch.equal(out[half:], b)
    q4, k4 = (x[:half].float().cpu().reshape(nb, t, heads, 64).transpose(1, 2) for x in (q, k))
    v4 = v[half:].float().cpu().reshape(nb, t, heads, 64).transpose(1, 2)
    assert rel_l2(out[half:], F.scaled_dot_product_attention(q4, k4, v4).transpose(1, 2).reshape(half, c)) < 2e-3


@pytest.mark.parametrize("nb,heads,tq,tk,kv_bdiv,pair", [(2, 5, 4096, 4096, 1, 0), (2, 2, 920, 920, 1, 0), (2, 3, 1024, 1024, 1, 1), (4, 2, 300, 145, 2, 0),
                                                         (1, 1, 130, 64, 1, 0), (1, 2, 2100, 2240, 1, 1), (2, 1, 129, 3600, 1, 0), (2, 2, 50, 129, 1, 1),
                                                         (1, 5, 14399, 14399, 1, 0), (2, 2, 2049, 2049, 1, 1)])
def test_flash_attn_kernels_agree_bitwise(ops, nb, heads, tq, tk, kv_bdiv, pair):
    """flash3_kernel (software-pipelined, LDS-DMA ring behind hand-counted waits) against flash_kernel on the same inputs: same
    fragment maps, rounding points and summation order, so every output bit must agree -- 1 to 64 key tiles, ragged last tiles
    (keys past tk are zero-filled by the buffer range check and masked), query blocks with idle waves, shared K/V, the pair form,
    strided views of a fused QKV buffer.  A miscounted wait or a wrong swizzle shows here as a differing bit."""
    g = torch.Generator().manual_seed(nb * 1000 + tq + tk)
    c = heads * 64
    q = dev((1.5 * torch.randn(nb * tq, 3 * c, generator=g)).half())[:, c:2 * c]
    kv = dev((1.2 * torch.randn((nb // kv_bdiv) * tk, 3 * c, generator=g)).half())
    k, v, v2 = kv[:, :c], kv[:, c:2 * c], kv[:, 2 * c:]
    outs = []
    for mode in (0, 1):
        ops.flash_pipelined(mode)
        try:
            o = torch.full((nb * tq, c), float("nan"), dtype=torch.float16, device=q.device)
            o2 = torch.full((nb * tq, c), float("nan"), dtype=torch.float16, device=q.device)
            if pair:
                ops.flash_attn(q, k, v, nbatch=nb, heads=heads, tq=tq, tk=tk, kv_bdiv=kv_bdiv, out=o, v2=v2, out2=o2)
            else:
                ops.flash_attn(q, k, v, nbatch=nb, heads=heads, tq=tq, tk=tk, kv_bdiv=kv_bdiv, out=o)
                o2.zero_()
            outs.append((o.cpu(), o2.cpu()))
        finally:
            ops.flash_pipelined(-1)
    assert not torch.isnan(outs[0][0]).any() and not torch.isnan(outs[0][1]).any()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("frames,heads", [(16, 2), (3, 1), (32, 1), (8, 5)])
def test_temporal_attn(ops, frames, heads):
    g = torch.Generator().manual_seed(frames)
    ns, hw = 3, 21
    c = heads * 64
    qkv = torch.randn(ns * frames * hw, 3 * c, generator=g).half()
    d = dev(qkv)
    out = ops.temporal_attn(d[:, :c], d[:, c:2 * c], d[:, 2 * c:], nsample=ns, frames=frames, hw=hw, heads=heads)

    def seq(t):  # [ns, F, hw, heads, 64] -> [ns*hw, heads, F, 64]
        return t.float().reshape(ns, frames, hw, heads, 64).permute(0, 2, 3, 1, 4).reshape(ns * hw, heads, frames, 64)

    ref = F.scaled_dot_product_attention(seq(qkv[:, :c]), seq(qkv[:, c:2 * c]), seq(qkv[:, 2 * c:]))
    ref = ref.reshape(ns, hw, heads, frames, 64).permute(0, 3, 1, 2, 4).reshape(ns * frames * hw, c)
    assert rel_l2(out, ref) < 2e-3
    assert (out.float().cpu() - ref).abs().max() < 1e-2


@pytest.mark.parametrize("c,frames,hw,ns", [(64, 16, 21, 2), (128, 8, 40, 3), (320, 16, 70, 2), (320, 32, 9, 1), (320, 8, 33, 2),
                                            (320, 16, 1024, 1)])
def test_temporal_qkv_attn_fused(ops, c, frames, hw, ns):
    """LayerNorm -> QKV -> attention over frames in one kernel (Q/K/V never in HBM) against torch (F.layer_norm, F.linear, SDPA
    per pixel over the frame axis) and against the unfused kernel chain; ragged pixel counts exercise the dead-pixel lanes"""
    from mvoc_amd.unet import Linear, pack_tfused_weights
    heads = c // 64
    g = torch.Generator().manual_seed(c + frames + hw)
    rows = ns * frames * hw
    x = (torch.randn(rows, c, generator=g) * 1.4 + 0.3).half()
    w = (torch.randn(3 * c, c, generator=g) / math.sqrt(c)).half()
    gm, bt = (1 + 0.3 * torch.randn(c, generator=g)).half(), (0.3 * torch.randn(c, generator=g)).half()
    lin = Linear(dev(w)).fold_layernorm(dev(gm), dev(bt))
    wp = pack_tfused_weights(lin.w_ln, heads)
    out = ops.temporal_qkv_attn(dev(x), wp, lin.ln, nsample=ns, frames=frames, hw=hw, heads=heads)
    # torch reference with the reference's fp16 rounding of q/k/v
    qkv = (F.layer_norm(x.float(), (c,), gm.float(), bt.float(), 1e-5).half().float() @ w.float().t()).half().float()

    def seq(t):  # [ns, F, hw, heads, 64] -> [ns*hw, heads, F, 64]
        return t.reshape(ns, frames, hw, heads, 64).permute(0, 2, 3, 1, 4).reshape(ns * hw, heads, frames, 64)

    ref = F.scaled_dot_product_attention(seq(qkv[:, :c]), seq(qkv[:, c:2 * c]), seq(qkv[:, 2 * c:]))
    ref = ref.reshape(ns, hw, heads, frames, 64).permute(0, 3, 1, 2, 4).reshape(rows, c)
    assert rel_l2(out, ref) < 3e-3, rel_l2(out, ref)
    assert (out.float().cpu() - ref).abs().max() < 2e-2
    # the unfused chain of this library (row statistics + folded GEMM + temporal attention kernel)
    q3 = lin.call_ln(dev(x), (dev(gm), dev(bt)))
    un = ops.temporal_attn(q3[:, :c], q3[:, c:2 * c], q3[:, 2 * c:], nsample=ns, frames=frames, hw=hw, heads=heads)
    assert rel_l2(out, un) < 2e-3


# ---- activation-stationary linear (csrc/xslin.hip) ------------------------------------------------------------------------
@pytest.mark.parametrize("k,n,m", [(320, 320, 4096), (320, 960, 1000), (64, 64, 129), (128, 384, 4097), (320, 2560, 300)])
def test_xs_linear_exact(ops, k, n, m):
    """small-integer operands: every product and partial sum is exact in fp32 and the result exact in fp16 -> bit-equal to the
    integer answer whatever the accumulation order (bias and residual included; ragged row counts)"""
    from mvoc_amd.unet import pack_xs_weights
    g = torch.Generator().manual_seed(k + n + m)
    x = torch.randint(-3, 4, (m, k), generator=g).half()
    w = torch.randint(-2, 3, (n, k), generator=g).half()
    b = torch.randint(-8, 9, (n,), generator=g).half()
    r = torch.randint(-16, 17, (m, n), generator=g).half()
    ref = x.double() @ w.double().t() + b.double()
    assert ref.abs().max() < 2048
    wp = pack_xs_weights(dev(w), dev(b))
    out = ops.xs_linear(dev(x), wp, n)
    assert torch.equal(out.cpu().double(), ref)
    out = ops.xs_linear(dev(x), wp, n, resid=dev(r))
    assert torch.equal(out.cpu().double(), ref + r.double())
    # a narrower store (padded weight rows) into a wider row pitch
    wide = torch.full((m, n + 64), 7.0, dtype=torch.float16, device="cuda")
    ops.xs_linear(dev(x), wp, n, n_store=n - 24, out=wide[:, 8:8 + n - 24])
    rs = torch.zeros((m, n + 40), dtype=torch.float16, device="cuda")  # residual rows of another pitch, trimmed last tile
    rs[:, 16:16 + n] = dev(r)
    out = ops.xs_linear(dev(x), wp, n, n_store=n - 24, resid=rs[:, 16:16 + n - 24])
    assert torch.equal(out.cpu().double(), (ref + r.double())[:, :n - 24])
    assert torch.equal(wide[:, 8:8 + n - 24].cpu().double(), ref[:, :n - 24])
    assert (wide[:, :8] == 7).all() and (wide[:, 8 + n - 24:] == 7).all()


@pytest.mark.parametrize("k,n,m,act", [(320, 320, 5000, "none"), (320, 960, 4096, "silu"), (128, 256, 4500, "gelu"), (320, 2560, 4100, "geglu"),
                                       (64, 128, 4096, "geglu")])
def test_xs_linear_matches_tiled_gemm(ops, k, n, m, act):
    """the activation-stationary kernel against torch and against the tiled GEMM of this library on the same operands: same
    fp16 rounding points (bias add, activation, residual), so the two differ by accumulation order only"""
    from mvoc_amd._ffi import ACT_GEGLU, ACT_GELU, ACT_NONE, ACT_SILU
    from mvoc_amd.unet import Linear, pack_geglu
    a = {"none": ACT_NONE, "silu": ACT_SILU, "gelu": ACT_GELU, "geglu": ACT_GEGLU}[act]
    g = torch.Generator().manual_seed(k + n + m)
    x = torch.randn(m, k, generator=g).half()
    w = (torch.randn(n, k, generator=g) / math.sqrt(k)).half()
    b = (0.2 * torch.randn(n, generator=g)).half()
    y = x.float() @ w.float().t() + b.float()
    if act == "geglu":
        ref = y[:, :n // 2].half().float() * F.gelu(y[:, n // 2:].half().float()).half().float()
        wd, bd = pack_geglu(dev(w), dev(b))
    else:
        ref = {"none": lambda t: t, "silu": F.silu, "gelu": F.gelu}[act](y.half().float())
        wd, bd = dev(w), dev(b)
    lin = Linear(wd, bd)
    kw = {} if act == "geglu" else {"resid": dev(x[:, :1].expand(m, n).contiguous())}
    if kw:
        ref = ref.half().float() + x[:, :1].float()
    assert lin._xs_ok(dev(x), dict(act=a, **kw))
    out = lin(dev(x), act=a, **kw)
    assert lin.wp is not None
    assert rel_l2(out, ref) < 1.5e-3, rel_l2(out, ref)
    Linear.use_xs = False
    try:
        old = lin(dev(x), act=a, **kw)
    finally:
        Linear.use_xs = True
    assert rel_l2(out, old) < 6e-4, rel_l2(out, old)
    assert (out.float() - old.float()).abs().max() < 2e-2


@pytest.mark.parametrize("k,n,m", [(320, 960, 4096), (320, 2560, 4200), (128, 384, 5000), (64, 192, 4096)])
def test_xs_linear_layernorm_fold(ops, k, n, m):
    """LayerNorm folded by normalising the register-resident rows: against F.layer_norm -> F.linear with the reference's fp16
    LayerNorm output, and against this library's row-statistics + folded-GEMM path"""
    from mvoc_amd._ffi import ACT_GEGLU, ACT_NONE
    from mvoc_amd.unet import Linear, pack_geglu
    geglu = n == 2560
    g = torch.Generator().manual_seed(k + n)
    x = (torch.randn(m, k, generator=g) * 1.7 + 0.4).half()
    w = (torch.randn(n, k, generator=g) / math.sqrt(k)).half()
    b = (0.2 * torch.randn(n, generator=g)).half()
    gm, bt = (1 + 0.3 * torch.randn(k, generator=g)).half(), (0.3 * torch.randn(k, generator=g)).half()
    y = F.layer_norm(x.float(), (k,), gm.float(), bt.float(), 1e-5).half().float() @ w.float().t() + b.float()
    if geglu:
        ref = y[:, :n // 2].half().float() * F.gelu(y[:, n // 2:].half().float()).half().float()
        wd, bd = pack_geglu(dev(w), dev(b))
    else:
        ref, wd, bd = y, dev(w), dev(b)
    lin = Linear(wd, bd).fold_layernorm(dev(gm), dev(bt))
    kw = {"act": ACT_GEGLU if geglu else ACT_NONE}
    out = lin.call_ln(dev(x), (dev(gm), dev(bt)), **kw)
    assert lin.wp_ln is not None
    assert rel_l2(out, ref) < 2.5e-3, rel_l2(out, ref)
    Linear.use_xs = False
    try:
        old = lin.call_ln(dev(x), (dev(gm), dev(bt)), **kw)
    finally:
        Linear.use_xs = True
    assert rel_l2(out, old) < 2.5e-3, rel_l2(out, old)


@pytest.mark.parametrize("n,mode", [(320, "resid"), (960, "ln"), (2560, "geglu"), (320, "silu")])
def test_xs_linear_two_row_groups(ops, n, mode):
    """m >= 131072 rows at K = 320 selects the 64-rows-per-wave form (256-row blocks, one residual buffer per wave refilled
    after the epilogue): exact-integer answers for the plain / residual forms, a torch fp32 reference computed on the host
    for the others (LayerNorm fold / GEGLU / SiLU + residual); the row count is ragged so the last block carries clamped rows"""
    from mvoc_amd._ffi import ACT_GEGLU, ACT_NONE, ACT_SILU
    from mvoc_amd.unet import Linear, pack_geglu, pack_xs_weights
    k, m = 320, 131072 + 77
    g = torch.Generator(device="cuda").manual_seed(n)
    if mode == "resid":
        x = torch.randint(-3, 4, (m, k), generator=g, device="cuda").half()
        w = torch.randint(-2, 3, (n, k), generator=g, device="cuda").half()
        b = torch.randint(-8, 9, (n,), generator=g, device="cuda").half()
        r = torch.randint(-16, 17, (m, n), generator=g, device="cuda").half()
        ref = x.float() @ w.float().t() + b.float()  # exact in fp32: |sum| < 2^11
        wp = pack_xs_weights(w, b)
        assert torch.equal(ops.xs_linear(x, wp, n).float(), ref)
        assert torch.equal(ops.xs_linear(x, wp, n, resid=r).float(), ref + r.float())
        return
    x = (torch.randn(m, k, generator=g, device="cuda") * 1.5 + 0.2).half()
    w = (torch.randn(n, k, generator=g, device="cuda") / math.sqrt(k)).half()
    b = (0.2 * torch.randn(n, generator=g, device="cuda")).half()
    gm, bt = (1 + 0.3 * torch.randn(k, generator=g, device="cuda")).half(), (0.3 * torch.randn(k, generator=g, device="cuda")).half()
    if mode == "geglu":
        w, b = pack_geglu(w, b)
    lin = Linear(w, b)
    if mode != "silu":
        lin.fold_layernorm(gm, bt)
    kw = {"act": {"ln": ACT_NONE, "geglu": ACT_GEGLU, "silu": ACT_SILU}[mode]}
    out = lin(x, resid=x, **kw) if mode == "silu" else lin.call_ln(x, (gm, bt), **kw)
    # independent reference: torch fp32 on the host for a row subset that covers every wave position of a block, both row
    # groups of a wave, block seams and the ragged last block
    rows = torch.cat([torch.arange(0, 300), torch.arange(65500, 65800), torch.arange(m - 400, m)])
    xs = x[rows].float().cpu()
    wt, bs = (w.float().cpu(), b.float().cpu())
    if mode == "silu":
        ref = F.silu((xs @ wt.t() + bs).half().float()).half().float() + xs
    else:
        y = (F.layer_norm(xs, (k,), gm.float().cpu(), bt.float().cpu(), 1e-5).half().float() @ wt.t() + bs).half().float()
        if mode == "geglu":  # packed rows: blocks of 64 = 32 value rows then their 32 gate rows
            y = y.reshape(len(rows), n // 64, 2, 32)
            ref = (y[:, :, 0] * F.gelu(y[:, :, 1])).reshape(len(rows), n // 2)
        else:
            ref = y
    got = out[rows.cuda()].float().cpu()
    assert rel_l2(got, ref) < 2.5e-3, rel_l2(got, ref)
    assert (got - ref).abs().max() < 3e-2 * max(1.0, float(ref.abs().max()))


def test_layernorm_fold_rows_with_large_offset(ops):
    """rows whose mean is large against their spread (outlier channels of a real checkpoint): the folded LayerNorm of the
    x-stationary linear, of the fused temporal kernel and of mvoc_row_stats_f16 takes its variance in two passes like
    F.layer_norm -- the one-pass E[x^2] - mu^2 form lost it to cancellation (round-2 advisor finding)"""
    from mvoc_amd.unet import Linear, pack_tfused_weights
    g = torch.Generator().manual_seed(77)
    m, k, n = 4096, 320, 960
    x = (40.0 + 0.5 * torch.randn(m, k, generator=g)).half()
    x[::7] = (-25.0 + 0.25 * torch.randn(x[::7].shape, generator=g)).half()
    w = (torch.randn(n, k, generator=g) / math.sqrt(k)).half()
    gm, bt = (1 + 0.3 * torch.randn(k, generator=g)).half(), (0.3 * torch.randn(k, generator=g)).half()
    ln = F.layer_norm(x.float(), (k,), gm.float(), bt.float(), 1e-5)
    ref = ln.half().float() @ w.float().t()
    lin = Linear(dev(w)).fold_layernorm(dev(gm), dev(bt))
    out = lin.call_ln(dev(x), (dev(gm), dev(bt)))          # K = 320: mvoc_xs_linear_f16 with normalize
    assert rel_l2(out, ref) < 3e-3, rel_l2(out, ref)
    st = ops.row_stats(dev(x), 1e-5).float().cpu()          # {mean, rstd} per row
    mu, var = x.float().mean(1), x.float().var(1, unbiased=False)
    assert (st[:, 0] - mu).abs().max() < 1e-3
    assert ((st[:, 1] - torch.rsqrt(var + 1e-5)) / torch.rsqrt(var + 1e-5)).abs().max() < 1e-4
    # fused temporal kernel on the same rows (16 frames x 256 pixels)
    wp = pack_tfused_weights(lin.w_ln, 5)
    o2 = ops.temporal_qkv_attn(dev(x), wp, lin.ln, nsample=1, frames=16, hw=256, heads=5)
    qkv = ref.half().float()

    def seq(t):
        return t.reshape(1, 16, 256, 5, 64).permute(0, 2, 3, 1, 4).reshape(256, 5, 16, 64)

    r2 = F.scaled_dot_product_attention(seq(qkv[:, :320]), seq(qkv[:, 320:640]), seq(qkv[:, 640:]))
    r2 = r2.reshape(1, 256, 5, 16, 64).permute(0, 3, 1, 2, 4).reshape(m, 320)
    assert rel_l2(o2, r2) < 4e-3, rel_l2(o2, r2)


def test_xs_linear_refuses(ops):
    from mvoc_amd.unet import pack_xs_weights
    x = torch.zeros(64, 96, dtype=torch.float16, device="cuda")
    w = torch.zeros(32, 96, dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError, match="xs_linear"):
        ops.xs_linear(x, pack_xs_weights(w), 32)


# ---- norms ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c,groups,rows,nsample,silu", [(320, 32, 64, 5, True), (64, 8, 37, 3, False), (2560, 32, 16, 2, True),
                                                        (960, 32, 100, 4, True), (128, 8, 4096, 2, False)])
def test_groupnorm(ops, c, groups, rows, nsample, silu):
    g = torch.Generator().manual_seed(c + rows)
    x = (torch.randn(nsample, rows, c, generator=g) * 2 + 0.5).half()
    gm, bt = (1 + 0.2 * torch.randn(c, generator=g)).half(), (0.2 * torch.randn(c, generator=g)).half()
    ref = F.group_norm(x.float().permute(0, 2, 1), groups, gm.float(), bt.float(), eps=1e-5)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 1)
    out = ops.groupnorm(dev(x.reshape(-1, c)), dev(gm), dev(bt), nsample=nsample, rows_per_sample=rows, groups=groups, eps=1e-5,
                        silu=silu)
    assert (out.float().cpu().reshape(nsample, rows, c) - ref).abs().max() < 1.5e-2
    assert rel_l2(out.reshape(nsample, rows, c), ref) < 2e-3


def test_groupnorm_concat(ops):
    g = torch.Generator().manual_seed(77)
    c1, c2, groups, rows, ns = 1280, 640, 32, 24, 3  # 60 channels per group: group 21 straddles the two sources
    x1, x2 = torch.randn(ns, rows, c1, generator=g).half(), (torch.randn(ns, rows, c2, generator=g) * 3).half()
    gm, bt = (1 + 0.2 * torch.randn(c1 + c2, generator=g)).half(), (0.2 * torch.randn(c1 + c2, generator=g)).half()
    ref = F.silu(F.group_norm(torch.cat([x1, x2], 2).float().permute(0, 2, 1), groups, gm.float(), bt.float(), eps=1e-5)).permute(0, 2, 1)
    out = ops.groupnorm(dev(x1.reshape(-1, c1)), dev(gm), dev(bt), x2=dev(x2.reshape(-1, c2)), nsample=ns, rows_per_sample=rows,
                        groups=groups, eps=1e-5, silu=True)
    assert rel_l2(out.reshape(ns, rows, c1 + c2), ref) < 2e-3


@pytest.mark.parametrize("m,n,k,resid,tile", [(2048, 320, 640, True, 82), (2048, 640, 320 * 2, True, 81), (1024, 1280, 1280, False, 81),
                                              (4096, 320, 960, True, 81), (2560, 640, 640, False, 82)])
def test_gemm_chan_sums_and_groupnorm_from_them(ops, m, n, k, resid, tile):
    """GroupNorm statistics from the producer's epilogue (gemm8.hip: stats_pass): the per-slab channel sums equal the sums of the
    STORED fp16 values (fp32 accumulation: 1e-5), and a GroupNorm fed with them returns what the three-pass GroupNorm returns
    on the same tensor, to the accumulation-order noise of its statistics"""
    g = torch.Generator().manual_seed(m + n + k)
    x = dev((torch.randn(m, k, generator=g) * 0.7).half())
    w = dev((torch.randn(n, k, generator=g) / k ** 0.5).half())
    b = dev(torch.randn(n, generator=g).half())
    r = dev((torch.randn(m, n, generator=g) * 2 + 1).half()) if resid else None
    out = ops.linear(x, w, b, resid=r, tile=tile, split_k=1, sums=True)
    cs = getattr(out, "chan_sums", None)
    assert cs is not None and tuple(cs.shape) == (m // 256, n, 2)
    o = out.float().reshape(m // 256, 256, n)
    assert torch.allclose(cs[..., 0], o.sum(1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(cs[..., 1], (o * o).sum(1), rtol=1e-5, atol=1e-3)
    plain = ops.linear(x, w, b, resid=r, tile=tile, split_k=1)
    assert torch.equal(plain, out) and getattr(plain, "chan_sums", None) is None
    gm, bt = dev((1 + 0.2 * torch.randn(n, generator=g)).half()), dev((0.2 * torch.randn(n, generator=g)).half())
    for rows in (256, m // 2, m):  # 4-D-like samples of one slab each, two samples, one 5-D-like sample
        y_s = ops.groupnorm(out, gm, bt, nsample=m // rows, rows_per_sample=rows, groups=32, eps=1e-5, silu=True)
        y_p = ops.groupnorm(plain, gm, bt, nsample=m // rows, rows_per_sample=rows, groups=32, eps=1e-5, silu=True)
        assert (y_s.float() - y_p.float()).abs().max() <= 4e-3 and rel_l2(y_s, y_p) < 1e-4
    # a two-source norm (decoder concat) takes both producers' sums
    out2 = ops.linear(x, w, b, tile=tile, split_k=1, sums=True)
    gm2, bt2 = torch.cat([gm, gm]), torch.cat([bt, bt])
    y_s = ops.groupnorm(out, gm2, bt2, x2=out2, nsample=m // 256, rows_per_sample=256, groups=32, eps=1e-5, silu=False)
    y_p = ops.groupnorm(plain, gm2, bt2, x2=out2.clone(), nsample=m // 256, rows_per_sample=256, groups=32, eps=1e-5, silu=False)
    assert rel_l2(y_s, y_p) < 1e-4


@pytest.mark.parametrize("m,n,k,resid,tile,split", [(1024, 1280, 11520, True, 81, 4), (5120, 1280, 3840, False, 81, 2), (4096, 640, 5760, True, 82, 2),
                                                    (1024, 1280, 2560, True, 11, 2), (1024, 1280, 23040, True, 0, 0)])
def test_split_k_reduction_emits_channel_sums(ops, m, n, k, resid, tile, split):
    """under-filled grids run split-K (the 8 x 8 / 16 x 16 levels): the reduction pass that writes the output also emits the
    per-slab channel sums (gemm.hip: splitk_reduce_sums_kernel) -- same contract as the eight-phase tiles' stats_pass, same stored
    tensor as the plain reduction, GroupNorm from them equal to the three-pass GroupNorm"""
    g = torch.Generator().manual_seed(m + n + k)
    x = dev((torch.randn(m, k, generator=g) * 0.7).half())
    w = dev((torch.randn(n, k, generator=g) / k ** 0.5).half())
    b = dev(torch.randn(n, generator=g).half())
    r = dev((torch.randn(m, n, generator=g) * 2 + 1).half()) if resid else None
    out = ops.linear(x, w, b, resid=r, tile=tile, split_k=split, sums=True)
    plain = ops.linear(x, w, b, resid=r, tile=tile, split_k=split)
    assert torch.equal(plain, out)
    cs = getattr(out, "chan_sums", None)
    assert cs is not None and tuple(cs.shape) == (m // 256, n, 2)
    o = out.float().reshape(m // 256, 256, n)
    assert torch.allclose(cs[..., 0], o.sum(1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(cs[..., 1], (o * o).sum(1), rtol=1e-5, atol=1e-3)
    gm, bt = dev((1 + 0.2 * torch.randn(n, generator=g)).half()), dev((0.2 * torch.randn(n, generator=g)).half())
    y_s = ops.groupnorm(out, gm, bt, nsample=1, rows_per_sample=m, groups=32, eps=1e-5, silu=True)
    y_p = ops.groupnorm(plain, gm, bt, nsample=1, rows_per_sample=m, groups=32, eps=1e-5, silu=True)
    assert (y_s.float() - y_p.float()).abs().max() <= 4e-3 and rel_l2(y_s, y_p) < 1e-4


@pytest.mark.parametrize("m,n,k,resid,tile,act", [(2048, 640, 640, True, 82, 0), (2048, 640, 640, True, 81, 0), (1000, 1280, 1280, True, 82, 0),
                                                  (4096, 512, 2048, True, 81, 0), (777, 1280, 640, False, 81, 0), (2048, 320, 960, False, 82, 0),
                                                  (1536, 640, 2560, True, 82, 0), (1024, 768, 640, True, 81, 1)])
def test_gemm_row_moments_and_layernorm_statistics_from_them(ops, m, n, k, resid, tile, act):
    """LayerNorm statistics from the producer's epilogue (gemm8.hip: rowmom_pass): per row and n-tile the sums equal the sums of the
    STORED fp16 values, the stored tensor is unchanged by the request, and {mean, rstd} merged from them equal mvoc_row_stats_f16 on
    the same tensor to fp32 noise -- n-tiles of 256 and 320 channels with a partial last tile, rows past a 256-row tile, a
    residual, an activation, a row with a mean far from its spread"""
    g = torch.Generator().manual_seed(m + n + k)
    x = dev((torch.randn(m, k, generator=g) * 0.7).half())
    w = dev((torch.randn(n, k, generator=g) / k ** 0.5).half())
    b = dev(torch.randn(n, generator=g).half())
    r = None
    if resid:
        rr = torch.randn(m, n, generator=g) * 2 + 1
        rr[5] += 300.0   # a row whose mean dwarfs its spread (the case a one-pass variance loses)
        r = dev(rr.half())
    out = ops.linear(x, w, b, resid=r, tile=tile, split_k=1, act=ops.ACT_SILU if act else ops.ACT_NONE, rowmom=True)
    rm = getattr(out, "row_moments", None)
    assert rm is not None
    mom, tw = rm
    assert tw == (256 if tile == 81 or act else 320) and tuple(mom.shape) == (m, (n + 255) // 256, 2)
    o = out.float()
    for t in range((n + tw - 1) // tw):
        sl = o[:, t * tw:min(n, (t + 1) * tw)]
        assert torch.allclose(mom[:, t, 0], sl.sum(1), rtol=1e-5, atol=2e-3)
        assert torch.allclose(mom[:, t, 1], (sl * sl).sum(1), rtol=1e-5, atol=2e-3)
    plain = ops.linear(x, w, b, resid=r, tile=tile, split_k=1, act=ops.ACT_SILU if act else ops.ACT_NONE)
    assert torch.equal(plain, out) and getattr(plain, "row_moments", None) is None
    st_m = ops.row_stats_of(out, 1e-5)
    st_p = ops.row_stats(plain, 1e-5)
    assert torch.allclose(st_m[:, 0], st_p[:, 0], rtol=1e-5, atol=1e-4)
    ref = o.double()
    rstd = 1.0 / torch.sqrt(ref.var(1, unbiased=False) + 1e-5)
    ordinary = torch.ones(m, dtype=torch.bool, device=out.device)
    if resid:
        ordinary[5] = False
        # the row whose mean is ~150 x its spread: a tile's sum of squares minus its squared sum cancels 4-5 digits of fp32 --
        # bounded here at 2e-3 on rstd (row_stats' two passes keep 1e-6); activations of the network sit at mean / spread < 10
        assert abs(float(st_m[5, 1]) / float(rstd[5]) - 1) < 2e-3, (float(st_m[5, 1]), float(rstd[5]))
    assert torch.allclose(st_m[ordinary, 1], st_p[ordinary, 1], rtol=2e-5, atol=0)
    assert torch.allclose(st_m[ordinary, 1].double(), rstd[ordinary], rtol=2e-5)
    # requests the kernel cannot honour leave the tensor without statistics: split-K slices, a column view
    if k >= 2048 and m <= 8192:  # (a forced split is honoured only where the workspace policy provides the scratch)
        sk = ops.linear(x, w, b, resid=r, tile=tile, split_k=2, rowmom=True)
        assert getattr(sk, "row_moments", None) is None
    assert getattr(ops.linear(x, w, b, resid=None, tile=tile, split_k=1, n_store=n - 64, rowmom=True), "row_moments", None) is None


@pytest.mark.parametrize("nsample,rows,c,n,sums,offset", [(6, 1024, 320, 320, False, 1.0), (2, 4096, 320, 320, True, 1.0), (20, 256, 128, 128, False, 1.0),
                                                         (1, 8192, 64, 64, True, 1.0), (3, 2048, 320, 320, False, 20.0), (2, 4096, 320, 320, True, 20.0)])
def test_groupnorm_folded_into_xs_linear(ops, nsample, rows, c, n, sums, offset):
    """GN -> proj_in with the norm folded into per-sample weights (mvoc_groupnorm_fold_xs_f16 + mvoc_xs_desc.wp_set_rows): against
    fp32 torch and against the GroupNorm kernels followed by the same linear; statistics from the producer's channel sums too.
    ``offset`` 20: group means 20-40 sigma off zero (offset / outlier channels of a residual stream) -- the folded constant takes its mean
    term from the ROUNDED scaled weight, so the weights' fp16 rounding meets |x - mean|, not |mean| (norm.hip: gn_fold_xs_kernel)"""
    g = torch.Generator().manual_seed(nsample * rows + c)
    m = nsample * rows
    x = torch.randn(m, c, generator=g) * (0.5 + torch.rand(nsample, 1, c, generator=g).repeat(1, rows, 1).reshape(m, c)) + \
        torch.randn(nsample, 1, c, generator=g).repeat(1, rows, 1).reshape(m, c)
    if offset != 1.0:  # one offset per (sample, GROUP): the group's mean sits `offset` sigma off zero, its variance stays ~ 1
        x = x + (offset * (1 + torch.rand(nsample, 1, 32, generator=g))).repeat_interleave(c // 32, 2).repeat(1, rows, 1).reshape(m, c)
    w = (torch.randn(n, c, generator=g) / c ** 0.5).half()
    b = torch.randn(n, generator=g).half()
    gm, bt = (1 + 0.3 * torch.randn(c, generator=g)).half(), (0.3 * torch.randn(c, generator=g)).half()
    if sums:  # x as the output of a GEMM that emits its channel sums (identity weights keep the values)
        xin = ops.linear(dev(x.half()), dev(torch.eye(c).half()), None, tile=81 if c % 64 == 0 and m >= 1024 else 0, split_k=1, sums=True)
        assert getattr(xin, "chan_sums", None) is not None
    else:
        xin = dev(x.half())
    xr = xin.float().cpu()
    ref = F.group_norm(xr.reshape(nsample, rows, c).permute(0, 2, 1), 32, gm.float(), bt.float(), 1e-6).permute(0, 2, 1).reshape(m, c) \
        @ w.float().t() + b.float()
    wp = ops.groupnorm_fold_xs(xin, dev(gm), dev(bt), dev(w), dev(b), nsample=nsample, rows_per_sample=rows, groups=32, eps=1e-6)
    out = ops.xs_linear(xin, wp, n, set_rows=rows)
    h = ops.groupnorm(xin, dev(gm), dev(bt), nsample=nsample, rows_per_sample=rows, groups=32, eps=1e-6, silu=False)
    from mvoc_amd.unet import pack_xs_weights
    two = ops.xs_linear(h, pack_xs_weights(dev(w), dev(b)), n)
    assert rel_l2(out, ref) < 1.5e-3 and rel_l2(two, ref) < 1.5e-3, (rel_l2(out, ref), rel_l2(two, ref))
    assert rel_l2(out, two) < 1.5e-3


@pytest.mark.parametrize("c", [64, 320, 512, 1280])
def test_layernorm(ops, c):
    g = torch.Generator().manual_seed(c)
    x = (torch.randn(1001, c, generator=g) * 1.5 + 0.3).half()
    gm, bt = (1 + 0.2 * torch.randn(c, generator=g)).half(), (0.2 * torch.randn(c, generator=g)).half()
    ref = F.layer_norm(x.float(), (c,), gm.float(), bt.float(), 1e-5)
    out = ops.layernorm(dev(x), dev(gm), dev(bt))
    assert (out.float().cpu() - ref).abs().max() < 1e-2
    assert rel_l2(out, ref) < 1e-3


# ---- PnP injection: bit-exact ---------------------------------------------------------------------------
def _mask_pair(frames, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (2, frames, h, w), generator=g, dtype=torch.int32)
    u8[:, :, : h // 2] = torch.where(torch.rand(2, frames, h // 2, w, generator=g) < 0.5, 255, 0).int()
    soft = (u8.float() / 255).half()
    hard = (u8 > 10)
    return soft, hard


@pytest.mark.parametrize("bg", [False, True])
def test_pnp_tokens_spatial_bit_exact(ops, bg):
    from oracle import pnp_ref
    g = torch.Generator().manual_seed(21)
    Fr, H, W, C, mh, mw = 3, 5, 7, 64, 10, 14
    q = torch.randn(5 * Fr, H * W, C, generator=g).half()
    k = torch.randn(5 * Fr, H * W, C, generator=g).half()
    q[4 * Fr, 0, :4] = torch.tensor([float("inf"), -0.0, float("nan"), 65504.0]).half()  # blend != select
    q[1 * Fr, 0, :4] = torch.tensor([-0.0, -0.0, 1.0, -65504.0]).half()
    soft, hard = _mask_pair(Fr, mh, mw, 5)
    rq, rk = pnp_ref.inject_qk_spatial(q, k, [hard[0], hard[1]], Fr, H, W, inject_background=bg)
    dq, dk = dev(q), dev(k)
    ops.pnp_blend_tokens(dq, dev(hard.half()), x2=dk, frames=Fr, height=H, width=W, channels=C, chunk_stride=Fr * H * W * C,
                         f_stride=H * W * C, p_stride=C, base_chunk0=bg)
    assert torch.equal(bits(dq), bits(rq)) and torch.equal(bits(dk), bits(rk))


@pytest.mark.parametrize("bg", [False, True])
def test_pnp_tokens_temporal_layout_bit_exact(ops, bg):
    """the reference's temporal layout [5*HW, F, C] through the stride parameters, soft float masks"""
    from oracle import pnp_ref
    g = torch.Generator().manual_seed(22)
    Fr, H, W, C = 4, 3, 5, 128
    q = torch.randn(5 * H * W, Fr, C, generator=g).half()
    k = torch.randn(5 * H * W, Fr, C, generator=g).half()
    soft, hard = _mask_pair(Fr, 6, 10, 6)
    rq, rk = pnp_ref.inject_qk_temporal(q, k, [soft[0], soft[1]], H, W, inject_background=bg)
    dq, dk = dev(q), dev(k)
    ops.pnp_blend_tokens(dq, dev(soft), x2=dk, frames=Fr, height=H, width=W, channels=C, chunk_stride=H * W * Fr * C,
                         f_stride=C, p_stride=Fr * C, base_chunk0=bg)
    assert torch.equal(bits(dq), bits(rq)) and torch.equal(bits(dk), bits(rk))


@pytest.mark.parametrize("hw", [(8, 8), (5, 7)])
def test_pnp_nchw_bit_exact(ops, hw):
    from oracle import pnp_ref
    g = torch.Generator().manual_seed(23)
    Fr, C = 3, 20
    H, W = hw
    x = torch.randn(5 * Fr, C, H, W, generator=g).half()
    soft, hard = _mask_pair(Fr, H, W, 7)
    ref = pnp_ref.inject_feature_nchw(x, [hard[0], hard[1]])
    dx = dev(x)
    ops.pnp_blend_nchw(dx, dev(hard.half()), frames=Fr, base_chunk0=True)
    assert torch.equal(bits(dx), bits(ref))


@pytest.mark.parametrize("kind", ["g1_spatial", "g2_temporal"])
@pytest.mark.parametrize("bg", [0, 1])
def test_pnp_against_reference_golden(ops, golden_dir, kind, bg):
    """Q/K after injection as captured from the REFERENCE's processors (tools/gen_golden.py), bit for bit"""
    g = np.load(os.path.join(golden_dir, f"{kind}_proc_bg{bg}.npz"))
    Fr, H, W = int(g["frames"]), int(g["height"]), int(g["width"])
    q, k = torch.from_numpy(g["q_off"][:, 0]), torch.from_numpy(g["k_off"][:, 0])  # pre-injection projections
    C = q.shape[-1]
    dq, dk = dev(q), dev(k)
    if kind == "g1_spatial":
        masks = torch.from_numpy(g["mask_bool"])[:, 0, 0].half()
        ops.pnp_blend_tokens(dq, dev(masks), x2=dk, frames=Fr, height=H, width=W, channels=C, chunk_stride=Fr * H * W * C,
                             f_stride=H * W * C, p_stride=C, base_chunk0=bool(bg))
    else:
        masks = torch.from_numpy(g["mask_float"])[:, 0, 0]
        ops.pnp_blend_tokens(dq, dev(masks), x2=dk, frames=Fr, height=H, width=W, channels=C, chunk_stride=H * W * Fr * C,
                             f_stride=C, p_stride=Fr * C, base_chunk0=bool(bg))
    assert np.array_equal(dq.cpu().numpy().view(np.uint16), g["q_on"][:, 0].view(np.uint16))
    assert np.array_equal(dk.cpu().numpy().view(np.uint16), g["k_on"][:, 0].view(np.uint16))


# ---- loop glue: bit-exact ------------------------------------------------------------------------------
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("cfg", [False, True])
def test_ddim_step_bit_exact(ops, inverse, cfg):
    from oracle import loops_ref, sched_ref
    from mvoc_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    g = torch.Generator().manual_seed(31)
    shp = (1, 4, 16, 32, 32)  # 65536 elements: double-rounding corner cases (~2^-9) must show up if mishandled
    x = torch.randn(shp, generator=g).half()
    vu, vc = torch.randn(shp, generator=g).half(), torch.randn(shp, generator=g).half()
    ref_s = (sched_ref.DDIMInverseSchedulerRef if inverse else sched_ref.DDIMSchedulerRef)()
    ref_s.set_timesteps(50)
    s = (DDIMInverseScheduler if inverse else DDIMScheduler)()
    s.set_timesteps(50, device="cuda")
    assert torch.equal(s.timesteps.cpu(), ref_s.timesteps)
    for t in (ref_s.timesteps[0], ref_s.timesteps[17], ref_s.timesteps[-1]):
        v = loops_ref.cfg_combine(vu, vc, 9.0) if cfg else vc
        ref = loops_ref.scheduler_step_5d(ref_s, v, t, x)
        out = s.step_fused(dev(x), dev(vc), t, v_uncond=dev(vu) if cfg else None, guidance_scale=9.0)
        assert torch.equal(bits(out), bits(ref)), int(t)


@pytest.mark.parametrize("rnf", [False, True])
@pytest.mark.parametrize("ratio", [0.0, 0.8, 0.01])
def test_latent_fusion_bit_exact(ops, rnf, ratio):
    from oracle import loops_ref
    g = torch.Generator().manual_seed(32)
    shp = (1, 4, 16, 30, 34)
    lat, bgl = torch.randn(shp, generator=g).half(), torch.randn(shp, generator=g).half()
    objs = torch.randn((2,) + shp, generator=g).half()
    masks = (torch.randint(0, 256, (2,) + shp, generator=g).float() / 255).half()
    ref = loops_ref.latent_fusion(lat, bgl, [objs[0], objs[1]], [masks[0], masks[1]], ratio, rnf)
    out = ops.latent_fusion(dev(lat), dev(bgl), dev(objs), dev(masks), ratio, rnf)
    assert torch.equal(bits(out), bits(ref))


# ---- stem -------------------------------------------------------------------------------------------------
def test_timestep_embedding(ops):
    from oracle.unet_ref import timestep_embedding
    t = torch.tensor([981.0, 1.0, 500.0, 8.0])
    out = ops.timestep_embedding(t.cuda(), 320)
    ref = timestep_embedding(t, 320)
    assert (out.float().cpu() - ref).abs().max() < 2e-3


def test_conv3x3_small_and_pool(ops):
    from mvoc_amd.unet import pack_conv3x3_small
    g = torch.Generator().manual_seed(41)
    n, cin, cout, h, w = 3, 4, 32, 12, 10
    x = torch.randn(n, cin, h, w, generator=g).half()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / 6).half()
    b = torch.randn(cout, generator=g).half()
    for stride, silu in ((1, True), (2, False)):
        ref = F.conv2d(x.float(), wt.float(), b.float(), stride=stride, padding=1)
        ref = F.silu(ref) if silu else ref
        out, ho, wo = ops.conv3x3_small(dev(_nhwc(x)), pack_conv3x3_small(dev(wt)), dev(b), nimg=n, h=h, wd=w, cin=cin, cout=cout,
                                        stride=stride, silu=silu)
        assert rel_l2(_from_rows(out, n, ho, wo), ref) < 1.5e-3
    ref = F.adaptive_avg_pool2d(x.float(), (8, 4))
    out = ops.adaptive_avgpool(dev(_nhwc(x)), nimg=n, h=h, w=w, c=cin, oh=8, ow=4)
    assert rel_l2(_from_rows(out, n, 8, 4), ref) < 1e-3


def test_temporal_encoder4_and_layout(ops):
    from oracle import unet_ref as U
    torch.manual_seed(0)
    enc = U.I2VGenXLTransformerTemporalEncoder(4, 2, 4, 16)
    U.init_weights_(enc, seed=3, scale_out=False)
    for p in enc.parameters():
        p.copy_(p.half().float() * 3)
    b, f, hw = 2, 5, 9
    x = torch.randn(b, 4, f, hw, 1)
    seq = x[..., 0].permute(0, 3, 2, 1).reshape(b * hw, f, 4)  # [b*hw, f, 4]
    ref = enc(seq.half().float()).reshape(b, hw, f, 4).permute(0, 2, 1, 3)  # [b, f, hw, 4]
    sd = {k: v.half().cuda() for k, v in enc.state_dict().items()}
    blob = torch.cat([sd[k].reshape(-1) for k in ("norm1.weight", "norm1.bias", "attn1.to_q.weight", "attn1.to_k.weight",
                                                  "attn1.to_v.weight", "attn1.to_out.0.weight", "attn1.to_out.0.bias",
                                                  "ff.net.0.proj.weight", "ff.net.0.proj.bias", "ff.net.2.weight", "ff.net.2.bias")])
    tok = torch.empty(b * f * hw, 4, dtype=torch.float16, device="cuda")
    ops.ncfhw_to_tokens(dev(x), tok)
    assert torch.equal(tok.cpu().reshape(b, f, hw, 4), x[..., 0].half().permute(0, 2, 3, 1))
    out = torch.zeros(b * f * hw, 8, dtype=torch.float16, device="cuda")
    ops.temporal_encoder4(tok, blob.contiguous(), out, b=b, f=f, hw=hw, coff=4)
    assert rel_l2(out[:, 4:].reshape(b, f, hw, 4), ref) < 3e-3
    back = ops.tokens_to_ncfhw(out[:, 4:].contiguous(), b, 4, f, hw, 1)
    assert torch.equal(back.cpu()[..., 0].permute(0, 2, 3, 1), out[:, 4:].cpu().reshape(b, f, hw, 4))


@pytest.mark.parametrize("nobj", [1, 3, 4])
@pytest.mark.parametrize("bg", [False, True])
def test_pnp_tokens_n_objects_bit_exact(ops, nobj, bg):
    """the batch layout [bg, obj_1..obj_n, uncond, cond] for n != 2 (the reference hard-codes 5 chunks,
    ``pnp_utils.py:592``; the oracle restates the same chain over a list of n masks)"""
    from oracle import pnp_ref
    g = torch.Generator().manual_seed(40 + nobj)
    Fr, H, W, C, nb = 2, 6, 5, 64, nobj + 3
    q = torch.randn(nb * Fr, H * W, C, generator=g).half()
    k = torch.randn(nb * Fr, H * W, C, generator=g).half()
    hard = torch.rand(nobj, Fr, 12, 10, generator=g) > 0.5
    rq, rk = pnp_ref.inject_qk_spatial(q, k, [hard[j] for j in range(nobj)], Fr, H, W, inject_background=bg)
    dq, dk = dev(q), dev(k)
    ops.pnp_blend_tokens(dq, dev(hard.half()), x2=dk, frames=Fr, height=H, width=W, channels=C, chunk_stride=Fr * H * W * C,
                         f_stride=H * W * C, p_stride=C, base_chunk0=bg)
    assert torch.equal(bits(dq), bits(rq)) and torch.equal(bits(dk), bits(rk))
    # feature form
    x = torch.randn(nb * Fr, 24, H, W, generator=g).half()
    hard2 = torch.rand(nobj, Fr, H, W, generator=g) > 0.5
    ref = pnp_ref.inject_feature_nchw(x, [hard2[j] for j in range(nobj)])
    dx = dev(x)
    ops.pnp_blend_nchw(dx, dev(hard2.half()), frames=Fr, base_chunk0=True)
    assert torch.equal(bits(dx), bits(ref))


@pytest.mark.parametrize("nobj", [1, 2, 4])
@pytest.mark.parametrize("bg", [False, True])
def test_pnp_cfg_off_layout_bit_exact(ops, nobj, bg):
    """classifier-free guidance OFF -- batch [bg, obj_1..obj_n, cond], ONE destination chunk (``ndst=1``).
    Spatial tokens (bool masks), temporal layout (soft masks) and the NCHW feature form, bit for bit against the oracle's
    generalisation (the reference hard-codes `// 5` and cannot run this layout)"""
    from oracle import pnp_ref
    g = torch.Generator().manual_seed(60 + nobj)
    Fr, H, W, C, nb = 3, 6, 5, 64, nobj + 2
    q = torch.randn(nb * Fr, H * W, C, generator=g).half()
    k = torch.randn(nb * Fr, H * W, C, generator=g).half()
    q[1, 0, :4] = torch.tensor([float("inf"), -0.0, float("nan"), 65504.0]).half()
    hard = torch.rand(nobj, Fr, 12, 10, generator=g) > 0.5
    rq, rk = pnp_ref.inject_qk_spatial(q, k, [hard[j] for j in range(nobj)], Fr, H, W, inject_background=bg, ndst=1)
    dq, dk = dev(q), dev(k)
    ops.pnp_blend_tokens(dq, dev(hard.half()), x2=dk, frames=Fr, height=H, width=W, channels=C, chunk_stride=Fr * H * W * C,
                         f_stride=H * W * C, p_stride=C, base_chunk0=bg, ndst=1)
    assert torch.equal(bits(dq), bits(rq)) and torch.equal(bits(dk), bits(rk))
    assert torch.equal(bits(dq[:(nb - 1) * Fr]), bits(q[:(nb - 1) * Fr]))  # sources untouched: only the last chunk is written
    # temporal layout [nb*HW, F, C], soft masks k/255
    qt = torch.randn(nb * H * W, Fr, C, generator=g).half()
    kt = torch.randn(nb * H * W, Fr, C, generator=g).half()
    soft = (torch.randint(0, 256, (nobj, Fr, 9, 7), generator=g).float() / 255).half()
    rqt, rkt = pnp_ref.inject_qk_temporal(qt, kt, [soft[j] for j in range(nobj)], H, W, inject_background=bg, ndst=1)
    dqt, dkt = dev(qt), dev(kt)
    ops.pnp_blend_tokens(dqt, dev(soft), x2=dkt, frames=Fr, height=H, width=W, channels=C, chunk_stride=H * W * Fr * C,
                         f_stride=C, p_stride=Fr * C, base_chunk0=bg, ndst=1)
    assert torch.equal(bits(dqt), bits(rqt)) and torch.equal(bits(dkt), bits(rkt))
    # feature form
    x = torch.randn(nb * Fr, 24, H, W, generator=g).half()
    hard2 = torch.rand(nobj, Fr, H, W, generator=g) > 0.5
    ref = pnp_ref.inject_feature_nchw(x, [hard2[j] for j in range(nobj)], ndst=1)
    dx = dev(x)
    ops.pnp_blend_nchw(dx, dev(hard2.half()), frames=Fr, base_chunk0=True, ndst=1)
    assert torch.equal(bits(dx), bits(ref))


@pytest.mark.parametrize("tile", [81, 82])
@pytest.mark.parametrize("shape", [(3, 64, 320, 9, 7), (2, 128, 320, 16, 16), (5, 64, 640, 5, 33), (1, 64, 160, 3, 3), (2, 192, 320, 40, 8)])
def test_conv3x3_g8_borders(ops, tile, shape):
    """eight-phase kernel: image / row borders come from the hardware range check of the LDS-DMA (rows outside the image carry
    an out-of-range offset).  Shapes cross image boundaries inside one 256-pixel tile (n*h*w not a multiple of the tile, rows
    narrower and wider than the tile), with a residual."""
    from mvoc_amd.unet import pack_conv3x3
    n, cin, cout, h, w = shape
    g = torch.Generator().manual_seed(sum(shape) + tile)
    x = torch.randn(n, cin, h, w, generator=g).half()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half()
    b = torch.randn(cout, generator=g).half()
    res = torch.randn(n, cout, h, w, generator=g).half()
    ref = F.conv2d(x.float(), wt.float(), b.float(), padding=1).half().float() + res.float()
    out, ho, wo = ops.conv3x3(dev(_nhwc(x)), pack_conv3x3(dev(wt)), dev(b), nimg=n, h=h, wd=w, n_store=cout, tile=tile,
                              resid=dev(_nhwc(res)))
    assert rel_l2(_from_rows(out, n, ho, wo), ref) < 1.5e-3
    # exact-integer check of the border masks: all-ones input and weights count the valid taps of every pixel
    xi = torch.ones(n, cin, h, w).half()
    wi = torch.zeros(cout, cin, 3, 3).half()
    wi[:, :8] = 1.0 / 8
    refi = F.conv2d(xi.float(), wi.float(), None, padding=1)
    outi, _, _ = ops.conv3x3(dev(_nhwc(xi)), pack_conv3x3(dev(wi)), dev(torch.zeros(cout).half()), nimg=n, h=h, wd=w,
                             n_store=cout, tile=tile)
    assert torch.equal(_from_rows(outi, n, h, w).float().cpu(), refi)


def test_conv3x3_g8_two_sources(ops):
    from mvoc_amd.unet import pack_conv3x3
    g = torch.Generator().manual_seed(77)
    n, c1, c2, cout, h, w = 3, 128, 64, 160, 12, 10
    x1, x2 = torch.randn(n, c1, h, w, generator=g).half(), torch.randn(n, c2, h, w, generator=g).half()
    wt = (torch.randn(cout, c1 + c2, 3, 3, generator=g) / 40).half()
    b = torch.randn(cout, generator=g).half()
    ref = F.conv2d(torch.cat([x1, x2], 1).float(), wt.float(), b.float(), padding=1)
    for tile in (0, 81, 82):
        out, _, _ = ops.conv3x3(dev(_nhwc(x1)), pack_conv3x3(dev(wt)), dev(b), nimg=n, h=h, wd=w, x2=dev(_nhwc(x2)), n_store=cout,
                                tile=tile)
        assert rel_l2(_from_rows(out, n, h, w), ref) < 1.5e-3
