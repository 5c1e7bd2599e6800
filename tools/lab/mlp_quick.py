"""Round-5 experiment, NOT in the library: the fused feed-forward kernel (tools/lab/xsmlp_experiment.hip.inc) against the two-kernel
path (xslin GEGLU ff1 + gemm8 ff2) and fp32 torch; timing of both.  To run it again: copy the .inc to mvoc_amd/csrc/xsmlp.hip, add it
to build.py: SOURCES (WITHOUT -amdgpu-mfma-vgpr-form), declare mvoc_xsmlp_desc {x, wp, c2, out, int64 m, int32 c, inner, ldo, normalize,
float ln_eps} + mvoc_xs_mlp_f16 in include/mvoc_hip.h and mvoc_amd/_ffi.py (XsMlpDesc); the packing helper and the ops wrapper are
below.  Result: profiles/r5/fused_feed_forward_experiment.txt."""
import sys, torch, math
sys.path.insert(0, '.')
import torch.nn.functional as F
from mvoc_amd import ops
from mvoc_amd.unet import H16, Linear, pack_geglu, pack_xs_weights
import ctypes as C
from mvoc_amd._ffi import check, lib
from mvoc_amd.ops import _chk, _rowmajor, _stream, XS_K
try:
    from mvoc_amd._ffi import XsMlpDesc
except ImportError:
    XsMlpDesc = None  # (the library of the tree does not carry the kernel)


def pack_xs_mlp_weights(wp1, w2):
    """the weight stream of mvoc_xs_mlp_f16: ``wp1`` = pack_xs_weights() of the LayerNorm-folded GEGLU projection ([2 inner / 32 tiles,
    c / 16 + 1, 512]: value tile, gate tile per 32 hidden channels), ``w2`` = ff2's [c, inner] -> [3 inner / 32 stages][c / 16 + 1][512].
    The W2 stage of chunk j holds W2[:, 32 j .. 32 j + 31] as c / 16 fragments, fragment f = (output tile f // 2, k step f % 2), element
    (lane = 32 hh + i, jj) = W2[32 tile + i][32 j + kmap] with kmap = 16 s + 4 hh + jj (jj < 4) | 16 s + 8 + 4 hh + jj - 4: the hidden
    channels a lane of the GEGLU accumulator holds, in the order it holds them (csrc/xsmlp.hip)"""
    c, inner = w2.shape[0], w2.shape[1]
    hc, nk, nt2 = inner // 32, c // 16, c // 32
    assert c % 32 == 0 and inner % 32 == 0 and tuple(wp1.shape) == (2 * hc, nk + 1, 512)
    s_, hh, jj = torch.meshgrid(torch.arange(2), torch.arange(2), torch.arange(8), indexing="ij")
    kmap = torch.where(jj < 4, 16 * s_ + 4 * hh + jj, 16 * s_ + 8 + 4 * hh + jj - 4).reshape(-1).to(w2.device)  # [(s, hh, jj)]
    f = w2.view(nt2, 32, hc, 32)[..., kmap].view(nt2, 32, hc, 2, 2, 8)          # [tile, i, chunk, s, hh, jj]
    f = f.permute(2, 0, 3, 4, 1, 5).reshape(hc, nk, 512)                          # piece = 2 tile + s, lane = 32 hh + i
    w2s = torch.zeros((hc, nk + 1, 512), dtype=H16, device=w2.device)
    w2s[:, :nk] = f
    # stages in the order the kernel's software pipeline consumes them: v0 g0 v1 g1 W2_0 v2 g2 W2_1 ... g_{hc-1} W2_{hc-2} W2_{hc-1}
    seq = [wp1[0], wp1[1]]
    for c in range(hc):
        if c >= 1:
            seq.append(w2s[c - 1])
        if c + 1 < hc:
            seq += [wp1[2 * (c + 1)], wp1[2 * (c + 1) + 1]]
    seq.append(w2s[hc - 1])
    assert len(seq) == 3 * hc
    return torch.stack(seq).contiguous()




def xs_mlp(x, wp, c2, inner, *, normalize=True, eps=1e-5, out=None):
    """x + ff2(GEGLU(ff1(LayerNorm(x)))) in one kernel (include/mvoc_hip.h: mvoc_xs_mlp_f16): x contiguous [m, c]; wp =
    unet.pack_xs_mlp_weights(...); c2 = ff2's bias fp32 [c]"""
    _chk(x, "x"), _chk(wp, "wp"), _chk(c2, "c2", torch.float32)
    m, c = x.shape
    if not x.is_contiguous() or c not in XS_K or inner % 32 or wp.numel() != (inner // 32) * 3 * (c // 16 + 1) * 512 or c2.numel() != c:
        raise RuntimeError("xs_mlp: x must be contiguous [m, c in (64, 128, 320)], wp the pack_xs_mlp_weights() stream, c2 fp32 [c]")
    if out is None:
        out = torch.empty_like(x)
    d = XsMlpDesc()
    d.x, d.wp, d.c2, d.out = x.data_ptr(), wp.data_ptr(), c2.data_ptr(), out.data_ptr()
    d.m, d.c, d.inner, d.ldo, d.normalize, d.ln_eps = m, c, inner, _rowmajor(out, "out"), int(bool(normalize)), eps
    check(lib.mvoc_xs_mlp_f16(C.byref(d), _stream()), "xs_mlp")
    return out



torch.manual_seed(0)
def run(m, c, bench=True):
    inner = 4 * c
    g = torch.Generator().manual_seed(m + c)
    x = (torch.randn(m, c, generator=g) * 1.3 + 0.4).half()
    w1 = (torch.randn(2 * inner, c, generator=g) / math.sqrt(c)).half(); b1 = torch.randn(2 * inner, generator=g).half()
    w2 = (torch.randn(c, inner, generator=g) / math.sqrt(inner)).half(); b2 = torch.randn(c, generator=g).half()
    gm, bt = (1 + 0.3 * torch.randn(c, generator=g)).half(), (0.3 * torch.randn(c, generator=g)).half()
    y = F.layer_norm(x.float(), (c,), gm.float(), bt.float(), 1e-5).half().float() @ w1.float().t() + b1.float()
    hh, gg = y.half().float().chunk(2, dim=-1)
    hid = (hh * F.gelu(gg).half().float()).half().float()
    ref = (hid @ w2.float().t() + b2.float()).half().float() + x.float()
    d = lambda t: t.cuda()
    wp, bp = pack_geglu(d(w1), d(b1))
    ff1 = Linear(wp, bp).fold_layernorm(d(gm), d(bt))
    ff2 = Linear(d(w2), d(b2))
    xd = d(x)
    ff1.wp_ln = pack_xs_weights(ff1.w_ln, ff1.ln[1])
    mw = pack_xs_mlp_weights(ff1.wp_ln, ff2.w)
    c2 = ff2.b.float().contiguous()
    fused = xs_mlp(xd, mw, c2, inner, normalize=True, eps=1e-5)
    two = ff2(ff1.call_ln(xd, (d(gm), d(bt)), act=ops.ACT_GEGLU), resid=xd)
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    print(f"m={m} c={c}: fused vs fp32 {rel(fused, ref):.2e}, two-kernel vs fp32 {rel(two, ref):.2e}, fused vs two-kernel {rel(fused, two.float().cpu()):.2e}", flush=True)
    if bench:
        def timed(fn):
            for _ in range(3): fn()
            best = 1e9
            for _ in range(4):
                torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): fn()
                e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
            return best
        tf = timed(lambda: xs_mlp(xd, mw, c2, inner, normalize=True, eps=1e-5))
        tt = timed(lambda: ff2(ff1.call_ln(xd, (d(gm), d(bt)), act=ops.ACT_GEGLU), resid=xd))
        fl = 2.0 * m * c * 3 * inner
        print(f"   fused {tf:8.1f} us = {fl / tf / 1e6:6.0f} TF/s | ff1 + ff2 {tt:8.1f} us = {fl / tt / 1e6:6.0f} TF/s | x{tt / tf:.2f}", flush=True)
run(300, 64, False); run(4200, 128, False); run(4099, 320, False)
run(65536, 320); run(327680, 320); run(196608, 320)
