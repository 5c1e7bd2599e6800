#!/usr/bin/env python3
"""Per-shape GEMM micro-benchmark (GPU box): records every implicit-GEMM launch of one UNet forward, then times each
unique shape under each tile config.  Usage: python tools/gemm_bench.py [B] [tiles csv]"""
import ctypes as C
import sys
import time
from collections import OrderedDict

import torch

sys.path.insert(0, ".")
from mvoc_amd import ops, _ffi
from mvoc_amd.unet import I2VGenXLUNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
def _tile(x):  # "81" or "81:3" = tile 81 with a forced 3-way split-K (encoded as 1000 * split + tile)
    t, _, sk = x.partition(":")
    return int(t) + 1000 * int(sk or 0)


TILES = [_tile(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
eng = I2VGenXLUNet(device="cuda:0").init_random(8888)
F, h, w = 16, 64, 64
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 4, F, h, w, generator=g).half().cuda()
il = torch.randn(B, 4, F, h, w, generator=g).half().cuda()
ie = torch.randn(B, F, 1024, generator=g).half().cuda()
eh = torch.randn(B, 77, 1024, generator=g).half().cuda()
fps = torch.full((B,), 8.0).cuda()
t = torch.tensor([981.0]).cuda()

rec = OrderedDict()
keep = []
orig = ops._gemm


def spy(d, dev=None, *rest, **kw):
    key = (d.a_mode, d.m, d.n, d.k, d.cin, d.c1, d.stride, d.upsample, d.act, bool(d.resid), bool(d.rowadd), d.hout)
    if key not in rec:
        dd = _ffi.GemmDesc()
        C.memmove(C.byref(dd), C.byref(d), C.sizeof(d))
        dd.workspace, dd.workspace_bytes = None, 0
        rec[key] = [dd, 0]
    rec[key][1] += 1
    orig(d, dev, *rest, **kw)


ops._gemm = spy
out = eng.forward_ext(x, t, fps, il, il, ie, eh)[0]
torch.cuda.synchronize()
ops._gemm = orig
# the recorded descriptors point at freed activations: re-point them at big scratch buffers of the right size
def need(d):
    rows_a = d.m if d.a_mode != 1 else d.nimg * d.hsrc * d.wsrc
    return rows_a * max(d.lda, 1) + 64, rows_a * max(d.lda2, 1) + 64, d.m * d.ldo + 64, d.m * max(d.ldr, 1) + 64
mx = [max(need(d)[i] for d, _ in rec.values()) for i in range(4)]
scratch = torch.empty(mx[0], device="cuda", dtype=torch.float16).normal_()
scratch2 = torch.empty(mx[1], device="cuda", dtype=torch.float16).normal_()
outbuf = torch.empty(mx[2], dtype=torch.float16, device="cuda")
res = torch.empty(mx[3], device="cuda", dtype=torch.float16).normal_()
wsbuf = torch.empty(8 * 8192 * 4096, device="cuda", dtype=torch.float32)
rows = []
tot_fl = 0
for key, (d, cnt) in rec.items():
    d.a = scratch.data_ptr()
    if d.a2:
        d.a2 = scratch2.data_ptr()
    d.out = outbuf.data_ptr()
    if d.resid:
        d.resid = res.data_ptr()
    fl = 2.0 * d.m * d.n * d.k
    if d.m <= 8192 and d.k >= 2048 and d.act != 1:
        d.workspace, d.workspace_bytes = wsbuf.data_ptr(), wsbuf.numel() * 4
    best = None
    per = {}
    for tile_code in TILES:
        tile, sk = tile_code % 1000, tile_code // 1000
        d.tile = tile
        d.split_k = sk
        if sk and not (d.m <= 8192 and d.k >= 2048 and d.act != 1 and d.k % (64 * sk) == 0):
            continue
        if tile == 66 and (d.act == 1 or d.n % 320):
            continue
        if tile == 67 and d.n % 256:
            continue
        if tile in (2, 12, 14, 62, 64) and (d.act == 1 or d.n % 160):
            continue
        if d.act == 1 and tile not in (0, 1, 3, 4, 11, 13, 61, 81):
            continue
        try:
            for _ in range(2):
                orig(d)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 5
            e0.record()
            for _ in range(n):
                orig(d)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
        except RuntimeError as e:
            continue
        per[tile_code] = us
        if best is None or us < best[1]:
            best = (tile_code, us)
    rows.append((key, cnt, fl, per, best))
    tot_fl += fl * cnt
tt = 0
print(f"{'mode':4} {'M':>7} {'N':>6} {'K':>6} {'cin':>5} act cnt   " + " ".join(f"t{t_:>1}:us/TF".rjust(14) for t_ in TILES) + "   total_ms(best)")
for key, cnt, fl, per, best in sorted(rows, key=lambda r: -r[2] * r[1]):
    s = " ".join((f"{per[t_]:8.1f}/{fl / per[t_] / 1e6:5.0f}" if t_ in per else " " * 14) for t_ in TILES)
    tt += best[1] * cnt
    print(f"{key[0]:4d} {key[1]:7d} {key[2]:6d} {key[3]:6d} {key[4]:5d} {key[8]:3d} {cnt:3d}   {s}   {best[1] * cnt / 1e3:7.2f}")
print(f"total gemm ms (best tiles) {tt / 1e3:.1f}  flops {tot_fl / 1e12:.2f} TF  -> {tot_fl / tt / 1e6:.0f} TF/s")
