#!/usr/bin/env python3
"""Per-SHAPE counters of the implicit-GEMM family (GPU box; driven by tools/pmc_shapes.sh, one rocprofv3 --pmc pass per counter
set): records every implicit-GEMM launch of one UNet forward at batch B, keeps the TOP shapes by flops x count, and launches each
REPS times between two marker kernels (mvoc_delay_us) -- the rows between an opening and its closing marker of the counter CSV are that shape's dispatches
(a split-K call = its slab kernel + its reduce kernel).  Without a profiler around it the same script times every shape with HIP
events and writes the manifest the passes are merged against.

usage: python3 tools/gemm_shapes_pmc.py <outdir> [B] [top] [reps]"""
import ctypes as C
import json
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, ".")
from mvoc_amd import ops, _ffi
from mvoc_amd.unet import I2VGenXLUNet

out = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 5
TOP = int(sys.argv[3]) if len(sys.argv) > 3 else 20
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 3
os.makedirs(out, exist_ok=True)
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
orig = ops._gemm


def spy(d, dev=None, *rest, **kw):
    key = (d.a_mode, d.m, d.n, d.k, d.cin, d.c1, d.stride, d.upsample, d.act, bool(d.resid), bool(d.rowadd), d.hout, bool(d.ln_rowsum), int(d.k_order))
    if key not in rec:
        dd = _ffi.GemmDesc()
        C.memmove(C.byref(dd), C.byref(d), C.sizeof(d))
        rec[key] = [dd, 0]
    rec[key][1] += 1
    orig(d, dev, *rest, **kw)


ops._gemm = spy
eng.forward_ext(x, t, fps, il, il, ie, eh)
torch.cuda.synchronize()
ops._gemm = orig


def need(d):
    rows_a = d.m if d.a_mode != 1 else d.nimg * d.hsrc * d.wsrc
    return rows_a * max(d.lda, 1) + 64, rows_a * max(d.lda2, 1) + 64, d.m * d.ldo + 64, d.m * max(d.ldr, 1) + 64


sel = sorted(rec.items(), key=lambda kv: -2.0 * kv[1][0].m * kv[1][0].n * kv[1][0].k * kv[1][1])[:TOP]
mx = [max(need(d)[i] for _, (d, _) in sel) for i in range(4)]
scratch = torch.empty(mx[0], device="cuda", dtype=torch.float16).normal_()
scratch2 = torch.empty(mx[1], device="cuda", dtype=torch.float16).normal_()
outbuf = torch.empty(mx[2], dtype=torch.float16, device="cuda")
res = torch.empty(mx[3], device="cuda", dtype=torch.float16).normal_()
stats = torch.zeros(1 << 24, device="cuda", dtype=torch.float32)  # chan_sums / row_moments / ln_stats targets of the recorded descs
wsbuf = torch.empty(8 * 8192 * 4096, device="cuda", dtype=torch.float32)
stream = torch.cuda.current_stream().cuda_stream
manifest = []
for key, (d, cnt) in sel:
    d.a = scratch.data_ptr()
    if d.a2:
        d.a2 = scratch2.data_ptr()
    d.out = outbuf.data_ptr()
    if d.resid:
        d.resid = res.data_ptr()
    for f_ in ("chan_sums", "row_moments"):
        if getattr(d, f_):
            setattr(d, f_, stats.data_ptr())
    if d.ln_stats:  # {mean, rstd} per row: zeros / anything finite
        d.ln_stats = stats.data_ptr()
    # (the spy copied the descriptor before ops._gemm attached its split-K scratch: same rule as there)
    d.workspace, d.workspace_bytes = None, 0
    if d.act != 1 and d.split_k != 1 and _ffi.lib.mvoc_gemm_workspace_bytes(d.m, d.n, d.k):
        d.workspace, d.workspace_bytes = wsbuf.data_ptr(), wsbuf.numel() * 4
    for _ in range(2):
        orig(d)
    torch.cuda.synchronize()
    _ffi.lib.mvoc_delay_us(1, stream)  # opening marker: the REPS launches of this shape sit between it and the closing marker
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        orig(d)
    e1.record()
    _ffi.lib.mvoc_delay_us(1, stream)  # closing marker (the next shape's warm-up launches follow it)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / REPS * 1e3
    rows_a = d.m if d.a_mode != 1 else d.nimg * d.hsrc * d.wsrc
    # compulsory bytes: source rows once + weights + stored outputs (+ residual)
    ns = d.n // 2 if d.act == 1 else (d.n_store or d.n)
    alg = 2 * (rows_a * d.cin if d.a_mode else d.m * d.k) + 2 * d.n * d.k + 2 * d.m * ns * (2 if d.resid else 1)
    manifest.append({"mode": d.a_mode, "M": d.m, "N": d.n, "K": d.k, "cin": d.cin, "c1": d.c1, "act": d.act, "resid": bool(d.resid), "upsample": d.upsample,
                     "ln": bool(d.ln_rowsum), "korder": int(d.k_order), "count": cnt, "reps": REPS, "us": us, "flop": 2.0 * d.m * d.n * d.k, "alg_bytes": alg})
torch.cuda.synchronize()
tag = os.environ.get("MVOC_PMC_PASS", "time")
json.dump(manifest, open(f"{out}/manifest_{tag}.json", "w"), indent=0)
print("done", len(manifest), "shapes")
