#!/usr/bin/env python3
"""Activation-stationary linear (csrc/xslin.hip) against the tiled GEMM on the K = 320 projections of the finest level (GPU box)."""
import math
import sys

import torch

sys.path.insert(0, ".")
from mvoc_amd._ffi import ACT_GEGLU, ACT_NONE  # noqa: E402
from mvoc_amd.unet import Linear, pack_geglu  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
k = 320
gm, bt = torch.ones(k, device="cuda").half(), torch.zeros(k, device="cuda").half()
for m in (65536, 327680):
    x = torch.randn(m, k, generator=g, device="cuda").half()
    res = torch.randn(m, k, generator=g, device="cuda").half()
    for name, n, act, ln, resid in (("to_out / proj 320->320 + resid", 320, ACT_NONE, False, True),
                                    ("proj_in 320->320", 320, ACT_NONE, False, False),
                                    ("LN + QKV 320->960", 960, ACT_NONE, True, False),
                                    ("LN + GEGLU ff1 320->2560", 2560, ACT_GEGLU, True, False)):
        w = (torch.randn(n, k, generator=g, device="cuda") / math.sqrt(k)).half()
        b = torch.zeros(n, device="cuda").half()
        if act == ACT_GEGLU:
            w, b = pack_geglu(w, b)
        lin = Linear(w, b)
        if ln:
            lin.fold_layernorm(gm, bt)
        kw = {"act": act}
        if resid:
            kw["resid"] = res
        fn = (lambda: lin.call_ln(x, (gm, bt), **kw)) if ln else (lambda: lin(x, **kw))
        Linear.use_xs = False
        told = timed(fn)
        Linear.use_xs = True
        tnew = timed(fn)
        fl = 2.0 * m * n * k
        byts = 2.0 * m * k + 2.0 * m * (n // 2 if act == ACT_GEGLU else n) + (2.0 * m * n if resid else 0)
        print(f"M={m:7d} {name:32s}: tiled {told:8.1f} us = {fl / told / 1e6:5.0f} TF/s | x-stationary {tnew:8.1f} us = "
              f"{fl / tnew / 1e6:5.0f} TF/s, {byts / tnew / 1e3:5.0f} GB/s algorithmic | x{told / tnew:.2f}", flush=True)
