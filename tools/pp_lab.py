#!/usr/bin/env python3
"""Diagnostic run of the persistent ping-pong GEMM (GPU box; MVOC_BUILD_LAB=1 python -m mvoc_amd.build first):
per-segment cycle shares of block 0 (in-kernel stamps) and the ablations (no LDS-DMA / zero-page sources) on three
shape classes.  Usage: MVOC_HIP_LIB=mvoc_amd/libmvoc_hip_lab.so python tools/pp_lab.py"""
import os
import sys

import torch

sys.path.insert(0, ".")
from mvoc_amd import ops  # noqa: E402
from mvoc_amd.unet import pack_conv3x3  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev).half()


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def cases():
    # 3x3 conv 1280 -> 1280 at 32x32 x 80 images (B = 5, L1 after downsample): M = 81920, K = 11520
    x = rnd(80 * 32 * 32, 1280)
    w = pack_conv3x3(rnd(1280, 1280, 3, 3) * 0.01)
    b = rnd(1280)
    yield "conv M=81920 N=1280 K=11520", 2.0 * 81920 * 1280 * 11520, lambda tile: ops.conv3x3(x, w, b, nimg=80, h=32, wd=32, n_store=1280, tile=tile, split_k=1)
    # GEGLU ff1 at L0: M = 327680, N = 2560, K = 320
    x2 = rnd(327680, 320)
    w2, b2 = rnd(2560, 320) * 0.05, rnd(2560)
    yield "geglu M=327680 N=2560 K=320", 2.0 * 327680 * 2560 * 320, lambda tile: ops.linear(x2, w2, b2, act=ops.ACT_GEGLU, tile=tile)
    # L1 projection with residual: M = 81920, N = 640, K = 2560 (ff2)
    x3, w3, b3, r3 = rnd(81920, 2560), rnd(640, 2560) * 0.02, rnd(640), rnd(81920, 640)
    yield "linear M=81920 N=640 K=2560 +resid", 2.0 * 81920 * 640 * 2560, lambda tile: ops.linear(x3, w3, b3, resid=r3, tile=tile)


TILES = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [67, 66, 91]
LAB = "lab" in os.environ.get("MVOC_HIP_LIB", "")
names = ["read", "issue", "vmwait", "barrier1", "mfma", "barrier2", "epilogue+rest", "total"]
for name, fl, fn in cases():
    print(f"== {name}")
    os.environ.pop("MVOC_PP_LAB", None)
    for tile in TILES:
        try:
            us = timed(lambda: fn(tile))
            print(f"   tile {tile}: {us:8.1f} us  {fl / us / 1e6:6.0f} TF/s")
        except RuntimeError as e:
            print(f"   tile {tile}: {str(e)[:80]}")
    for bits, label in () if not LAB else ((0, "stamps only"), (1, "no LDS-DMA in the K loop"), (2, "all sources -> zero page")):
        st = torch.zeros(16, dtype=torch.int64, device=dev)
        os.environ["MVOC_PP_LAB"] = f"{bits},{st.data_ptr()}"
        us = timed(lambda: fn(91))
        v = st.cpu().tolist()
        print(f"   lab[{label}]: {us:8.1f} us  {fl / us / 1e6:6.0f} TF/s (stamped build)")
        for grp in range(2):
            tot = max(v[grp * 8 + 7], 1)
            print(f"      group {'AB'[grp]}: " + "  ".join(f"{n} {100.0 * v[grp * 8 + i] / tot:4.1f}%" for i, n in enumerate(names[:7])) +
                  f"  | total {tot} cycles")
    os.environ.pop("MVOC_PP_LAB", None)
