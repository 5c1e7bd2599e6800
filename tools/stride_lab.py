#!/usr/bin/env python3
"""Does the row stride of the activation operand matter (L2 channel camping)?  Times plain linears whose A operand is a
column slice of a wider buffer (row stride = K + pad halfs).  GPU box."""
import sys

import torch

sys.path.insert(0, ".")
from mvoc_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


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


for (m, n, k) in ((81920, 640, 2560), (327680, 320, 1280), (20480, 1280, 5120), (327680, 960, 320), (81920, 1920, 640)):
    w = torch.randn(n, k, generator=g, device=dev).half() * 0.02
    b = torch.randn(n, generator=g, device=dev).half()
    fl = 2.0 * m * n * k
    line = f"M={m} N={n} K={k}: "
    for pad in (0, 8, 32, 64, 136):
        big = torch.randn(m, k + pad, generator=g, device=dev).half()
        x = big[:, :k]
        for tile in (0, 66):
            try:
                us = timed(lambda: ops.linear(x, w, b, tile=tile))
                line += f" pad{pad}/t{tile} {fl / us / 1e6:5.0f}"
            except RuntimeError as e:
                line += f" pad{pad}/t{tile}  n/a"
        del big
    print(line, flush=True)
