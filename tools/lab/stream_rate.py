#!/usr/bin/env python3
"""What a 2-reads-1-write streaming pass over the finest level's tensors reaches (the K = 320 projections + residual are HBM-bound:
x [327 680, 320] + residual in, the same shape out = 629 MB), beside the two kernels that run those projections.  GPU box."""
import sys
import torch
sys.path.insert(0, ".")
from mvoc_amd import ops
from mvoc_amd.unet import Linear


def timed(fn, n=20):
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


for m in (65536, 327680):
    c = 320
    x = torch.randn(m, c, device="cuda").half()
    r = torch.randn(m, c, device="cuda").half()
    o = torch.empty_like(x)
    nbytes = 3 * m * c * 2
    us = timed(lambda: torch.add(x, r, out=o))
    print(f"M={m}: torch add (2 reads + 1 write, {nbytes / 1e6:.0f} MB): {us:7.1f} us = {nbytes / us / 1e6:.2f} TB/s")
    us = timed(lambda: ops.add(x, r))
    print(f"M={m}: mvoc_add_f16: {us:7.1f} us = {nbytes / us / 1e6:.2f} TB/s")
    us = timed(lambda: o.copy_(x))
    print(f"M={m}: copy (1 read + 1 write, {2 * m * c * 2 / 1e6:.0f} MB): {us:7.1f} us = {2 * m * c * 2 / us / 1e6:.2f} TB/s")
    w = (torch.randn(c, c, device="cuda") / c ** 0.5).half()
    b = torch.randn(c, device="cuda").half()
    lin = Linear(w, b)
    for rows in (0, 1 << 30):  # 0: the 320-wide eight-phase tile takes the residual projections; huge: xslin keeps them
        Linear.resid_tiled_rows = rows if rows else 1
        us = timed(lambda: lin(x, resid=r))
        print(f"M={m}: to_out + residual via {'gemm8 320-wide' if rows == 0 else 'xslin'}: {us:7.1f} us = {nbytes / us / 1e6:.2f} TB/s, {2.0 * m * c * c / us / 1e6:.0f} TF/s")
