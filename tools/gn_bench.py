#!/usr/bin/env python3
"""GroupNorm micro-benchmark (GPU box): the norm shapes of one composition forward (B=5, 16x64x64), GB/s of compulsory
traffic (read x + write y) and of the traffic the three-kernel implementation actually moves (2 reads + 1 write)."""
import sys
import torch
sys.path.insert(0, ".")
from mvoc_amd import ops

B, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 5), 16
cases = []
for C, hw in ((320, 4096), (640, 1024), (1280, 256), (1280, 64)):
    cases.append((f"5D C={C} hw={hw}", B, F * hw, C, None))
    cases.append((f"4D C={C} hw={hw}", B * F, hw, C, None))
cases.append(("4D concat 640+320 hw=4096", B * F, 4096, 640, 320))
for name, ns, rows, c, c2 in cases:
    x = torch.randn(ns * rows, c, device="cuda", dtype=torch.float16)
    x2 = torch.randn(ns * rows, c2, device="cuda", dtype=torch.float16) if c2 else None
    ct = c + (c2 or 0)
    g, b = torch.ones(ct, device="cuda", dtype=torch.float16), torch.zeros(ct, device="cuda", dtype=torch.float16)
    out = torch.empty(ns * rows, ct, device="cuda", dtype=torch.float16)
    for _ in range(3):
        ops.groupnorm(x, g, b, x2=x2, nsample=ns, rows_per_sample=rows, groups=32, eps=1e-5, silu=True, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        ops.groupnorm(x, g, b, x2=x2, nsample=ns, rows_per_sample=rows, groups=32, eps=1e-5, silu=True, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    mb = ns * rows * ct * 2 / 1e6
    print(f"{name:28s} {mb:7.1f} MB  {us:8.1f} us   compulsory {2 * mb / us * 1e3 / 1e3:6.2f} TB/s   moved {3 * mb / us * 1e3 / 1e3:6.2f} TB/s")
