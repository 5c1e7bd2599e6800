#!/usr/bin/env python3
"""xslin on the HBM-bound to_out + residual shape with its lab switches (lab library: MVOC_XS_LAB=1 no output stores, =2 the same bytes
stored as 1 KB runs): which side of the traffic holds the kernel at 3.5-4 TB/s?  GPU box; MVOC_HIP_LIB must be the lab library."""
import os, sys
import torch
sys.path.insert(0, ".")
from mvoc_amd.unet import Linear
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
m, c = 327680, 320
x = torch.randn(m, c, device="cuda").half(); r = torch.randn(m, c, device="cuda").half()
lin = Linear((torch.randn(c, c, device="cuda") / c ** 0.5).half(), torch.randn(c, device="cuda").half())
Linear.use_ws = False; Linear.resid_tiled_rows = 1 << 30
for rg in ("1", "2"):
    os.environ["MVOC_XS_RG"] = rg
print(f"lab={os.environ.get('MVOC_XS_LAB', '0')} rg={os.environ.get('MVOC_XS_RG_FORCE', 'auto')}: with residual {timed(lambda: lin(x, resid=r)):7.1f} us | without {timed(lambda: lin(x)):7.1f} us")
