#!/usr/bin/env python3
"""flash3 vs flash_kernel, bitwise: run once per MVOC_FLASH3 value (the switch is read once per process), compare the saved outputs."""
import os, subprocess, sys
import torch
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ".")
    from mvoc_amd import ops
    torch.manual_seed(0)
    outs = []
    for (nb, heads, tq, tk, pair) in ((4, 5, 4096, 4096, 0), (2, 5, 920, 920, 0), (4, 10, 1024, 1024, 1), (2, 5, 300, 145, 0), (2, 20, 256, 256, 1), (1, 5, 130, 64, 0), (2, 5, 4096, 4096, 1)):
        c = heads * 64
        q = torch.randn(nb * tq, c, device="cuda").half() * 1.5
        k = torch.randn(nb * tk, c, device="cuda").half() * 1.5
        v = torch.randn(nb * tk, c, device="cuda").half()
        if pair:
            v2 = torch.randn(nb * tk, c, device="cuda").half()
            o = torch.empty(nb * tq, c, device="cuda", dtype=torch.float16); o2 = torch.empty_like(o)
            ops.flash_attn(q, k, v, nbatch=nb, heads=heads, tq=tq, tk=tk, out=o, v2=v2, out2=o2)
            outs += [o.cpu(), o2.cpu()]
        else:
            outs.append(ops.flash_attn(q, k, v, nbatch=nb, heads=heads, tq=tq, tk=tk).cpu())
    torch.save(outs, sys.argv[2])
else:
    for v in ("0", "1"):
        subprocess.check_call([sys.executable, __file__, "child", f"/tmp/f3_{v}.pt"], env=dict(os.environ, MVOC_FLASH3=v))
    a, b = torch.load("/tmp/f3_0.pt"), torch.load("/tmp/f3_1.pt")
    for i, (x, y) in enumerate(zip(a, b)):
        print(i, "bit-identical" if torch.equal(x, y) else f"DIFFERENT max {float((x.float() - y.float()).abs().max()):.3e} nan {int(torch.isnan(y).sum())}")
