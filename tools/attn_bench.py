#!/usr/bin/env python3
"""flash / temporal attention micro-benchmark on the UNet's shapes (GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from mvoc_amd import ops


def timeit(fn, n=10):
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


B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for (c, hw) in ((320, 4096), (640, 1024), (1280, 256), (1280, 64)):
    nb, heads = 16 * B, c // 64
    qkv = torch.randn(nb * hw, 3 * c, device="cuda").half()
    us = timeit(lambda: ops.flash_attn(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], nbatch=nb, heads=heads, tq=hw, tk=hw))
    fl = 4.0 * nb * heads * hw * hw * 64
    print(f"self  nb={nb} heads={heads} T={hw}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")
    if hasattr(ops.AttnDesc, "v2") and nb % 2 == 0:  # the PnP destination pair: one softmax(q k^T), two value tensors
        half = nb // 2 * hw
        out = torch.empty(nb * hw, c, device="cuda", dtype=torch.float16)
        us2 = timeit(lambda: ops.flash_attn(qkv[:half, :c], qkv[:half, c:2 * c], qkv[:half, 2 * c:], nbatch=nb // 2, heads=heads, tq=hw, tk=hw,
                                            out=out[:half], v2=qkv[half:, 2 * c:], out2=out[half:]))
        print(f"pair  nb={nb // 2}x2 heads={heads} T={hw}: {us2:8.1f} us  (two plain launches of the same work: {us:8.1f} us)  {fl / us2 / 1e6:7.1f} TF/s equivalent")
    q = torch.randn(nb * hw, c, device="cuda").half()
    kv = torch.randn(B * 145, 2 * c, device="cuda").half()
    us = timeit(lambda: ops.flash_attn(q, kv[:, :c], kv[:, c:], nbatch=nb, heads=heads, tq=hw, tk=145, kv_bdiv=16))
    fl = 4.0 * nb * heads * hw * 145 * 64
    print(f"cross nb={nb} heads={heads} T={hw}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s   ({(q.numel() * 2 * 2) / us / 1e3:.0f} GB/s q+o)")
    qkv = torch.randn(B * 16 * hw, 3 * c, device="cuda").half()
    us = timeit(lambda: ops.temporal_attn(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], nsample=B, frames=16, hw=hw, heads=heads))
    print(f"temporal B={B} hw={hw} heads={heads}: {us:8.1f} us  {(B * 16 * hw * c * 2 * 4) / us / 1e3:.0f} GB/s")
