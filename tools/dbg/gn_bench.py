"""GroupNorm(+SiLU) per shape: three-pass (own statistics) and from the producer's channel sums; GB/s = (read + write) / time"""
import sys, torch
sys.path.insert(0, ".")
from mvoc_amd import ops
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for (c, hw) in ((320, 4096), (640, 1024), (1280, 256), (1280, 64)):
    rows = 16 * hw
    x = torch.randn(B * rows, c, device="cuda").half()
    g, b = torch.ones(c, device="cuda").half(), torch.zeros(c, device="cuda").half()
    out = torch.empty_like(x)
    us3 = timeit(lambda: ops.groupnorm(x, g, b, nsample=B, rows_per_sample=rows, groups=32, eps=1e-5, silu=True, out=out))
    cs = torch.zeros((B * rows // 256, c, 2), dtype=torch.float32, device="cuda")
    xf = x.float().reshape(-1, 256, c)
    cs[:, :, 0] = xf.sum(1); cs[:, :, 1] = (xf * xf).sum(1)
    x.chan_sums = cs
    us1 = timeit(lambda: ops.groupnorm(x, g, b, nsample=B, rows_per_sample=rows, groups=32, eps=1e-5, silu=True, out=out))
    del x.chan_sums
    by = 2.0 * x.numel() * 2
    cp = timeit(lambda: out.copy_(x))
    print(f"B={B} c={c} rows/sample={rows}: own statistics {us3:7.1f} us | from channel sums {us1:7.1f} us ({by / us1 / 1e3:5.0f} GB/s read+write) | torch copy {cp:7.1f} us ({by / cp / 1e3:5.0f} GB/s)")
# the 4-D norms of the spatial transformers: one sample per frame
for (c, hw) in ((320, 4096), (640, 1024), (1280, 256)):
    ns = 16 * B
    x = torch.randn(ns * hw, c, device="cuda").half()
    g, b = torch.ones(c, device="cuda").half(), torch.zeros(c, device="cuda").half()
    out = torch.empty_like(x)
    us3 = timeit(lambda: ops.groupnorm(x, g, b, nsample=ns, rows_per_sample=hw, groups=32, eps=1e-6, silu=False, out=out))
    cs = torch.zeros((ns * hw // 256, c, 2), dtype=torch.float32, device="cuda")
    xf = x.float().reshape(-1, 256, c)
    cs[:, :, 0] = xf.sum(1); cs[:, :, 1] = (xf * xf).sum(1)
    x.chan_sums = cs
    us1 = timeit(lambda: ops.groupnorm(x, g, b, nsample=ns, rows_per_sample=hw, groups=32, eps=1e-6, silu=False, out=out))
    print(f"4-D B={B} c={c} rows/sample={hw} x {ns} samples: own statistics {us3:7.1f} us | from channel sums {us1:7.1f} us")
