"""self-attention at the reference's native latent size (90 x 160 = 14 400 tokens) and 4 096, B = 1 and the pair form"""
import sys, torch
sys.path.insert(0, ".")
from mvoc_amd import ops
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (nb, heads, T) in ((16, 5, 14400), (32, 5, 14400), (80, 5, 4096), (16, 5, 4096), (16, 10, 3600)):
    c = heads * 64
    qkv = torch.randn(nb * T, 3 * c, device="cuda").half()
    us = timeit(lambda: ops.flash_attn(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], nbatch=nb, heads=heads, tq=T, tk=T))
    fl = 4.0 * nb * heads * T * T * 64
    half = nb // 2 * T
    out = torch.empty(nb * T, c, device="cuda", dtype=torch.float16)
    us2 = timeit(lambda: ops.flash_attn(qkv[:half, :c], qkv[:half, c:2 * c], qkv[:half, 2 * c:], nbatch=nb // 2, heads=heads, tq=T, tk=T, out=out[:half], v2=qkv[half:, 2 * c:], out2=out[half:]))
    print(f"nb={nb} heads={heads} T={T}: self {us:9.1f} us {fl / us / 1e6:7.1f} TF/s | pair {us2:9.1f} us {fl / us2 / 1e6:7.1f} TF/s equivalent")
