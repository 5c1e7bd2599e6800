import torch, math, time, sys, os, subprocess
sys.path.insert(0, '.')
if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    for d in ("0", "200", "450", "700"):
        print("== MVOC_G4_DELAY", d, flush=True)
        subprocess.run([sys.executable, __file__], env=dict(os.environ, MVOC_G4_DELAY=d))
    sys.exit(0)
from mvoc_amd import ops
torch.manual_seed(0)
def check(m, n, k, resid=True):
    x = torch.randint(-3, 4, (m, k)).half().cuda(); w = torch.randint(-3, 4, (n, k)).half().cuda(); b = torch.randint(-8, 9, (n,)).half().cuda()
    r = torch.randint(-4, 5, (m, n)).half().cuda() if resid else None
    ref = x.float() @ w.float().t() + b.float() + (r.float() if resid else 0)
    out = ops.linear(x, w, b, resid=r, tile=84)
    torch.cuda.synchronize()
    bad = (out.float() != ref).sum().item()
    if bad: print(f"exact m{m} n{n} k{k}: wrong {bad}")
for m, n, k in [(256, 128, 64), (1000, 320, 320), (81920, 640, 640), (4099, 1312, 512)]:
    check(m, n, k)
def bench(m, n, k, resid, tiles=(81, 84, 82)):
    x = (torch.randn(m, k) * 0.7).half().cuda(); w = (torch.randn(n, k) / k ** 0.5).half().cuda(); b = torch.randn(n).half().cuda()
    r = torch.randn(m, n).half().cuda() if resid else None
    res = []
    for tile in tiles:
        if tile == 82 and n % 320: continue
        for _ in range(3): ops.linear(x, w, b, resid=r, tile=tile, split_k=1)
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.linear(x, w, b, resid=r, tile=tile, split_k=1)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        res.append(f"t{tile} {best:7.1f}us {2*m*n*k/best/1e6:6.0f}TF")
    print(f"M={m} N={n} K={k} resid={int(resid)}: " + "  ".join(res), flush=True)
for shp in [(81920, 640, 640, 1), (81920, 640, 640, 0), (327680, 320, 960, 1), (81920, 1920, 640, 0), (20480, 1280, 1280, 1),
            (81920, 640, 1920, 1), (20480, 1280, 5120, 1), (16384, 640, 640, 1), (4096, 1280, 1280, 1)]:
    bench(*shp, tiles=(81, 84))
