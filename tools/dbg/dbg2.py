import sys, torch
sys.path.insert(0, '.')
from mvoc_amd import ops
for tile in (15, 14, 12, 11):
  for m in (333, 128, 2048):
    bad_runs = 0
    for rep in range(5):
        g = torch.Generator().manual_seed(tile * 1000 + m)
        n, k = 320, 320
        x = torch.randint(-3, 4, (m, k), generator=g).float()
        w = torch.randint(-3, 4, (n, k), generator=g).float()
        b = torch.randint(-8, 9, (n,), generator=g).float()
        out = ops.linear(x.half().cuda(), w.half().cuda(), b.half().cuda(), tile=tile)
        ref = x @ w.t() + b
        d = (out.float().cpu() != ref)
        if d.any():
            bad_runs += 1
            if rep == 0:
                idx = d.nonzero()
                print(tile, m, 'mismatch count', int(d.sum()), 'rows', sorted(set(idx[:, 0].tolist()))[:20], 'cols', sorted(set(idx[:, 1].tolist()))[:20])
    print(tile, m, 'bad runs', bad_runs)
