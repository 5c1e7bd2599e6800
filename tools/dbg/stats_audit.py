"""which GroupNorm / LayerNorm-statistics calls of a UNet forward still run their own statistics pass (eager, no graphs)"""
import collections, sys, torch
sys.path.insert(0, ".")
import bench
from mvoc_amd import ops
dev = torch.device("cuda:0")
job = bench.Job(dev, 16, 64, use_graphs=False)
log = collections.Counter()
g0, r0 = ops.groupnorm, ops.row_stats
def gn(x, gamma, beta, *, nsample, rows_per_sample, groups, eps, silu, x2=None, out=None):
    cs = ops.chan_sums_of(x, rows_per_sample)
    cs2 = ops.chan_sums_of(x2, rows_per_sample) if x2 is not None else None
    ok = cs is not None and (x2 is None or cs2 is not None)
    log[("gn", MODE, nsample * rows_per_sample, x.shape[1] + (x2.shape[1] if x2 is not None else 0), "cat" if x2 is not None else "", "sums" if ok else "OWN PASS")] += 1
    return g0(x, gamma, beta, nsample=nsample, rows_per_sample=rows_per_sample, groups=groups, eps=eps, silu=silu, x2=x2, out=out)
def rs(x, eps=1e-5):
    log[("row_stats", MODE, x.shape[0], x.shape[1], "", "OWN PASS")] += 1
    return r0(x, eps)
ops.groupnorm, ops.row_stats = gn, rs
import mvoc_amd.unet as U
for m in (U,):
    if hasattr(m, "ops"): pass
MODE = "inv"
job.inversion_step()
MODE = "comp"
job.composition_step()
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: (kv[0][1], kv[0][0], -kv[0][2])):
    print(f"{k[1]:5s} {k[0]:9s} rows {k[2]:7d} C {k[3]:5d} {k[4]:3s} {k[5]:9s} x{v}")
