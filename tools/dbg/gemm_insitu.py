"""GEMM launches of one eager UNet forward: duration in situ (between its producer and its consumer) against the same call
replayed alone right afterwards on the same buffers.  usage: gemm_insitu.py [comp|inv]"""
import collections, copy, ctypes as C, sys, torch
sys.path.insert(0, ".")
import bench
from mvoc_amd import ops
from mvoc_amd._ffi import lib
mode = sys.argv[1] if len(sys.argv) > 1 else "comp"
dev = torch.device("cuda:0")
job = bench.Job(dev, 16, 64, use_graphs=False)
step = job.composition_step if mode == "comp" else job.inversion_step
step(); step()
torch.cuda.synchronize()
rec = []
g0 = ops._gemm
def spy(d, dev=None, out=None, sums=False):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g0(d, dev, out, sums)
    e1.record()
    dd = type(d)()
    C.memmove(C.byref(dd), C.byref(d), C.sizeof(d))
    rec.append((dd, e0, e1))
ops._gemm = spy
keep = []
# keep every tensor alive so the replays see valid buffers: simplest is to disable freeing by holding the allocator (no empty_cache) and
# replaying immediately after the forward -- the caching allocator does not return memory, buffers stay mapped (contents may be stale)
step()
torch.cuda.synchronize()
ops._gemm = g0
rows = collections.OrderedDict()
for d, e0, e1 in rec:
    key = (d.a_mode, d.m, d.n, d.k, d.act, bool(d.resid), bool(d.ln_rowsum), bool(d.rowadd))
    rows.setdefault(key, []).append((d, e0.elapsed_time(e1) * 1e3))
tot_i = tot_a = 0.0
print(f"{'mode':>4} {'M':>7} {'N':>6} {'K':>6} act res ln ra  cnt   in situ us   alone us   ratio")
for key, lst in rows.items():
    d = lst[0][0]
    d.chan_sums = 0
    st = ops._stream()
    for _ in range(2): lib.mvoc_gemm_f16(C.byref(d), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): lib.mvoc_gemm_f16(C.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    alone = e0.elapsed_time(e1) / 5 * 1e3
    insitu = sum(t for _, t in lst) / len(lst)
    tot_i += insitu * len(lst); tot_a += alone * len(lst)
    print(f"{key[0]:4d} {key[1]:7d} {key[2]:6d} {key[3]:6d} {key[4]:3d} {int(key[5]):3d} {int(key[6]):2d} {int(key[7]):2d} {len(lst):4d} {insitu:12.1f} {alone:10.1f} {insitu / alone:7.2f}")
print(f"total: in situ {tot_i / 1e3:.1f} ms, alone {tot_a / 1e3:.1f} ms, ratio {tot_i / tot_a:.3f}  ({mode} step, {len(rec)} GEMM launches)")
