import math, sys, torch
sys.path.insert(0, ".")
from mvoc_amd._ffi import ACT_GEGLU, ACT_NONE, ACT_GELU, ACT_SILU
from mvoc_amd.unet import Linear, pack_geglu
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator(device="cuda").manual_seed(0)
k, m, n = 320, 327680, 2560
x = torch.randn(m, k, generator=g, device="cuda").half()
w = (torch.randn(n, k, generator=g, device="cuda") / math.sqrt(k)).half()
b = torch.zeros(n, device="cuda").half()
for name, act in (("none", ACT_NONE), ("silu", ACT_SILU), ("gelu", ACT_GELU), ("geglu", ACT_GEGLU)):
    lin = Linear(w, b)
    for xs in (False, True):
        Linear.use_xs = xs
        t = timed(lambda: lin(x, act=act))
        print(f"{name:6s} xs={xs}: {t:8.1f} us  {2.0*m*n*k/t/1e6:5.0f} TF/s", flush=True)

import os
if "lab" in os.environ.get("MVOC_HIP_LIB", ""):
    st = torch.zeros(8, dtype=torch.int64, device="cuda")
    os.environ["MVOC_XS_STAMPS"] = str(st.data_ptr())
    for name, act in (("none", ACT_NONE), ("gelu", ACT_GELU), ("geglu", ACT_GEGLU)):
        lin = Linear(w, b)
        Linear.use_xs = True
        lin(x, act=act); torch.cuda.synchronize()
        v = st.cpu().tolist()
        tot = max(v[4], 1)
        print(f"{name:6s} wave 0 of block 0: DMA wait {100*v[0]/tot:4.1f}%  barrier {100*v[1]/tot:4.1f}%  issue+reads+MFMA {100*v[2]/tot:4.1f}%  "
              f"epilogue(+prologue) {100*v[3]/tot:4.1f}% | {tot} cycles (100 MHz s_memtime), {tot / max(v[5],1):.0f} per stage")
