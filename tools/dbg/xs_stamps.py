import math, os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["MVOC_HIP_LIB"] = os.path.join(os.getcwd(), "mvoc_amd/libmvoc_hip_lab.so")
st = torch.zeros(16, dtype=torch.int64, device="cuda")
os.environ["MVOC_XS_STAMPS"] = str(st.data_ptr())
from mvoc_amd._ffi import ACT_GEGLU, ACT_NONE
from mvoc_amd.unet import Linear, pack_geglu
g = torch.Generator(device="cuda").manual_seed(0)
k = 320
gm, bt = torch.ones(k, device="cuda").half(), torch.zeros(k, device="cuda").half()
for m in (65536, 327680):
    x = torch.randn(m, k, generator=g, device="cuda").half()
    res = torch.randn(m, k, generator=g, device="cuda").half()
    for name, n, act, ln, resid in (("to_out 320->320 + resid", 320, ACT_NONE, False, True), ("LN + QKV 320->960", 960, ACT_NONE, True, False),
                                    ("LN + GEGLU ff1 320->2560", 2560, ACT_GEGLU, True, False)):
        w = (torch.randn(n, k, generator=g, device="cuda") / math.sqrt(k)).half()
        b = torch.zeros(n, device="cuda").half()
        if act == ACT_GEGLU:
            w, b = pack_geglu(w, b)
        lin = Linear(w, b)
        if ln:
            lin.fold_layernorm(gm, bt)
        kw = {"act": act}
        if resid:
            kw["resid"] = res
        fn = (lambda: lin.call_ln(x, (gm, bt), **kw)) if ln else (lambda: lin(x, **kw))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        v = st.cpu().tolist()
        tot, T = max(v[4], 1), max(v[5], 1)
        print(f"M={m} {name:28s}: block 0 wave 0: {T} stages, ticks per stage: DMA wait {v[0]/T:.0f} | barrier {v[1]/T:.0f} | reads + MFMAs {v[2]/T:.0f} | epilogue {v[3]/T:.0f} (incl. prologue) | block {tot}")
