"""per-phase s_memtime ticks of flash3_kernel (block 0, wave 0): library built by `tools/dbg/attn_ab.sh f3dbg -DMVOC_F3_STAMPS`"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["MVOC_HIP_LIB"] = os.path.join(os.getcwd(), "tools/lab/libmvoc_attn_f3dbg.so")
os.environ["MVOC_FLASH3"] = "1"
from mvoc_amd import ops
L = C.CDLL(os.environ["MVOC_HIP_LIB"])
c, hw = 320, 4096
for nb in (16, 80):
    qkv = torch.randn(nb * hw, 3 * c, device="cuda").half()
    for _ in range(3):
        ops.flash_attn(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], nbatch=nb, heads=5, tq=hw, tk=hw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.flash_attn(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], nbatch=nb, heads=5, tq=hw, tk=hw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print(f"nb={nb}: {us:.1f} us, {4.0 * hw * hw * 64 * 5 * nb / us / 1e6:.0f} TF/s (stamped build)")
    h = (C.c_ulonglong * 8)()
    L.mvoc_f3_stamps_read(h)
    nt = h[6]
    print(f"nb={nb}: block 0 wave 0, ticks per KV tile: barrier wait {h[0]/nt:.0f} | LDS write + load issue {h[1]/nt:.0f} | mask + K reads + row max (3 S MFMAs) {h[2]/nt:.0f} | "
          f"probabilities + P V + rest of S {h[3]/nt:.0f} | loop overhead {h[4]/nt:.0f} | total {sum(h[:5])/nt:.0f}; kernel {h[5]} ticks in {h[7] * 10} ns: shader clock {h[5] / (h[7] * 10):.2f} GHz")
