import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["MVOC_HIP_LIB"] = os.path.join(os.getcwd(), "tools/lab/libmvoc_fldbg.so")
from mvoc_amd import ops, _ffi
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
    h = (C.c_ulonglong * 8)()
    _ffi.lib._lib.mvoc_flash_dbg_read(h) if hasattr(_ffi.lib, "_lib") else C.CDLL(os.environ["MVOC_HIP_LIB"]).mvoc_flash_dbg_read(h)
    nt = h[6]
    print(f"nb={nb}: {us:.1f} us, {4.0 * hw * hw * 64 * 5 * nb / us / 1e6:.0f} TF/s; block 0 wave 0, ticks per KV tile: barrier-1 wait {h[0]/nt:.0f} | LDS write + barrier-2 {h[1]/nt:.0f} | loads issue + S^T (8 reads, 8 MFMA) {h[2]/nt:.0f} | softmax {h[3]/nt:.0f} | P^T pack + PV (16 tr reads, 8 MFMA) {h[4]/nt:.0f} | total {h[5]/nt:.0f}")
