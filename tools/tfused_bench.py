#!/usr/bin/env python3
"""Fused LN->QKV->frame-attention kernel against the unfused chain at the 320-channel level (GPU box)."""
import math
import sys

import torch

sys.path.insert(0, ".")
from mvoc_amd import ops  # noqa: E402
from mvoc_amd.unet import Linear, pack_tfused_weights  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
c, heads = 320, 5
w = (torch.randn(3 * c, c, generator=g, device="cuda") / math.sqrt(c)).half()
gm, bt = torch.ones(c, device="cuda").half(), torch.zeros(c, device="cuda").half()
lin = Linear(w).fold_layernorm(gm, bt)
wp = pack_tfused_weights(lin.w_ln, heads)
for ns, frames, hw in ((1, 16, 4096), (5, 16, 4096), (1, 32, 9216)):
    rows = ns * frames * hw
    x = torch.randn(rows, c, generator=g, device="cuda").half()
    fl = 2.0 * rows * 3 * c * c + 4.0 * rows * frames * c

    def unfused():
        q3 = lin.call_ln(x, (gm, bt))
        return ops.temporal_attn(q3[:, :c], q3[:, c:2 * c], q3[:, 2 * c:], nsample=ns, frames=frames, hw=hw, heads=heads)

    def fused():
        return ops.temporal_qkv_attn(x, wp, lin.ln, nsample=ns, frames=frames, hw=hw, heads=heads)

    tu, tf = timed(unfused), timed(fused)
    print(f"B={ns} F={frames} hw={hw}: unfused (row_stats + QKV GEMM + tattn) {tu:8.1f} us = {fl / tu / 1e6:6.0f} TF/s | "
          f"fused {tf:8.1f} us = {fl / tf / 1e6:6.0f} TF/s = {100 * fl / tf / 1e6 / 2500:4.1f} % of 2.5 PF | x{tu / tf:.2f}", flush=True)

import os
if "lab" in os.environ.get("MVOC_HIP_LIB", ""):
    st = torch.zeros(16, dtype=torch.int64, device="cuda")
    os.environ["MVOC_TF_STAMPS"] = str(st.data_ptr())
    ns, frames, hw = 5, 16, 4096
    x = torch.randn(ns * frames * hw, c, generator=g, device="cuda").half()
    ops.temporal_qkv_attn(x, wp, lin.ln, nsample=ns, frames=frames, hw=hw, heads=heads)
    torch.cuda.synchronize()
    v = st.cpu().tolist()
    names = ["bar_M", "issue", "mfma", "vmwait", "bar_E", "epilogues", "prologue"]
    for grp in range(2):
        tot = max(v[grp * 8 + 7], 1)
        print(f"group {'AB'[grp]}: " + "  ".join(f"{n} {100.0 * v[grp * 8 + i] / tot:4.1f}%" for i, n in enumerate(names)) + f" | total {tot} cycles, {tot / 30:.0f} per stage")
