#!/usr/bin/env python3
"""run one GEMM shape a few times (for rocprofv3 --pmc): python3 tools/one_gemm.py mode M N K cin [tile]"""
import sys
import torch
sys.path.insert(0, ".")
from mvoc_amd import ops
mode, M, N, K, cin = [int(x) for x in sys.argv[1:6]]
tile = int(sys.argv[6]) if len(sys.argv) > 6 else 0
dev = "cuda"
w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
b = torch.randn(N, device=dev).half()
if mode == 0:
    x = torch.randn(M, K, device=dev).half()
    f = lambda: ops.linear(x, w, b, tile=tile)
elif mode == 1:
    hw = 64
    nimg = M // (hw * hw)
    x = torch.randn(M, cin, device=dev).half()
    f = lambda: ops.conv3x3(x, w, b, nimg=nimg, h=hw, wd=hw, n_store=N, tile=tile)
else:
    x = torch.randn(M, cin, device=dev).half()
    f = lambda: ops.tconv3(x, w, b, nvid=M // (16 * 4096), frames=16, hw=4096, tile=tile)
for _ in range(5):
    f()
torch.cuda.synchronize()
