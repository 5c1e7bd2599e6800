#!/usr/bin/env python3
"""Conditioning prep on the GPU box: 64 conditioning frames through the ViT-H/14 vision tower as one batch vs the
reference's per-frame calls, and the text tower on 3 prompts."""
import sys
import time

import torch

sys.path.insert(0, ".")
from mvoc_amd.clip import CLIPTextModel, CLIPVisionModelWithProjection  # noqa: E402


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


v = CLIPVisionModelWithProjection().init_random(1)
t = CLIPTextModel().init_random(2)
px = torch.randn(64, 3, 224, 224, device="cuda").half()
ids = torch.randint(0, 49408, (3, 77))
one = timed(lambda: v(px))
loop = timed(lambda: [v(px[i:i + 1]) for i in range(64)], n=1)
fl = 64 * 257 * 2 * (32 * (4 * 1280 * 1280 + 2 * 1280 * 5120) + 0) + 64 * 32 * 4 * 257 * 257 * 1280
print(f"vision tower, 64 frames: one batched pass {one:.1f} ms ({fl / one / 1e9:.0f} TFLOP/s) | 64 passes of batch 1 {loop:.1f} ms | x{loop / one:.1f}")
print(f"text tower, 3 prompts x 77 tokens: {timed(lambda: t(ids)):.2f} ms")
a, b = v(px)[5:6].float(), v(px[5:6]).float()  # the GEMM picks tiles / split-K by M: accumulation order differs, values agree
print(f"batched vs per-frame pass, frame 5: rel-L2 {float((a - b).norm() / b.norm()):.2e}")
