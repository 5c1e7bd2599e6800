#!/usr/bin/env python3
"""Inversion step time vs the number of clips batched into one loop (I2VGenXLPipeline.invert_many), 16 x 512 x 512."""
import sys, time
import torch
sys.path.insert(0, ".")
from mvoc_amd.pipeline import I2VGenXLPipeline
from mvoc_amd.schedulers import DDIMInverseScheduler
pipe = I2VGenXLPipeline.synthetic(device="cuda:0", seed=8888, use_graphs=True)
sched = DDIMInverseScheduler(); sched.set_timesteps(50)
pipe.scheduler = sched
pipe._guidance_scale = 1.0
F, h = 16, 64
for n in (1, 2, 3, 4):
    conds = [pipe._stock_conditioning("", "", f"clip{j}", F, h * 8, h * 8, 8, None, None, None, None) for j in range(n)]
    cond = {k: torch.cat([c[k] for c in conds]).contiguous() for k in conds[0]}
    lat = torch.randn(n, 4, F, h, h, device="cuda", dtype=torch.float16)
    st = pipe._make_stock_step(("b", n), lat, cond, 1.0)
    table, index = sched.coef_table("cuda", 1.0)
    def step(i):
        t = int(sched.timesteps[i % 50]); st["t"].fill_(float(t)); st["coef"].copy_(table[index[t]]); st["run"]()
    for i in range(3): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(6): step(i)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 6 * 1e3
    print(f"clips per loop {n}: {ms:7.2f} ms per iteration = {ms / n:6.2f} ms per clip-step  ({n * 1e3 / ms:5.1f} clip-steps/s)")
