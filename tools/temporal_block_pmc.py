#!/usr/bin/env python3
"""The temporal-attention BLOCK north_star names (TransformerTemporalModel: GroupNorm -> proj_in -> 2 x (LN -> QKV -> frame attention ->
to_out + residual) -> LN -> GEGLU ff1 -> ff2 + residual -> proj_out + residual; pnp_utils.py:170-220, 720-887) at the finest level
(C = 320, 64 x 64 x 16 frames), batch 5, run alone -- for `bash tools/pmc_kernel.sh <tag> _ tools/temporal_block_pmc.py [inject]`:
every kernel the script launches inside the marked region belongs to the block, so the summary's "ALL kernels together" line is
the block's MFMA utilisation by the counter north_star asks for.  `inject`: the composition's form of an up-block site (Q/K
injection on attn1: QKV GEMM + blend + tattn instead of the fused kernel)."""
import sys
import types

import torch

sys.path.insert(0, ".")
from mvoc_amd import ops, pnp_utils  # noqa: E402
from mvoc_amd.unet import I2VGenXLUNet  # noqa: E402

inject = len(sys.argv) > 1 and sys.argv[1] == "inject"
B, F, H, W = 5, 16, 64, 64
eng = I2VGenXLUNet().init_random(seed=1)
site = eng.up_blocks[3].temp_attentions[0] if inject else eng.down_blocks[0].temp_attentions[0]
if inject:
    from mvoc_amd.schedulers import DDIMScheduler
    s = DDIMScheduler(); s.set_timesteps(50)
    pipe = types.SimpleNamespace(unet=eng)
    pnp_utils.modify_diffuser_attention_forward(eng)
    pnp_utils.register_temp_attention_pnp(pipe, s.timesteps[:50], False)
    g = torch.Generator().manual_seed(0)
    m = (torch.rand(2, F, H, W, generator=g) > 0.7)
    masks = [(m[j].half()[None, None].repeat(1, 4, 1, 1, 1), m[j][None, None].repeat(1, 4, 1, 1, 1)) for j in range(2)]
    pnp_utils.register_time_all(pipe, 861, masks)
x = torch.randn(B * F * H * W, 320, device="cuda").half()
x = ops.linear(x, torch.eye(320, device="cuda").half(), None, sums=True)  # as a producer leaves it: with its channel sums
for _ in range(2):
    y = site.forward(eng, x, (B, F, H, W))
torch.cuda.synchronize()
ops.delay_us(1)
for _ in range(4):
    y = site.forward(eng, x, (B, F, H, W))
ops.delay_us(1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    y = site.forward(eng, x, (B, F, H, W))
e1.record(); torch.cuda.synchronize()
print(f"temporal block {'with Q/K injection ' if inject else ''}B={B} C=320 {F}x{H}x{W}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per block")
