#!/usr/bin/env python3
"""The CPU side of the metric MEASURED at BASELINE configs[1]'s own size: the oracle
(oracle/: fp32 PyTorch CPU ops, the primitives the reference's diffusers path calls) on the host cores of the node.

  1. thread sweep: one cfg-1 inversion step (8 frames, 32 x 32 latents, B = 1) at 32 / 64 / 128 / 256 threads (capped by the host),
     1 warm-up + 2 measured steps each -> profiles/r6/cpu_threads.txt
  2. cfg 2 in full size: 1 warm-up + 2 measured steps of the B = 1 inversion step (UNet + inverse-DDIM update) and of the B = 5
     composition step (UNet with the PnP hooks of a Q/K-injection step + CFG + DDIM update) at 16 x 64 x 64, at the sweep's best
     thread count -> job-mix steps/s = 4 / (3 t_B1 + t_B5) -> profiles/r6/cpu_baseline_cfg2_measured.json

CPU only (no GPU call); the sweep ~20 min (256 threads: 300 s per step), the cfg-2 steps ~8 min.
usage: python3 tools/cpu_baseline_full.py <outdir> [--skip-b5] [--threads N: that thread count only, no sweep file]"""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import loops_ref, sched_ref, unet_ref as U  # noqa: E402
from oracle.pnp_model_ref import PnPState, install_pnp  # noqa: E402
from mvoc_amd.flops import unet_flops  # noqa: E402  (pure Python: no GPU, no library call)
from mvoc_amd.unet_spec import UNetConfig  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cpu_full"
os.makedirs(out, exist_ok=True)
torch.set_grad_enabled(False)
host = os.cpu_count() or 1


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


TOY = os.environ.get("MVOC_CPU_FULL_TOY") == "1"  # script self-test on the toy network (seconds); never a measurement


def build_model():
    with torch.device("meta"):
        model = U.I2VGenXLUNet(U.UNetConfig.small4() if TOY else U.UNetConfig())
    model = model.to_empty(device="cpu")
    noise = torch.randn(1 << 22) * 0.02
    for p in model.parameters():
        n = p.numel()
        p.view(-1).copy_(noise.repeat((n + noise.numel() - 1) // noise.numel())[:n])
    return model


def timed(fn, warm=1, n=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.time()
        fn()
        ts.append(time.time() - t0)
    return ts


model = build_model()
cfg = UNetConfig(**U.UNetConfig.small4().to_dict()) if TOY else UNetConfig()
CD = model.config.cross_attention_dim
g = torch.Generator().manual_seed(0)

# ---- 1. thread sweep on one cfg-1 step ----------------------------------------------------------------------------------------
f1, hw1 = 8, 32
x1 = torch.randn(1, 4, f1, hw1, hw1, generator=g).half()
il1 = torch.randn(1, 4, f1, hw1, hw1, generator=g)
ie1, eh1, fps1 = torch.randn(1, 1, CD, generator=g), torch.randn(1, 77, CD, generator=g), torch.tensor([8])
inv = sched_ref.DDIMInverseSchedulerRef()
inv.set_timesteps(50)
fl1 = unet_flops(cfg, 1, f1, hw1, hw1)["total"]


def cfg1_step():
    t = int(inv.timesteps[3])
    noise = model(x1.float(), t, fps1, il1, ie1, eh1)[0].half()
    return loops_ref.scheduler_step_5d(inv, noise, t, x1)


sweep = []
lines = [f"host: {host} hardware threads, {cpu_model()}; torch {torch.__version__}; one cfg-1 inversion step (B = 1, {f1} frames, {hw1} x {hw1} latents, "
         f"{fl1 / 1e12:.2f} TFLOP) on the oracle (fp32 PyTorch CPU ops), 1 warm-up + 2 measured", "threads  s/step (two runs)   TFLOP/s"]
FIXED = int(sys.argv[sys.argv.index("--threads") + 1]) if "--threads" in sys.argv else 0  # skip the sweep (it was taken before)
for th in (8, 16, 32, 64, 128, 256):
    if th > host or (FIXED and th != FIXED):
        continue
    torch.set_num_threads(th)
    ts = timed(cfg1_step)
    best = min(ts)
    sweep.append({"threads": th, "s_per_step": ts, "tflops": fl1 / best / 1e12})
    lines.append(f"{th:7d}  {ts[0]:7.2f} {ts[1]:7.2f}   {fl1 / best / 1e12:6.3f}")
    print(lines[-1], flush=True)
best_th = max(sweep, key=lambda r: r["tflops"])["threads"]
lines.append(f"best: {best_th} threads")
if not FIXED:
    open(f"{out}/cpu_threads.txt", "w").write("\n".join(lines) + "\n")

# ---- 2. cfg 2 at full size ----------------------------------------------------------------------------------------------------
torch.set_num_threads(best_th)
F_, h = (4, 8) if TOY else (16, 64)
res = {"host_cores": host, "cpu_model": cpu_model(), "threads": best_th, "kind": "port", "thread_sweep": sweep,
       "what": "oracle (oracle/unet_ref.py + pnp_model_ref.py + loops_ref.py + sched_ref.py, fp32 PyTorch CPU ops) at BASELINE configs[1]'s size "
               "16 x 64 x 64: 1 warm-up + 2 measured steps each of the B = 1 inversion step and the B = 5 composition step (BASELINE.md section 4)"}
x = torch.randn(1, 4, F_, h, h, generator=g).half()
il = torch.randn(1, 4, F_, h, h, generator=g)


def b1_step():
    t = int(inv.timesteps[3])
    noise = model(x.float(), t, fps1, il, ie1, eh1)[0].half()
    return loops_ref.scheduler_step_5d(inv, noise, t, x)


ts = timed(b1_step)
fl = unet_flops(cfg, 1, F_, h, h)["total"]
res["b1_inversion_step_s"] = ts
res["b1_tflops"] = fl / min(ts) / 1e12
print("B=1", ts, flush=True)
json.dump(res, open(f"{out}/cpu_baseline_cfg2_measured.json", "w"), indent=1)
if "--skip-b5" not in sys.argv:
    fwd = sched_ref.DDIMSchedulerRef()
    fwd.set_timesteps(50)
    st = PnPState(conv_schedule=fwd.timesteps[:5], spatial_schedule=fwd.timesteps[:50], temporal_schedule=fwd.timesteps[:50])
    install_pnp(model, st)
    u8 = torch.randint(0, 256, (2, F_, h, h), generator=g)
    masks = [((u8[j].float() / 255).half()[None, None].repeat(1, 4, 1, 1, 1), (u8[j] > 10)[None, None].repeat(1, 4, 1, 1, 1)) for j in range(2)]
    x5 = torch.randn(5, 4, F_, h, h, generator=g).half().float()
    il5 = torch.randn(5, 4, F_, h, h, generator=g)
    ie5, eh5, fps5 = torch.randn(5, F_, CD, generator=g), torch.randn(5, 77, CD, generator=g), torch.tensor([8] * 5)
    tq = int(fwd.timesteps[10])  # a Q/K-injection-only step: 45 of the job's 50

    def b5_step():
        st.t, st.masks = tq, masks
        noise = model.forward_ext(x5, tq, fps5, il5, il5, ie5, eh5)[0]
        return loops_ref.scheduler_step_5d(fwd, loops_ref.cfg_combine(noise[3:4].half(), noise[4:5].half(), 9.0), tq, x5[4:5].half())

    ts5 = timed(b5_step)
    fl5 = unet_flops(cfg, 5, F_, h, h)["total"]
    res["b5_composition_step_s"] = ts5
    res["b5_tflops"] = fl5 / min(ts5) / 1e12
    res["job_mix_steps_per_s"] = 4.0 / (3 * min(ts) + min(ts5))
    print("B=5", ts5, flush=True)
json.dump(res, open(f"{out}/cpu_baseline_cfg2_measured.json", "w"), indent=1)
print(json.dumps(res))
