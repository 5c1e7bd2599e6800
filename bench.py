#!/usr/bin/env python3
"""bench.py -- UNet3D denoising steps/sec of MVOC's hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1], "boat_surf demo: background + 2 objects, 16x512x512, 50 DDIM steps, fp16"):
the job is 3 DDIM inversions (background + 2 objects, UNet batch 1, cfg 1.0) and one PnP composition (UNet batch 5
= [bg, obj1, obj2, uncond, cond], cfg 9.0, the five injection families on boat_surf's schedule: Q/K injection on all
50 steps, resnet / temporal-conv / conv_out feature injection on the first 5), 50 steps each = 200 UNet denoising
steps.  The composition's 50 steps are visited in an order that spreads its two step kinds evenly (one feature-injection
step per ten composition steps), so a short run samples the job instead of its first steps only.  One bench "step" is ONE UNet denoising step (UNet forward + injections + CFG + DDIM
update + latent hand-off, all inputs resident in HBM); the K timed steps cycle through the job's mix
[inversion, inversion, inversion, composition], so ``value`` = the job's average steps/s.  Synthetic latents /
conditioning / seeded weights of the exact architecture (no checkpoint or dataset is reachable).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: one rank per GPU.  Launched by torch.distributed.run (RANK / WORLD_SIZE set) this process is one rank; launched plainly it
starts the N ranks itself as child processes (`python -m torch.distributed.run ... bench.py <same flags>`, BEFORE anything here
touches the GPU), relays rank 0's JSON line and exits with the children's status; N beyond the visible devices is refused.

Multi-GPU: the per-object inversions and per-entry compositions are independent (reference loops at
inverse.py:136, composite.py:87), so every rank runs its own shard of steps with NO data-path collective;
torch.distributed (RCCL) is used only for the barrier and the max-over-ranks timing.  scaling = weak.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP16_TFLOPS = 2500.0  # dense MFMA fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0     # HBM3E, same guide


def _rate(fam, name, div):
    """work (flops or bytes) per second of one kernel family, scaled by `div` (1e9 with ms -> T/s, 1e6 -> G/s)"""
    f = fam.get(name)
    return f["work"] / max(f["ms"], 1e-9) / div if f and f["ms"] > 0 else 0.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--latent", type=int, default=64, help="latent height = width (64 <-> 512x512 frames)")
    ap.add_argument("--latent-h", type=int, default=0, help="diagnostic: latent height when not square (with --latent-w; the "
                                                             "reference's own demo size 1280x720 is --latent-h 90 --latent-w 160)")
    ap.add_argument("--latent-w", type=int, default=0)
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--mix", default="job", choices=["job", "inv", "comp"],
                    help="diagnostics: time only inversion (B=1) or only composition (B=5) steps; the metric is --mix job")
    ap.add_argument("--workload", default="boat_surf", choices=["boat_surf", "longclip", "demo"],
                    help="boat_surf = the metric (BASELINE configs[1], per-object shards across GPUs, weak scaling); longclip = "
                         "BASELINE configs[3]: ONE 32-frame 768x768 clip, frame axis sharded over the GPUs (strong scaling); "
                         "demo = diagnostic: END-TO-END wall-clock of the boat_surf-shaped job through the drop-in drivers "
                         "(3 x inverse.py + composite.py, VAE / CLIP / file IO included; tools/demo_job.py)")
    ap.add_argument("--sequential-inversions", action="store_true",
                    help="the job's three source inversions one step after the other on one stream (the definition of rounds 1-3; "
                         "the default runs the three clips' batch-1 steps concurrently on three streams and reports this form beside it)")
    ap.add_argument("--batch-inversions", action="store_true",
                    help="diagnostic: the 3 source inversions of the job share one UNet call per step (batch 3, "
                         "I2VGenXLPipeline.invert_many) instead of three calls at batch 1; --steps must be a multiple of 4")
    ap.add_argument("--exchange", default="a2a", choices=["a2a", "allgather"], help="longclip: frame<->pixel exchange form")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="launcher self-test (runs WITHOUT a GPU, gloo): the ranks rendezvous, take the barrier / max-reduce "
                         "bracket of the timed region around sleeps instead of UNet steps and rank 0 prints a line labelled as "
                         "such -- covers `--gpus N` launched plainly (tests/test_config_launch_cpu.py); never a measurement")
    ap.add_argument("--pmc-pass", action="store_true",
                    help="run under `rocprofv3 --pmc ...` (tools/pmc_bench.sh): eager launches, no priming, the K timed steps "
                         "bracketed by two marker kernels so that the counter rows of exactly these steps can be cut out")
    return ap.parse_args()


class Job:
    """the boat_surf job on synthetic data: an inversion stream (B=1) and a composition stream (B=5)"""

    def __init__(self, device, frames, latent, use_graphs, latent_w=None):
        from mvoc_amd.pipeline import I2VGenXLPipeline
        from mvoc_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
        from mvoc_amd import pnp_utils
        self.device = device
        self.F, self.h, self.w = frames, latent, (latent_w or latent)
        t0 = time.time()
        self.pipe = I2VGenXLPipeline.synthetic(device=device, seed=8888, use_graphs=use_graphs)
        torch.cuda.synchronize()
        self.build_s = time.time() - t0
        pipe, dev, F, h, w = self.pipe, device, frames, latent, self.w
        H, W = h * 8, w * 8
        g = torch.Generator().manual_seed(8888)
        shape = (1, 4, F, h, w)
        # ---- inversion stream (inverse.py: cfg 1.0, prompt "") ------------------------------------------------
        self.inv_sched = DDIMInverseScheduler()
        self.inv_sched.set_timesteps(50)
        pipe.scheduler = self.inv_sched
        pipe._guidance_scale = 1.0
        self.inv_cond = pipe._stock_conditioning("", "", "first-frame", F, H, W, 8, None, None, None, None)
        self.inv_latents = torch.randn(shape, generator=g).to(dev, torch.float16)
        self.inv_state = pipe._make_stock_step("bench-inv", self.inv_latents, self.inv_cond, 1.0)
        self.inv_table, self.inv_index = self.inv_sched.coef_table(dev, 1.0)
        self.inv_i = 0
        self.batch_inversions = False
        self.concurrent = False
        self.inv3_state = None
        # ---- composition stream (composite.py on the boat_surf entry of group_composite/group_config.json) -----
        self.sched = DDIMScheduler()
        self.sched.set_timesteps(50)
        ts = self.sched.timesteps
        pnp_utils.modify_diffuser_attention_forward(pipe.unet)
        pnp_utils.register_temp_attention_pnp(pipe, ts[:int(50 * 1.0)], False)
        pnp_utils.register_spatial_attention_pnp(pipe, ts[:int(50 * 1.0)], False)
        pnp_utils.register_temp_conv_injection(pipe, ts[:int(50 * 0.1)])
        pnp_utils.register_out_conv_injection(pipe, ts[:int(50 * 0.1)])
        pnp_utils.register_resnet_injection(pipe, ts[:int(50 * 0.1)])
        gm = np.load(os.path.join(REPO, "tests", "golden", "g9_boat_surf_masks.npz"))
        masks = []
        for name in ("boat_mask", "surf_mask"):
            u8 = torch.from_numpy(gm[f"{name}_64x64_float_u8"]).float()
            if (h, w) != (64, 64) or frames != 16:
                u8 = torch.nn.functional.interpolate(u8[None], size=(h, w), mode="nearest")[0]
                u8 = u8[torch.arange(frames) % u8.shape[0]]
            fl = (u8 / 255).to(torch.float16)[None, None].repeat(1, 4, 1, 1, 1).to(dev)
            bl = (u8 > 10)[None, None].repeat(1, 4, 1, 1, 1).to(dev)
            masks.append((fl, bl))
        c = pipe.conditioner
        pe, ne = c.encode_prompt("windsurf,sailboat,sky,ocean", "Chaotic, chaotic colors")
        inv_pe, _ = c.encode_prompt("", "")
        il = lambda k: c.image_latents(k, F, H, W)
        emb = lambda k: torch.cat([c.encode_image(f"{k}-{i}") for i in range(F)], 1)
        main_emb = emb("main")
        cond = dict(
            encoder_hidden_states=torch.cat([inv_pe.repeat(3, 1, 1), ne, pe]).contiguous(),
            image_embeddings=torch.cat([emb("bg"), emb("obj1"), emb("obj2"), torch.zeros_like(main_emb), main_emb]).contiguous(),
            image_latents_first=torch.cat([il("bg"), il("obj1"), il("obj2"), il("main"), il("main")]).contiguous(),
            image_latents=torch.cat([il("bg"), il("obj1"), il("obj2"), il("main"), il("main")]).contiguous(),
            fps=torch.full((5,), 8.0, dtype=torch.float32, device=dev))
        pipe.scheduler = self.sched
        pipe._guidance_scale = 9.0
        self.comp_latents = torch.randn(shape, generator=g).to(dev, torch.float16)
        self.comp_state = pipe.make_composition_state(self.comp_latents, cond, masks, 9.0)
        self.comp_table, self.comp_index = self.sched.coef_table(dev, 9.0)
        # inverted latents of the three sources at every timestep: resident in HBM (3 x 50 x 524 KB)
        self.src = {(s, int(t)): torch.randn(shape, generator=g).to(dev, torch.float16) for s in range(3) for t in ts}
        self.comp_i = 0
        self._hooks_live = False
        self.mix = "job"

    def enable_batched_inversions(self):
        """the job's three source inversions (bg, obj1, obj2) as ONE loop at UNet batch 3"""
        pipe = self.pipe
        saved, pipe._guidance_scale = pipe._guidance_scale, 1.0  # inverse.py's cfg: no CFG duplication of the conditioning
        conds = [pipe._stock_conditioning("", "", f"source-{j}", self.F, self.h * 8, self.w * 8, 8, None, None, None, None)
                 for j in range(3)]
        pipe._guidance_scale = saved
        cond = {k: torch.cat([c[k] for c in conds]).contiguous() for k in conds[0]}
        lat = torch.cat([self.inv_latents, self.inv_latents.flip(2), self.inv_latents.flip(3)])
        self.inv3_state = pipe._make_stock_step("bench-inv3", lat, cond, 1.0)
        self.batch_inversions = True

    def enable_concurrent_inversions(self):
        """the job's three source inversions (bg, obj1, obj2) as three batch-1 loops running AT THE SAME TIME on three HIP
        streams (I2VGenXLPipeline.invert_concurrent; `inverse.py --concurrent_entries 3`, the driver's default): every clip
        replays its own captured iteration while the other clips' kernels take the CUs a batch-1 launch leaves idle.  Captured
        with mvoc_gemm_desc.concurrency = 3 as the driver does: NOT the launches of the one-by-one form (the under-filled GEMMs
        keep K in one piece, no split-K reduce), latents within rel-L2 5e-5 per step of it (fp16 rounding noise), not bit-identical"""
        pipe = self.pipe
        saved, pipe._guidance_scale = pipe._guidance_scale, 1.0
        saved_sched, pipe.scheduler = pipe.scheduler, self.inv_sched
        from mvoc_amd import ops
        self.inv_states = []
        with ops.gemm_concurrency(3):  # as invert_concurrent captures them: under-filled batch-1 GEMMs keep K in one piece
            for j in (0, 1, 2):
                cond = pipe._stock_conditioning("", "", f"source-{j}", self.F, self.h * 8, self.w * 8, 8, None, None, None, None)
                lat = self.inv_latents.flip(1 + j) if j else self.inv_latents
                self.inv_states.append(pipe._make_stock_step(f"bench-inv-{j}", lat, cond, 1.0))
        pipe._guidance_scale, pipe.scheduler = saved, saved_sched
        self.inv_streams = [torch.cuda.Stream(device=self.device) for _ in range(3)]
        self.concurrent = True

    def concurrent_inversion_step(self, j):
        """one step of clip j on its own stream (the three clips of a mix period share the period's timestep)"""
        if self._hooks_live:
            from mvoc_amd import pnp_utils
            pnp_utils.register_time_all(self.pipe, None, None)
            self._hooks_live = False
        if j == 0:
            cur = torch.cuda.current_stream()
            for s_ in self.inv_streams:  # the period's inversions start behind the previous composition step
                s_.wait_stream(cur)
        t = int(self.inv_sched.timesteps[(self.inv_i // 3) % 50])
        self.inv_i += 1
        st = self.inv_states[j]
        with torch.cuda.stream(self.inv_streams[j]):
            st["t"].fill_(float(t))
            st["coef"].copy_(self.inv_table[self.inv_index[t]])
            st["run"]()
            snap = st["latents"].clone()  # the per-step snapshot invert() hands to the latent cache
        return snap

    def join_inversions(self):
        cur = torch.cuda.current_stream()
        for s_ in self.inv_streams:
            cur.wait_stream(s_)

    def inversion_step(self):
        if self.batch_inversions:
            return self.batched_inversion_step()
        if self._hooks_live:  # the two stages share one engine here: clear the composition's hook state
            from mvoc_amd import pnp_utils
            pnp_utils.register_time_all(self.pipe, None, None)
            self._hooks_live = False
        t = int(self.inv_sched.timesteps[self.inv_i % 50])
        self.inv_i += 1
        st = self.inv_state
        st["t"].fill_(float(t))
        st["coef"].copy_(self.inv_table[self.inv_index[t]])
        st["run"]()
        return st["latents"].clone()  # the per-step snapshot invert() appends / hands to the latent cache

    def batched_inversion_step(self):
        if self._hooks_live:
            from mvoc_amd import pnp_utils
            pnp_utils.register_time_all(self.pipe, None, None)
            self._hooks_live = False
        t = int(self.inv_sched.timesteps[self.inv_i % 50])
        self.inv_i += 1
        st = self.inv3_state
        st["t"].fill_(float(t))
        st["coef"].copy_(self.inv_table[self.inv_index[t]])
        st["run"]()
        return st["latents"].clone()

    @staticmethod
    def comp_schedule_index(j):
        """the j-th composition step of a run -> index into the demo's 50-step schedule: a permutation of 0..49 in which the
        5 feature-injection steps (indices 0..4) come as every tenth step (j % 10 == 9) and the 45 Q/K-only steps fill the
        rest, i.e. any 10 consecutive composition steps hold the job's 1 : 9 proportion"""
        k = j % 50
        return k // 10 if k % 10 == 9 else 5 + k - k // 10

    def composition_step(self):
        i = self.comp_schedule_index(self.comp_i)
        self.comp_i += 1
        t = int(self.sched.timesteps[i])
        bg, o1, o2 = self.src[(0, t)], self.src[(1, t)], self.src[(2, t)]
        fuse = (0.0, False, [o1, o2]) if i < 1 else None  # fusion_step [0, 1], random_noise_ratio 0.0
        self._hooks_live = True
        self.pipe.composition_step(self.comp_state, t, bg, [o1, o2], self.comp_table[self.comp_index[t]], fuse)

    def is_comp(self, k):
        return {"job": k % 4 == 3, "inv": False, "comp": True}[self.mix]

    def step(self, k):
        if self.is_comp(k):
            if self.concurrent:
                self.join_inversions()  # (a job composes after its inversions; nothing of the two stages overlaps here)
            self.composition_step()
        elif self.concurrent and self.mix == "job":
            self.concurrent_inversion_step(k % 4)
        elif not self.batch_inversions:
            self.inversion_step()
        elif k % 4 == 0:  # steps k, k+1, k+2 of the mix are the three sources' inversion steps: one batched call
            self.batched_inversion_step()


def roofline_leg(job, steps):
    """Repeat the timed steps eagerly with every launch bracketed by HIP events on the launch stream
    (mvoc_prof_*).  A delay kernel is queued first so the host runs ahead and brackets see no launch gaps."""
    from mvoc_amd import ops
    from mvoc_amd.flops import unet_flops
    pipe = job.pipe
    saved = pipe.use_graphs
    # eager twins of the captured iterations
    inv_body = job.inv_state["run"].fn if hasattr(job.inv_state["run"], "fn") else job.inv_state["run"]
    pipe.use_graphs = False
    inv_run = job.inv_state["run"]
    job.inv_state["run"] = inv_body
    job.inv_i, job.comp_i = 0, 0
    # compulsory bytes of every implicit-GEMM launch (source activation once + weights + stored outputs + residual)
    alg_bytes = [0.0]
    real_gemm = ops._gemm

    def spy(d, dev=None, *rest, **kw):
        if d.a_mode == 1:
            a_elems = d.nimg * d.hsrc * d.wsrc * d.cin
        else:
            a_elems = d.m * d.cin
        ns = d.n // 2 if d.act == 1 else (d.n_store if d.n_store > 0 else d.n)
        alg_bytes[0] += 2.0 * (a_elems + d.n * d.k + d.m * ns * (2 if d.resid else 1))
        return real_gemm(d, dev, *rest, **kw)

    real_xs = ops.xs_linear

    def spy_xs(x, wp, n, **kw):  # the K = 320 projections run in the same family through their own entry point
        m, k = x.shape
        ns = n // 2 if kw.get("act", 0) == 1 else (kw.get("n_store") or n)
        alg_bytes[0] += 2.0 * (m * k + n * k + m * ns * (2 if kw.get("resid") is not None else 1))
        return real_xs(x, wp, n, **kw)

    ops._gemm = spy
    ops.xs_linear = spy_xs
    ops.prof_reset()
    ops.prof_enable(True)
    try:
        for k in range(steps):
            ops.delay_us(20000 if job.is_comp(k) else 60000)
            job.step(k)
        torch.cuda.synchronize()
    finally:
        ops._gemm = real_gemm
        ops.xs_linear = real_xs
    ops.prof_enable(False)
    fam = ops.prof_collect()
    ops.prof_reset()
    pipe.use_graphs = saved
    job.inv_state["run"] = inv_run
    cfg = pipe.unet.config
    n_inv = sum(1 for k in range(steps) if not job.is_comp(k))
    n_comp = steps - n_inv
    n_feat = sum(1 for j in range(n_comp) if j % 10 == 9)  # feature-injection steps run on the 3 source chunks
    alg = (n_inv * unet_flops(cfg, 1, job.F, job.h, job.w)["total"] + (n_comp - n_feat) * unet_flops(cfg, 5, job.F, job.h, job.w)["total"]
           + n_feat * unet_flops(cfg, 3, job.F, job.h, job.w)["total"])
    g = fam["gemm"]
    achieved = g["work"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
    total_ms = sum(v["ms"] for v in fam.values())
    # HBM traffic of this kernel family: PMC counters cannot be read inside the timed process (FETCH_SIZE and WRITE_SIZE need
    # separate rocprofv3 --pmc passes), so tools/pmc_bench.sh re-runs `bench.py --pmc-pass` (the same step mix, eager) per
    # counter and cuts out the rows between the two marker kernels.  Its summary is used ONLY if it was taken with the
    # library that is loaded now (digest of csrc) -- a stale summary reports null, not an old number.
    traffic, traffic_src = None, None
    try:
        digest = open(os.path.join(REPO, "mvoc_amd", "libmvoc_hip.so.stamp")).read().strip()
    except OSError:
        digest = None
    cands = []
    for rnd in sorted(os.listdir(os.path.join(REPO, "profiles")), reverse=True):
        f = os.path.join(REPO, "profiles", rnd, "pmc_gemm_traffic.json")
        if os.path.exists(f):
            cands.append((f, f"profiles/{rnd}/pmc_gemm_traffic.json"))
    traffic_note = "no PMC summary committed for this build (tools/pmc_bench.sh)"
    for f, rel in cands:
        pj = json.load(open(f))
        if digest is not None and pj.get("lib_digest") == digest and pj.get("mix") == "3 inversion : 1 composition":
            traffic, traffic_src, traffic_note = pj["hbm_bytes_per_launch"], rel, pj.get("note")
            break
        traffic_note = f"{rel} was taken with another build of the library: not reported"
    return {
        "bound": "mfma", "kernel": "implicit-GEMM family: gemm8_kernel (eight-phase 256-pixel tiles; its share of the family's time is in the rocprofv3 summary under profiles/) + gemm_glds_kernel / gemm_kernel (the general tiles) for linear / conv3x3 / temporal conv, + xslin_kernel (K = 320 projections)",
        "achieved": round(achieved, 2), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP16_TFLOPS, 4),
        "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC, separate rocprofv3 --pmc passes over the same step mix)",
        "traffic_source": traffic_src, "traffic_note": traffic_note,
        "algorithmic_bytes_per_launch_avg": round(alg_bytes[0] / max(g["launches"], 1)),
        "launches": int(g["launches"]), "avg_launch_us": round(1e3 * g["ms"] / max(g["launches"], 1), 2),
        "flops_per_launch_avg": g["work"] / max(g["launches"], 1),
        "share_of_gpu_time": round(g["ms"] / total_ms, 4) if total_ms else None,
        "algorithmic_tflop_all_steps": round(alg / 1e12, 2),
        "executed_matrix_tflop_all_steps": round(sum(fam[k]["work"] for k in ("gemm", "flash_attn", "temporal_fused") if k in fam) / 1e12, 2),
        "by_family_ms": {k: round(v["ms"], 3) for k, v in fam.items()},
        "by_family_launches": {k: int(v["launches"]) for k, v in fam.items()},
        "flash_attn_tflops": round(_rate(fam, "flash_attn", 1e9), 2),
        "temporal_fused_tflops": round(_rate(fam, "temporal_fused", 1e9), 2),
        "temporal_attn_gbps": round(_rate(fam, "temporal_attn", 1e6), 1),
        "groupnorm_gbps": round(_rate(fam, "groupnorm", 1e6), 1),
        "pnp_gbps": round(_rate(fam, "pnp", 1e6), 1),
        # every family against ITS roofline: MFMA-bound families vs the dense fp16 peak, the rest vs the HBM peak
        "family_roofline_frac": {
            "gemm": round(achieved / PEAK_FP16_TFLOPS, 4),
            "flash_attn": round(_rate(fam, "flash_attn", 1e9) / PEAK_FP16_TFLOPS, 4),
            "temporal_fused": round(_rate(fam, "temporal_fused", 1e9) / PEAK_FP16_TFLOPS, 4),
            "temporal_attn": round(_rate(fam, "temporal_attn", 1e6) / PEAK_HBM_GBPS, 4),
            "groupnorm": round(_rate(fam, "groupnorm", 1e6) / PEAK_HBM_GBPS, 4),
            "layernorm": round(_rate(fam, "layernorm", 1e6) / PEAK_HBM_GBPS, 4),
            "pnp": round(_rate(fam, "pnp", 1e6) / PEAK_HBM_GBPS, 4),
        },
        "note": "HIP events around every launch of an eager repeat of the timed steps; achieved = sum(2*m*n*k) / sum(duration)",
    }


def demo(args):
    """north_star's ">= 8x end-to-end wall-clock vs the CPU reference on the boat_surf demo at 1 GPU": the whole job through
    i2vgen-xl/inverse.py (x3 source clips) + i2vgen-xl/composite.py, everything included (model build, PNG decode, VAE encode /
    decode, CLIP towers, 150 + 50 denoising steps, ddim_latents_{t}.pt and result files).  The CPU side cannot be run in full
    (about 8.2 PFLOP of UNet work): it is the cfg-1 CPU sample of the default bench line scaled by FLOPs, stated as such."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import demo_job
    from mvoc_amd.flops import unet_flops
    from mvoc_amd.unet_spec import UNetConfig
    torch.set_grad_enabled(False)
    res = demo_job.run(frames=args.frames, size=(args.latent_h or args.latent) * 8, steps=50)
    cfg = UNetConfig()
    h = args.latent_h or args.latent
    f1, f5, f3 = (unet_flops(cfg, b, args.frames, h, h)["total"] for b in (1, 5, 3))
    job_flops = 150 * f1 + 45 * f5 + 5 * f3
    out = {"metric": "boat_surf demo job end-to-end wall-clock (3 inversions + 1 composition, 50 steps each)", "value": res["wall_s"],
           "unit": "s", "n_gpus": 1, "higher_is_better": False, "dtype": "f16", "data": "synthetic", "vs_baseline": None,
           "config": {"workload": res["job"]}, "detail": res,
           "unet_pflop_of_the_job": round(job_flops / 1e15, 3),
           "unet_tflops_over_the_whole_wall_clock": round(job_flops / res["wall_s"] / 1e12, 1)}
    if not args.no_cpu_baseline:
        try:
            cb = cpu_baseline(args.frames, h)
            cpu_tflops = cb["cfg1_steps_per_s"] * unet_flops(cfg, 1, 8, 32, 32)["total"] / 1e12
            out["cpu_baseline"] = dict(cb, end_to_end_estimate_s=round(job_flops / 1e12 / cpu_tflops, 0),
                                       end_to_end_note="UNet work of the job / the oracle's measured cfg-1 rate on this host "
                                                       "(VAE / CLIP / IO of the CPU path not included: a lower bound of its wall-clock)")
            out["speedup_vs_cpu_estimate"] = round(out["cpu_baseline"]["end_to_end_estimate_s"] / res["wall_s"], 1)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)


def cpu_baseline(frames, latent):
    """the CPU restatement of the same path (oracle/: fp32 PyTorch CPU ops, the primitives diffusers
    calls) timed on this node's host cores.  The bounded sample is BASELINE.json configs[0] IN FULL: one 8-frame 256x256
    clip, 10-step DDIM inversion (UNet forward + inverse-DDIM update per step, oracle/loops_ref.invert_loop), 24.8 TFLOP;
    ``value`` converts it to the metric's unit by FLOPs (one job-mix step of configs[1] = 41.9 TFLOP)."""
    from oracle import loops_ref, sched_ref, unet_ref as U
    from mvoc_amd.flops import unet_flops
    from mvoc_amd.unet_spec import UNetConfig
    # PyTorch CPU ops stop scaling (and regress) far below the hardware threads of the GPU host on these op sizes -- measured on the
    # 256-thread host of the GPU box (profiles/r6/cpu_threads.txt, tools/cpu_baseline_full.py: one cfg-1 step at 8 / 16 / 32 / 64 / 128 /
    # 256 threads = 0.81 / 1.11 / 0.82 / 0.48 / 0.14 / 0.008 TFLOP/s): 16 threads is what is used and what `cores` reports;
    # `host_cores` is what the node has
    host = os.cpu_count() or 1
    cores = min(host, 16)
    torch.set_num_threads(cores)
    f1, hw1, steps1 = 8, 32, 10
    cfg = UNetConfig()
    with torch.device("meta"):
        model = U.I2VGenXLUNet(U.UNetConfig())
    model = model.to_empty(device="cpu")
    with torch.no_grad():
        noise = torch.randn(1 << 22) * 0.02  # cheap init (values are irrelevant to the timing, but must be finite/normal)
        for p in model.parameters():
            n = p.numel()
            p.view(-1).copy_(noise.repeat((n + noise.numel() - 1) // noise.numel())[:n])
        x = torch.randn(1, 4, f1, hw1, hw1).half()
        il = torch.randn(1, 4, f1, hw1, hw1)
        ie = torch.randn(1, 1, 1024)
        eh = torch.randn(1, 77, 1024)
        fps = torch.tensor([8])

        def unet_fn(inp, t):
            return model(inp.float(), int(t), fps, il, ie, eh)[0].half()

        t0 = time.time()
        loops_ref.invert_loop(unet_fn, sched_ref.DDIMInverseSchedulerRef(), x, steps1, 1.0)
        dt = time.time() - t0
    fl_cfg1 = unet_flops(cfg, 1, f1, hw1, hw1)["total"]
    fl_step = (3 * unet_flops(cfg, 1, frames, latent, latent)["total"] + unet_flops(cfg, 5, frames, latent, latent)["total"]) / 4
    cfg1_sps = steps1 / dt
    # the cfg-2 steps MEASURED once on the GPU box's host (1 warm-up + 2 steps at B = 1 and B = 5): quoted beside
    # the extrapolation, never recomputed by the default run (8 min of host time)
    measured = None
    mf = os.path.join(REPO, "profiles", "r6", "cpu_baseline_cfg2_measured.json")
    if os.path.exists(mf):
        mj = json.load(open(mf))
        if mj.get("job_mix_steps_per_s"):
            measured = {"job_mix_steps_per_s": round(mj["job_mix_steps_per_s"], 6), "threads": mj["threads"], "cpu_model": mj["cpu_model"],
                        "b1_inversion_step_s": [round(x, 2) for x in mj["b1_inversion_step_s"]],
                        "b5_composition_step_s": [round(x, 2) for x in mj["b5_composition_step_s"]],
                        "source": "profiles/r6/cpu_baseline_cfg2_measured.json (tools/cpu_baseline_full.py; thread sweep: profiles/r6/cpu_threads.txt)"}
    return {
        "measured_cfg2": measured,
        "value": round(cfg1_sps * fl_cfg1 / fl_step, 6), "unit": "steps/s", "cores": cores, "host_cores": host, "kind": "port",
        "sample": f"BASELINE configs[0] in full: {steps1}-step DDIM inversion of one {f1}-frame {hw1 * 8}x{hw1 * 8} clip on the oracle "
                  f"(oracle/unet_ref.py + loops_ref.py, fp32 PyTorch CPU ops, {cores} threads) = {steps1 * fl_cfg1 / 1e12:.1f} TFLOP in "
                  f"{dt:.1f} s = {cfg1_sps:.3f} cfg1-steps/s; scaled by FLOPs ({fl_cfg1 / 1e12:.2f} -> {fl_step / 1e12:.2f} TFLOP) to the "
                  f"job-mix step -- `value` is therefore an EXTRAPOLATION of the cfg 2 rate by this run (1 warm-up + 2 measured steps at "
                  f"B = 1 and B = 5 cost ~8 min of host time: taken once, `measured_cfg2`, profiles/r6/cpu_baseline_cfg2_measured.json; thread "
                  f"count from the sweep in profiles/r6/cpu_threads.txt)",
        "extrapolated": True,
        "sample_seconds": round(dt, 2), "cfg1_steps_per_s": round(cfg1_sps, 4),
    }


def longclip(args, rank, world, device, dist):
    """BASELINE configs[3]: DDIM inversion steps of one long clip, frame-sharded (mvoc_amd/frame_shard.py).  Extra
    diagnostic line, not the metric: strong scaling, value = steps/s of the ONE clip."""
    from mvoc_amd.pipeline import I2VGenXLPipeline
    from mvoc_amd.schedulers import DDIMInverseScheduler
    from mvoc_amd.flops import unet_flops
    frames = args.frames if args.frames != 16 else 32
    latent = args.latent if args.latent != 64 else 96
    # eager when sharded: the RCCL exchanges are issued by torch.distributed between the library's launches
    pipe = I2VGenXLPipeline.synthetic(device=device, seed=8888, use_graphs=(world == 1 and not args.no_graphs))
    shard = None
    if world > 1:
        from mvoc_amd.frame_shard import FrameShard
        shard = FrameShard(exchange=args.exchange)
        pipe.unet.set_frame_shard(shard)
    sched = DDIMInverseScheduler()
    sched.set_timesteps(50)
    pipe.scheduler = sched
    pipe._guidance_scale = 1.0
    H = W = latent * 8
    cond = pipe._stock_conditioning("", "", "first-frame", frames, H, W, 8, None, None, None, None)
    g = torch.Generator().manual_seed(8888)
    lat = torch.randn((1, 4, frames, latent, latent), generator=g).to(device, torch.float16)
    st = pipe._make_stock_step("longclip", lat, cond, 1.0)
    table, index = sched.coef_table(device, 1.0)

    def step(i):
        t = int(sched.timesteps[i % 50])
        st["t"].fill_(float(t))
        st["coef"].copy_(table[index[t]])
        st["run"]()

    for i in range(max(args.warmup, 2)):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if shard is not None:
        shard.bytes_sent = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    fl = unet_flops(pipe.unet.config, 1, frames, latent, latent)["total"]
    if rank == 0:
        print(json.dumps({
            "metric": "UNet3D denoising steps/sec, one long clip, frame-axis shard", "value": round(args.steps / dt, 4),
            "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 2),
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"DDIM inversion of ONE clip, {frames} frames x {H}x{W}, UNet batch 1, frame axis sharded "
                                   f"{frames // world} frames per GPU, temporal sections pixel-sharded ({args.exchange})",
                       "frames": frames, "height": H, "width": W, "exchange": args.exchange if world > 1 else None,
                       "tflop_per_step": round(fl / 1e12, 2), "end_to_end_tflops": round(fl * args.steps / dt / 1e12, 2),
                       "exchange_payload_mb_per_rank_per_step": None if shard is None else round(shard.bytes_sent / args.steps / 1e6, 1),
                       "hip_graphs": world == 1 and not args.no_graphs}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def visible_gpus_without_hip():
    """GPUs this process may use, counted WITHOUT a HIP / HSA call (torch.cuda.device_count() can fall through to hipGetDeviceCount,
    which initialises the runtime in a parent that is about to start child ranks): KFD topology nodes that have SIMDs, cut down by
    the *_VISIBLE_DEVICES lists the runtime would apply"""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None  # (no KFD topology in this container's sysfs: the caller falls back to the runtime's own count)
    for f in nodes:
        try:
            props = dict(line.split()[:2] for line in open(f) if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without torchrun around it: start the N ranks as CHILD processes -- this parent makes no
    HIP call (devices are counted through sysfs) and never replaces itself -- relay what rank 0 prints and return the launcher's
    exit status (non-zero as soon as one rank fails: torch.distributed.run tears the others down).  The launcher picks its own
    rendezvous port (--standalone), on 127.0.0.1."""
    import subprocess
    n = args.gpus
    if not args.selftest_launch:
        visible = visible_gpus_without_hip()
        if visible is None:  # sysfs tells nothing here: ask the runtime (on builds without amdsmi this initialises HIP in this parent,
            visible = torch.cuda.device_count()  # which starts children and never replaces itself -- within this pool's rule)
        if n > visible and os.environ.get("MVOC_BENCH_OVERSUBSCRIBE") != "1":  # (the one-GPU test box runs two ranks on its GPU)
            raise SystemExit(f"bench.py: --gpus {n} but only {visible} device(s) visible")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout:  # rank 0's JSON line (other ranks print nothing on stdout); stderr passes through
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def rank_identity(rank, local_rank, device):
    """what lets a reader of the N-GPU line confirm that N ranks ran on N distinct GPUs: device index, name, uuid / PCI address"""
    pr = torch.cuda.get_device_properties(device)
    uuid = getattr(pr, "uuid", None)
    pci = None
    if hasattr(pr, "pci_bus_id"):
        pci = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{getattr(pr, 'pci_device_id', 0):02x}"
    return {"rank": rank, "local_rank": local_rank, "device_index": device.index, "device_name": pr.name, "cus": pr.multi_processor_count,
            "uuid": None if uuid is None else str(uuid), "pci": pci, "pid": os.getpid()}


def selftest_launch(args, rank, world):
    """the barrier / max-over-ranks bracket of the timed region around sleeps: checks the launch path, measures nothing"""
    import torch.distributed as dist
    if os.environ.get("MVOC_BENCH_SELFTEST_FAIL_RANK") == str(rank):
        raise SystemExit(3)  # (the test of "exit non-zero if any rank fails")
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    if world > 1:
        dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test (no kernels ran; not a measurement)", "value": None, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(float(tt.item()) / args.steps * 1e3, 3),
                          "scaling": "weak", "data": "none", "config": {"workload": "sleep"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.selftest_launch:
        return selftest_launch(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    local_rank %= max(torch.cuda.device_count(), 1)  # (tests launch two ranks on a one-GPU box)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("MVOC_BENCH_BACKEND", "nccl")  # RCCL; "gloo" only for the one-GPU test of this code path
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # the metric workload uses it for barrier + max-reduce only
        else:
            dist.init_process_group(backend)
    # N ranks on N distinct GPUs: gathered BEFORE anything is timed (a launcher that put two ranks on one device would report a
    # scaling figure of something else); MVOC_BENCH_OVERSUBSCRIBE=1 is the one-GPU test box
    ident = rank_identity(rank, local_rank, device)
    idents = [ident]
    if dist is not None:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        # (one node: two ranks share a GPU iff they report the same device index; the bus address / uuid ride along as evidence and
        # would separate two nodes -- a runtime that reports one uuid for every device must not make a valid run look oversubscribed)
        keys = {(i["device_index"], i["pci"] or i["uuid"]) for i in idents}
        if len(keys) != world and os.environ.get("MVOC_BENCH_OVERSUBSCRIBE") != "1":
            raise SystemExit(f"bench.py: {world} ranks on {len(keys)} distinct device(s): {idents}")

    if args.workload == "longclip":
        return longclip(args, rank, world, device, dist)
    if args.workload == "demo":
        return demo(args)
    if args.pmc_pass:
        args.no_graphs = args.no_roofline = args.no_cpu_baseline = True
        args.warmup = 0
    lat_h, lat_w = (args.latent_h or args.latent), (args.latent_w or args.latent_h or args.latent)
    job = Job(device, args.frames, lat_h, not args.no_graphs, lat_w)
    job.mix = args.mix
    if args.pmc_pass:
        from mvoc_amd import ops
        torch.cuda.synchronize()
        ops.delay_us(1)  # marker kernel (delay_kernel) in front of the timed steps
        for k in range(args.steps):
            job.step(k)
        ops.delay_us(1)  # ... and behind them
        torch.cuda.synchronize()
        print(json.dumps({"pmc_pass": True, "steps": args.steps, "mix": "3 inversion : 1 composition"}), flush=True)
        return
    if args.batch_inversions:
        if args.steps % 4 or args.mix != "job":
            raise SystemExit("--batch-inversions needs --mix job and --steps % 4 == 0 (one batched call = 3 inversion steps)")
        job.enable_batched_inversions()
        args.no_roofline = True  # the roofline leg brackets the metric's own (unbatched) steps
    concurrent = (not args.sequential_inversions and not args.batch_inversions and not args.no_graphs and args.mix == "job"
                  and args.steps % 4 == 0)
    if concurrent:
        job.enable_concurrent_inversions()
    # prime every graph variant the timed region will replay, then W untimed warm-up steps
    for k in range(4):
        job.step(k)
    for j in (9, 19):  # the feature-injection variants (schedule index 0 with latent fusion, 1 without): captured before timing
        job.comp_i = j
        job.composition_step()
    job.comp_i = 0
    def timed_region():
        for k in range(args.warmup):
            job.step(k)
        if job.concurrent:
            job.join_inversions()
        job.inv_i, job.comp_i = 0, 0
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            job.step(k)
        if job.concurrent:
            job.join_inversions()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        own[0] = dt_
        if dist is not None:
            tt = torch.tensor([dt_], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        return dt_

    own = [0.0]
    dt = timed_region()
    ident["ms_per_step"] = round(own[0] / args.steps * 1e3, 3)  # this rank's own clock around the same K steps (the line's is the MAX)
    idents = [ident]
    if dist is not None:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
    dt_seq = None
    if concurrent:  # the same K steps in the rounds 1-3 form (one inversion step after the other), reported beside the metric
        job.concurrent = False
        dt_seq = timed_region()
        job.concurrent = True

    # per-kind timing (informational), same graphs
    def timed(fn, n):
        torch.cuda.synchronize()
        a = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - a) / n * 1e3

    was_concurrent, job.concurrent = job.concurrent, False
    inv_ms = timed(job.inversion_step, 3)
    inv_conc_ms = None
    if was_concurrent:
        def period():
            for j in range(3):
                job.concurrent_inversion_step(j)
            job.join_inversions()
        job.inv_i = 0
        inv_conc_ms = timed(period, 3) / 3
    job.comp_i = 0
    comp_ms = timed(job.composition_step, 2)       # Q/K-injection-only steps: 45 of the job's 50
    job.comp_i = 19
    compf_ms = timed(lambda: (setattr(job, "comp_i", 19), job.composition_step()), 2)  # feature-injection step: 5 of 50

    out = None
    if rank == 0:
        from mvoc_amd.flops import unet_flops
        cfg = job.pipe.unet.config
        f1 = unet_flops(cfg, 1, args.frames, lat_h, lat_w)["total"]
        f5 = unet_flops(cfg, 5, args.frames, lat_h, lat_w)["total"]
        f3 = unet_flops(cfg, 3, args.frames, lat_h, lat_w)["total"]
        out = {
            "metric": "UNet3D denoising steps/sec, 16x512^2 frames, inversion+compose",
            "value": round(world * args.steps / dt, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {
                "workload": ("[--batch-inversions: the 3 inversion steps of each mix period run as ONE UNet call at batch 3] " if args.batch_inversions else "") +
                            ("[the 3 inversion steps of a mix period are one step of each of the job's three source clips (bg, obj1, obj2): "
                             "independent batch-1 loops run on three HIP streams at the same time, captured with the GEMM concurrency hint 3 "
                             "(mvoc_gemm_desc.concurrency: under-filled batch-1 GEMMs keep K in one piece instead of split-K + reduce, so "
                             "the launches DIFFER from the one-after-the-other form and latents agree with it to rel-L2 5e-5 per step, "
                             "not bit for bit); `sequential_inversions` below is the rounds-1-3 form of the same K steps: un-hinted "
                             "launches, one clip after the other] " if was_concurrent else "") +
                            f"boat_surf demo job mix: 3 DDIM-inversion steps (UNet batch 1, cfg 1.0) : 1 PnP composition step "
                            f"(UNet batch 5 = bg+2 objects+uncond+cond, cfg 9.0; the demo's schedule: Q/K injection on every step, "
                            f"resnet / temporal-conv / conv_out feature injection on 5 of 50 -- every tenth composition step of the run; "
                            f"at those the uncond / cond chunks are dead code (conv_out injection overwrites their output) and the "
                            f"UNet runs on the 3 source chunks), "
                            f"{args.frames} frames x {lat_w * 8}x{lat_h * 8}, 50-step DDIM schedules, fp16",
                "frames": args.frames, "height": lat_h * 8, "width": lat_w * 8,
                "unet_params": "1.42 B (I2VGen-XL architecture, seeded synthetic weights)",
                "parallelism": (f"{world} independent shards, one per GPU, no data-path collective; every rank runs the whole job mix, i.e. "
                                f"{3 if was_concurrent else 1} concurrent source clip(s) per rank" if world > 1 else
                                f"single GPU, {3 if was_concurrent else 1} concurrent source clip(s)"),
                "ranks": idents,
                "collective_backend": None if dist is None else {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                                                 "used_for": "barrier + max-over-ranks of the timed region + this table only"},
                "hip_graphs": not args.no_graphs,
                "loop_invariant_conditioning": "context tokens, cross-attention K/V of them and the image-latent stem are computed once "
                                               "per loop (I2VGenXLUNet.prepare_conditioning), bit-identical to recomputing them per step",
                "inversion_step_ms": round(inv_ms, 3),
                "inversion_step_ms_three_clips_concurrent": None if inv_conc_ms is None else round(inv_conc_ms, 3),
                "sequential_inversions": None if dt_seq is None else {
                    "value": round(world * args.steps / dt_seq, 4), "ms_per_step": round(dt_seq / args.steps * 1e3, 3),
                    "note": "the same K steps with the three inversion steps of a period one after the other on one stream (rounds 1-3)"},
                "composition_step_ms": round(comp_ms, 3),
                "composition_feature_injection_step_ms": round(compf_ms, 3),
                "timed_composition_steps": {"qk_injection_only": sum(1 for j in range(sum(1 for k in range(args.steps) if job.is_comp(k))) if j % 10 != 9),
                                            "feature_injection": sum(1 for j in range(sum(1 for k in range(args.steps) if job.is_comp(k))) if j % 10 == 9)},
                "job_average_ms_per_step": round((150 * (inv_conc_ms if inv_conc_ms is not None else inv_ms) + 45 * comp_ms + 5 * compf_ms) / 200, 3),
                "work_not_executed": "every step delivers the reference loop's result; what the engine does not run, because nothing reads it "
                                     "or it is computed twice (each switchable, each with a parity test): (1) composition, conv_out-"
                                     "injection steps (5 of 50): the two destination chunks (unet.prune_dead_chunks, round 2); (2) "
                                     "composition, Q/K-only steps: the three source chunks behind the last injection site "
                                     "(pipeline.prune_source_tail: rest of the last temporal transformer, conv_norm_out, conv_out); (3) "
                                     "composition: the unconditional chunk up to the first cross-attention -- identical to the "
                                     "conditional chunk there (pipeline.share_cfg_prefix); (4) both step kinds: 5 of the 9 taps of the "
                                     "three Upsample2D convs (sub-pixel form: four 2 x 2 parity kernels with summed weights). "
                                     "tflop_per_*_step / end_to_end_tflops below are the REFERENCE forward's FLOP model (SURVEY 8d), "
                                     "i.e. work delivered per second, not matrix work executed: roofline.achieved counts the 2 m n k of the "
                                     "launches that ran",
                "tflop_per_inversion_step": round(f1 / 1e12, 2), "tflop_per_composition_step": round(f5 / 1e12, 2),
                "executed_tflops": None,  # filled from the roofline leg: the matrix work that RAN per second of the timed region
                "end_to_end_tflops": round(sum(f1 if not job.is_comp(k) else (f3 if (sum(1 for q in range(k) if job.is_comp(q)) % 10 == 9) else f5)
                                               for k in range(args.steps)) / dt / 1e12, 2),
            },
        }
    # the two extra legs must never cost the metric line: a failure is reported in place of the object
    if rank == 0 and not args.no_roofline:
        try:
            job.concurrent = False  # (per-launch HIP-event brackets: one stream, one launch at a time)
            out["roofline"] = roofline_leg(job, args.steps)
            # (the eager repeat executes the launches of the timed steps: its matrix work over the TIMED region's clock)
            out["config"]["executed_tflops"] = round(out["roofline"]["executed_matrix_tflop_all_steps"] / dt, 2)
        except Exception as e:  # noqa: BLE001
            out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args.frames, args.latent)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
