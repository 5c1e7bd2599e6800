#!/usr/bin/env python3
"""End-to-end wall-clock of the boat_surf-shaped job through the drop-in drivers (GPU box): what BASELINE.json's
north_star states its ">= 8x end-to-end" target on (reference call sites: i2vgen-xl/inverse.py:111-227 for the three source
clips, i2vgen-xl/composite.py:72-224 for the composition).

A synthetic data tree of the demo's shape is written to a scratch directory (3 source clips of 16 frames at 512x512 as PNGs,
2 x 16 object masks, an edited first frame), then this repo's `i2vgen-xl/inverse.py` main() runs the three 50-step inversions
and `i2vgen-xl/composite.py` main() the 50-step composition -- seeded synthetic UNet / VAE / CLIP weights of the real
architectures (MVOC_SYNTHETIC_VAE=1, MVOC_SYNTHETIC_CLIP=1), files written exactly as the reference writes them
(ddim_latents_{t}.pt per step, video.gif + video_{i:05d}.png).  Shares of the wall-clock are measured by wrapping the VAE /
CLIP / file entry points with synchronising timers (IO excluded from the metric and reported separately).

Used by `bench.py --workload demo`; prints one JSON object.  usage: python tools/demo_job.py [--frames 16] [--size 512] [--steps 50]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Shares:
    """accumulates synchronised wall-clock per category; nested calls are charged to the innermost category only"""

    def __init__(self):
        self.t = {}
        self.stack = []

    def wrap(self, obj, name, cat):
        fn = getattr(obj, name)
        shares = self

        def timed(*a, **kw):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            shares.stack.append(0.0)
            try:
                return fn(*a, **kw)
            finally:
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                inner = shares.stack.pop()
                shares.t[cat] = shares.t.get(cat, 0.0) + dt - inner
                if shares.stack:
                    shares.stack[-1] += dt

        setattr(obj, name, timed)


def write_tree(root, frames, size):
    from PIL import Image
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:size, 0:size]
    for ci, name in enumerate(("bg_clip", "obj1_clip", "obj2_clip")):
        d = os.path.join(root, "demo", name, name)
        os.makedirs(d)
        for i in range(frames):
            img = np.stack([(xx * (1 + ci) + 7 * i) % 256, (yy * 2 + 13 * i) % 256, ((xx + yy) // 2 + 31 * ci) % 256], -1).astype(np.uint8)
            img = (img // 2 + rng.integers(0, 128, img.shape, dtype=np.uint8))
            Image.fromarray(img).save(os.path.join(d, f"{i:05d}.png"))
    ef = os.path.join(root, "demo", "bg_clip", "edited_first_frame")
    os.makedirs(ef)
    Image.fromarray(rng.integers(0, 255, (size, size, 3), dtype=np.uint8)).save(os.path.join(ef, "00000.png"))
    for mi, mname in enumerate(("m1", "m2")):
        md = os.path.join(root, "demo", "bg_clip", mname)
        os.makedirs(md)
        for i in range(frames):
            m = np.zeros((size, size), np.uint8)
            y0, x0 = size // 8 + 4 * i + mi * size // 3, size // 4 + 2 * i
            m[y0:y0 + size // 3, x0:x0 + size // 3] = 255
            Image.fromarray(m).save(os.path.join(md, f"{i:05d}.png"))


def run(frames=16, size=512, steps=50, keep=False):
    os.environ["MVOC_SYNTHETIC_VAE"] = os.environ["MVOC_SYNTHETIC_CLIP"] = "1"
    sys.path[:0] = [os.path.join(REPO, "i2vgen-xl"), REPO]
    for m in ("utils", "pnp_utils", "inverse", "composite", "pipelines", "pipelines.pipeline_i2vgen_xl"):
        sys.modules.pop(m, None)
    import composite
    import inverse
    import utils as ref_utils
    from mvoc_amd.config import OmegaConf
    from mvoc_amd import latent_cache, pipeline as pl
    root = tempfile.mkdtemp(prefix="mvoc_demo_")
    t_tree = time.perf_counter()
    write_tree(root, frames, size)
    t_tree = time.perf_counter() - t_tree
    dev = torch.device("cuda:0")
    sh = Shares()
    # file IO: frame / mask decoding + resize, latent files, result files
    for mod, names in ((inverse, ("load_video_frames", "export_to_gif")), (composite, ("_frames", "load_image", "export_to_gif")),
                       (ref_utils, ("load_ddim_latents_at_t",))):
        for n in names:
            if hasattr(mod, n):
                sh.wrap(mod, n, "file_io")
    sh.wrap(latent_cache.LatentCache, "flush", "file_io")
    sh.wrap(pl.SyntheticConditioner, "encode_video", "vae")
    sh.wrap(pl.SyntheticConditioner, "decode", "vae")
    sh.wrap(pl.SyntheticConditioner, "image_latents", "vae")
    sh.wrap(pl.SyntheticConditioner, "encode_images", "clip")
    sh.wrap(pl.SyntheticConditioner, "encode_image", "clip")
    sh.wrap(pl.SyntheticConditioner, "encode_prompt", "clip")
    sh.wrap(pl.I2VGenXLPipeline, "synthetic", "model_build")
    sh.wrap(inverse, "build_pipeline", "model_build")

    it = OmegaConf.load(os.path.join(REPO, "tests", "data", "inversion_template.yaml"))
    it.data_dir = root
    it.image_size = [size, size]
    it.n_frames = frames
    it.inverse_config.n_steps = steps
    entries = [{"active": True, "force_recompute_latents": True, "video_name": n, "video_dir": os.path.join(root, "demo", n),
                "recon_config": {"enable_recon": False}} for n in ("bg_clip", "obj1_clip", "obj2_clip")]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inverse.main(it, entries, dev, synthetic=True)
    torch.cuda.synchronize()
    t_inv = time.perf_counter() - t0
    inv_shares, sh.t = dict(sh.t), {}

    ct = OmegaConf.load(os.path.join(REPO, "tests", "data", "composite_template.yaml"))
    ct.data_dir = root
    ct.image_size = [size, size]
    ct.n_frames = frames
    ct.n_steps = steps
    lat = "inversions/i2vgen-xl/{}/ddim_latents"
    centry = {"active": True, "task_name": "demo", "video_name": "bg_clip", "editing_prompt": "windsurf,sailboat,sky,ocean",
              "editing_negative_prompt": "Chaotic, chaotic colors", "edited_video_name": "out",
              "edited_first_frame_path": "demo/bg_clip/edited_first_frame/00000.png",
              "ddim_init_latents_t_idx": 0, "pnp_f_t": 0.1, "pnp_spatial_attn_t": 1.0, "pnp_temp_attn_t": 1.0, "random_noise_ratio": 0.0,
              "fusion_step": [0, 1], "obj_mask_path": ["demo/bg_clip/m1", "demo/bg_clip/m2"], "obj_width_height": [[size, size], [size, size]],
              "obj_ddim_latents_path": [lat.format("obj1_clip"), lat.format("obj2_clip")], "bg_ddim_latents_path": lat.format("bg_clip"),
              "edited_contorl_frame_path_main": "demo/bg_clip/bg_clip", "edited_contorl_frame_path_background": "demo/bg_clip/bg_clip",
              "edited_contorl_frame_path": ["demo/obj1_clip/obj1_clip", "demo/obj2_clip/obj2_clip"]}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    composite.main(ct, [centry], dev, synthetic=True)
    torch.cuda.synchronize()
    t_comp = time.perf_counter() - t0
    comp_shares = dict(sh.t)
    out_root = os.path.join(root, "Results", "demo", "i2vgen-xl", "bg_clip", "out")
    files = sorted(os.listdir(os.path.join(out_root, os.listdir(out_root)[0])))
    n_lat = sum(len(os.listdir(os.path.join(root, lat.format(n)))) for n in ("bg_clip", "obj1_clip", "obj2_clip"))
    if not keep:
        shutil.rmtree(root, ignore_errors=True)

    def stage(total, shares):
        other = sum(shares.values())
        d = {k: round(v, 3) for k, v in sorted(shares.items())}
        d["denoising_loops_and_host"] = round(total - other, 3)
        return d

    build = inv_shares.get("model_build", 0.0) + comp_shares.get("model_build", 0.0)
    return {
        "job": f"3 x {steps}-step DDIM inversion ({frames} frames, {size}x{size}) + 1 x {steps}-step PnP composition (bg + 2 objects), "
               f"drop-in drivers i2vgen-xl/inverse.py + composite.py, seeded synthetic UNet / VAE / CLIP weights",
        "wall_s": round(t_inv + t_comp, 3),
        "wall_s_without_model_build": round(t_inv + t_comp - build, 3),
        "inverse_py_s": round(t_inv, 3), "composite_py_s": round(t_comp, 3),
        "inverse_py_shares_s": stage(t_inv, inv_shares), "composite_py_shares_s": stage(t_comp, comp_shares),
        "files": {"ddim_latents_pt": n_lat, "result_files": files[:3] + (["..."] if len(files) > 3 else []), "n_result_files": len(files)},
        "synthetic_input_tree_s": round(t_tree, 3),
    }


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    print(json.dumps(run(a.frames, a.size, a.steps, a.keep)), flush=True)
