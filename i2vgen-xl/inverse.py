#!/usr/bin/env python3
"""Stage 1 of MVOC on MI355X: DDIM inversion of every active entry of a group config.  Same CLI, config keys,
output files (``{output_dir}/ddim_latents_{t}.pt``) and call sequence as the reference's ``i2vgen-xl/inverse.py``.

    PYTHONPATH=.. python inverse.py --template_config configs/group_inversion/template.yaml \
                                    --configs_json configs/group_inversion/group_config.json [--synthetic] [--shard i/n]

Entries are independent (reference loop ``inverse.py:136``); ``--shard i/n`` (or RANK/WORLD_SIZE from torchrun)
gives every n-th active entry to this process, one process per GPU, no collectives.
"""
import argparse
import json
import logging
import os
from pathlib import Path

import torch
from PIL import Image

from common import PRETRAINED_MODEL_PATH
from mvoc_amd.config import OmegaConf
from mvoc_amd.launch import my_entries, pick_device
from mvoc_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
from pipelines.pipeline_i2vgen_xl import I2VGenXLPipeline
from utils import export_to_gif, load_ddim_latents_at_t, load_video_frames, seed_everything

logger = logging.getLogger(__name__)


def ddim_inversion(config, first_frame, frame_list, pipe, inverse_scheduler, g):
    pipe.scheduler = inverse_scheduler
    video_latents_at_0 = pipe.encode_vae_video(frame_list, device=pipe._execution_device, height=config.image_size[1],
                                               width=config.image_size[0])
    ddim_latents = pipe.invert(prompt=config.prompt, image=first_frame, height=config.image_size[1], width=config.image_size[0],
                               num_frames=config.n_frames, num_inference_steps=config.n_steps, guidance_scale=config.cfg,
                               negative_prompt=config.negative_prompt, target_fps=config.target_fps, latents=video_latents_at_0,
                               generator=g, return_dict=False, output_dir=config.output_dir)
    return ddim_latents[0]


def ddim_sampling(config, first_frame, ddim_latents_at_T, pipe, ddim_scheduler, ddim_init_latents_t_idx, g, output_type="pil"):
    pipe.scheduler = ddim_scheduler
    return pipe(prompt=config.prompt, image=first_frame, height=config.image_size[1], width=config.image_size[0],
                num_frames=config.n_frames, num_inference_steps=config.n_steps, guidance_scale=config.cfg,
                negative_prompt=config.negative_prompt, target_fps=config.target_fps, latents=ddim_latents_at_T, generator=g,
                return_dict=True, ddim_init_latents_t_idx=ddim_init_latents_t_idx, output_type=output_type,
                decode_chunk_size=1).frames[0]


def build_pipeline(device, synthetic):
    from mvoc_amd.vae import attach_vae
    if synthetic or not os.path.isdir(os.path.join(PRETRAINED_MODEL_PATH, "unet")):
        if not synthetic:
            logger.warning(f"{PRETRAINED_MODEL_PATH}/unet not found: using seeded synthetic UNet weights")
        pipe = I2VGenXLPipeline.synthetic(device=device)
    else:
        pipe = I2VGenXLPipeline.from_pretrained(PRETRAINED_MODEL_PATH, torch_dtype=torch.float16, variant="fp16", device=device)
    # VAE encode / decode either side of the loops (frames in, frames out): the checkpoint's vae/ when present; with
    # MVOC_SYNTHETIC_VAE=1 seeded weights of the same architecture; otherwise the drivers keep latents
    attach_vae(pipe, PRETRAINED_MODEL_PATH, synthetic=os.environ.get("MVOC_SYNTHETIC_VAE") == "1")
    # CLIP towers for the prompt / image embeddings (conditioning prep): the checkpoint's image_encoder/ + text_encoder/ (+
    # tokenizer/) when present; MVOC_SYNTHETIC_CLIP=1: seeded weights of the same architecture (the vision tower then runs,
    # prompts stay on the seeded stand-in because no tokenizer exists without the checkpoint)
    from mvoc_amd.clip import attach_clip
    attach_clip(pipe, PRETRAINED_MODEL_PATH, synthetic=os.environ.get("MVOC_SYNTHETIC_CLIP") == "1")
    return pipe


def batched_inversions(pipe, inverse_scheduler, template_config, configs_list, batch, concurrent=False, hint=1):
    """--batch_entries N: invert up to N pending clips of identical shape / step count in one batched loop
    (``I2VGenXLPipeline.invert_many``) before the per-entry pass below, which then finds their latents on disk.
    --concurrent_entries N (``concurrent``): the same grouping, but every clip keeps its own batch-1 loop and the N loops run at
    the same time on N HIP streams (``I2VGenXLPipeline.invert_concurrent``).  ``hint`` is passed on as its ``concurrency_hint``:
    1 = exactly the launches of the one-by-one pass, files bit-identical to it; N > 1 (the driver's default: --concurrent_entries,
    the SAME value for every group whatever its size, so a clip's files do not depend on how many others were pending) = the
    under-filled GEMMs keep K in one piece and sum in another order: latents within rel-L2 5e-5 per step of the one-by-one
    pass (fp16 rounding noise), not bit-identical."""
    pending, done = [], set()  # done: output directories this pass produces (the per-entry pass must not invert them again,
    for entry in configs_list:  #       also not under force_recompute_latents)
        if not entry["active"]:
            continue
        config = OmegaConf.merge(template_config, OmegaConf.create(entry))
        inv = config.inverse_config
        if (os.path.exists(config.output_dir) and not config.get("force_recompute_latents", False)) or inv.cfg > 1:
            continue
        config.video_frames_path = os.path.join(config.video_dir, config.video_name)
        _, frame_list = load_video_frames(config.video_frames_path, config.n_frames, config.image_size)
        first_frame = frame_list[0]
        if inv.inverse_static_video:
            frame_list = [frame_list[0]] * config.n_frames
        if inv.null_image_inversion:
            first_frame = Image.new("RGB", (config.image_size[0], config.image_size[1]), (0, 0, 0))
        key = (tuple(inv.image_size), inv.n_frames, inv.n_steps, inv.target_fps, str(inv.negative_prompt))
        pending.append((key, inv, first_frame, frame_list))
        done.add(os.path.abspath(config.output_dir))
    pipe.scheduler = inverse_scheduler
    while pending:
        key = pending[0][0]
        group = [p for p in pending if p[0] == key][:batch]
        pending = [p for p in pending if all(p is not g_ for g_ in group)]
        inv0 = group[0][1]
        lat = [pipe.encode_vae_video(fl, device=pipe._execution_device, height=inv0.image_size[1], width=inv0.image_size[0])
               for _, _, _, fl in group]
        logger.info(f"{'concurrent' if concurrent else 'batched'} inversion of {len(group)} clips: {[g_[1].output_dir for g_ in group]}")
        run = pipe.invert_concurrent if concurrent else pipe.invert_many
        run([g_[1].prompt for g_ in group], [g_[2] for g_ in group], lat, [g_[1].output_dir for g_ in group],
            height=inv0.image_size[1], width=inv0.image_size[0], target_fps=inv0.target_fps, num_frames=inv0.n_frames,
            num_inference_steps=inv0.n_steps, guidance_scale=inv0.cfg, negative_prompt=inv0.negative_prompt,
            **({"concurrency_hint": int(hint)} if concurrent else {}))
    return done


def main(template_config, configs_list, device, synthetic=False, frame_shard=None, batch_entries=1, concurrent_entries=3,
         concurrency_hint=-1):
    pipe = build_pipeline(device, synthetic)
    if frame_shard is not None:
        pipe.enable_frame_shard(frame_shard)
    g = torch.Generator().manual_seed(template_config.seed)
    inverse_scheduler = DDIMInverseScheduler.from_pretrained(PRETRAINED_MODEL_PATH, subfolder="scheduler")
    ddim_scheduler = DDIMScheduler.from_pretrained(PRETRAINED_MODEL_PATH, subfolder="scheduler")
    inverted = set()
    if batch_entries > 1:
        inverted = batched_inversions(pipe, inverse_scheduler, template_config, configs_list, batch_entries)
    elif concurrent_entries > 1 and frame_shard is None:
        inverted = batched_inversions(pipe, inverse_scheduler, template_config, configs_list, concurrent_entries, concurrent=True,
                                      hint=concurrent_entries if concurrency_hint < 1 else concurrency_hint)
    for entry in configs_list:
        if not entry["active"]:
            logger.info(f"Skipping config_entry: {entry}")
            continue
        config = OmegaConf.merge(template_config, OmegaConf.create(entry))
        config.video_path = os.path.join(config.video_dir, config.video_name + ".mp4")
        config.video_frames_path = os.path.join(config.video_dir, config.video_name)
        logger.info(f"config: {OmegaConf.to_yaml(config)}")
        _, frame_list = load_video_frames(config.video_frames_path, config.n_frames, config.image_size)
        first_frame = frame_list[0]
        if config.inverse_config.inverse_static_video:
            frame_list = [frame_list[0]] * config.n_frames
        if config.inverse_config.null_image_inversion:
            first_frame = Image.new("RGB", (config.image_size[0], config.image_size[1]), (0, 0, 0))
        if os.path.abspath(config.output_dir) in inverted:
            logger.info(f"### {config.output_dir}: inverted by the grouped pass above")
        elif os.path.exists(config.output_dir) and not config.get("force_recompute_latents", False):
            logger.info(f"### Skipping !!! {config.output_dir} already exists. ")
        else:
            ddim_inversion(config.inverse_config, first_frame, frame_list, pipe, inverse_scheduler, g)
        recon = config.recon_config
        if recon.enable_recon:
            ddim_scheduler.set_timesteps(recon.n_steps)
            t0 = ddim_scheduler.timesteps[recon.ddim_init_latents_t_idx]
            # the start latent comes from the in-HBM latent cache when this process produced it (every rank of a frame
            # shard holds it; only rank 0's writer thread touches the files), from disk otherwise
            lat = pipe.latent_cache.get(recon.ddim_latents_path, int(t0))
            is_writer = frame_shard is None or frame_shard.rank == 0  # all ranks compute; one writes the outputs
            try:
                video = ddim_sampling(recon, first_frame, lat, pipe, ddim_scheduler, recon.ddim_init_latents_t_idx, g)
                if is_writer:
                    os.makedirs(config.output_dir, exist_ok=True)
                    export_to_gif(video, os.path.join(config.output_dir, "ddim_reconstruction.gif"))
            except NotImplementedError as e:  # no VAE decoder on this path: keep the reconstructed latents instead
                logger.warning(f"reconstruction decoded to latents only ({e})")
                video = ddim_sampling(recon, first_frame, lat, pipe, ddim_scheduler, recon.ddim_init_latents_t_idx, g, "latent")
                if is_writer:
                    os.makedirs(config.output_dir, exist_ok=True)
                    torch.save(video.cpu(), os.path.join(config.output_dir, "ddim_reconstruction_latents.pt"))
        if frame_shard is not None:
            frame_shard.barrier()  # rank 0's files of this entry are complete before any rank moves on

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--template_config", type=str, default="configs/group_inversion/template.yaml")
    ap.add_argument("--configs_json", type=str, default="configs/group_inversion/group_config.json")
    ap.add_argument("--synthetic", action="store_true", help="seeded synthetic UNet weights + conditioning (no checkpoint)")
    ap.add_argument("--shard", type=str, default=None, help="i/n: process entry k when k %% n == i")
    ap.add_argument("--frame_shard", action="store_true",
                    help="under torchrun: every rank works on the SAME entries, each clip's frame axis sharded over the ranks "
                         "(RCCL; for clips too long for one GPU's latency budget) instead of one entry per rank")
    ap.add_argument("--batch_entries", type=int, default=1,
                    help="invert up to N clips of identical shape in one batched UNet loop (one GPU, cfg 1.0 inversions)")
    ap.add_argument("--concurrent_entries", type=int, default=3,
                    help="invert up to N pending clips of identical shape at the same time, each in its own batch-1 loop on its own "
                         "HIP stream (1 = one by one)")
    ap.add_argument("--concurrency_hint", type=int, default=-1,
                    help="GEMM scheduling hint of the concurrent loops (mvoc_gemm_desc.concurrency), the same for every group. "
                         "1: every clip runs exactly the launches of the one-by-one pass, ddim_latents_{t}.pt bit-identical to it. "
                         "-1 (default): = --concurrent_entries: under-filled GEMMs keep K in one piece (6 %% faster); the files "
                         "then differ from the one-by-one pass by fp16 rounding noise (rel-L2 5e-5 per step), whatever the "
                         "number of pending clips")
    args = ap.parse_args()
    template_config = OmegaConf.load(args.template_config)
    logging.basicConfig(level=logging.DEBUG if template_config.debug else logging.INFO,
                        format="%(asctime)s - %(levelname)s - [%(funcName)s] - %(message)s")
    assert Path(args.configs_json).exists()
    with open(args.configs_json) as f:
        configs_list = json.load(f)
    device = pick_device(template_config.device, args.shard)
    torch.set_grad_enabled(False)
    seed_everything(template_config.seed)
    if args.frame_shard and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        from mvoc_amd.frame_shard import FrameShard
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", device_id=device)
        main(template_config, configs_list, device, args.synthetic, frame_shard=FrameShard())
        dist.barrier()
        dist.destroy_process_group()
    else:
        main(template_config, my_entries(configs_list, args.shard), device, args.synthetic, batch_entries=args.batch_entries,
             concurrent_entries=args.concurrent_entries, concurrency_hint=args.concurrency_hint)
