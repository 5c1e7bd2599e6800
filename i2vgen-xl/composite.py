#!/usr/bin/env python3
"""Stage 2 of MVOC on MI355X: PnP composition sampling for every active entry of a group config.  Same CLI, config
keys, hook registration order and output naming as the reference's ``i2vgen-xl/composite.py``.

    PYTHONPATH=.. python composite.py --template_config configs/group_composite/template.yaml \
                                      --configs_json configs/group_composite/group_config.json [--synthetic] [--shard i/n]
"""
import argparse
import json
import logging
import os
from functools import partial
from pathlib import Path

import torch
from PIL import Image

from common import PRETRAINED_MODEL_PATH
from mvoc_amd.config import OmegaConf
from mvoc_amd.launch import my_entries, pick_device
from mvoc_amd.schedulers import DDIMScheduler
from pipelines.pipeline_i2vgen_xl import I2VGenXLPipeline, I2VGenXLUnetExtension
from pnp_utils import (modify_diffuser_attention_forward, register_out_conv_injection, register_resnet_injection,
                       register_spatial_attention_pnp, register_temp_attention_pnp, register_temp_conv_injection)
from utils import export_to_gif, load_image, seed_everything

logger = logging.getLogger(__name__)


def init_pnp(pipe, scheduler, config):
    """injection schedules are prefixes of the FULL timestep list (``composite.py:38-60``), registered in the
    reference's order: forward patch, temporal attn, spatial attn, temporal conv, conv_out, resnet"""
    n = config.n_steps
    conv_t, spa_t, tmp_t = int(n * config.pnp_f_t), int(n * config.pnp_spatial_attn_t), int(n * config.pnp_temp_attn_t)
    conv_ts = scheduler.timesteps[:conv_t] if conv_t >= 0 else []
    spa_ts = scheduler.timesteps[:spa_t] if spa_t >= 0 else []
    tmp_ts = scheduler.timesteps[:tmp_t] if tmp_t >= 0 else []
    modify_diffuser_attention_forward(pipe.unet)
    register_temp_attention_pnp(pipe, tmp_ts, config.inject_background)
    register_spatial_attention_pnp(pipe, spa_ts, config.inject_background)
    register_temp_conv_injection(pipe, conv_ts)
    register_out_conv_injection(pipe, conv_ts)
    register_resnet_injection(pipe, conv_ts)
    logger.debug(f"conv/spatial/temporal injection steps: {conv_t}/{spa_t}/{tmp_t}")


def _frames(folder, n, size):
    out = []
    for i in range(n):
        out.append(load_image(os.path.join(folder, f"{i:0>5d}.png")).resize(tuple(size), resample=Image.Resampling.LANCZOS))
    return out


def output_suffix(config):
    return ("ddim_init_latents_t_idx_" + str(config.ddim_init_latents_t_idx) + "_nsteps_" + str(config.n_steps) + "_cfg_"
            + str(config.cfg) + "_pnpf" + str(config.pnp_f_t) + "_pnps" + str(config.pnp_spatial_attn_t) + "_pnpt"
            + str(config.pnp_temp_attn_t) + "_ratio" + str(config.random_noise_ratio) + "noise_fusion_step"
            + f"{config.fusion_step[0]}-{config.fusion_step[1]}")


def main(template_config, configs_list, device, synthetic=False):
    from inverse import build_pipeline
    pipe = build_pipeline(device, synthetic)
    ddim_scheduler = DDIMScheduler.from_pretrained(PRETRAINED_MODEL_PATH, subfolder="scheduler")
    for entry in configs_list:
        if not entry["active"]:
            continue
        config = OmegaConf.merge(template_config, OmegaConf.create(entry))
        d = config.data_dir
        config.video_path = os.path.join(config.video_dir, config.video_name + ".mp4")
        config.video_frames_path = os.path.join(config.video_dir, config.video_name)
        config.edited_first_frame_path = os.path.join(d, config.edited_first_frame_path)
        config.obj_mask_path = [os.path.join(d, p) for p in config.obj_mask_path]
        config.obj_ddim_latents_path = [os.path.join(d, p) for p in config.obj_ddim_latents_path]
        config.bg_ddim_latents_path = os.path.join(d, config.bg_ddim_latents_path)
        config.edited_contorl_frame_path_main = os.path.join(d, config.edited_contorl_frame_path_main)
        config.edited_contorl_frame_path_background = os.path.join(d, config.edited_contorl_frame_path_background)
        config.edited_contorl_frame_path = [os.path.join(d, p) for p in config.edited_contorl_frame_path]
        logger.info(f"config: {OmegaConf.to_yaml(config)}")
        main_1st = load_image(config.edited_first_frame_path).resize(tuple(config.image_size), resample=Image.Resampling.LANCZOS)
        main_frames = _frames(config.edited_contorl_frame_path_main, config.n_frames, config.image_size)
        obj_frames = [_frames(p, config.n_frames, config.image_size) for p in config.edited_contorl_frame_path]
        bg_frames = _frames(config.edited_contorl_frame_path_background, config.n_frames, config.image_size)
        ddim_scheduler.set_timesteps(config.n_steps)
        init_pnp(pipe, ddim_scheduler, config)
        pipe.register_modules(scheduler=ddim_scheduler)
        pipe.unet.forward = partial(I2VGenXLUnetExtension.forward, pipe.unet)
        out_type = "pil"
        kw = dict(prompt=config.editing_prompt, main_first_image=main_1st, main_image_list=main_frames,
                  background_first_image=bg_frames[0], background_image_list=bg_frames,
                  objs_first_image=[f[0] for f in obj_frames], objs_image_list=obj_frames, height=config.image_size[1],
                  width=config.image_size[0], num_frames=config.n_frames, num_inference_steps=config.n_steps,
                  guidance_scale=config.cfg, negative_prompt=config.editing_negative_prompt, target_fps=config.target_fps,
                  generator=torch.Generator().manual_seed(config.seed), return_dict=True,
                  ddim_init_latents_t_idx=config.ddim_init_latents_t_idx, ddim_inv_prompt=config.ddim_inv_prompt,
                  obj_mask=config.obj_mask_path, obj_width_height=config.obj_width_height,
                  random_noise_ratio=config.random_noise_ratio, bg_inv_latents_path=config.bg_ddim_latents_path,
                  obj_ddim_latents_path=config.obj_ddim_latents_path,
                  obj_ddim_latents_idx_offset=config.obj_ddim_latents_idx_offset,
                  obj_random_noise_fusion=config.obj_random_noise_fusion, fusion_steps=config.fusion_step)
        output_dir = os.path.join(config.output_dir, output_suffix(config))
        os.makedirs(output_dir, exist_ok=True)
        try:
            video = pipe.sample_with_pnp_pipeline_with_edit_prompt_extraction_with_attn_injection(output_type=out_type, **kw).frames[0]
        except NotImplementedError as e:  # no VAE decoder on this path: keep the composed latents
            logger.warning(f"composition decoded to latents only ({e})")
            lat = pipe.sample_with_pnp_pipeline_with_edit_prompt_extraction_with_attn_injection(output_type="latent", **kw).frames
            torch.save(lat.cpu(), os.path.join(output_dir, "video_latents.pt"))
            continue
        video = [f.resize(tuple(config.image_size), resample=Image.LANCZOS) for f in video]
        export_to_gif(video, os.path.join(output_dir, "video.gif"))
        for i, f in enumerate(video):
            f.save(os.path.join(output_dir, f"video_{i:05d}.png"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--template_config", type=str, default="configs/group_composite/template.yaml")
    ap.add_argument("--configs_json", type=str, default="configs/group_composite/group_config.json")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--shard", type=str, default=None)
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
    main(template_config, my_entries(configs_list, args.shard), device, args.synthetic)
