"""ORACLE (test infrastructure, not product): the three denoising-loop bodies of MVOC's pipeline,
restated on plain tensors.

* inversion loop body      ``i2vgen-xl/pipelines/pipeline_i2vgen_xl.py:1940-2000``
* reconstruction loop body ``...:1167-1202``
* composition loop body    ``...:1636-1734`` (latent noise-fusion ``:1639-1665``, batch assembly ``:1675-1680``,
                           CFG on the two trailing chunks ``:1713-1717``, DDIM update ``:1723-1731``)

The inline glue arithmetic has no reference test; it is restated here op by op in the tensor dtype
(python-float factors stay fp32 "opmath" scalars as on the device path) and checked by known-answer
tests in ``tests/test_loops.py`` (e.g. ``random_noise_ratio=0``, ``fusion_step=[0,1]`` => step-0 latents
are exactly the mask-paste of the loaded latents).  Quirks kept: ``fusion_counter`` is never incremented
(``:1634,1649``), so every fusion step re-reads the first fusion timestep's object latents.
"""
import torch


def _pmul(scalar: float, x: torch.Tensor) -> torch.Tensor:
    """python-float x tensor on the device path: scalar held as fp32, one rounding to x.dtype."""
    return (x.float() * torch.tensor(scalar, dtype=torch.float32)).to(x.dtype)


def cfg_combine(uncond, cond, guidance_scale: float):
    """``noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond)`` (``:1188, 1717, 1970``)."""
    return uncond + _pmul(guidance_scale, cond - uncond)


def latent_fusion(latents, bg_latents, obj_latents, float_masks, mix_ratio: float, obj_random_noise_fusion=False):
    """``pipeline_i2vgen_xl.py:1644-1663``.  All tensors ``[1,4,F,h,w]``; masks float in latents.dtype."""
    latents = _pmul(mix_ratio, latents) + _pmul(1.0 - mix_ratio, bg_latents)
    for obj, m in zip(obj_latents, float_masks):
        inv_obj = obj * m
        background = latents * (1.0 - m)
        if obj_random_noise_fusion:
            fg = latents * m
            fusion = _pmul(mix_ratio, fg) + _pmul(1 - mix_ratio, inv_obj)
        else:
            fusion = inv_obj
        latents = background + fusion
    return latents


def scheduler_step_5d(scheduler, noise_pred, t, latents):
    """permute to ``[B*F,C,h,w]``, ``scheduler.step``, permute back (``:1723-1731``).  Elementwise, so
    the permutes do not change values."""
    b, c, f, h, w = latents.shape
    lat = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    npred = noise_pred.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    out = scheduler.step(npred, t, lat).prev_sample
    return out.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4)


def invert_loop(unet_fn, scheduler, latents, n_steps, guidance_scale=1.0):
    """``invert`` (``:1940-2003``).  ``unet_fn(latent_model_input, t) -> noise_pred``.
    Returns (dict t -> latent at noise level t, stacked ``[1,n,4,F,h,w]`` noisiest first)."""
    scheduler.set_timesteps(n_steps)
    do_cfg = guidance_scale > 1.0
    saved, seq = {}, []
    for t in scheduler.timesteps:
        inp = torch.cat([latents] * 2) if do_cfg else latents
        noise_pred = unet_fn(inp, t)
        if do_cfg:
            u, c = noise_pred.chunk(2)
            noise_pred = cfg_combine(u, c, guidance_scale)
        latents = scheduler_step_5d(scheduler, noise_pred, t, latents)
        saved[int(t)] = latents.clone()
        seq.append(latents.clone())
    return saved, torch.stack(list(reversed(seq)), 1)


def sample_loop(unet_fn, scheduler, latents, n_steps, guidance_scale=9.0, ddim_init_latents_t_idx=0):
    """``__call__`` loop (``:1140-1202``)."""
    scheduler.set_timesteps(n_steps)
    scheduler.timesteps = scheduler.timesteps[ddim_init_latents_t_idx:]
    do_cfg = guidance_scale > 1.0
    for t in scheduler.timesteps:
        inp = torch.cat([latents] * 2) if do_cfg else latents
        noise_pred = unet_fn(inp, t)
        if do_cfg:
            u, c = noise_pred.chunk(2)
            noise_pred = cfg_combine(u, c, guidance_scale)
        latents = scheduler_step_5d(scheduler, noise_pred, t, latents)
    return latents


def composition_loop(unet_fn, scheduler, latents, bg_latents_at, obj_latents_at, float_masks, n_steps,
                     guidance_scale=9.0, ddim_init_latents_t_idx=0, fusion_steps=(0, 1), random_noise_ratio=0.0,
                     obj_random_noise_fusion=False, obj_ddim_latents_idx_offset=None, on_step=None):
    """``sample_with_pnp_...`` loop (``:1552-1734``).

    ``bg_latents_at(t)`` / ``obj_latents_at(j, t)`` return ``[1,4,F,h,w]`` latents at noise level t (the
    ``ddim_latents_{t}.pt`` files).  ``unet_fn(x[5,4,F,h,w], t) -> [5,4,F,h,w]``; ``on_step(i, t)`` is the
    ``register_time_all`` hook point.  Returns final latents.  ``guidance_scale <= 1``: the CFG-off generalisation
    (not runnable in the reference): batch ``[bg, objs.., cond]``, no CFG combine."""
    n_obj = len(float_masks)
    do_cfg = guidance_scale > 1.0
    offs = obj_ddim_latents_idx_offset or [0] * n_obj
    scheduler.set_timesteps(n_steps)
    full_ts = scheduler.timesteps.clone()
    scheduler.timesteps = scheduler.timesteps[ddim_init_latents_t_idx:]
    fusion_ts = [[full_ts[offs[j]:][k] for k in range(*fusion_steps)] for j in range(n_obj)]
    fusion_counter = 0  # never incremented in the reference
    for i, t in enumerate(scheduler.timesteps):
        bg = bg_latents_at(int(t))
        if fusion_steps[0] <= i < fusion_steps[1]:
            objs = [obj_latents_at(j, int(fusion_ts[j][fusion_counter])) for j in range(n_obj)]
            latents = latent_fusion(latents, bg, objs, float_masks, random_noise_ratio, obj_random_noise_fusion)
        else:
            objs = [obj_latents_at(j, int(t)) for j in range(n_obj)]
        inp = torch.cat([bg] + objs + [latents] * (2 if do_cfg else 1))
        if on_step is not None:
            on_step(i, int(t))
        noise_pred = unet_fn(inp, t)
        chunks = noise_pred.chunk(n_obj + (3 if do_cfg else 2))
        noise_pred = cfg_combine(chunks[-2], chunks[-1], guidance_scale) if do_cfg else chunks[-1]
        latents = scheduler_step_5d(scheduler, noise_pred, t, latents)
    return latents
