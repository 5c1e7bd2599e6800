"""DDIM / inverse-DDIM schedulers with the diffusers protocol the reference drives
(``set_timesteps``, ``.timesteps``, ``scale_model_input``, ``step(...).prev_sample``, ``.order``,
``.init_noise_sigma``, deep-copyable -- reference ``i2vgen-xl/pipelines/pipeline_i2vgen_xl.py:1140-1141, 1196,
1552-1554, 1728, 1914, 1979``; construction at ``i2vgen-xl/inverse.py:123-131``, ``composite.py:82-85``).

Config = the ``scheduler_config.json`` shipped with ``ali-vilab/i2vgen-xl``: 1000 train steps,
``squaredcos_cap_v2`` betas rescaled to zero terminal SNR, ``v_prediction``, ``timestep_spacing='leading'``,
``steps_offset=1``, ``set_alpha_to_one``, eta 0.

The update itself is one HIP kernel (``mvoc_ddim_step_f16``) that also folds classifier-free guidance and
reproduces the reference's fp16 op-by-op rounding bit for bit.  Its five fp32 coefficients per step live in a
device table built at ``set_timesteps`` so that a captured hipGraph can be replayed for every step without
host->device traffic.
"""
import math

import numpy as np
import torch

from . import ops


def _sqrt32(t):
    """correctly rounded fp32 square root: ``torch.sqrt`` on CPU differs by an ulp between hosts (vectorised kernel), which
    made the schedule tables host-dependent; numpy's is IEEE, like CUDA's sqrtf on the reference's platform"""
    return torch.from_numpy(np.asarray(np.sqrt(torch.as_tensor(t, dtype=torch.float32).detach().cpu().numpy())))


def _alphas_cumprod(n=1000):
    def bar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2

    betas = torch.tensor([min(1 - bar((i + 1) / n) / bar(i / n), 0.999) for i in range(n)], dtype=torch.float32)
    root = _sqrt32(torch.cumprod(1.0 - betas, dim=0))
    r0, rT = root[0].clone(), root[-1].clone()
    root = (root - rT) * (r0 / (r0 - rT))
    ab = root ** 2
    betas = 1 - torch.cat([ab[0:1], ab[1:] / ab[:-1]])  # the rescale returns betas; alphas = 1 - betas again
    return torch.cumprod(1.0 - betas, dim=0)


class _StepOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0
    _inverse = False

    def __init__(self, num_train_timesteps=1000, steps_offset=1):
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.alphas_cumprod = _alphas_cumprod(num_train_timesteps)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.initial_alpha_cumprod = torch.tensor(1.0)
        self.num_inference_steps = None
        self.timesteps = None
        self._device = None
        self._tables = {}

    @classmethod
    def from_pretrained(cls, path=None, subfolder=None, **_):
        """The reference builds both schedulers from the checkpoint's scheduler_config.json; that file only
        carries the constants hard-wired above, so nothing is read."""
        return cls()

    def __deepcopy__(self, memo):
        new = type(self)(self.num_train_timesteps, self.steps_offset)
        if self.num_inference_steps is not None:
            new.set_timesteps(self.num_inference_steps, self._device)
            new.timesteps = self.timesteps.clone()
        return new

    def set_timesteps(self, num_inference_steps, device=None):
        if self.num_train_timesteps % num_inference_steps and num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps must not exceed num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round().astype(np.int64)
        if not self._inverse:
            ts = ts[::-1].copy()
        # kept on the host: the loops use them for control flow (`t in schedule`, file names) only
        self.timesteps = torch.from_numpy(ts + self.steps_offset)
        self._device = device
        self._tables = {}

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _alphas_for(self, t):
        ratio = self.num_train_timesteps // self.num_inference_steps
        if self._inverse:
            cur = min(t - ratio, self.num_train_timesteps - 1)
            a_from = self.alphas_cumprod[cur] if cur >= 0 else self.initial_alpha_cumprod
            a_to = self.alphas_cumprod[t]
        else:
            prev = t - ratio
            a_from = self.alphas_cumprod[t]
            a_to = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return a_from, a_to

    def coefficients(self, t, guidance_scale=1.0):
        a_from, a_to = self._alphas_for(int(t))
        return [float(_sqrt32(a_from)), float(_sqrt32(1 - a_from)), float(_sqrt32(a_to)), float(_sqrt32(1 - a_to)),
                float(np.float32(guidance_scale))]

    def coef_table(self, device, guidance_scale=1.0):
        """fp32 [n_all_timesteps, 5] on the device + {t: row}; rows follow a full set_timesteps() schedule"""
        key = (str(device), float(guidance_scale))
        if key not in self._tables:
            ratio = self.num_train_timesteps // self.num_inference_steps
            all_ts = [int(i * ratio + self.steps_offset) for i in range(self.num_inference_steps)]
            rows = torch.tensor([self.coefficients(t, guidance_scale) for t in all_ts], dtype=torch.float32)
            self._tables[key] = (rows.to(device), {t: i for i, t in enumerate(all_ts)})
        return self._tables[key]

    def step_fused(self, sample, v_cond, timestep, v_uncond=None, guidance_scale=1.0, out=None):
        table, index = self.coef_table(sample.device, guidance_scale)
        return ops.ddim_step(sample, v_cond, table[index[int(timestep)]], v_uncond=v_uncond, out=out)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        if eta != 0.0:
            raise NotImplementedError("eta != 0 is never used by the reference (prepare_extra_step_kwargs default)")
        prev = self.step_fused(sample.contiguous(), model_output.contiguous(), timestep)
        return _StepOutput(prev) if return_dict else (prev,)


class DDIMInverseScheduler(DDIMScheduler):
    _inverse = True
