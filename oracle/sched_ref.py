"""ORACLE (test infrastructure, not product): DDIM and inverse-DDIM schedulers as the
reference drives them (``scheduler.set_timesteps / step`` at
``i2vgen-xl/pipelines/pipeline_i2vgen_xl.py:1140,1196,1552,1728,1914,1979``;
``DDIMInverseScheduler/DDIMScheduler.from_pretrained`` at ``i2vgen-xl/inverse.py:123-131``).

The algorithm lives in the un-vendored dependency diffusers==0.27.2 (``environment.yaml:58``); the
scheduler config is the one shipped with ``ali-vilab/i2vgen-xl`` (recalled): 1000 train steps,
``squaredcos_cap_v2`` betas, ``v_prediction``, ``rescale_betas_zero_snr``, ``timestep_spacing='leading'``,
``steps_offset=1``, ``set_alpha_to_one``, no clipping, eta=0.  **Parity unpinned**: the reference holds no
test or golden vector for it; the only anchor is the comment ``0 for 981, 3 for 921, 9 for 801, 20 for 581``
at ``i2vgen-xl/configs/group_composite/template.yaml:43``, which ``tests/test_host_cpu.py`` checks (the timestep list), beside the closed-form properties tested there.

dtype behaviour restated: sample / model_output keep their dtype (fp16 on the GPU path); the alpha
factors are fp32 scalars, so every product and sum is rounded to the tensor dtype once per op.
"""
import math

import numpy as np
import torch


def _sqrt32(t: torch.Tensor) -> torch.Tensor:
    """correctly rounded fp32 square root (what CUDA's sqrtf / ``tensor ** 0.5`` gives on the reference's platform).
    ``torch.sqrt`` on CPU is NOT bit-reproducible across hosts (the vectorised kernel differs by an ulp between the build
    container and the GPU box), which moved ``alphas_cumprod[1]`` by 2 ulp; numpy's is IEEE."""
    return torch.from_numpy(np.asarray(np.sqrt(t.detach().to(torch.float32).cpu().numpy())))


def make_alphas_cumprod(num_train_timesteps=1000, rescale_zero_snr=True) -> torch.Tensor:
    def alpha_bar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2

    betas = []
    for i in range(num_train_timesteps):
        t1, t2 = i / num_train_timesteps, (i + 1) / num_train_timesteps
        betas.append(min(1 - alpha_bar(t2) / alpha_bar(t1), 0.999))
    betas = torch.tensor(betas, dtype=torch.float32)
    if rescale_zero_snr:
        ab_sqrt = _sqrt32(torch.cumprod(1.0 - betas, dim=0))
        s0, sT = ab_sqrt[0].clone(), ab_sqrt[-1].clone()
        ab_sqrt = (ab_sqrt - sT) * (s0 / (s0 - sT))
        ab = ab_sqrt ** 2
        alphas = torch.cat([ab[0:1], ab[1:] / ab[:-1]])
        betas = 1 - alphas
    return torch.cumprod(1.0 - betas, dim=0)


def _smul(scalar_f32: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """fp32 0-dim scalar x tensor as the GPU eager kernel does it: the scalar stays fp32 (opmath), the
    product is formed in fp32 and rounded to x.dtype once.  (CPU eager would first round the scalar to
    x.dtype, which is not what the reference's device path computes.)"""
    return (x.float() * scalar_f32.float()).to(x.dtype)


class _Out:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class DDIMSchedulerRef:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, steps_offset=1):
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.alphas_cumprod = make_alphas_cumprod(num_train_timesteps)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, t=None):
        return sample

    def step(self, model_output, timestep, sample, eta=0.0, **_):
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        x0 = _smul(_sqrt32(a_t), sample) - _smul(_sqrt32(b_t), model_output)
        eps = _smul(_sqrt32(a_t), model_output) + _smul(_sqrt32(b_t), sample)
        direction = _smul(_sqrt32((1 - a_prev)), eps)
        return _Out(_smul(_sqrt32(a_prev), x0) + direction)


class DDIMInverseSchedulerRef(DDIMSchedulerRef):
    def __init__(self, num_train_timesteps=1000, steps_offset=1):
        super().__init__(num_train_timesteps, steps_offset)
        self.initial_alpha_cumprod = torch.tensor(1.0)

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        ts = (np.arange(0, n) * ratio).round().copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, timestep, sample, eta=0.0, **_):
        nxt = int(timestep)  # noise level the sample is moved TO
        cur = min(nxt - self.num_train_timesteps // self.num_inference_steps, self.num_train_timesteps - 1)
        a_cur = self.alphas_cumprod[cur] if cur >= 0 else self.initial_alpha_cumprod
        a_nxt = self.alphas_cumprod[nxt]
        b_cur = 1 - a_cur
        x0 = _smul(_sqrt32(a_cur), sample) - _smul(_sqrt32(b_cur), model_output)
        eps = _smul(_sqrt32(a_cur), model_output) + _smul(_sqrt32(b_cur), sample)
        direction = _smul(_sqrt32((1 - a_nxt)), eps)
        return _Out(_smul(_sqrt32(a_nxt), x0) + direction)
