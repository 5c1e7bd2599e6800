"""VAE encode / decode either side of the denoising loops on MI355X.

Reference: diffusers ``AutoencoderKL`` as the pipeline uses it -- ``encode_vae_video`` (16 single-frame encodes,
``i2vgen-xl/pipelines/pipeline_i2vgen_xl.py:893-920``), ``prepare_image_latents`` (``:860-890``), ``decode_latents``
(``:771-791``, ``decode_chunk_size=1``), image pre/post-processing (``VaeImageProcessor`` ``:443, 908, 1208``;
``_center_crop_wide`` ``:2054-2076``; ``tensor2vid`` ``:82-100``).

Same design as the UNet engine: plain Python nodes holding packed fp16 weights under diffusers' state_dict key names, every
computation a libmvoc_hip kernel on channels-last rows ``[n*h*w, C]`` -- the 3x3 convs are the implicit GEMM (the encoder's
``Downsample2D(padding=0)`` = bottom/right zero row folded into the gather, ``pad_mode=1``; ``Upsample2D`` = nearest index in
the gather), GroupNorm(+SiLU) the two-pass kernels, the mid block's single-head attention (head_dim = 512, T = h*w tokens)
three GEMMs around a row-softmax kernel: S = (q/sqrt(C)) k^T, P = softmax(S), O = P v + b_v (P's rows sum to one, so the
value bias moves behind the product) with v^T produced directly as W_v x^T.  Frames are batched (the reference decodes one
frame at a time only to save memory: per-frame results are identical).
"""
import math
import os

import numpy as np
import torch

from . import ops
from ._ffi import check, lib
from .unet import H16, Linear, _pad_rows, pack_conv3x3, pack_conv3x3_small


class VaeConfig:
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, scaling_factor=0.18215):
        self.in_channels, self.out_channels = in_channels, out_channels
        self.block_out_channels = tuple(block_out_channels)
        self.layers_per_block, self.latent_channels = layers_per_block, latent_channels
        self.norm_num_groups, self.scaling_factor = norm_num_groups, scaling_factor

    @staticmethod
    def from_any(c):
        if isinstance(c, VaeConfig):
            return c
        keys = ("in_channels", "out_channels", "block_out_channels", "layers_per_block", "latent_channels", "norm_num_groups",
                "scaling_factor")
        d = c if isinstance(c, dict) else c.__dict__
        return VaeConfig(**{k: d[k] for k in keys if k in d})


def param_shapes(cfg):
    """diffusers AutoencoderKL state_dict: key -> shape"""
    boc, lc, L = cfg.block_out_channels, cfg.latent_channels, cfg.layers_per_block
    sh = {}

    def conv(k, co, ci, ks):
        sh[k + ".weight"], sh[k + ".bias"] = (co, ci, ks, ks), (co,)

    def norm(k, c):
        sh[k + ".weight"], sh[k + ".bias"] = (c,), (c,)

    def lin(k, co, ci):
        sh[k + ".weight"], sh[k + ".bias"] = (co, ci), (co,)

    def resnet(k, ci, co):
        norm(k + ".norm1", ci), conv(k + ".conv1", co, ci, 3), norm(k + ".norm2", co), conv(k + ".conv2", co, co, 3)
        if ci != co:
            conv(k + ".conv_shortcut", co, ci, 1)

    def mid(k, c):
        resnet(k + ".resnets.0", c, c), resnet(k + ".resnets.1", c, c)
        a = k + ".attentions.0"
        norm(a + ".group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(a + "." + n, c, c)

    conv("encoder.conv_in", boc[0], cfg.in_channels, 3)
    ci = boc[0]
    for i, c in enumerate(boc):
        for j in range(L):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", ci if j == 0 else c, c)
        if i != len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3)
        ci = c
    mid("encoder.mid_block", boc[-1])
    norm("encoder.conv_norm_out", boc[-1]), conv("encoder.conv_out", 2 * lc, boc[-1], 3)
    conv("quant_conv", 2 * lc, 2 * lc, 1), conv("post_quant_conv", lc, lc, 1)
    rev = list(reversed(boc))
    conv("decoder.conv_in", rev[0], lc, 3)
    mid("decoder.mid_block", rev[0])
    ci = rev[0]
    for i, c in enumerate(rev):
        for j in range(L + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", ci if j == 0 else c, c)
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
        ci = c
    norm("decoder.conv_norm_out", rev[-1]), conv("decoder.conv_out", cfg.out_channels, rev[-1], 3)
    return sh


class _Conv3:
    def __init__(self, sd, k):
        w = sd[k + ".weight"].to(H16)
        self.cout, self.cin = w.shape[0], w.shape[1]
        self.w = pack_conv3x3(w)
        self.b = torch.zeros(self.w.shape[0], dtype=H16, device=w.device)
        self.b[:self.cout] = sd[k + ".bias"].to(H16)

    def __call__(self, x, n, h, w, **kw):
        return ops.conv3x3(x, self.w, self.b, nimg=n, h=h, wd=w, n_store=self.cout, **kw)


class _Resnet:
    def __init__(self, sd, k, groups):
        g = lambda s: sd[k + s].to(H16).contiguous()
        self.groups = groups
        self.norm1, self.norm2 = (g(".norm1.weight"), g(".norm1.bias")), (g(".norm2.weight"), g(".norm2.bias"))
        self.conv1, self.conv2 = _Conv3(sd, k + ".conv1"), _Conv3(sd, k + ".conv2")
        self.shortcut = None
        if k + ".conv_shortcut.weight" in sd:
            w = g(".conv_shortcut.weight")
            self.shortcut = Linear(w.reshape(w.shape[0], w.shape[1]), g(".conv_shortcut.bias"))

    def __call__(self, x, n, h, w):
        t = ops.groupnorm(x, *self.norm1, nsample=n, rows_per_sample=h * w, groups=self.groups, eps=1e-6, silu=True)
        t, _, _ = self.conv1(t, n, h, w)
        t = ops.groupnorm(t, *self.norm2, nsample=n, rows_per_sample=h * w, groups=self.groups, eps=1e-6, silu=True)
        sc = self.shortcut(x) if self.shortcut is not None else x
        out, _, _ = self.conv2(t, n, h, w, resid=sc)
        return out


class _Attention:
    def __init__(self, sd, k, groups):
        g = lambda s: sd[k + s].to(H16).contiguous()
        self.groups = groups
        self.norm = (g(".group_norm.weight"), g(".group_norm.bias"))
        c = g(".to_q.weight").shape[0]
        scale = 1.0 / math.sqrt(c)
        self.to_q = Linear((g(".to_q.weight").float() * scale).to(H16), (g(".to_q.bias").float() * scale).to(H16))
        self.to_k = Linear(g(".to_k.weight"), g(".to_k.bias"))
        self.wv = _pad_rows(g(".to_v.weight"))  # used as the A operand: v^T = W_v x^T
        self.bv = g(".to_v.bias")
        self.to_out = Linear(g(".to_out.0.weight"), g(".to_out.0.bias"))
        self.c = c

    def __call__(self, x, n, h, w):
        T, c = h * w, self.c
        if T % 32:
            raise RuntimeError(f"VAE attention: h*w = {T} tokens must be a multiple of 32 (image sides multiples of 64)")
        t = ops.groupnorm(x, *self.norm, nsample=n, rows_per_sample=T, groups=self.groups, eps=1e-6, silu=False)
        q, k = self.to_q(t), self.to_k(t)
        o = torch.empty((n * T, c), dtype=H16, device=x.device)
        for i in range(n):
            sl = slice(i * T, (i + 1) * T)
            s = ops.linear(q[sl], k[sl])                      # [T, T] scores (q carries 1/sqrt(C))
            ops.softmax_rows(s)
            vt = ops.linear(self.wv[:c], t[sl])               # [C, T] = W_v x^T
            ops.linear(s, vt, self.bv, out=o[sl])             # P v + b_v
        return self.to_out(o, resid=x)


class _Mid:
    def __init__(self, sd, k, groups):
        self.r0, self.r1 = _Resnet(sd, k + ".resnets.0", groups), _Resnet(sd, k + ".resnets.1", groups)
        self.attn = _Attention(sd, k + ".attentions.0", groups)

    def __call__(self, x, n, h, w):
        return self.r1(self.attn(self.r0(x, n, h, w), n, h, w), n, h, w)


class AutoencoderKL:
    """MI355X engine with diffusers' ``AutoencoderKL`` state_dict and the call surface the pipeline uses"""

    def __init__(self, config=None, device="cuda:0"):
        self.config = VaeConfig.from_any(config) if config is not None else VaeConfig()
        self.device = torch.device(device)
        self.dtype = H16
        self._loaded = False

    # ---- weights ----------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd):
        exp = param_shapes(self.config)
        missing = [k for k in exp if k not in sd]
        if missing:
            raise KeyError(f"VAE state_dict is missing {len(missing)} keys, e.g. {missing[:3]}")
        for k, shp in exp.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{k}: expected shape {shp}, got {tuple(sd[k].shape)}")
        self._build({k: sd[k].detach().to(self.device, H16) for k in exp})
        return self

    def init_random(self, seed=1234):
        g = torch.Generator(device=self.device).manual_seed(seed)
        sd = {}
        for k, shp in param_shapes(self.config).items():
            if len(shp) >= 2:
                fan = 1
                for s in shp[1:]:
                    fan *= s
                t = (torch.rand(shp, generator=g, device=self.device) * 2 - 1) / math.sqrt(fan)
                if k.endswith("conv2.weight") or k.endswith("to_out.0.weight"):
                    t *= 0.5
            elif "norm" in k:
                t = (1.0 if k.endswith("weight") else 0.0) + 0.1 * (torch.rand(shp, generator=g, device=self.device) * 2 - 1)
            else:
                t = 0.05 * (torch.rand(shp, generator=g, device=self.device) * 2 - 1)
            sd[k] = t.to(H16)
        self._build(sd)
        return self

    @classmethod
    def from_pretrained(cls, path, device="cuda:0", variant="fp16"):
        from safetensors.torch import load_file
        cfg = None
        cf = os.path.join(path, "vae", "config.json")  # diffusers' AutoencoderKL config (the keys VaeConfig.from_any keeps)
        if os.path.exists(cf):
            import json
            with open(cf) as fh:
                cfg = VaeConfig.from_any(json.load(fh))
        for name in (f"diffusion_pytorch_model.{variant}.safetensors", "diffusion_pytorch_model.safetensors"):
            f = os.path.join(path, "vae", name)
            if os.path.exists(f):
                return cls(cfg, device=device).load_state_dict(load_file(f))
        raise FileNotFoundError(f"no VAE weights under {path}/vae (expected diffusers safetensors)")

    def _build(self, sd):
        cfg = self.config
        boc, g, L = cfg.block_out_channels, cfg.norm_num_groups, cfg.layers_per_block
        f16 = lambda k: sd[k].to(H16).contiguous()
        e = "encoder"
        self.enc_conv_in = (pack_conv3x3_small(f16(f"{e}.conv_in.weight")), f16(f"{e}.conv_in.bias"))
        self.enc_down = []
        for i in range(len(boc)):
            res = [_Resnet(sd, f"{e}.down_blocks.{i}.resnets.{j}", g) for j in range(L)]
            down = _Conv3(sd, f"{e}.down_blocks.{i}.downsamplers.0.conv") if i != len(boc) - 1 else None
            self.enc_down.append((res, down))
        self.enc_mid = _Mid(sd, f"{e}.mid_block", g)
        self.enc_norm_out = (f16(f"{e}.conv_norm_out.weight"), f16(f"{e}.conv_norm_out.bias"))
        self.enc_conv_out = _Conv3(sd, f"{e}.conv_out")
        lc = cfg.latent_channels
        self.quant = (f16("quant_conv.weight").reshape(2 * lc, 2 * lc), f16("quant_conv.bias"))
        self.post_quant = (f16("post_quant_conv.weight").reshape(lc, lc), f16("post_quant_conv.bias"))
        d = "decoder"
        self.dec_conv_in = (pack_conv3x3_small(f16(f"{d}.conv_in.weight")), f16(f"{d}.conv_in.bias"))
        self.dec_mid = _Mid(sd, f"{d}.mid_block", g)
        self.dec_up = []
        for i in range(len(boc)):
            res = [_Resnet(sd, f"{d}.up_blocks.{i}.resnets.{j}", g) for j in range(L + 1)]
            up = _Conv3(sd, f"{d}.up_blocks.{i}.upsamplers.0.conv") if i != len(boc) - 1 else None
            self.dec_up.append((res, up))
        self.dec_norm_out = (f16(f"{d}.conv_norm_out.weight"), f16(f"{d}.conv_norm_out.bias"))
        self.dec_conv_out = _Conv3(sd, f"{d}.conv_out")
        self._loaded = True

    # ---- forward ----------------------------------------------------------------------------------------------------
    def _need(self):
        if not self._loaded:
            raise RuntimeError("AutoencoderKL: load_state_dict() or init_random() first")

    @torch.no_grad()
    def encode_moments(self, images):
        """images [n,3,H,W] fp16 in [-1,1] (H, W multiples of 64) -> (mean, logvar) [n,4,H/8,W/8] fp16 (logvar unclamped)"""
        self._need()
        cfg = self.config
        images = images.to(self.device, H16)
        n, c, H, W = images.shape
        x = ops.image_to_tokens(images)
        w0, b0 = self.enc_conv_in
        x, h, w = ops.conv3x3_small(x, w0, b0, nimg=n, h=H, wd=W, cin=c, cout=w0.shape[0])
        for res, down in self.enc_down:
            for r in res:
                x = r(x, n, h, w)
            if down is not None:
                x, h, w = down(x, n, h, w, stride=2, pad_mode=1)
        x = self.enc_mid(x, n, h, w)
        x = ops.groupnorm(x, *self.enc_norm_out, nsample=n, rows_per_sample=h * w, groups=cfg.norm_num_groups, eps=1e-6, silu=True)
        m, _, _ = self.enc_conv_out(x, n, h, w)
        m = ops.conv1x1_small(m, *self.quant)
        lc = cfg.latent_channels
        mom = ops.tokens_to_image(m, n, 2 * lc, h, w)
        return mom[:, :lc].contiguous(), mom[:, lc:].contiguous()

    @torch.no_grad()
    def encode_sample(self, images, noise=None, generator=None):
        """``vae.encode(x).latent_dist.sample()``: mean + exp(0.5 * clamp(logvar, -30, 20)) * noise, fp16 [n,4,h,w]"""
        mean, logvar = self.encode_moments(images)
        if noise is None:  # diffusers' randn_tensor: on the generator's device, then moved
            gdev = generator.device if generator is not None else mean.device
            noise = torch.randn(mean.shape, generator=generator, device=gdev, dtype=torch.float32)
        out = torch.empty_like(mean)
        nz = noise.to(self.device, H16).contiguous()  # (held until the launch is enqueued)
        check(lib.mvoc_gaussian_sample_f16(mean.data_ptr(), logvar.data_ptr(), nz.data_ptr(), out.data_ptr(), mean.numel(),
                                           ops._stream()), "gaussian_sample")
        return out

    @torch.no_grad()
    def decode(self, z):
        """z [n,4,h,w] fp16 (already divided by the scaling factor) -> images [n,3,8h,8w] fp16"""
        self._need()
        cfg = self.config
        z = z.to(self.device, H16)
        n, lc, h, w = z.shape
        x = ops.conv1x1_small(ops.image_to_tokens(z), *self.post_quant)
        w0, b0 = self.dec_conv_in
        x, _, _ = ops.conv3x3_small(x, w0, b0, nimg=n, h=h, wd=w, cin=lc, cout=w0.shape[0])
        x = self.dec_mid(x, n, h, w)
        for res, up in self.dec_up:
            for r in res:
                x = r(x, n, h, w)
            if up is not None:
                x, h, w = up(x, n, h, w, upsample_to=(2 * h, 2 * w))
        x = ops.groupnorm(x, *self.dec_norm_out, nsample=n, rows_per_sample=h * w, groups=cfg.norm_num_groups, eps=1e-6, silu=True)
        co = self.dec_conv_out
        buf = torch.empty((n * h * w, 4 * ((co.cout + 3) // 4)), dtype=H16, device=self.device)
        co(x, n, h, w, out=buf)
        return ops.tokens_to_image(buf, n, co.cout, h, w)


# ---- the pipeline's glue around the VAE (pipeline_i2vgen_xl.py) -----------------------------------------------------------
def center_crop_wide(image, resolution):
    """``_center_crop_wide`` (``:2054-2076``): BOX-resize so that the image covers ``resolution`` (W, H), then centre crop"""
    from PIL import Image
    scale = min(image.size[0] / resolution[0], image.size[1] / resolution[1])
    image = image.resize((round(image.width // scale), round(image.height // scale)), resample=Image.BOX)
    x1 = (image.width - resolution[0]) // 2
    y1 = (image.height - resolution[1]) // 2
    return image.crop((x1, y1, x1 + resolution[0], y1 + resolution[1]))


def preprocess_image(image):
    """``VaeImageProcessor(do_resize=False).preprocess``: PIL RGB -> [1,3,H,W] float32 in [-1,1]"""
    a = np.asarray(image.convert("RGB")).astype(np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1)[None] * 2.0 - 1.0


def tensor2vid(video, output_type="pil"):
    """``tensor2vid`` + ``VaeImageProcessor.postprocess`` (``:82-100``): [B,3,F,H,W] in [-1,1] -> per batch entry a list of PIL
    frames ("pil"), an array [F,H,W,3] in [0,1] ("np") or a tensor [F,3,H,W] ("pt")"""
    outs = []
    for b in range(video.shape[0]):
        v = (video[b].permute(1, 0, 2, 3).float().cpu() / 2 + 0.5).clamp(0, 1)  # [F,3,H,W]
        if output_type == "pt":
            outs.append(v)
            continue
        a = v.permute(0, 2, 3, 1).numpy()
        if output_type == "np":
            outs.append(a)
        elif output_type == "pil":
            from PIL import Image
            outs.append([Image.fromarray((f * 255).round().astype("uint8")) for f in a])
        else:
            raise ValueError(f"{output_type} does not exist. Please choose one of ['np', 'pt', 'pil]")
    return np.stack(outs) if output_type == "np" else (torch.stack(outs) if output_type == "pt" else outs)


class VaeCodec:
    """``encode_vae_video`` / ``prepare_image_latents`` / ``decode_latents`` of the reference pipeline on the HIP VAE; plugs
    into ``mvoc_amd.pipeline`` conditioners (``encode_video`` / ``image_latents`` / ``decode``)."""

    def __init__(self, vae):
        self.vae = vae
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)

    def _frames_tensor(self, frames, height, width):
        return torch.cat([preprocess_image(center_crop_wide(f, (width, height))) for f in frames]).to(self.vae.device, H16)

    def _scaled(self, lat):
        out = torch.empty_like(lat)
        check(lib.mvoc_scale_f16(lat.data_ptr(), out.data_ptr(), lat.numel(), float(self.vae.config.scaling_factor), ops._stream()), "scale")
        return out

    def encode_video(self, frames, height, width, generator=None):
        """list of PIL frames -> [1,4,F,h,w] fp16 (``:893-920``: per frame ``latent_dist.sample() * scaling_factor``)"""
        lat = self._scaled(self.vae.encode_sample(self._frames_tensor(frames, height, width), generator=generator))
        return lat[None].permute(0, 2, 1, 3, 4).contiguous()

    def image_latents(self, image, num_frames, height, width, generator=None):
        """``prepare_image_latents`` (``:860-890``): frame 0 = scaled latent of the image, frames k >= 1 = k/(F-1)"""
        first = self._scaled(self.vae.encode_sample(self._frames_tensor([image], height, width), generator=generator))[:, :, None]
        if num_frames > 1:
            ramp = torch.cat([torch.ones_like(first) * ((k + 1) / (num_frames - 1)) for k in range(num_frames - 1)], dim=2)
            first = torch.cat([first, ramp], dim=2)
        return first.contiguous()

    def decode(self, latents):
        """``decode_latents`` (``:771-791``): [B,4,F,h,w] -> video [B,3,F,H,W] float32"""
        b, c, f, h, w = latents.shape
        lat = latents.to(self.vae.device, H16).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w).contiguous()
        z = torch.empty_like(lat)
        check(lib.mvoc_scale_f16(lat.data_ptr(), z.data_ptr(), lat.numel(), float(1 / self.vae.config.scaling_factor), ops._stream()), "scale")
        img = self.vae.decode(z)
        return img.reshape(b, f, -1, img.shape[2], img.shape[3]).permute(0, 2, 1, 3, 4).float()


def attach_vae(pipe, pretrained_path=None, synthetic=False, seed=1234):
    """give ``pipe``'s conditioner a HIP VAE: the checkpoint's ``vae/`` when it exists, seeded synthetic weights of the exact
    architecture when ``synthetic`` (no checkpoint is reachable in the build environment), else nothing (the drivers then keep
    latents instead of frames)"""
    vae = None
    if pretrained_path and os.path.isdir(os.path.join(pretrained_path, "vae")):
        vae = AutoencoderKL.from_pretrained(pretrained_path, device=pipe.device)
    elif synthetic:
        vae = AutoencoderKL(device=pipe.device).init_random(seed)
    if vae is not None:
        pipe.vae = vae
        pipe.conditioner.vae = VaeCodec(vae)
    return vae
