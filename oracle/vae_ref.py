"""ORACLE (test infrastructure, not product): CPU restatement of the VAE either side of MVOC's denoising loops.

The reference calls diffusers' ``AutoencoderKL`` (``i2vgen-xl/pipelines/pipeline_i2vgen_xl.py:771-791`` ``decode_latents``,
``:860-890`` ``prepare_image_latents``, ``:893-920`` ``encode_vae_video``; image pre/post-processing through
``VaeImageProcessor`` ``:443, 908, 1208`` and ``_center_crop_wide`` ``:2054-2076``).  diffusers 0.27.2 is not vendored and
not installable here, the checkpoint's ``vae/config.json`` is not on disk: the module tree below is **[recalled]** --
``AutoencoderKL(block_out_channels=(128,256,512,512), layers_per_block=2, latent_channels=4, norm_num_groups=32,
scaling_factor=0.18215)`` with ``DownEncoderBlock2D`` / ``UpDecoderBlock2D`` / ``UNetMidBlock2D(attention_head_dim=512)``,
state_dict key names as diffusers writes them.  **Parity unpinned** like the rest of the diffusers half (oracle/__init__.py):
what the tests hold the HIP path to is this restatement on identical weights.

Plain ``torch.nn`` modules in fp32; ``tiny()`` is a narrow configuration for fast tests.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class VaeConfig:
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, scaling_factor=0.18215):
        self.in_channels, self.out_channels = in_channels, out_channels
        self.block_out_channels = tuple(block_out_channels)
        self.layers_per_block, self.latent_channels = layers_per_block, latent_channels
        self.norm_num_groups, self.scaling_factor = norm_num_groups, scaling_factor

    @staticmethod
    def tiny(**kw):
        d = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=1, norm_num_groups=8)
        d.update(kw)
        return VaeConfig(**d)

    def to_dict(self):
        return dict(self.__dict__)


class ResnetBlock2D(nn.Module):
    """diffusers ResnetBlock2D with temb_channels=None, eps 1e-6, swish, output_scale_factor 1"""

    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    """diffusers Attention as UNetMidBlock2D builds it for the VAE: one head of dim C, GroupNorm first, biases on q/k/v,
    residual connection, rescale_output_factor 1"""

    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        res = x
        t = x.view(b, c, h * w)
        t = self.group_norm(t).transpose(1, 2)  # [b, hw, c]
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]  # one head
        o = self.to_out[1](self.to_out[0](o))
        return o.transpose(1, 2).reshape(b, c, h, w) + res


class MidBlock(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, groups), ResnetBlock2D(c, c, groups)])
        self.attentions = nn.ModuleList([Attention(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Downsample2D(nn.Module):
    """padding=0 form: F.pad(x, (0,1,0,1)) then a stride-2 3x3 conv"""

    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Block(nn.Module):
    def __init__(self, cin, cout, n, groups, down=False, up=False):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if down else None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        cin = boc[0]
        for i, c in enumerate(boc):
            self.down_blocks.append(_Block(cin, c, cfg.layers_per_block, g, down=i != len(boc) - 1))
            cin = c
        self.mid_block = MidBlock(boc[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        rev = list(reversed(boc))
        self.conv_in = nn.Conv2d(cfg.latent_channels, rev[0], 3, padding=1)
        self.mid_block = MidBlock(rev[0], g)
        self.up_blocks = nn.ModuleList()
        cin = rev[0]
        for i, c in enumerate(rev):
            self.up_blocks.append(_Block(cin, c, cfg.layers_per_block + 1, g, up=i != len(rev) - 1))
            cin = c
        self.conv_norm_out = nn.GroupNorm(g, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKL(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        self.config = cfg or VaeConfig()
        self.encoder, self.decoder = Encoder(self.config), Decoder(self.config)
        lc = self.config.latent_channels
        self.quant_conv = nn.Conv2d(2 * lc, 2 * lc, 1)
        self.post_quant_conv = nn.Conv2d(lc, lc, 1)

    def encode_moments(self, x):
        """[n,3,H,W] in [-1,1] -> (mean, logvar clamped to [-30, 20]) each [n,4,H/8,W/8] (DiagonalGaussianDistribution)"""
        m = self.quant_conv(self.encoder(x))
        mean, logvar = m.chunk(2, dim=1)
        return mean, logvar.clamp(-30.0, 20.0)

    def encode_sample(self, x, noise):
        """``latent_dist.sample()`` with the noise given explicitly: mean + exp(0.5 logvar) * noise"""
        mean, logvar = self.encode_moments(x)
        return mean + torch.exp(0.5 * logvar) * noise

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def init_weights_(model, seed=123):
    """seeded synthetic weights: U(+-1/sqrt(fan_in)) matrices (second conv of every resnet and the attention output damped so
    ~30 residual layers stay O(1)), norm gains ~1"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.ndim >= 2:
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) / math.sqrt(p[0].numel()))
                if name.endswith("conv2.weight") or name.endswith("to_out.0.weight"):
                    p.mul_(0.5)
            elif "norm" in name:
                p.copy_((1.0 if name.endswith("weight") else 0.0) + 0.1 * (torch.rand(p.shape, generator=g) * 2 - 1))
            else:
                p.copy_(0.05 * (torch.rand(p.shape, generator=g) * 2 - 1))
    return model


# ---- the loop-side glue of the reference (pipeline_i2vgen_xl.py) ---------------------------------------------------------
def decode_latents(vae, latents):
    """``decode_latents`` (``:771-791``): [B,4,F,h,w] -> video [B,3,F,H,W] fp32 in about [-1,1]"""
    latents = 1 / vae.config.scaling_factor * latents
    b, c, f, h, w = latents.shape
    lat = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    img = torch.cat([vae.decode(lat[i:i + 1]) for i in range(lat.shape[0])])  # decode_chunk_size = 1
    return img[None].reshape((b, f, -1) + img.shape[2:]).permute(0, 2, 1, 3, 4).float()


def encode_frames(vae, frames, noise):
    """``encode_vae_video`` (``:893-920``) after the PIL part: frames [F,3,H,W] in [-1,1] -> [1,4,F,h,w] scaled latents"""
    lat = torch.stack([vae.encode_sample(frames[i:i + 1], noise[i:i + 1])[0] * vae.config.scaling_factor for i in range(frames.shape[0])])
    return lat[None].permute(0, 2, 1, 3, 4)
