"""ORACLE (test infrastructure, not product): PyTorch CPU restatement of the
I2VGen-XL 3D UNet as MVOC uses it.

The arithmetic of this network lives in the un-vendored third-party dependency
``diffusers==0.27.2`` (reference ``environment.yaml:58``); the reference repo
re-implements the forwards of four of its module classes and of the UNet itself:

* TransformerTemporalModel.forward   -> ``i2vgen-xl/pnp_utils.py:170-220``
* BasicTransformerBlock.forward      -> ``i2vgen-xl/pnp_utils.py:222-346``
* Attention.forward (dispatch)       -> ``i2vgen-xl/pnp_utils.py:348-385``
* Transformer2DModel.forward         -> ``i2vgen-xl/pnp_utils.py:387-548``
* AttnProcessor2_0 (+ injection)     -> ``i2vgen-xl/pnp_utils.py:565-704, 720-887``
* ResnetBlock2D.forward              -> ``i2vgen-xl/pnp_utils.py:902-1020``
* TemporalConvLayer.forward          -> ``i2vgen-xl/pnp_utils.py:1042-1057``
* UNet forward (extension)           -> ``i2vgen-xl/pipelines/pipeline_i2vgen_xl.py:109-362``

Those restatements are PINNED by ``tests/golden/*.npz`` (the reference's own
functions run against these modules, see ``tools/gen_golden.py``).  Module
constructors, parameter shapes and the stock ``I2VGenXLUNet.forward`` follow
diffusers 0.27.2 from memory: **parity unpinned** for that half.

Attribute names and state_dict keys equal diffusers' so that (a) the reference's
hook code can walk this tree and (b) a real ``i2vgen-xl`` checkpoint's key names
map 1:1 onto it.
"""
import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# config
# --------------------------------------------------------------------------------------
class UNetConfig:
    """ali-vilab/i2vgen-xl ``unet/config.json`` (recalled); ``tiny()`` is a 2-level toy."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
                 up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
                 layers_per_block=2, norm_num_groups=32, cross_attention_dim=1024, attention_head_dim=64,
                 transformer_in_heads=8, context_pool=32):
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.block_out_channels = tuple(block_out_channels)
        self.down_block_types = tuple(down_block_types)
        self.up_block_types = tuple(up_block_types)
        self.layers_per_block = layers_per_block
        self.norm_num_groups = norm_num_groups
        self.cross_attention_dim = cross_attention_dim
        self.attention_head_dim = attention_head_dim
        self.transformer_in_heads = transformer_in_heads
        self.context_pool = context_pool

    @staticmethod
    def tiny(**kw):
        d = dict(block_out_channels=(64, 128), down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
                 up_block_types=("UpBlock3D", "CrossAttnUpBlock3D"), layers_per_block=1, norm_num_groups=8,
                 cross_attention_dim=64, attention_head_dim=64, transformer_in_heads=2, context_pool=8)
        d.update(kw)
        return UNetConfig(**d)

    @staticmethod
    def small4(**kw):
        """4-level toy with the full attribute tree the reference's hooks address (up_blocks[1..3], ...)."""
        d = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=2, norm_num_groups=8,
                 cross_attention_dim=64, attention_head_dim=64, transformer_in_heads=2, context_pool=8)
        d.update(kw)
        return UNetConfig(**d)

    def to_dict(self):
        return dict(self.__dict__)


# --------------------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------------------
class AttnProcessor2_0:
    """Stock SDPA processor (diffusers 0.27.2).  The reference's two PnP processors are this
    plus the masked Q/K injection (``pnp_utils.py:565-704``)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 scale=1.0, **_unused):
        b = hidden_states.shape[0]
        q = attn.to_q(hidden_states)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        k = attn.to_k(ctx)
        v = attn.to_v(ctx)
        hd = k.shape[-1] // attn.heads
        q = q.view(b, -1, attn.heads, hd).transpose(1, 2)
        k = k.view(b, -1, attn.heads, hd).transpose(1, 2)
        v = v.view(b, -1, attn.heads, hd).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, -1, attn.heads * hd).to(q.dtype)
        o = attn.to_out[0](o)
        o = attn.to_out[1](o)
        if attn.residual_connection:
            o = o + hidden_states
        return o / attn.rescale_output_factor


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False, out_bias=True):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.inner_dim = inner
        self.scale = dim_head ** -0.5
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(0.0)])
        self.processor = AttnProcessor2_0()

    def prepare_attention_mask(self, *a, **k):  # never reached on this path (masks are None)
        raise NotImplementedError

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, height=None, width=None,
                **kw):
        # pnp_utils.py:348-385: pass height/width only to processors that accept them
        import inspect
        if "height" in inspect.signature(self.processor.__call__).parameters:
            return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                                  attention_mask=attention_mask, height=height, width=width, **kw)
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, g = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(g)


class GELU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)

    def forward(self, x):
        return F.gelu(self.proj(x))


class FeedForward(nn.Module):
    def __init__(self, dim, inner_dim=None, activation_fn="geglu"):
        super().__init__()
        inner_dim = inner_dim or dim * 4
        act = GEGLU(dim, inner_dim) if activation_fn == "geglu" else GELU(dim, inner_dim)
        self.net = nn.ModuleList([act, nn.Dropout(0.0), nn.Linear(inner_dim, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    """norm_type='layer_norm' branch of ``pnp_utils.py:222-346``."""

    def __init__(self, dim, heads, dim_head, cross_attention_dim=None, double_self_attention=False):
        super().__init__()
        self.norm_type = "layer_norm"
        self.only_cross_attention = False
        self.pos_embed = None
        self._chunk_size = None
        self._chunk_dim = 0
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads=heads, dim_head=dim_head)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, cross_attention_dim=None if double_self_attention else cross_attention_dim,
                               heads=heads, dim_head=dim_head)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                timestep=None, cross_attention_kwargs=None, class_labels=None, height=None, width=None, **_):
        x = hidden_states
        x = self.attn1(self.norm1(x), encoder_hidden_states=None, attention_mask=attention_mask,
                       height=height, width=width) + x
        x = self.attn2(self.norm2(x), encoder_hidden_states=encoder_hidden_states,
                       attention_mask=encoder_attention_mask) + x
        x = self.ff(self.norm3(x)) + x
        return x


class Transformer2DModel(nn.Module):
    """use_linear_projection=True, continuous-input branch of ``pnp_utils.py:387-548``."""

    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, norm_num_groups=32):
        super().__init__()
        inner = heads * dim_head
        self.is_input_continuous = True
        self.is_input_vectorized = False
        self.is_input_patches = False
        self.use_linear_projection = True
        self.caption_projection = None
        self.gradient_checkpointing = False
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim=cross_attention_dim)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, hidden_states, encoder_hidden_states=None, cross_attention_kwargs=None, return_dict=True, **_):
        n, c, h, w = hidden_states.shape
        res = hidden_states
        x = self.norm(hidden_states).permute(0, 2, 3, 1).reshape(n, h * w, c)
        x = self.proj_in(x)
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states=encoder_hidden_states, height=h, width=w)
        x = self.proj_out(x).reshape(n, h, w, c).permute(0, 3, 1, 2).contiguous()
        return (x + res,)


class TransformerTemporalModel(nn.Module):
    """``pnp_utils.py:170-220``; both attentions are self-attention over the frame axis."""

    def __init__(self, heads, dim_head, in_channels, norm_num_groups=32):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, double_self_attention=True)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, hidden_states, encoder_hidden_states=None, num_frames=1, cross_attention_kwargs=None,
                return_dict=True, **_):
        bf, c, h, w = hidden_states.shape
        b = bf // num_frames
        res = hidden_states
        x = hidden_states.reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)  # [b,c,f,h,w]
        x = self.norm(x)  # statistics over (c/groups, f, h, w)
        x = x.permute(0, 3, 4, 2, 1).reshape(b * h * w, num_frames, c)
        x = self.proj_in(x)
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states=encoder_hidden_states, height=h, width=w)
        x = self.proj_out(x)
        x = x.reshape(b, h, w, num_frames, c).permute(0, 3, 4, 1, 2).contiguous().reshape(bf, c, h, w)
        return (x + res,)


# --------------------------------------------------------------------------------------
# conv blocks
# --------------------------------------------------------------------------------------
class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x, output_size=None, **_):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return self.conv(x)


class Downsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x, **_):
        return self.conv(x)


class ResnetBlock2D(nn.Module):
    """``pnp_utils.py:902-1020`` without the injection branch (time_embedding_norm='default')."""

    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.nonlinearity = nn.SiLU()
        self.upsample = None
        self.downsample = None
        self.skip_time_act = False
        self.time_embedding_norm = "default"
        self.output_scale_factor = 1.0
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, input_tensor, temb, scale=1.0):
        h = self.conv1(self.nonlinearity(self.norm1(input_tensor)))
        h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + h) / self.output_scale_factor


class TemporalConvLayer(nn.Module):
    """``pnp_utils.py:1042-1057``: four (GN, SiLU, Conv3d(3,1,1)) stages + identity."""

    def __init__(self, dim, norm_num_groups=32, zero_init_last=False):
        super().__init__()

        def stage(with_dropout):
            mods = [nn.GroupNorm(norm_num_groups, dim), nn.SiLU()]
            if with_dropout:
                mods.append(nn.Dropout(0.0))
            mods.append(nn.Conv3d(dim, dim, (3, 1, 1), padding=(1, 0, 0)))
            return nn.Sequential(*mods)

        self.conv1 = stage(False)
        self.conv2 = stage(True)
        self.conv3 = stage(True)
        self.conv4 = stage(True)
        if zero_init_last:  # diffusers zero-inits conv4; randomised here so the layer is exercised
            nn.init.zeros_(self.conv4[-1].weight)
            nn.init.zeros_(self.conv4[-1].bias)

    def forward(self, hidden_states, num_frames=1):
        bf, c, h, w = hidden_states.shape
        x = hidden_states.reshape(bf // num_frames, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        y = self.conv4(self.conv3(self.conv2(self.conv1(x))))
        x = x + y
        return x.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class CrossAttnDownBlock3D(nn.Module):
    has_cross_attention = True

    def __init__(self, cin, cout, temb, layers, heads, dim_head, ctx_dim, groups, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups) for i in range(layers)])
        self.temp_convs = nn.ModuleList([TemporalConvLayer(cout, groups) for _ in range(layers)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, dim_head, cout, ctx_dim, groups) for _ in range(layers)])
        self.temp_attentions = nn.ModuleList([TransformerTemporalModel(heads, dim_head, cout, groups) for _ in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, num_frames=1, cross_attention_kwargs=None, **_):
        outs = ()
        x = hidden_states
        for r, tc, a, ta in zip(self.resnets, self.temp_convs, self.attentions, self.temp_attentions):
            x = r(x, temb)
            x = tc(x, num_frames=num_frames)
            x = a(x, encoder_hidden_states=encoder_hidden_states, return_dict=False)[0]
            x = ta(x, num_frames=num_frames, return_dict=False)[0]
            outs += (x,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                x = d(x)
            outs += (x,)
        return x, outs


class DownBlock3D(nn.Module):
    has_cross_attention = False

    def __init__(self, cin, cout, temb, layers, groups, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups) for i in range(layers)])
        self.temp_convs = nn.ModuleList([TemporalConvLayer(cout, groups) for _ in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None

    def forward(self, hidden_states, temb=None, num_frames=1, **_):
        outs = ()
        x = hidden_states
        for r, tc in zip(self.resnets, self.temp_convs):
            x = r(x, temb)
            x = tc(x, num_frames=num_frames)
            outs += (x,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                x = d(x)
            outs += (x,)
        return x, outs


class UNetMidBlock3DCrossAttn(nn.Module):
    has_cross_attention = True

    def __init__(self, c, temb, heads, dim_head, ctx_dim, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups), ResnetBlock2D(c, c, temb, groups)])
        self.temp_convs = nn.ModuleList([TemporalConvLayer(c, groups), TemporalConvLayer(c, groups)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, dim_head, c, ctx_dim, groups)])
        self.temp_attentions = nn.ModuleList([TransformerTemporalModel(heads, dim_head, c, groups)])

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, num_frames=1, cross_attention_kwargs=None, **_):
        x = self.resnets[0](hidden_states, temb)
        x = self.temp_convs[0](x, num_frames=num_frames)
        for a, ta, r, tc in zip(self.attentions, self.temp_attentions, self.resnets[1:], self.temp_convs[1:]):
            x = a(x, encoder_hidden_states=encoder_hidden_states, return_dict=False)[0]
            x = ta(x, num_frames=num_frames, return_dict=False)[0]
            x = r(x, temb)
            x = tc(x, num_frames=num_frames)
        return x


class _UpBase(nn.Module):
    def _make(self, cin, cout, prev, temb, layers, groups, add_upsample):
        res = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = prev if i == 0 else cout
            res.append(ResnetBlock2D(rin + skip, cout, temb, groups))
        self.resnets = nn.ModuleList(res)
        self.temp_convs = nn.ModuleList([TemporalConvLayer(cout, groups) for _ in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None


class UpBlock3D(_UpBase):
    has_cross_attention = False

    def __init__(self, cin, cout, prev, temb, layers, groups, add_upsample):
        super().__init__()
        self._make(cin, cout, prev, temb, layers, groups, add_upsample)

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, upsample_size=None, num_frames=1, **_):
        x = hidden_states
        for r, tc in zip(self.resnets, self.temp_convs):
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            x = torch.cat([x, skip], dim=1)
            x = r(x, temb)
            x = tc(x, num_frames=num_frames)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                x = u(x, upsample_size)
        return x


class CrossAttnUpBlock3D(_UpBase):
    has_cross_attention = True

    def __init__(self, cin, cout, prev, temb, layers, heads, dim_head, ctx_dim, groups, add_upsample):
        super().__init__()
        self._make(cin, cout, prev, temb, layers, groups, add_upsample)
        self.attentions = nn.ModuleList([Transformer2DModel(heads, dim_head, cout, ctx_dim, groups) for _ in range(layers)])
        self.temp_attentions = nn.ModuleList([TransformerTemporalModel(heads, dim_head, cout, groups) for _ in range(layers)])

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None,
                upsample_size=None, num_frames=1, cross_attention_kwargs=None, **_):
        x = hidden_states
        for r, tc, a, ta in zip(self.resnets, self.temp_convs, self.attentions, self.temp_attentions):
            skip = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            x = torch.cat([x, skip], dim=1)
            x = r(x, temb)
            x = tc(x, num_frames=num_frames)
            x = a(x, encoder_hidden_states=encoder_hidden_states, return_dict=False)[0]
            x = ta(x, num_frames=num_frames, return_dict=False)[0]
        if self.upsamplers is not None:
            for u in self.upsamplers:
                x = u(x, upsample_size)
        return x


# --------------------------------------------------------------------------------------
# embeddings / stem
# --------------------------------------------------------------------------------------
def timestep_embedding(timesteps: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)`` -> [cos | sin], fp32."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half
    ang = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


class Timesteps(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, t):
        return timestep_embedding(t, self.dim)


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear_1 = nn.Linear(cin, cout)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(cout, cout)

    def forward(self, x, cond=None):
        return self.linear_2(self.act(self.linear_1(x)))


class I2VGenXLTransformerTemporalEncoder(nn.Module):
    """LN -> self-attn (+res) -> FeedForward(gelu) on the *un-normed* sum (+res)."""

    def __init__(self, dim, heads, dim_head, ff_inner):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads=heads, dim_head=dim_head)
        self.ff = FeedForward(dim, inner_dim=ff_inner, activation_fn="gelu")

    def forward(self, x):
        x = self.attn1(self.norm1(x)) + x
        return self.ff(x) + x


class I2VGenXLUNet(nn.Module):
    def __init__(self, cfg: Optional[UNetConfig] = None):
        super().__init__()
        cfg = cfg or UNetConfig()
        self.config = cfg
        boc = cfg.block_out_channels
        ic = cfg.in_channels
        g = cfg.norm_num_groups
        hd = cfg.attention_head_dim
        ctx = cfg.cross_attention_dim
        temb = boc[0] * 4

        self.conv_in = nn.Conv2d(ic + ic, boc[0], 3, padding=1)
        self.transformer_in = TransformerTemporalModel(cfg.transformer_in_heads, hd, boc[0], g)
        self.image_latents_proj_in = nn.Sequential(
            nn.Conv2d(4, ic * 4, 3, padding=1), nn.SiLU(),
            nn.Conv2d(ic * 4, ic * 4, 3, padding=1), nn.SiLU(),
            nn.Conv2d(ic * 4, ic, 3, padding=1))
        self.image_latents_temporal_encoder = I2VGenXLTransformerTemporalEncoder(ic, 2, ic, ic * 4)
        self.image_latents_context_embedding = nn.Sequential(
            nn.Conv2d(4, ic * 8, 3, padding=1), nn.SiLU(),
            nn.AdaptiveAvgPool2d((cfg.context_pool, cfg.context_pool)),
            nn.Conv2d(ic * 8, ic * 16, 3, stride=2, padding=1), nn.SiLU(),
            nn.Conv2d(ic * 16, ctx, 3, stride=2, padding=1))
        self.time_proj = Timesteps(boc[0])
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        self.context_embedding = nn.Sequential(nn.Linear(ctx, temb), nn.SiLU(), nn.Linear(temb, ctx * ic))
        self.fps_embedding = nn.Sequential(nn.Linear(boc[0], temb), nn.SiLU(), nn.Linear(temb, temb))

        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i, t in enumerate(cfg.down_block_types):
            in_c, out_c = out_c, boc[i]
            last = i == len(boc) - 1
            if t == "CrossAttnDownBlock3D":
                self.down_blocks.append(CrossAttnDownBlock3D(in_c, out_c, temb, cfg.layers_per_block, out_c // hd, hd,
                                                             ctx, g, not last))
            else:
                self.down_blocks.append(DownBlock3D(in_c, out_c, temb, cfg.layers_per_block, g, not last))
        self.mid_block = UNetMidBlock3DCrossAttn(boc[-1], temb, boc[-1] // hd, hd, ctx, g)

        self.up_blocks = nn.ModuleList()
        self.num_upsamplers = 0
        rev = list(reversed(boc))
        out_c = rev[0]
        for i, t in enumerate(cfg.up_block_types):
            last = i == len(boc) - 1
            prev, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            if not last:
                self.num_upsamplers += 1
            if t == "CrossAttnUpBlock3D":
                self.up_blocks.append(CrossAttnUpBlock3D(in_c, out_c, prev, temb, cfg.layers_per_block + 1,
                                                         out_c // hd, hd, ctx, g, not last))
            else:
                self.up_blocks.append(UpBlock3D(in_c, out_c, prev, temb, cfg.layers_per_block + 1, g, not last))
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    # -- shared trunk ---------------------------------------------------------------------
    def _embeddings(self, sample, timestep, fps):
        b, _, f, _, _ = sample.shape
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float64 if isinstance(t, float) else torch.int64, device=sample.device)
        elif t.ndim == 0:
            t = t[None].to(sample.device)
        t = t.expand(b)
        t_emb = self.time_embedding(self.time_proj(t).to(self.dtype))
        fps = fps.expand(fps.shape[0])
        fps_emb = self.fps_embedding(self.time_proj(fps).to(self.dtype))
        return (t_emb + fps_emb).repeat_interleave(f, dim=0)

    def _stem_and_blocks(self, sample, image_latents_first, emb, context_emb, num_frames):
        b, c, f, h, w = sample.shape
        up_factor = 2 ** self.num_upsamplers
        forward_upsample_size = any(s % up_factor != 0 for s in (h, w))
        il = image_latents_first.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
        il = self.image_latents_proj_in(il)
        il = il.reshape(b, f, c, h, w).permute(0, 3, 4, 1, 2).reshape(b * h * w, f, c)
        il = self.image_latents_temporal_encoder(il)
        il = il.reshape(b, h, w, f, c).permute(0, 4, 3, 1, 2)
        x = torch.cat([sample, il], dim=1)
        x = x.permute(0, 2, 1, 3, 4).reshape(b * f, 2 * c, h, w)
        x = self.conv_in(x)
        x = self.transformer_in(x, num_frames=num_frames, return_dict=False)[0]

        skips = (x,)
        for blk in self.down_blocks:
            if blk.has_cross_attention:
                x, res = blk(hidden_states=x, temb=emb, encoder_hidden_states=context_emb, num_frames=num_frames)
            else:
                x, res = blk(hidden_states=x, temb=emb, num_frames=num_frames)
            skips += res
        x = self.mid_block(x, emb, encoder_hidden_states=context_emb, num_frames=num_frames)
        upsample_size = None
        for i, blk in enumerate(self.up_blocks):
            n = len(blk.resnets)
            res, skips = skips[-n:], skips[:-n]
            if i != len(self.up_blocks) - 1 and forward_upsample_size:
                upsample_size = skips[-1].shape[2:]
            if blk.has_cross_attention:
                x = blk(hidden_states=x, temb=emb, res_hidden_states_tuple=res, encoder_hidden_states=context_emb,
                        upsample_size=upsample_size, num_frames=num_frames)
            else:
                x = blk(hidden_states=x, temb=emb, res_hidden_states_tuple=res, upsample_size=upsample_size,
                        num_frames=num_frames)
        x = self.conv_out(self.conv_act(self.conv_norm_out(x)))
        return x.reshape(b, f, *x.shape[1:]).permute(0, 2, 1, 3, 4)

    def _latent_context_tokens(self, frame_latents):
        """[b,4,h,w] -> [b, 64, ctx] tokens (``pipeline_i2vgen_xl.py:221-227``)."""
        e = self.image_latents_context_embedding(frame_latents)
        n, c, hh, ww = e.shape
        return e.permute(0, 2, 3, 1).reshape(n, hh * ww, c)

    # -- stock diffusers forward (used by invert / __call__; unpinned) ----------------------
    def forward(self, sample, timestep, fps, image_latents, image_embeddings=None, encoder_hidden_states=None,
                cross_attention_kwargs=None, return_dict=True, **_):
        b, c, f, h, w = sample.shape
        emb = self._embeddings(sample, timestep, fps)
        lat_tok = self._latent_context_tokens(image_latents[:, :, 0])
        img_tok = self.context_embedding(image_embeddings).view(-1, self.config.in_channels,
                                                                self.config.cross_attention_dim)
        ctx = torch.cat([encoder_hidden_states, lat_tok, img_tok], dim=1).repeat_interleave(f, dim=0)
        out = self._stem_and_blocks(sample, image_latents, emb, ctx, f)
        return (out,)

    # -- the reference's extension forward (pinned by golden G7) -----------------------------
    def forward_ext(self, sample, timestep, fps, image_latents_first, image_latents, image_embeddings=None,
                    encoder_hidden_states=None, multi_frame_guidance=False, return_dict=True, **_):
        """``pipeline_i2vgen_xl.py:109-362``.  Context is assembled per frame as
        text(77) || latent tokens(64) || CLIP-image tokens(4); with ``multi_frame_guidance=False`` every
        frame uses frame 0's image embedding and latents (``:151, :212``)."""
        b, c, f, h, w = sample.shape
        if not multi_frame_guidance:
            image_embeddings = image_embeddings[:, 0:1, :].repeat(1, f, 1)
        emb = self._embeddings(sample, timestep, fps)
        per_frame = []
        for i in range(image_latents.size(2)):
            lat_tok = self._latent_context_tokens(image_latents[:, :, i if multi_frame_guidance else 0])
            img_tok = self.context_embedding(image_embeddings[:, i, :].unsqueeze(1)).view(
                -1, self.config.in_channels, self.config.cross_attention_dim)
            per_frame.append(torch.cat([encoder_hidden_states, lat_tok, img_tok], dim=1).unsqueeze(1))
        ctx = torch.cat(per_frame, dim=1)
        ctx = ctx.reshape(ctx.shape[0] * ctx.shape[1], ctx.shape[2], ctx.shape[3])
        out = self._stem_and_blocks(sample, image_latents_first, emb, ctx, f)
        return (out,)


def init_weights_(model: nn.Module, seed: int = 8888, scale_out: bool = True):
    """Seeded synthetic weights (there is no checkpoint in this environment).  PyTorch default init
    keeps activations O(1) through ~150 residual layers only if the residual branches are damped, so the
    last projection of every residual branch is scaled by 0.5; GroupNorm/LayerNorm affine get a small
    random perturbation so gamma/beta are exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.ndim >= 2:
                fan_in = p[0].numel()
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * (1.0 / math.sqrt(fan_in)))
            elif "norm" in name or ".conv1.0." in name or ".conv2.0." in name or ".conv3.0." in name or ".conv4.0." in name:
                if name.endswith("weight"):
                    p.copy_(1.0 + 0.1 * (torch.rand(p.shape, generator=g) * 2 - 1))
                else:
                    p.copy_(0.1 * (torch.rand(p.shape, generator=g) * 2 - 1))
            else:
                p.copy_(0.05 * (torch.rand(p.shape, generator=g) * 2 - 1))
        if scale_out:
            for name, p in model.named_parameters():
                if p.ndim >= 2 and (name.endswith("proj_out.weight") or name.endswith("conv2.weight")
                                    or name.endswith("conv4.3.weight") or name.endswith("to_out.0.weight")
                                    or name.endswith("ff.net.2.weight")):
                    p.mul_(0.5)
    return model


def count_params(model):
    return sum(p.numel() for p in model.parameters())
