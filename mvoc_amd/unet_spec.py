"""Parameter table of the I2VGen-XL 3D UNet (diffusers==0.27.2 ``I2VGenXLUNet`` state_dict key names and
shapes) -- what ``from_pretrained`` of the reference (``i2vgen-xl/inverse.py:113-118``) would hand over.

Used to (a) validate / load a diffusers checkpoint state_dict, (b) create synthetic weights of the exact
architecture directly on the device (there is no checkpoint or network in the build environment).
"""
from collections import OrderedDict


class UNetConfig:
    """``ali-vilab/i2vgen-xl`` unet/config.json."""

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
        if attention_head_dim != 64:
            raise ValueError("the HIP attention kernels are built for head_dim 64 (I2VGen-XL)")
        if in_channels != 4:
            raise ValueError("the fused stem kernels are built for 4 latent channels")

    @classmethod
    def from_any(cls, cfg):
        if isinstance(cfg, cls):
            return cls(**cfg.__dict__)
        if isinstance(cfg, dict):
            return cls(**cfg)
        return cls(**{k: getattr(cfg, k) for k in cls().__dict__})

    @classmethod
    def from_diffusers(cls, d):
        """``<checkpoint>/unet/config.json`` as diffusers 0.27.2 writes it (``I2VGenXLUNet.register_to_config``: sample_size,
        in / out_channels, down / up_block_types, block_out_channels, layers_per_block, norm_num_groups, cross_attention_dim,
        attention_head_dim, num_attention_heads + ``_class_name`` / ``_diffusers_version``) -> UNetConfig.  Keys this engine has no
        use for are dropped; ``attention_head_dim`` may be a list (one entry per block, all equal in I2VGen-XL);
        ``transformer_in_heads`` / ``context_pool`` are constants of the architecture in diffusers (8 heads, 32 x 32 context pool) and
        only read when a (toy) checkpoint states them."""
        known = cls().__dict__
        kw = {k: d[k] for k in known if k in d and d[k] is not None}
        hd = kw.get("attention_head_dim")
        if isinstance(hd, (list, tuple)):
            if len(set(hd)) != 1:
                raise ValueError(f"unet/config.json: attention_head_dim {hd} differs between blocks: not I2VGen-XL")
            kw["attention_head_dim"] = int(hd[0])
        name = d.get("_class_name")
        if name not in (None, "I2VGenXLUNet"):
            raise ValueError(f"unet/config.json describes a {name}, not an I2VGenXLUNet")
        return cls(**kw)


def _lin(sd, name, cin, cout, bias=True):
    sd[name + ".weight"] = (cout, cin)
    if bias:
        sd[name + ".bias"] = (cout,)


def _norm(sd, name, c):
    sd[name + ".weight"] = (c,)
    sd[name + ".bias"] = (c,)


def _conv(sd, name, cin, cout, k=3):
    sd[name + ".weight"] = (cout, cin, k, k)
    sd[name + ".bias"] = (cout,)


def _attn(sd, name, dim, kv_dim, inner):
    _lin(sd, name + ".to_q", dim, inner, bias=False)
    _lin(sd, name + ".to_k", kv_dim, inner, bias=False)
    _lin(sd, name + ".to_v", kv_dim, inner, bias=False)
    _lin(sd, name + ".to_out.0", inner, dim)


def _basic_block(sd, name, dim, ctx_dim):
    _norm(sd, name + ".norm1", dim)
    _attn(sd, name + ".attn1", dim, dim, dim)
    _norm(sd, name + ".norm2", dim)
    _attn(sd, name + ".attn2", dim, ctx_dim if ctx_dim else dim, dim)
    _norm(sd, name + ".norm3", dim)
    _lin(sd, name + ".ff.net.0.proj", dim, dim * 8)
    _lin(sd, name + ".ff.net.2", dim * 4, dim)


def _transformer(sd, name, cin, inner, ctx_dim):
    _norm(sd, name + ".norm", cin)
    _lin(sd, name + ".proj_in", cin, inner)
    _basic_block(sd, name + ".transformer_blocks.0", inner, ctx_dim)
    _lin(sd, name + ".proj_out", inner, cin)


def _resnet(sd, name, cin, cout, temb):
    _norm(sd, name + ".norm1", cin)
    _conv(sd, name + ".conv1", cin, cout)
    _lin(sd, name + ".time_emb_proj", temb, cout)
    _norm(sd, name + ".norm2", cout)
    _conv(sd, name + ".conv2", cout, cout)
    if cin != cout:
        _conv(sd, name + ".conv_shortcut", cin, cout, k=1)


def _temp_conv(sd, name, c):
    for i, conv_idx in ((1, 2), (2, 3), (3, 3), (4, 3)):
        _norm(sd, f"{name}.conv{i}.0", c)
        sd[f"{name}.conv{i}.{conv_idx}.weight"] = (c, c, 3, 1, 1)
        sd[f"{name}.conv{i}.{conv_idx}.bias"] = (c,)


def param_shapes(cfg: UNetConfig) -> "OrderedDict[str, tuple]":
    sd = OrderedDict()
    boc, ic, hd, ctx = cfg.block_out_channels, cfg.in_channels, cfg.attention_head_dim, cfg.cross_attention_dim
    temb = boc[0] * 4
    _conv(sd, "conv_in", 2 * ic, boc[0])
    _transformer(sd, "transformer_in", boc[0], cfg.transformer_in_heads * hd, None)
    _conv(sd, "image_latents_proj_in.0", 4, ic * 4)
    _conv(sd, "image_latents_proj_in.2", ic * 4, ic * 4)
    _conv(sd, "image_latents_proj_in.4", ic * 4, ic)
    _norm(sd, "image_latents_temporal_encoder.norm1", ic)
    _attn(sd, "image_latents_temporal_encoder.attn1", ic, ic, 2 * ic)
    _lin(sd, "image_latents_temporal_encoder.ff.net.0.proj", ic, ic * 4)
    _lin(sd, "image_latents_temporal_encoder.ff.net.2", ic * 4, ic)
    _conv(sd, "image_latents_context_embedding.0", 4, ic * 8)
    _conv(sd, "image_latents_context_embedding.3", ic * 8, ic * 16)
    _conv(sd, "image_latents_context_embedding.5", ic * 16, ctx)
    _lin(sd, "time_embedding.linear_1", boc[0], temb)
    _lin(sd, "time_embedding.linear_2", temb, temb)
    _lin(sd, "context_embedding.0", ctx, temb)
    _lin(sd, "context_embedding.2", temb, ctx * ic)
    _lin(sd, "fps_embedding.0", boc[0], temb)
    _lin(sd, "fps_embedding.2", temb, temb)
    out_c = boc[0]
    for i, t in enumerate(cfg.down_block_types):
        in_c, out_c = out_c, boc[i]
        for j in range(cfg.layers_per_block):
            _resnet(sd, f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c, temb)
        for j in range(cfg.layers_per_block):
            _temp_conv(sd, f"down_blocks.{i}.temp_convs.{j}", out_c)
        if t == "CrossAttnDownBlock3D":
            for j in range(cfg.layers_per_block):
                _transformer(sd, f"down_blocks.{i}.attentions.{j}", out_c, out_c, ctx)
            for j in range(cfg.layers_per_block):
                _transformer(sd, f"down_blocks.{i}.temp_attentions.{j}", out_c, out_c, None)
        if i != len(boc) - 1:
            _conv(sd, f"down_blocks.{i}.downsamplers.0.conv", out_c, out_c)
    c = boc[-1]
    for j in range(2):
        _resnet(sd, f"mid_block.resnets.{j}", c, c, temb)
    for j in range(2):
        _temp_conv(sd, f"mid_block.temp_convs.{j}", c)
    _transformer(sd, "mid_block.attentions.0", c, c, ctx)
    _transformer(sd, "mid_block.temp_attentions.0", c, c, None)
    rev = list(reversed(boc))
    out_c = rev[0]
    layers = cfg.layers_per_block + 1
    for i, t in enumerate(cfg.up_block_types):
        prev, out_c = out_c, rev[i]
        in_c = rev[min(i + 1, len(boc) - 1)]
        for j in range(layers):
            skip = in_c if j == layers - 1 else out_c
            rin = prev if j == 0 else out_c
            _resnet(sd, f"up_blocks.{i}.resnets.{j}", rin + skip, out_c, temb)
        for j in range(layers):
            _temp_conv(sd, f"up_blocks.{i}.temp_convs.{j}", out_c)
        if i != len(boc) - 1:
            _conv(sd, f"up_blocks.{i}.upsamplers.0.conv", out_c, out_c)
        if t == "CrossAttnUpBlock3D":
            for j in range(layers):
                _transformer(sd, f"up_blocks.{i}.attentions.{j}", out_c, out_c, ctx)
            for j in range(layers):
                _transformer(sd, f"up_blocks.{i}.temp_attentions.{j}", out_c, out_c, None)
    _norm(sd, "conv_norm_out", boc[0])
    _conv(sd, "conv_out", boc[0], cfg.out_channels)
    return sd
