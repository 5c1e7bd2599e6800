"""Conditioning prep: the CLIP vision and text towers of I2VGen-XL on the HIP operators.

The reference calls transformers' ``CLIPVisionModelWithProjection`` once per conditioning image
(``pipeline_i2vgen_xl.py:739-769`` ``_encode_image``; 16 frames x (background + objects + main) = 64 calls of batch 1 in the
composition entry, ``:1417-1427, 1501-1541``) and ``CLIPTextModel`` once per prompt (``:552-737`` ``encode_prompt``, 2-3
prompts).  Here a tower is one batched pass: all images of a job go through the ViT as one [B*257, 1280] row matrix.

Per layer (pre-LN transformer, the same for both towers): LayerNorm folded into the fused QKV projection
(``unet.Linear.fold_layernorm``), ``mvoc_flash_attn_f16`` (head_dim 64 for the text tower, causal; ViT-H's 80-wide heads are
zero-padded to 96 by the projection weights, scale 1/sqrt(80)), out-projection + residual in the GEMM epilogue, LayerNorm
folded into fc1 + GELU, fc2 + residual.  State-dict keys are transformers' (with or without the ``text_model.`` prefix older
versions write).  Architecture numbers of the checkpoint's ``image_encoder`` / ``text_encoder`` (OpenCLIP ViT-H/14) are
[recalled]: the checkpoint is not on disk here; ``config.json`` beside the weights overrides them when present.

Tokenisation (``CLIPTokenizer``: string work on the host) is not part of this path: ``encode_prompt`` takes token ids, or a
tokenizer object the caller loaded from the checkpoint.
"""
import ctypes as C
import json
import math
import os

import numpy as np
import torch

from . import ops
from ._ffi import ACT_GELU, check, lib
from .unet import Linear

H16 = torch.float16
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)  # CLIPImageProcessor's image_mean / image_std
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class ClipVisionConfig:
    def __init__(self, hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224,
                 patch_size=14, projection_dim=1024, hidden_act="gelu", layer_norm_eps=1e-5, **_):
        self.hidden_size, self.intermediate_size, self.num_hidden_layers = hidden_size, intermediate_size, num_hidden_layers
        self.num_attention_heads, self.image_size, self.patch_size = num_attention_heads, image_size, patch_size
        self.projection_dim, self.hidden_act, self.layer_norm_eps = projection_dim, hidden_act, layer_norm_eps


class ClipTextConfig:
    def __init__(self, hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, vocab_size=49408,
                 max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5, **_):
        self.hidden_size, self.intermediate_size, self.num_hidden_layers = hidden_size, intermediate_size, num_hidden_layers
        self.num_attention_heads, self.vocab_size = num_attention_heads, vocab_size
        self.max_position_embeddings, self.hidden_act, self.layer_norm_eps = max_position_embeddings, hidden_act, layer_norm_eps


def _cfg(cls, c):
    if c is None:
        return cls()
    if isinstance(c, cls):
        return c
    d = c if isinstance(c, dict) else (c.to_dict() if hasattr(c, "to_dict") else dict(c.__dict__))
    return cls(**d)


def _pad_heads_rows(w, heads, d, dp):
    """[heads*d, ...] -> [heads*dp, ...]: each head's rows followed by dp - d zero rows"""
    if d == dp:
        return w
    out = torch.zeros((heads, dp) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    out[:, :d] = w.view((heads, d) + tuple(w.shape[1:]))
    return out.view((heads * dp,) + tuple(w.shape[1:]))


class _Layer:
    def __init__(self, sd, k, heads, eps):
        g = lambda n: sd[f"{k}.{n}"].to(H16)
        c = g("self_attn.q_proj.weight").shape[0]
        self.heads, self.d = heads, c // heads
        self.dp = 64 if self.d <= 64 else 96
        if self.d not in (64, 80):
            raise NotImplementedError(f"CLIP head dim {self.d}: the attention kernel serves 64 and 80 (padded to 96)")
        pad = lambda n: _pad_heads_rows(g(n), heads, self.d, self.dp)
        wqkv = torch.cat([pad(f"self_attn.{p}_proj.weight") for p in "qkv"], 0)
        bqkv = torch.cat([pad(f"self_attn.{p}_proj.bias") for p in "qkv"], 0)
        self.ln1 = (g("layer_norm1.weight").contiguous(), g("layer_norm1.bias").contiguous())
        self.ln2 = (g("layer_norm2.weight").contiguous(), g("layer_norm2.bias").contiguous())
        self.qkv = Linear(wqkv, bqkv).fold_layernorm(*self.ln1, eps=eps)
        wo = g("self_attn.out_proj.weight")  # [c, heads*d] -> zero columns where the padded head dims sit
        self.out = Linear(_pad_heads_rows(wo.t().contiguous(), heads, self.d, self.dp).t().contiguous(), g("self_attn.out_proj.bias"))
        self.fc1 = Linear(g("mlp.fc1.weight"), g("mlp.fc1.bias")).fold_layernorm(*self.ln2, eps=eps)
        self.fc2 = Linear(g("mlp.fc2.weight"), g("mlp.fc2.bias"))
        self.eps = eps

    def __call__(self, x, nbatch, t, causal):
        hd = self.heads * self.dp
        qkv = self.qkv.call_ln(x, self.ln1)
        a = ops.flash_attn(qkv[:, :hd], qkv[:, hd:2 * hd], qkv[:, 2 * hd:], nbatch=nbatch, heads=self.heads, tq=t, tk=t,
                           head_dim=self.dp, causal=causal, scale=1.0 / math.sqrt(self.d))
        x = self.out(a, resid=x)
        h = self.fc1.call_ln(x, self.ln2, act=ACT_GELU)
        return self.fc2(h, resid=x)


class _Tower:
    prefix = ""

    def __init__(self, config=None, device="cuda:0"):
        self.config = _cfg(self.config_class, config)
        if self.config.hidden_act != "gelu":
            raise NotImplementedError(f"CLIP hidden_act {self.config.hidden_act!r}: I2VGen-XL's towers (OpenCLIP ViT-H) use 'gelu'")
        self.device, self.dtype, self._loaded = torch.device(device), H16, False

    def load_state_dict(self, sd):
        p = self.prefix
        sd = {(k[len(p):] if k.startswith(p) else k): v for k, v in sd.items()}
        exp = self.param_shapes()
        missing = [k for k in exp if k not in sd]
        if missing:
            raise KeyError(f"{type(self).__name__} state_dict is missing {len(missing)} keys, e.g. {missing[:3]}")
        for k, shp in exp.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{k}: expected shape {shp}, got {tuple(sd[k].shape)}")
        self._build({k: sd[k].detach().to(self.device, H16) for k in exp})
        self._loaded = True
        return self

    def _layer_shapes(self, k):
        c, i = self.config.hidden_size, self.config.intermediate_size
        s = {}
        for p in ("q", "k", "v", "out"):
            s[f"{k}.self_attn.{p}_proj.weight"], s[f"{k}.self_attn.{p}_proj.bias"] = (c, c), (c,)
        for n in ("layer_norm1", "layer_norm2"):
            s[f"{k}.{n}.weight"], s[f"{k}.{n}.bias"] = (c,), (c,)
        s[f"{k}.mlp.fc1.weight"], s[f"{k}.mlp.fc1.bias"], s[f"{k}.mlp.fc2.weight"], s[f"{k}.mlp.fc2.bias"] = (i, c), (i,), (c, i), (c,)
        return s

    def init_random(self, seed=4321):
        """seeded synthetic weights of the exact architecture (no checkpoint is reachable in the build environment), generated
        on the device (ViT-H is 630 M parameters: 15 s per tower through the CPU generator, 0.1 s here)"""
        g = torch.Generator(device=self.device).manual_seed(seed)
        dev = self.device
        sd = {}
        for k, shp in self.param_shapes().items():
            if "norm" in k or "layrnorm" in k:
                t = (1.0 if k.endswith("weight") else 0.0) + 0.1 * (torch.rand(shp, generator=g, device=dev) * 2 - 1)
            elif k.endswith("bias"):
                t = 0.05 * (torch.rand(shp, generator=g, device=dev) * 2 - 1)
            elif "embedding" in k and not k.endswith("patch_embedding.weight"):
                t = 0.5 * torch.randn(shp, generator=g, device=dev)
            else:
                fan = int(np.prod(shp[1:]))
                t = (torch.rand(shp, generator=g, device=dev) * 2 - 1) * math.sqrt(3.0 / fan)
                if k.endswith("out_proj.weight") or k.endswith("fc2.weight"):
                    t *= 0.5
            sd[k] = t.to(H16)
        return self.load_state_dict(sd)

    @classmethod
    def from_pretrained(cls, path, device="cuda:0", variant="fp16"):
        from safetensors.torch import load_file
        d = os.path.join(path, cls.subfolder)
        cfg = None
        if os.path.exists(os.path.join(d, "config.json")):
            cfg = json.load(open(os.path.join(d, "config.json")))
        for name in (f"model.{variant}.safetensors", "model.safetensors"):
            f = os.path.join(d, name)
            if os.path.exists(f):
                return cls(cfg, device=device).load_state_dict(load_file(f))
        raise FileNotFoundError(f"no CLIP weights under {d} (expected transformers safetensors)")

    def _need(self):
        if not self._loaded:
            raise RuntimeError(f"{type(self).__name__}: load_state_dict() or init_random() first")


class CLIPVisionModelWithProjection(_Tower):
    """``image_encoder`` of the checkpoint: pixel_values [B,3,224,224] -> image_embeds [B, projection_dim]"""
    config_class, prefix, subfolder = ClipVisionConfig, "", "image_encoder"

    def param_shapes(self):
        c = self.config
        n = (c.image_size // c.patch_size) ** 2 + 1
        s = {"vision_model.embeddings.class_embedding": (c.hidden_size,),
             "vision_model.embeddings.patch_embedding.weight": (c.hidden_size, 3, c.patch_size, c.patch_size),
             "vision_model.embeddings.position_embedding.weight": (n, c.hidden_size),
             "vision_model.pre_layrnorm.weight": (c.hidden_size,), "vision_model.pre_layrnorm.bias": (c.hidden_size,),
             "vision_model.post_layernorm.weight": (c.hidden_size,), "vision_model.post_layernorm.bias": (c.hidden_size,),
             "visual_projection.weight": (c.projection_dim, c.hidden_size)}
        for i in range(c.num_hidden_layers):
            s.update(self._layer_shapes(f"vision_model.encoder.layers.{i}"))
        return s

    def _build(self, sd):
        c = self.config
        k = 3 * c.patch_size ** 2
        self.kpad = (k + 63) // 64 * 64
        w = torch.zeros((c.hidden_size, self.kpad), dtype=H16, device=self.device)
        w[:, :k] = sd["vision_model.embeddings.patch_embedding.weight"].reshape(c.hidden_size, k)
        self.patch = Linear(w)
        self.cls = sd["vision_model.embeddings.class_embedding"].contiguous()
        self.pos = sd["vision_model.embeddings.position_embedding.weight"].contiguous()
        self.pre = (sd["vision_model.pre_layrnorm.weight"].contiguous(), sd["vision_model.pre_layrnorm.bias"].contiguous())
        self.post = (sd["vision_model.post_layernorm.weight"].contiguous(), sd["vision_model.post_layernorm.bias"].contiguous())
        self.layers = [_Layer(sd, f"vision_model.encoder.layers.{i}", c.num_attention_heads, c.layer_norm_eps)
                       for i in range(c.num_hidden_layers)]
        self.proj = Linear(sd["visual_projection.weight"])

    @torch.no_grad()
    def __call__(self, pixel_values):
        """-> image_embeds [B, projection_dim] fp16 (``CLIPVisionModelWithProjection(...).image_embeds``)"""
        self._need()
        c = self.config
        x = pixel_values.to(self.device, H16).contiguous()
        b = x.shape[0]
        if tuple(x.shape[1:]) != (3, c.image_size, c.image_size):
            raise ValueError(f"pixel_values must be [B,3,{c.image_size},{c.image_size}], got {tuple(x.shape)}")
        g = c.image_size // c.patch_size
        t = g * g + 1
        st = torch.cuda.current_stream().cuda_stream
        cols = torch.empty((b * g * g, self.kpad), dtype=H16, device=self.device)
        check(lib.mvoc_clip_patches_f16(x.data_ptr(), cols.data_ptr(), b, c.image_size, c.patch_size, self.kpad, st), "clip_patches")
        pe = self.patch(cols)
        h = torch.empty((b * t, c.hidden_size), dtype=H16, device=self.device)
        check(lib.mvoc_clip_embed_f16(pe.data_ptr(), None, self.cls.data_ptr(), self.pos.data_ptr(), h.data_ptr(), b * t, t,
                                      c.hidden_size, st), "clip_embed")
        h = ops.layernorm(h, *self.pre, eps=c.layer_norm_eps)
        for layer in self.layers:
            h = layer(h, b, t, False)
        pooled = h.view(b, t, c.hidden_size)[:, 0].contiguous()  # the class token's row of every image
        pooled = ops.layernorm(pooled, *self.post, eps=c.layer_norm_eps)
        return self.proj(pooled)


class CLIPTextModel(_Tower):
    """``text_encoder`` of the checkpoint: input_ids [B, T<=77] -> last_hidden_state [B, T, hidden]"""
    config_class, prefix, subfolder = ClipTextConfig, "text_model.", "text_encoder"

    def param_shapes(self):
        c = self.config
        s = {"embeddings.token_embedding.weight": (c.vocab_size, c.hidden_size),
             "embeddings.position_embedding.weight": (c.max_position_embeddings, c.hidden_size),
             "final_layer_norm.weight": (c.hidden_size,), "final_layer_norm.bias": (c.hidden_size,)}
        for i in range(c.num_hidden_layers):
            s.update(self._layer_shapes(f"encoder.layers.{i}"))
        return s

    def _build(self, sd):
        c = self.config
        self.tok = sd["embeddings.token_embedding.weight"].contiguous()
        self.pos = sd["embeddings.position_embedding.weight"].contiguous()
        self.final = (sd["final_layer_norm.weight"].contiguous(), sd["final_layer_norm.bias"].contiguous())
        self.layers = [_Layer(sd, f"encoder.layers.{i}", c.num_attention_heads, c.layer_norm_eps) for i in range(c.num_hidden_layers)]

    @torch.no_grad()
    def __call__(self, input_ids):
        """-> last_hidden_state [B, T, hidden] fp16 (``CLIPTextModel(ids)[0]``: causal self-attention, final LayerNorm)"""
        self._need()
        c = self.config
        ids = torch.as_tensor(input_ids).to(self.device, torch.int32).contiguous()
        b, t = ids.shape
        if t > c.max_position_embeddings or int(ids.max()) >= c.vocab_size or int(ids.min()) < 0:
            raise ValueError("input_ids out of range for this text tower")
        h = torch.empty((b * t, c.hidden_size), dtype=H16, device=self.device)
        check(lib.mvoc_clip_embed_f16(self.tok.data_ptr(), ids.data_ptr(), None, self.pos.data_ptr(), h.data_ptr(), b * t, t,
                                      c.hidden_size, torch.cuda.current_stream().cuda_stream), "clip_embed")
        for layer in self.layers:
            h = layer(h, b, t, True)
        return ops.layernorm(h, *self.final, eps=c.layer_norm_eps).view(b, t, c.hidden_size)


# ---- the pipeline-side glue (pipeline_i2vgen_xl.py) -----------------------------------------------------------------------
def clip_pixel_values(images, size=224):
    """``_resize_bilinear`` (``:2040-2051``: PIL BILINEAR to the feature extractor's crop size) then ``_encode_image``'s
    host half (``:742-756``: pil_to_numpy /255, CLIP mean / std normalisation, no crop / rescale) -> float32 [n,3,size,size]"""
    from PIL import Image
    out = []
    for im in images:
        a = np.asarray(im.convert("RGB").resize((size, size), Image.BILINEAR), dtype=np.float32) / 255.0
        out.append((a - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32))
    return torch.from_numpy(np.stack(out).transpose(0, 3, 1, 2).copy())


class ClipCodec:
    """the conditioner half the towers back: ``encode_images`` (one batched ViT pass over every conditioning frame of a job)
    and ``encode_prompt`` (token ids -> hidden states)"""

    def __init__(self, vision=None, text=None, tokenizer=None, batch=64):
        self.vision, self.text, self.tokenizer, self.batch = vision, text, tokenizer, batch

    def encode_images(self, images):
        """list of PIL images -> [n, 1, 1024] fp16; ``batch`` images per tower pass"""
        px = clip_pixel_values(images, self.vision.config.image_size)
        out = [self.vision(px[i:i + self.batch]) for i in range(0, len(images), self.batch)]
        return torch.cat(out)[:, None]

    def token_ids(self, prompt):
        if self.tokenizer is None:
            raise RuntimeError("ClipCodec has no tokenizer: pass token ids (or prompt_embeds) instead of strings")
        t = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True, return_tensors="pt")
        return t.input_ids

    def encode_prompt(self, prompt, negative_prompt=None):
        """-> (prompt_embeds, negative_prompt_embeds) [1,77,1024] each; both prompts in one tower pass"""
        ids = [p if torch.is_tensor(p) else self.token_ids(p) for p in (prompt, negative_prompt if negative_prompt is not None else "")]
        hs = self.text(torch.cat([torch.as_tensor(i).view(1, -1) for i in ids]))
        return hs[:1], hs[1:]


def attach_clip(pipe, pretrained_path=None, synthetic=False, seed=4321):
    """give ``pipe``'s conditioner HIP CLIP towers: the checkpoint's ``image_encoder`` / ``text_encoder`` (+ its tokenizer through
    transformers, host side) when they exist, seeded synthetic weights of the exact architecture when ``synthetic``"""
    vision = text = tok = None
    if pretrained_path and os.path.isdir(os.path.join(pretrained_path, "image_encoder")):
        vision = CLIPVisionModelWithProjection.from_pretrained(pretrained_path, device=pipe.device)
    elif synthetic:
        vision = CLIPVisionModelWithProjection(device=pipe.device).init_random(seed)
    if pretrained_path and os.path.isdir(os.path.join(pretrained_path, "text_encoder")):
        text = CLIPTextModel.from_pretrained(pretrained_path, device=pipe.device)
        if os.path.isdir(os.path.join(pretrained_path, "tokenizer")):
            from transformers import CLIPTokenizer
            tok = CLIPTokenizer.from_pretrained(os.path.join(pretrained_path, "tokenizer"))
    elif synthetic:
        text = CLIPTextModel(device=pipe.device).init_random(seed + 1)
    if vision is not None or text is not None:
        pipe.conditioner.clip = ClipCodec(vision, text, tok)
    return vision, text
