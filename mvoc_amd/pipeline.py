"""MVOC's three denoising loops on the MI355X engine, behind the reference pipeline's method names.

Reference (``i2vgen-xl/pipelines/pipeline_i2vgen_xl.py``):
  * ``I2VGenXLPipeline.invert``                       ``:1752-2018``  (loop body ``:1940-2000``)
  * ``I2VGenXLPipeline.__call__``                     ``:980-1216``   (loop body ``:1167-1202``)
  * ``I2VGenXLPipeline.sample_with_pnp_pipeline_with_edit_prompt_extraction_with_attn_injection``
                                                      ``:1220-1748``  (loop body ``:1636-1734``)

Scope: the per-step work -- UNet forward, PnP injections, latent fusion, CFG, DDIM update,
latent cache hand-off.  The once-per-clip encoders either side of the loops (CLIP text/vision, VAE) are "next"
rows: they enter through a ``conditioner`` object (``encode_prompt / encode_image / image_latents /
encode_video / decode``); ``SyntheticConditioner`` supplies tensors of the right shapes where no checkpoint
exists.  Everything inside a loop iteration runs on the GPU through libmvoc_hip; with ``use_graphs`` the whole
iteration is one captured hipGraph replay (~2000 kernel launches otherwise).
"""
import hashlib
import logging
import os
from copy import deepcopy

import torch

from . import ops
from .latent_cache import LatentCache
from .pnp_utils import register_time_all
from .schedulers import DDIMScheduler

logger = logging.getLogger(__name__)
H16 = torch.float16


class PipelineOutput:
    def __init__(self, frames=None, inverted_latents=None):
        self.frames = frames
        self.inverted_latents = inverted_latents


class SyntheticConditioner:
    """Deterministic stand-in for CLIP text/vision + VAE (SURVEY section 8d synthetic inputs): embeddings are
    N(0,1) draws seeded by a hash of the prompt / image bytes; image latents are frame 0 = 0.18215*N(0,1)
    followed by the frame-position ramp k/(F-1) exactly as ``prepare_image_latents`` (``:860-890``) builds it."""

    cross_attention_dim = 1024
    vae_scale_factor = 8

    def __init__(self, device, cross_attention_dim=1024, vae=None, clip=None):
        """``vae``: a ``mvoc_amd.vae.VaeCodec`` -- video / image latents and decoding then go through the HIP VAE
         instead of the seeded stand-ins; ``clip``: a ``mvoc_amd.clip.ClipCodec`` -- image / prompt embeddings
        then come from the HIP CLIP towers"""
        self.device = torch.device(device)
        self.cross_attention_dim = cross_attention_dim
        self.vae = vae
        self.clip = clip

    def _gen(self, *keys):
        h = hashlib.sha256("|".join(str(k) for k in keys).encode()).digest()
        return torch.Generator().manual_seed(int.from_bytes(h[:7], "little"))

    @staticmethod
    def _image_key(image):
        if image is None:
            return "none"
        if hasattr(image, "tobytes"):
            return hashlib.sha256(image.tobytes()).hexdigest()
        return str(image)

    def encode_prompt(self, prompt, negative_prompt=None):
        c = self.clip
        if c is not None and c.text is not None and (c.tokenizer is not None or torch.is_tensor(prompt)):
            return c.encode_prompt(prompt, negative_prompt)
        pe = torch.randn(1, 77, self.cross_attention_dim, generator=self._gen("p", prompt)).to(self.device, H16)
        ne = torch.randn(1, 77, self.cross_attention_dim, generator=self._gen("p", negative_prompt or "")).to(self.device, H16)
        return pe, ne

    def encode_image(self, image):
        return self.encode_images([image])

    def encode_images(self, images):
        """-> [n, 1, 1024]: every image of the list in one batched pass of the vision tower (the reference loops
        ``_encode_image`` per frame, ``pipeline_i2vgen_xl.py:1417-1427, 1501-1541``)"""
        if self.clip is not None and self.clip.vision is not None and all(hasattr(im, "convert") for im in images):
            return self.clip.encode_images(images)
        return torch.cat([torch.randn(1, 1, self.cross_attention_dim, generator=self._gen("i", self._image_key(im))).to(self.device, H16)
                          for im in images])

    def image_latents(self, image, num_frames, height, width):
        if self.vae is not None and hasattr(image, "convert"):
            return self.vae.image_latents(image, num_frames, height, width)
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor
        first = 0.18215 * torch.randn(1, 4, 1, h, w, generator=self._gen("l", self._image_key(image), h, w))
        if num_frames > 1:
            ramp = torch.cat([torch.full((1, 4, 1, h, w), (k + 1) / (num_frames - 1)) for k in range(num_frames - 1)], 2)
            first = torch.cat([first, ramp], 2)
        return first.to(self.device, H16)

    def encode_video(self, frames, height, width):
        if self.vae is not None and all(hasattr(f, "convert") for f in frames):
            return self.vae.encode_video(frames, height, width)
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor
        lat = [0.18215 * 4 * torch.randn(4, h, w, generator=self._gen("v", self._image_key(f), h, w)) for f in frames]
        return torch.stack(lat, 1)[None].to(self.device, H16)

    def decode(self, latents):
        """-> video [B,3,F,H,W] float32 in about [-1,1] (``decode_latents``, ``pipeline_i2vgen_xl.py:771-791``)"""
        if self.vae is None:
            raise NotImplementedError("this conditioner has no VAE (SyntheticConditioner(vae=VaeCodec(...))): pass output_type='latent'")
        return self.vae.decode(latents)


class GraphedStep:
    """Capture one loop iteration (a python callable working on static device buffers) into a hipGraph."""

    def __init__(self, fn, warmup=2, preserve=()):
        """``preserve``: tensors the iteration updates in place; they are restored after the eager warm-up runs"""
        self.fn = fn
        saved = [t.clone() for t in preserve]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            fn()
        for t, s in zip(preserve, saved):
            t.copy_(s)

    def __call__(self):
        self.graph.replay()


class I2VGenXLPipeline:
    """Drop-in surface of the reference pipeline for the denoising path."""

    def __init__(self, unet, scheduler=None, conditioner=None, use_graphs=True):
        self.unet = unet
        self.scheduler = scheduler or DDIMScheduler()
        self.conditioner = conditioner or SyntheticConditioner(unet.device, unet.config.cross_attention_dim)
        self.use_graphs = use_graphs
        self.vae_scale_factor = 8
        self._guidance_scale = 1.0
        self._graphs = {}
        self.max_cached_graphs = 4
        self._concurrent_states = {}  # invert_concurrent: one captured iteration per concurrent clip
        self._streams = []
        # composition loop: the UNet's source chunks are never read behind the last injection site (unet.prune_source_tail)
        self.prune_source_tail = os.environ.get("MVOC_PRUNE_SOURCE_TAIL", "1") != "0"  # (=0: A/B)
        # ... and its unconditional / conditional chunks are one computation up to the first cross-attention (unet.shared_prefix_chunks)
        self.share_cfg_prefix = os.environ.get("MVOC_SHARE_CFG_PREFIX", "1") != "0"  # (=0: A/B)
        self.latent_cache = LatentCache(unet.device)

    # ---- reference plumbing ------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path, torch_dtype=H16, variant="fp16", device="cuda:0", **kw):
        """``I2VGenXLPipeline.from_pretrained(PRETRAINED_MODEL_PATH, torch_dtype=fp16, variant="fp16")`` of the reference drivers
        (``inverse.py:113-118``, ``composite.py:76-85``): the UNet from ``<path>/unet/config.json`` (the I2VGen-XL default when the file
        is absent) + ``<path>/unet/diffusion_pytorch_model[.fp16].safetensors`` (diffusers layout).  The checkpoint's VAE and CLIP
        towers are attached by the drivers (``mvoc_amd.vae.attach_vae`` / ``mvoc_amd.clip.attach_clip``)."""
        import json
        from safetensors.torch import load_file
        from .unet import I2VGenXLUNet
        from .unet_spec import UNetConfig
        cfg = None
        cf = os.path.join(path, "unet", "config.json")
        if os.path.exists(cf):
            with open(cf) as fh:
                cfg = UNetConfig.from_diffusers(json.load(fh))
        for name in (f"diffusion_pytorch_model.{variant}.safetensors", "diffusion_pytorch_model.safetensors"):
            f = os.path.join(path, "unet", name)
            if os.path.exists(f):
                unet = I2VGenXLUNet(cfg, device=device).load_state_dict(load_file(f))
                return cls(unet, **kw)
        raise FileNotFoundError(f"no UNet weights under {path}/unet (expected diffusers safetensors); for synthetic "
                                "weights use mvoc_amd.pipeline.I2VGenXLPipeline.synthetic()")

    @classmethod
    def synthetic(cls, config=None, device="cuda:0", seed=8888, **kw):
        from .unet import I2VGenXLUNet
        return cls(I2VGenXLUNet(config, device=device).init_random(seed), **kw)

    def to(self, device):
        return self

    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def _execution_device(self):
        return self.unet.device

    @property
    def device(self):
        return self.unet.device

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1

    def encode_vae_video(self, video, device=None, height=576, width=1024):
        return self.conditioner.encode_video(video, height, width)

    def prepare_latents(self, batch, channels, num_frames, height, width, dtype, device, generator, latents=None):
        shape = (batch, channels, num_frames, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            g = generator if isinstance(generator, torch.Generator) and generator.device.type == "cpu" else None
            latents = torch.randn(shape, generator=g, dtype=torch.float32).to(dtype)
        return (latents.to(device, dtype) * self.scheduler.init_noise_sigma).contiguous()

    # ---- conditioning ---------------------------------------------------------------------------------
    def _stock_conditioning(self, prompt, negative_prompt, image, num_frames, height, width, target_fps, prompt_embeds,
                            negative_prompt_embeds, image_embeddings, image_latents):
        c = self.conditioner
        if prompt_embeds is None:
            prompt_embeds, negative_prompt_embeds = c.encode_prompt(prompt, negative_prompt)
        if image_embeddings is None:
            if getattr(c, "clip", None) is not None and hasattr(image, "convert"):
                from .vae import center_crop_wide
                # :1116-1120 -- the stock entry crops before the CLIP resize (the composition entries resize the uncropped frame)
                image_embeddings = c.encode_image(center_crop_wide(image, (width, width)))
            else:
                image_embeddings = c.encode_image(image)
        if image_latents is None:
            image_latents = c.image_latents(image, num_frames, height, width)
        if self.do_classifier_free_guidance:
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])
            image_embeddings = torch.cat([torch.zeros_like(image_embeddings), image_embeddings])  # :766
            image_latents = torch.cat([image_latents] * 2)
        nb = prompt_embeds.shape[0]
        fps = torch.full((nb,), float(target_fps), dtype=torch.float32, device=self.device)
        return dict(encoder_hidden_states=prompt_embeds.to(self.device, H16).contiguous(),
                    image_embeddings=image_embeddings.to(self.device, H16).contiguous(),
                    image_latents=image_latents.to(self.device, H16).contiguous(), fps=fps)

    def enable_frame_shard(self, shard):
        """Frame-shard the UNet of this pipeline over the ranks of ``shard`` (``mvoc_amd.frame_shard.FrameShard``; BASELINE
        configs[3]: one long clip on the 8 GPUs of a node).  Every rank runs the same loop on the same full latents (the
        scheduler update is replicated, 4 channels); the UNet computes F/world frames per rank.  Loop iterations run
        eagerly (torch.distributed issues the RCCL exchanges between the library's launches); rank 0 alone writes
        ``ddim_latents_{t}.pt`` files."""
        self.unet.set_frame_shard(shard)
        self.use_graphs = False
        self._graphs = {}
        self._concurrent_states = {}
        self.latent_cache.write_files = shard.rank == 0
        return self

    # ---- one loop iteration each (static buffers so that they can be graph-captured) -------------------
    def _make_stock_step(self, key, latents, cond, guidance_scale):
        """iteration of invert / __call__: [cat x2] -> UNet -> CFG + (inverse-)DDIM update, in place on `state`.
        The conditioning lives in buffers owned by the state (``load_cond`` refreshes them), so one captured iteration
        serves every later call of the same shape."""
        st = {"latents": latents.clone(), "t": torch.zeros(1, dtype=torch.float32, device=self.device),
              "coef": torch.zeros(5, dtype=torch.float32, device=self.device),
              "cond": {k: v.clone() for k, v in cond.items()}}
        do_cfg = guidance_scale > 1
        own = st["cond"]
        # loop-invariant conditioning (context tokens, cross-attention K/V, image-latent stem): once per loop, not per step
        shape = (latents.shape[0] * (2 if do_cfg else 1),) + tuple(latents.shape[1:])

        def prepare():
            return self.unet.prepare_conditioning(shape, own["fps"], own["image_latents"], own["image_latents"],
                                                  own["image_embeddings"], own["encoder_hidden_states"], False)

        prepared = prepare()

        def load_cond(new):
            """new conditioning of the same shapes -> the state's static buffers (no re-capture)"""
            for k, v in new.items():
                own[k].copy_(v)
            prepared.copy_from(prepare())

        def body():
            x = st["latents"]
            inp = torch.cat([x, x]) if do_cfg else x
            noise = self.unet.forward(inp, st["t"], own["fps"], image_latents=own["image_latents"],
                                      image_embeddings=own["image_embeddings"],
                                      encoder_hidden_states=own["encoder_hidden_states"], conditioning=prepared)[0]
            if do_cfg:
                ops.ddim_step(x, noise[1:2].contiguous(), st["coef"], v_uncond=noise[0:1].contiguous(), out=x)
            else:
                ops.ddim_step(x, noise, st["coef"], out=x)

        st["load_cond"] = load_cond
        st["run"] = GraphedStep(body, preserve=(st["latents"],)) if self.use_graphs else body
        return st

    def _run_stock_loop(self, latents, cond, num_inference_steps, guidance_scale, first_idx=0, on_step=None):
        sched = self.scheduler
        sched.set_timesteps(num_inference_steps, device=self.device)
        sched.timesteps = sched.timesteps[first_idx:]
        table, index = sched.coef_table(self.device, guidance_scale)
        # keyed by what the captured iteration bakes in (shapes, CFG layout, frame shard, graphs on/off) -- NOT by the
        # conditioning tensors' addresses: those change with every invert() / __call__()
        flags = self.unet.injection_flags()
        if any(flags):  # eager mode raises on the batch layout; a cached graph would silently replay a clean iteration
            raise RuntimeError("stock loop with PnP hooks armed: call register_time_all(pipe, None, None) first")
        key = ("stock", tuple(latents.shape), guidance_scale > 1, getattr(self.unet, "shard_generation", 0) if self.unet.shard is not None else 0,
               bool(self.use_graphs), flags, tuple((k, tuple(v.shape)) for k, v in sorted(cond.items())))
        st = self._graphs.pop(key, None)
        if st is None:
            if len(self._graphs) >= self.max_cached_graphs:  # each entry pins a UNet graph + its private activation pool
                self._graphs.pop(next(iter(self._graphs)))     # least recently used (hits are re-inserted at the end)
            st = self._make_stock_step(key, latents, cond, guidance_scale)
        else:
            st["load_cond"](cond)
        self._graphs[key] = st
        st["latents"].copy_(latents)
        for i, t in enumerate(sched.timesteps):
            st["t"].fill_(float(t))
            st["coef"].copy_(table[index[int(t)]])
            st["run"]()
            if on_step is not None:
                on_step(i, int(t), st["latents"])
        return st["latents"].clone()

    # ---- reference methods -----------------------------------------------------------------------------
    @torch.no_grad()
    def invert(self, prompt=None, image=None, height=704, width=1280, target_fps=16, num_frames=16,
               num_inference_steps=50, guidance_scale=9.0, negative_prompt=None, eta=0.0, num_videos_per_prompt=1,
               decode_chunk_size=1, generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None,
               output_type="pil", return_dict=True, cross_attention_kwargs=None, clip_skip=1, output_dir=None,
               image_embeddings=None, image_latents=None):
        """DDIM inversion; writes ``ddim_latents_{t}.pt`` per step and returns ``[1, steps, 4, F, h, w]`` noisiest first."""
        self._guidance_scale = guidance_scale
        cond = self._stock_conditioning(prompt, negative_prompt, image, num_frames, height, width, target_fps, prompt_embeds,
                                        negative_prompt_embeds, image_embeddings, image_latents)
        latents = self.prepare_latents(1, 4, num_frames, height, width, H16, self.device, generator, latents)
        seq = []

        def on_step(i, t, lat):
            snap = lat.clone()
            seq.append(snap)
            self.latent_cache.put(output_dir, t, snap)  # device-resident hand-off + async torch.save

        self._run_stock_loop(latents, cond, num_inference_steps, guidance_scale, on_step=on_step)
        self.latent_cache.flush()
        inverted = torch.stack(list(reversed(seq)), 1)
        if not return_dict:
            return inverted
        return PipelineOutput(inverted_latents=inverted)

    @torch.no_grad()
    def invert_many(self, prompts, images, latents, output_dirs, height=704, width=1280, target_fps=16, num_frames=16,
                    num_inference_steps=50, guidance_scale=1.0, negative_prompt=None):
        """DDIM-invert several source clips of the same shape in ONE batched loop: the per-object inversions of a
        composition job (background + N objects) are independent (``inverse.py:136-190`` runs them one after another),
        so on one GPU they can share every weight read -- UNet batch n instead of n passes at batch 1.  Same files, same
        return value per clip as ``invert``; guidance 1.0 only (the setting of ``group_inversion/template.yaml``)."""
        if guidance_scale > 1:
            raise NotImplementedError("invert_many batches the cfg = 1.0 inversions of inverse.py; use invert() for CFG")
        n = len(prompts)
        if not (len(images) == len(latents) == len(output_dirs) == n and n > 0):
            raise ValueError("invert_many: prompts, images, latents and output_dirs must have the same length")
        self._guidance_scale = guidance_scale
        conds = [self._stock_conditioning(p, negative_prompt, im, num_frames, height, width, target_fps, None, None, None, None)
                 for p, im in zip(prompts, images)]
        cond = {k: torch.cat([c[k] for c in conds]).contiguous() for k in conds[0]}
        lat = torch.cat([self.prepare_latents(1, 4, num_frames, height, width, H16, self.device, None, l) for l in latents])
        seqs = [[] for _ in range(n)]

        def on_step(i, t, cur):
            for j in range(n):
                snap = cur[j:j + 1].clone()
                seqs[j].append(snap)
                self.latent_cache.put(output_dirs[j], t, snap)

        self._run_stock_loop(lat, cond, num_inference_steps, guidance_scale, on_step=on_step)
        self.latent_cache.flush()
        return [torch.stack(list(reversed(s_)), 1) for s_ in seqs]

    @torch.no_grad()
    def invert_concurrent(self, prompts, images, latents, output_dirs, height=704, width=1280, target_fps=16, num_frames=16,
                          num_inference_steps=50, guidance_scale=1.0, negative_prompt=None, concurrency_hint=True):
        """DDIM-invert several source clips AT THE SAME TIME, each in its own batch-1 loop on its own HIP stream.  The
        per-object inversions of a composition job are independent (``inverse.py:136-190`` runs them one after another; on a
        node they shard one per GPU): on one GPU their kernels interleave, and wherever a batch-1 launch cannot fill the chip --
        the 16x16 / 8x8 levels run 80-240 workgroups on 256 CUs, the small norm / statistics kernels are latency-bound -- another
        clip's kernels take the idle CUs (32.5 -> 26.7 ms per clip-step at three clips, profiles/r4).  Same return value per clip
        as ``invert``.

        ``concurrency_hint``: the value every GEMM of the captured iterations carries in ``mvoc_gemm_desc.concurrency``.
        ``False`` / ``1``: every clip runs exactly the launches of ``invert`` -- latents and ``ddim_latents_{t}.pt`` files are
        BIT-IDENTICAL to the one-by-one pass (asserted at production width in tests/test_fullwidth_gpu.py).  ``True`` (default):
        the hint is the number of clips of this call; an ``int`` fixes it whatever the group size (the driver passes its
        ``--concurrent_entries`` so that a clip's files do not depend on how many other clips happened to be pending).  With a hint
        > 1 the GEMMs whose batch-1 grid cannot fill the chip keep their K in one piece (no split-K slabs / reduce pass: the other
        clips fill the idle CUs; 25.5 ms per clip-step) and therefore sum in another order than under ``invert``: the latents then
        differ from the one-by-one pass by fp16 rounding of another summation order (rel-L2 5e-5 after a step on the 1.42 B
        network; same test), inside the per-step tolerance but NOT bit-identical."""
        n = len(prompts)
        if not (len(images) == len(latents) == len(output_dirs) == n and n > 0):
            raise ValueError("invert_concurrent: prompts, images, latents and output_dirs must have the same length")
        if self.unet.shard is not None:
            raise NotImplementedError("invert_concurrent: frame-sharded clips run one at a time")
        self._guidance_scale = guidance_scale
        sched = self.scheduler
        sched.set_timesteps(num_inference_steps, device=self.device)
        table, index = sched.coef_table(self.device, guidance_scale)
        if any(self.unet.injection_flags()):
            raise RuntimeError("stock loop with PnP hooks armed: call register_time_all(pipe, None, None) first")
        states, seqs = [], [[] for _ in range(n)]
        for j in range(n):
            cond = self._stock_conditioning(prompts[j], negative_prompt, images[j], num_frames, height, width, target_fps, None, None,
                                            None, None)
            lat = self.prepare_latents(1, 4, num_frames, height, width, H16, self.device, None, latents[j])
            hint = (n if concurrency_hint is True else max(1, int(concurrency_hint)))
            key = ("stock-concurrent", j, hint, tuple(lat.shape), guidance_scale > 1, bool(self.use_graphs),
                   tuple((k, tuple(v.shape)) for k, v in sorted(cond.items())))
            st = self._concurrent_states.pop(key, None)
            if st is None:
                with ops.gemm_concurrency(hint):  # (baked into the captured iteration: tile / split-K choices)
                    st = self._make_stock_step(key, lat, cond, guidance_scale)
            else:
                st["load_cond"](cond)
            self._concurrent_states[key] = st  # (re-)inserted at the end: least recently used first
            st["latents"].copy_(lat)
            states.append(st)
        # each entry pins a UNet graph and its private activation pool: keep this call's states plus at most as many older ones
        # as `_graphs` may hold (a job that alternates two shapes keeps both, a long multi-shape job does not grow without bound)
        while len(self._concurrent_states) > n + self.max_cached_graphs:
            self._concurrent_states.pop(next(iter(self._concurrent_states)))
        while len(self._streams) < n:
            self._streams.append(torch.cuda.Stream(device=self.device))
        cur = torch.cuda.current_stream()
        for s_ in self._streams[:n]:
            s_.wait_stream(cur)
        for t in sched.timesteps:
            row = table[index[int(t)]]
            for j, (st, s_) in enumerate(zip(states, self._streams)):
                with torch.cuda.stream(s_):  # everything of clip j -- the two small fills, the replay, the snapshot -- on stream j
                    st["t"].fill_(float(t))
                    st["coef"].copy_(row)
                    st["run"]()
                    snap = st["latents"].clone()
                    seqs[j].append(snap)
                    self.latent_cache.put(output_dirs[j], int(t), snap)
        for s_ in self._streams[:n]:
            cur.wait_stream(s_)
        self.latent_cache.flush()
        return [torch.stack(list(reversed(q)), 1) for q in seqs]

    @torch.no_grad()
    def __call__(self, prompt=None, image=None, height=704, width=1280, target_fps=16, num_frames=16,
                 num_inference_steps=50, guidance_scale=9.0, negative_prompt=None, eta=0.0, num_videos_per_prompt=1,
                 decode_chunk_size=1, generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None,
                 output_type="pil", return_dict=True, cross_attention_kwargs=None, clip_skip=1, ddim_init_latents_t_idx=0,
                 image_embeddings=None, image_latents=None):
        self._guidance_scale = guidance_scale
        cond = self._stock_conditioning(prompt, negative_prompt, image, num_frames, height, width, target_fps, prompt_embeds,
                                        negative_prompt_embeds, image_embeddings, image_latents)
        latents = self.prepare_latents(1, 4, num_frames, height, width, H16, self.device, generator, latents)
        latents = self._run_stock_loop(latents, cond, num_inference_steps, guidance_scale, first_idx=ddim_init_latents_t_idx)
        frames = latents if output_type == "latent" else self._to_video(latents, output_type)
        return PipelineOutput(frames=frames) if return_dict else (frames,)

    def _to_video(self, latents, output_type):
        """``decode_latents`` + ``tensor2vid`` (``pipeline_i2vgen_xl.py:1207-1208``): per batch entry a list of PIL frames"""
        from .vae import tensor2vid
        return tensor2vid(self.conditioner.decode(latents), output_type)

    # ---- composition --------------------------------------------------------------------------------------
    def make_composition_state(self, latents, cond, masks, guidance_scale):
        """static buffers + the captured iteration variants of the composition loop.
        cond: dict(encoder_hidden_states [n,77,D], image_embeddings [n,F,D], image_latents_first, image_latents, fps).
        The state keeps its OWN copy of ``cond``: the hoisted conditioning (``prepare_conditioning``) and the shared-CFG-prefix
        decision below are taken once from these values, so a caller that later rewrites its tensors in place cannot make the
        captured iterations disagree with them -- new conditioning = a new state."""
        cond = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in cond.items()}
        n_obj = len(masks)
        do_cfg = guidance_scale > 1  # off: batch [bg, objs.., cond], one destination chunk for the injections
        nb = n_obj + (3 if do_cfg else 2)
        dev = self.device
        st = {"latents": latents.clone(), "inp": torch.empty((nb,) + tuple(latents.shape[1:]), dtype=H16, device=dev),
              "t": torch.zeros(1, dtype=torch.float32, device=dev), "coef": torch.zeros(5, dtype=torch.float32, device=dev),
              "masks": masks, "variants": {}, "cond": cond, "n_obj": n_obj,
              "fusion_masks": torch.stack([m[0].to(dev, H16) for m in masks]).contiguous(),
              "fusion_objs": torch.empty((n_obj,) + tuple(latents.shape), dtype=H16, device=dev)}

        prepared = self.unet.prepare_conditioning(tuple(st["inp"].shape), cond["fps"], cond["image_latents_first"],
                                                  cond["image_latents"], cond["image_embeddings"],
                                                  cond["encoder_hidden_states"], False)

        # classifier-free guidance: the unconditional and the conditional chunk receive the same latent (below); when their image
        # latents and fps are equal too -- the reference builds both from the main image, pipeline_i2vgen_xl.py:1676-1690 -- they
        # differ only in what the cross-attentions see, and the UNet shares their common prefix (unet.shared_prefix_chunks)
        share = bool(self.share_cfg_prefix and do_cfg and
                     all(torch.equal(cond[k][nb - 2], cond[k][nb - 1]) for k in ("image_latents_first", "image_latents", "fps")))
        st["share_cfg_prefix"] = share

        def body():
            x = st["latents"]
            if do_cfg:
                st["inp"][nb - 2].copy_(x[0])
            st["inp"][nb - 1].copy_(x[0])
            u = self.unet
            saved, u.prune_source_tail = u.prune_source_tail, bool(self.prune_source_tail)  # this loop reads the destination chunks only
            saved_sp, u.shared_prefix_chunks = u.shared_prefix_chunks, (2 if share else 0)
            try:
                noise = u.forward_ext(st["inp"], st["t"], cond["fps"], cond["image_latents_first"], cond["image_latents"],
                                      cond["image_embeddings"], cond["encoder_hidden_states"], multi_frame_guidance=False,
                                      conditioning=prepared)[0]
            finally:
                u.prune_source_tail, u.shared_prefix_chunks = saved, saved_sp
            ops.ddim_step(x, noise[nb - 1:nb].contiguous(), st["coef"],
                          v_uncond=noise[nb - 2:nb - 1].contiguous() if do_cfg else None, out=x)

        st["body"] = body
        st["nb"] = nb
        return st

    def composition_step(self, st, t, bg_latents, obj_latents, table_row, fuse=None):
        """one iteration of ``:1636-1734`` on device-resident latents; ``fuse`` = (mix_ratio, obj_random_noise_fusion,
        fusion object latents) on fusion steps"""
        if fuse is not None:
            mix, rnf, fobjs = fuse
            for j, o in enumerate(fobjs):
                st["fusion_objs"][j].copy_(o)
            ops.latent_fusion(st["latents"], bg_latents, st["fusion_objs"], st["fusion_masks"], mix, rnf, out=st["latents"])
            obj_latents = fobjs
        st["inp"][0].copy_(bg_latents[0])
        for j, o in enumerate(obj_latents):
            st["inp"][1 + j].copy_(o[0])
        st["t"].fill_(float(t))
        st["coef"].copy_(table_row)
        register_time_all(self, int(t), st["masks"])
        if not self.use_graphs:
            st["body"]()
            return
        # a captured iteration bakes in EVERY site's injecting() decision (the reference allows a schedule per site) and
        # the device copies of the masks: both are part of the variant key
        u = self.unet
        vkey = (u.injection_flags(), u.mask_key(st["masks"]), bool(u.pair_destinations), bool(u.prune_dead_chunks),
                bool(self.prune_source_tail), bool(st.get("share_cfg_prefix")))
        g = st["variants"].get(vkey)
        if g is None:
            g = st["variants"][vkey] = GraphedStep(st["body"], preserve=(st["latents"],))
        g()

    @torch.no_grad()
    def sample_with_pnp_pipeline_with_edit_prompt_extraction_with_attn_injection(
            self, prompt=None, main_first_image=None, main_image_list=None, background_first_image=None,
            background_image_list=None, objs_first_image=None, objs_image_list=None, height=704, width=1280, target_fps=16,
            num_frames=16, num_inference_steps=50, guidance_scale=9.0, negative_prompt=None, eta=0.0,
            num_videos_per_prompt=1, decode_chunk_size=1, generator=None, latents=None, prompt_embeds=None,
            negative_prompt_embeds=None, output_type="pil", return_dict=True, cross_attention_kwargs=None, clip_skip=1,
            fusion_steps=(0, 3), ddim_init_latents_t_idx=1, ddim_inv_prompt=None, obj_mask=None, obj_width_height=None,
            obj_ddim_latents_idx_offset=None, obj_random_noise_fusion=False, random_noise_ratio=0.0,
            bg_inv_latents_path=None, obj_ddim_latents_path=None, obj_masks_tensors=None):
        """PnP composition sampling.  ``obj_mask``: list of mask paths (preprocessed by ``mvoc_amd.utils.mask_preprocess``)
        or pass ``obj_masks_tensors`` = list of (float [1,4,F,h,w], bool [1,4,F,h,w]) directly."""
        from .utils import mask_preprocess
        self._guidance_scale = guidance_scale
        # guidance_scale <= 1: classifier-free guidance off.  The reference's hooks hard-code the batch of 5
        # (pnp_utils.py:592,747,784,972,1061,1115: `// 5`) and cannot run this; here the batch is [bg, objs.., cond], the
        # injections write the single trailing chunk and the DDIM update takes the conditional prediction as it is
        do_cfg = guidance_scale > 1
        c = self.conditioner
        n_obj = len(obj_ddim_latents_path)
        assert obj_mask is None or len(obj_mask) == n_obj
        # conditioning, assembled in the reference's batch order [bg, obj_1.., uncond, cond] (:1387, 1476, 1498, 1540)
        pe, ne = (prompt_embeds, negative_prompt_embeds) if prompt_embeds is not None else c.encode_prompt(prompt, negative_prompt)
        inv_pe, _ = c.encode_prompt(ddim_inv_prompt, negative_prompt)
        ehs = torch.cat([inv_pe.repeat(n_obj + 1, 1, 1)] + ([ne] if do_cfg else []) + [pe])
        main_lat = c.image_latents(main_first_image, num_frames, height, width)
        bg_lat = c.image_latents(background_first_image, num_frames, height, width)
        obj_first = [c.image_latents(im, num_frames, height, width) for im in objs_first_image]
        first_all = torch.cat([bg_lat] + obj_first + [main_lat] * (2 if do_cfg else 1))
        obj_lat = [c.image_latents(frames[0], num_frames, height, width) for frames in objs_image_list]
        bg_lat2 = c.image_latents(background_image_list[0], num_frames, height, width)
        lat_all = torch.cat([bg_lat2] + obj_lat + [main_lat] * (2 if do_cfg else 1))

        # every conditioning frame of the job (background, objects, main: 4 x 16 in the demo) through the vision tower at once
        lists = [background_image_list] + list(objs_image_list) + [main_image_list]
        allf = [f for fr in lists for f in fr]
        # (a conditioner without the batched entry -- the documented interface is encode_image -- is called per frame)
        flat = c.encode_images(allf) if hasattr(c, "encode_images") else torch.cat([c.encode_image(f) for f in allf])  # [sum F, 1, 1024]
        embs, o = [], 0
        for fr in lists:
            embs.append(flat[o:o + len(fr)].transpose(0, 1))  # [1, F, 1024]
            o += len(fr)
        main_emb = embs[-1]
        emb_all = torch.cat(embs[:-1] + ([torch.zeros_like(main_emb)] if do_cfg else []) + [main_emb])
        fps = torch.full((n_obj + (3 if do_cfg else 2),), float(target_fps), dtype=torch.float32, device=self.device)
        cond = dict(encoder_hidden_states=ehs.to(self.device, H16).contiguous(), image_embeddings=emb_all.to(self.device, H16).contiguous(),
                    image_latents_first=first_all.to(self.device, H16).contiguous(), image_latents=lat_all.to(self.device, H16).contiguous(), fps=fps)

        sched = self.scheduler
        sched.set_timesteps(num_inference_steps, device=self.device)
        full = deepcopy(sched)
        sched.timesteps = sched.timesteps[ddim_init_latents_t_idx:]
        offs = obj_ddim_latents_idx_offset or [0] * n_obj
        fusion_ts = [[int(full.timesteps[offs[j]:][k]) for k in range(*fusion_steps)] for j in range(n_obj)]
        latents = self.prepare_latents(1, 4, num_frames, height, width, H16, self.device, generator, latents)
        if obj_masks_tensors is None:
            obj_masks_tensors = [mask_preprocess(m, self.device, H16, 1, 4, num_frames, downscale=8) for m in obj_mask]
        st = self.make_composition_state(latents, cond, obj_masks_tensors, guidance_scale)
        table, index = sched.coef_table(self.device, guidance_scale)
        cache = self.latent_cache
        fusion_counter = 0  # never incremented in the reference (:1634, 1649)
        for i, t in enumerate(sched.timesteps):
            t = int(t)
            bg = cache.get(bg_inv_latents_path, t)
            fuse = None
            if fusion_steps[0] <= i < fusion_steps[1]:
                fobjs = [cache.get(obj_ddim_latents_path[j], fusion_ts[j][fusion_counter]) for j in range(n_obj)]
                fuse = (random_noise_ratio, obj_random_noise_fusion, fobjs)
                objs = fobjs
            else:
                objs = [cache.get(obj_ddim_latents_path[j], t) for j in range(n_obj)]
            self.composition_step(st, t, bg, objs, table[index[t]], fuse)
        latents = st["latents"].clone()
        frames = latents if output_type == "latent" else self._to_video(latents, output_type)
        return PipelineOutput(frames=frames) if return_dict else (frames,)
