"""GPU: the NON-synthetic door of the drop-in drivers (reference ``inverse.py:113-131``, ``composite.py:76-85``):
``I2VGenXLPipeline.from_pretrained(PRETRAINED_MODEL_PATH, torch_dtype=fp16, variant="fp16")`` + the checkpoint's ``vae/``,
``image_encoder/``, ``text_encoder/``, ``tokenizer/`` on a TOY checkpoint written here in the diffusers / transformers directory
layout (config.json + *.fp16.safetensors; no real checkpoint is reachable in the build environment).

``inverse.py`` WITHOUT ``--synthetic`` must (a) read ``unet/config.json`` (the toy widths are not the I2VGen-XL default: with the
default config ``load_state_dict`` rejects the shapes), (b) write ``ddim_latents_{t}.pt`` bit-identical to a pipeline assembled by
``load_state_dict`` on the very same tensors."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VAE_CFG = dict(in_channels=3, out_channels=3, block_out_channels=[64, 64, 128, 128], layers_per_block=1, latent_channels=4, norm_num_groups=8,
               scaling_factor=0.18215)
VIS_CFG = dict(hidden_size=320, intermediate_size=640, num_hidden_layers=2, num_attention_heads=4, image_size=56, patch_size=14, projection_dim=64)
TXT_CFG = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1, vocab_size=56, max_position_embeddings=77)


def _rand_sd(shapes, seed):
    """fp16 tensors of the given shapes: matrices U(+-1/sqrt(fan_in)), norm gains ~ 1, the rest small"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in shapes.items():
        if len(shp) >= 2 and "embedding" not in k:
            fan = int(np.prod(shp[1:]))
            t = (torch.rand(shp, generator=g) * 2 - 1) / fan ** 0.5
        elif "norm" in k and k.endswith("weight"):
            t = 1.0 + 0.1 * (torch.rand(shp, generator=g) * 2 - 1)
        else:
            t = 0.05 * (torch.rand(shp, generator=g) * 2 - 1)
        sd[k] = t.half().contiguous()
    return sd


def _toy_tokenizer(d):
    """a CLIPTokenizer directory (vocab.json + merges.txt) over single characters: 56 tokens"""
    os.makedirs(d)
    chars = list("abcdefghijklmnopqrstuvwxyz,")
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    assert len(vocab) == TXT_CFG["vocab_size"]
    json.dump(vocab, open(os.path.join(d, "vocab.json"), "w"))
    open(os.path.join(d, "merges.txt"), "w").write("#version: 0.2\n")
    json.dump({"model_max_length": 77}, open(os.path.join(d, "tokenizer_config.json"), "w"))


def _write_checkpoint(root):
    from safetensors.torch import save_file
    from oracle import unet_ref as U
    from mvoc_amd import clip as mc, vae as mv
    o = U.I2VGenXLUNet(U.UNetConfig.small4())
    U.init_weights_(o, seed=5)
    unet_sd = {k: v.half().contiguous() for k, v in o.state_dict().items()}
    ucfg = o.config.to_dict()
    # as diffusers writes it: class name, version, a per-block list for attention_head_dim, keys this engine has no use for
    ucfg_json = dict(ucfg, _class_name="I2VGenXLUNet", _diffusers_version="0.27.2", attention_head_dim=[ucfg["attention_head_dim"]] * 4,
                     sample_size=32, num_attention_heads=None)
    sds = {"unet": unet_sd, "vae": _rand_sd(mv.param_shapes(mv.VaeConfig.from_any(VAE_CFG)), 1),
           "image_encoder": _rand_sd(mc.CLIPVisionModelWithProjection(VIS_CFG).param_shapes(), 2),
           "text_encoder": _rand_sd(mc.CLIPTextModel(TXT_CFG).param_shapes(), 3)}
    cfgs = {"unet": ucfg_json, "vae": dict(VAE_CFG, _class_name="AutoencoderKL"), "image_encoder": VIS_CFG, "text_encoder": TXT_CFG}
    names = {"unet": "diffusion_pytorch_model.fp16.safetensors", "vae": "diffusion_pytorch_model.fp16.safetensors",
             "image_encoder": "model.fp16.safetensors", "text_encoder": "model.fp16.safetensors"}
    for sub, sd in sds.items():
        os.makedirs(os.path.join(root, sub))
        json.dump(cfgs[sub], open(os.path.join(root, sub, "config.json"), "w"))
        save_file(sd, os.path.join(root, sub, names[sub]))
    _toy_tokenizer(os.path.join(root, "tokenizer"))
    os.makedirs(os.path.join(root, "scheduler"))
    json.dump({"_class_name": "DDIMScheduler", "num_train_timesteps": 1000, "steps_offset": 1}, open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    return ucfg, sds


def test_inverse_py_without_synthetic_reads_the_checkpoint(tmp_path, monkeypatch):
    from PIL import Image
    sys.path[:0] = [os.path.join(REPO, "i2vgen-xl"), REPO]
    for m in ("utils", "pnp_utils", "inverse", "composite", "pipelines", "pipelines.pipeline_i2vgen_xl"):
        sys.modules.pop(m, None)
    import inverse
    from mvoc_amd.config import OmegaConf
    ckpt = str(tmp_path / "checkpoints" / "i2vgen-xl")
    ucfg, sds = _write_checkpoint(ckpt)
    monkeypatch.setattr(inverse, "PRETRAINED_MODEL_PATH", ckpt)
    rng = np.random.default_rng(0)
    d = tmp_path / "demo" / "clipA" / "clipA"
    d.mkdir(parents=True)
    for i in range(4):
        Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)).save(d / f"{i:05d}.png")

    def run(inv_dir):
        tmpl = OmegaConf.load(os.path.join(REPO, "tests", "data", "inversion_template.yaml"))
        tmpl.data_dir = str(tmp_path)
        tmpl.inv_dir = inv_dir
        tmpl.inverse_config.prompt = "sailboat,ocean"  # goes through the checkpoint's tokenizer + text tower
        entries = [{"active": True, "video_name": "clipA", "video_dir": str(tmp_path / "demo" / "clipA"), "image_size": [64, 64], "n_frames": 4,
                    "recon_config": {"enable_recon": True, "ddim_init_latents_t_idx": 1}}]
        inverse.seed_everything(tmpl.seed)  # as the driver's __main__ does: the VAE samples its latent distribution from the global RNG
        inverse.main(tmpl, entries, torch.device("cuda:0"), synthetic=False, concurrent_entries=1)
        return tmp_path / inv_dir / "i2vgen-xl" / "clipA"

    seen = {}
    real_build = inverse.build_pipeline

    def spy_build(device, synthetic):
        pipe = real_build(device, synthetic)
        seen["pipe"] = pipe
        return pipe

    monkeypatch.setattr(inverse, "build_pipeline", spy_build)
    a = run("inv_from_pretrained")
    pipe = seen["pipe"]
    # (a) the checkpoint's configs were read, every component came from it
    assert tuple(pipe.unet.config.block_out_channels) == tuple(ucfg["block_out_channels"]) != (320, 640, 1280, 1280)
    assert pipe.unet.config.cross_attention_dim == ucfg["cross_attention_dim"]
    assert tuple(pipe.vae.config.block_out_channels) == (64, 64, 128, 128)
    c = pipe.conditioner.clip
    assert c.vision.config.hidden_size == 320 and c.text.config.hidden_size == 64 and c.tokenizer is not None
    assert (a / "ddim_reconstruction.gif").exists()  # VAE decode of the reconstruction ran (frames out)

    # (b) the same tensors through load_state_dict
    def manual_build(device, synthetic):
        from mvoc_amd import clip as mc, vae as mv
        from mvoc_amd.pipeline import I2VGenXLPipeline
        from mvoc_amd.unet import I2VGenXLUNet
        from transformers import CLIPTokenizer
        p = I2VGenXLPipeline(I2VGenXLUNet(ucfg, device=device).load_state_dict(sds["unet"]))
        p.vae = mv.AutoencoderKL(VAE_CFG, device=device).load_state_dict(sds["vae"])
        p.conditioner.vae = mv.VaeCodec(p.vae)
        p.conditioner.clip = mc.ClipCodec(mc.CLIPVisionModelWithProjection(VIS_CFG, device=device).load_state_dict(sds["image_encoder"]),
                                          mc.CLIPTextModel(TXT_CFG, device=device).load_state_dict(sds["text_encoder"]),
                                          CLIPTokenizer.from_pretrained(os.path.join(ckpt, "tokenizer")))
        return p

    monkeypatch.setattr(inverse, "build_pipeline", manual_build)
    b = run("inv_load_state_dict")
    files = sorted(os.listdir(a / "ddim_latents"))
    assert files == sorted(f"ddim_latents_{t}.pt" for t in (1, 201, 401, 601, 801)) == sorted(os.listdir(b / "ddim_latents"))
    for f in files:
        x, y = torch.load(a / "ddim_latents" / f), torch.load(b / "ddim_latents" / f)
        assert x.dtype == torch.float16 and torch.isfinite(x.float()).all() and x.float().abs().max() > 0
        assert torch.equal(x, y), f
    # a checkpoint directory without unet/ weights fails loudly
    from mvoc_amd.pipeline import I2VGenXLPipeline
    os.remove(os.path.join(ckpt, "unet", "diffusion_pytorch_model.fp16.safetensors"))
    with pytest.raises(FileNotFoundError):
        I2VGenXLPipeline.from_pretrained(ckpt, torch_dtype=torch.float16, variant="fp16")
