"""GPU: the HIP VAE (mvoc_amd.vae) against the CPU oracle's restatement of diffusers' AutoencoderKL
(oracle/vae_ref.py -- parity unpinned like the rest of the diffusers half: no diffusers, no checkpoint) on identical
fp16-rounded weights, plus the new kernels it needs against torch.

Tolerance: encoder moments / decoder images rel-L2 <= 3e-3, max-abs <= 2e-2 * max|ref| (the UNet-forward bound of SURVEY 8d:
fp16 kernels with fp32 accumulation vs an fp32 evaluation, ~25 conv / norm layers deep); elementwise kernels bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def dev(t):
    return t.to("cuda", torch.float16).contiguous()


def _metrics(out, ref):
    out, ref = out.float().cpu(), ref.float().cpu()
    assert out.shape == ref.shape and torch.isfinite(out).all()
    return float((out - ref).norm() / ref.norm()), float((out - ref).abs().max() / ref.abs().max())


def _pair(cfg):
    from oracle import vae_ref as V
    from mvoc_amd.vae import AutoencoderKL
    o = V.init_weights_(V.AutoencoderKL(cfg), seed=5)
    for p in o.parameters():
        p.copy_(p.half().float())
    eng = AutoencoderKL(cfg.to_dict()).load_state_dict(o.state_dict())
    return o, eng


@pytest.mark.parametrize("which,n,size", [("tiny", 2, 64), ("tiny", 1, 128), ("full", 1, 128)])
def test_encode_and_decode_vs_oracle(which, n, size):
    from oracle import vae_ref as V
    cfg = V.VaeConfig.tiny() if which == "tiny" else V.VaeConfig()
    o, eng = _pair(cfg)
    g = torch.Generator().manual_seed(size + n)
    x = (torch.rand(n, 3, size, size, generator=g) * 2 - 1).half().float()
    mean_ref, logvar_ref = o.encode_moments(x)
    mean, logvar = eng.encode_moments(x)
    assert tuple(mean.shape) == (n, 4, size // 8, size // 8)
    for name, a, b in (("mean", mean, mean_ref), ("logvar", logvar.float().clamp(-30, 20), logvar_ref)):
        rel, mx = _metrics(a, b)
        print(f"VAE {which} encode {name} n={n} {size}x{size}: rel-L2 {rel:.2e}, max-abs/max {mx:.2e}")
        assert rel <= 3e-3 and mx <= 2e-2, (name, rel, mx)
    z = torch.randn(n, 4, size // 8, size // 8, generator=g).half().float()
    img_ref = o.decode(z)
    img = eng.decode(z)
    assert tuple(img.shape) == (n, 3, size, size)
    rel, mx = _metrics(img, img_ref)
    print(f"VAE {which} decode n={n} {size}x{size}: rel-L2 {rel:.2e}, max-abs/max {mx:.2e}")
    assert rel <= 3e-3 and mx <= 2e-2, (rel, mx)


def test_pipeline_glue_matches_oracle_glue():
    """encode_vae_video / prepare_image_latents / decode_latents through VaeCodec (PIL in, latents [1,4,F,h,w]; latents in,
    video [B,3,F,H,W] out) against the oracle's restatement of the same glue; tensor2vid's three output types"""
    from PIL import Image
    from oracle import vae_ref as V
    from mvoc_amd.vae import VaeCodec, center_crop_wide, preprocess_image, tensor2vid
    o, eng = _pair(V.VaeConfig.tiny())
    codec = VaeCodec(eng)
    rng = np.random.default_rng(0)
    frames = [Image.fromarray(rng.integers(0, 255, (96, 160, 3), dtype=np.uint8)) for _ in range(3)]
    H = W = 64
    xs = torch.cat([preprocess_image(center_crop_wide(f, (W, H))) for f in frames])
    assert tuple(xs.shape) == (3, 3, H, W) and float(xs.min()) >= -1 and float(xs.max()) <= 1
    gen = torch.Generator().manual_seed(9)
    noise = torch.randn(3, 4, H // 8, W // 8, generator=gen)
    lat = codec.encode_video(frames, H, W, generator=torch.Generator().manual_seed(9))
    assert tuple(lat.shape) == (1, 4, 3, H // 8, W // 8) and lat.dtype == torch.float16
    ref = V.encode_frames(o, xs.half().float(), noise.half().float())
    rel, _ = _metrics(lat, ref)
    print(f"encode_vae_video vs oracle: rel-L2 {rel:.2e}")
    assert rel <= 4e-3, rel
    il = codec.image_latents(frames[0], 4, H, W, generator=torch.Generator().manual_seed(9))
    assert tuple(il.shape) == (1, 4, 4, H // 8, W // 8)
    assert torch.equal(il[:, :, 0], lat[:, :, 0])  # same image, same noise
    for k in (1, 2, 3):
        assert torch.equal(il[:, :, k], torch.full_like(il[:, :, k], k / 3))
    z = torch.randn(1, 4, 3, H // 8, W // 8, generator=gen).half()
    video = codec.decode(z.cuda())
    vref = V.decode_latents(o, z.float())
    assert video.dtype == torch.float32 and tuple(video.shape) == (1, 3, 3, H, W)
    rel, mx = _metrics(video, vref)
    print(f"decode_latents vs oracle: rel-L2 {rel:.2e}")
    assert rel <= 4e-3, rel
    pil = tensor2vid(video, "pil")
    assert len(pil) == 1 and len(pil[0]) == 3 and pil[0][0].size == (W, H)
    assert tensor2vid(video, "np").shape == (1, 3, H, W, 3) and tuple(tensor2vid(video, "pt").shape) == (1, 3, 3, H, W)


def test_pipeline_returns_frames_with_a_vae():
    """__call__ with output_type='pil' decodes through the HIP VAE and returns the reference's structure: .frames[0] is the
    list of PIL frames (pipeline_i2vgen_xl.py:1207-1216); without a VAE the conditioner refuses"""
    from oracle import unet_ref as U
    from mvoc_amd.pipeline import I2VGenXLPipeline, SyntheticConditioner
    from mvoc_amd.unet import I2VGenXLUNet
    from mvoc_amd.vae import AutoencoderKL, VaeCodec
    from mvoc_amd.schedulers import DDIMScheduler
    import mvoc_amd.vae as mv
    eng = I2VGenXLUNet(U.UNetConfig.small4().to_dict()).init_random(3)
    pipe = I2VGenXLPipeline(eng, DDIMScheduler(), conditioner=SyntheticConditioner(eng.device, 64))
    kw = dict(prompt="a", image="img", height=64, width=64, num_frames=2, num_inference_steps=2, guidance_scale=7.5, target_fps=8)
    with pytest.raises(NotImplementedError):
        pipe(output_type="pil", **kw)
    vae = AutoencoderKL(mv.VaeConfig(block_out_channels=(64, 64, 128, 128), layers_per_block=1, norm_num_groups=8)).init_random(2)
    pipe.conditioner.vae = VaeCodec(vae)
    frames = pipe(output_type="pil", **kw).frames
    assert len(frames) == 1 and len(frames[0]) == 2 and frames[0][0].size == (64, 64) and frames[0][0].mode == "RGB"


# ---- the kernels the VAE adds ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tile", [0, 11, 13, 81, 82])
def test_conv3x3_bottom_right_padding_exact(tile):
    """Downsample2D(padding=0): F.pad(x, (0,1,0,1)) + conv2d(stride=2, padding=0) == the gather's pad_mode=1, bit-exact on
    integer data (any shifted tap or a wrong border shows as a wrong integer)"""
    from mvoc_amd import ops
    from mvoc_amd.unet import pack_conv3x3
    g = torch.Generator().manual_seed(tile)
    n, c, cout, h, w = (4, 128, 128, 16, 16) if tile < 80 else (16, 128, 256, 64, 64)
    x = torch.randint(-1, 2, (n, c, h, w), generator=g).float()
    wt = torch.randint(-1, 2, (cout, c, 3, 3), generator=g).float()
    b = torch.randint(-4, 5, (cout,), generator=g).float()
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), wt, b, stride=2, padding=0)
    assert ref.abs().max() < 2048
    rows = x.permute(0, 2, 3, 1).reshape(n * h * w, c)
    out, ho, wo = ops.conv3x3(dev(rows), pack_conv3x3(dev(wt)), dev(b), nimg=n, h=h, wd=w, stride=2, pad_mode=1, n_store=cout, tile=tile,
                              split_k=1)
    assert (ho, wo) == (h // 2, w // 2)
    assert torch.equal(out.float().cpu().reshape(n, ho, wo, cout).permute(0, 3, 1, 2), ref)


def test_softmax_rows_and_small_kernels():
    from mvoc_amd import ops
    from mvoc_amd._ffi import check, lib
    g = torch.Generator().manual_seed(1)
    s = (torch.randn(300, 1024, generator=g) * 3).half()
    s[5, 7] = 40.0  # a dominant score
    d = dev(s)
    ops.softmax_rows(d)
    ref = torch.softmax(s.float(), dim=-1)
    assert (d.float().cpu() - ref).abs().max() < 1e-3 and abs(float(d.float().sum(dim=1).mean()) - 1) < 2e-3
    x, w, b = torch.randn(1000, 8, generator=g).half(), torch.randn(8, 8, generator=g).half(), torch.randn(8, generator=g).half()
    out = ops.conv1x1_small(dev(x), dev(w), dev(b))
    assert (out.float().cpu() - (x.float() @ w.float().t() + b.float())).abs().max() < 4e-3
    img = torch.randn(3, 5, 6, 7, generator=g).half()
    tok = ops.image_to_tokens(dev(img))
    assert torch.equal(tok.cpu(), img.permute(0, 2, 3, 1).reshape(-1, 5))
    assert torch.equal(ops.tokens_to_image(tok, 3, 5, 6, 7).cpu(), img)
    # DiagonalGaussianDistribution.sample() and python-float scaling: the fp16 eager chains, bit for bit (exp: <= 1 ulp)
    mean, logvar, noise = (torch.randn(4096, generator=g).half() for _ in range(3))
    logvar[:3] = torch.tensor([-40.0, 30.0, 0.0]).half()
    out = torch.empty(4096, dtype=torch.float16, device="cuda")
    dm, dl, dn = dev(mean), dev(logvar), dev(noise)
    check(lib.mvoc_gaussian_sample_f16(dm.data_ptr(), dl.data_ptr(), dn.data_ptr(), out.data_ptr(), 4096, ops._stream()), "sample")
    ref = mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise  # CPU half ops: fp32 compute, one rounding per op
    o32, r32 = out.cpu().float(), ref.float()
    assert float(((o32 - r32).abs() / r32.abs().clamp_min(1e-3)).max()) <= 1.5e-3  # at most an fp16 ulp (expf vs torch.exp)
    assert float((o32 != r32).float().mean()) < 0.02
    sc = torch.empty(4096, dtype=torch.float16, device="cuda")
    check(lib.mvoc_scale_f16(dm.data_ptr(), sc.data_ptr(), 4096, 0.18215, ops._stream()), "scale")
    assert torch.equal(sc.cpu(), mean * 0.18215)


def test_mask_preprocess_on_device_matches_g9(golden_dir):
    """mask preprocessing with the resize / threshold / scaling on the GPU (PNG decode on the host) against G9 =
    the reference's own mask_preprocess output on the boat_surf masks -- bit for bit, float and bool"""
    import os
    from mvoc_amd.utils import mask_preprocess
    g = np.load(os.path.join(golden_dir, "g9_boat_surf_masks.npz"))
    for name in ("boat_mask", "surf_mask"):
        fl, bl = mask_preprocess(os.path.join(golden_dir, "boat_surf_masks", name), "cuda:0", torch.float16, 1, 4, 16, downscale=8)
        assert fl.is_cuda and fl.dtype == torch.float16 and bl.dtype == torch.bool and tuple(fl.shape) == (1, 4, 16, 90, 160)
        want = (torch.from_numpy(g[f"{name}_90x160_float_u8"]).float() / 255).half()
        for c in range(4):
            assert torch.equal(fl[0, c].cpu(), want)
            assert torch.equal(bl[0, c].cpu(), torch.from_numpy(g[f"{name}_90x160_bool"]))
    one = os.path.join(golden_dir, "boat_surf_masks", "surf_mask", "00002.png")
    fl, bl = mask_preprocess(one, "cuda:0", torch.float16, 1, 4, 3, downscale=8)
    assert tuple(fl.shape) == (1, 4, 3, 90, 160) and torch.equal(bl[0, 0, 1].cpu(), torch.from_numpy(g["surf_mask_90x160_bool"][2]))
