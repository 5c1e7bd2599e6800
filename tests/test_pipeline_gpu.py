"""GPU: the three denoising loops (mvoc_amd.pipeline) against the oracle's loop restatement on a toy UNet, the
graph-captured iteration against the eager one, and the drop-in drivers (i2vgen-xl/inverse.py, composite.py) end to end
on a tiny synthetic group config.

Tolerances (fp16 HIP vs fp32 CPU UNet inside an fp16 loop): latents after n<=5 DDIM steps max-abs <= 3e-2 (the
per-step UNet tolerance of test_unet_gpu.py accumulated; values are O(1))."""
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pair():
    from oracle import unet_ref as U
    from mvoc_amd.unet import I2VGenXLUNet
    o = U.I2VGenXLUNet(U.UNetConfig.small4())
    U.init_weights_(o, seed=9)
    for p in o.parameters():
        p.copy_(p.half().float())
    eng = I2VGenXLUNet(o.config.to_dict()).load_state_dict(o.state_dict())
    return o, eng


def _cond(g, b, f, h, w, cd=64):
    return dict(pe=torch.randn(1, 7, cd, generator=g).half(), ne=torch.randn(1, 7, cd, generator=g).half(),
                ie=torch.randn(1, 1, cd, generator=g).half(), il=torch.randn(1, 4, f, h, w, generator=g).half())


@pytest.mark.parametrize("cfg", [1.0, 7.5])
@pytest.mark.parametrize("graphs", [False, True])
def test_invert_and_sample_vs_oracle(cfg, graphs, tmp_path):
    from oracle import loops_ref, sched_ref
    from mvoc_amd.pipeline import I2VGenXLPipeline
    from mvoc_amd.schedulers import DDIMInverseScheduler, DDIMScheduler
    o, eng = _pair()
    g = torch.Generator().manual_seed(3)
    f, h, w = 3, 8, 8
    c = _cond(g, 1, f, h, w)
    x0 = torch.randn(1, 4, f, h, w, generator=g).half()
    pipe = I2VGenXLPipeline(eng, DDIMInverseScheduler(), use_graphs=graphs)
    out_dir = str(tmp_path / "lat")
    inv = pipe.invert(height=h * 8, width=w * 8, num_frames=f, num_inference_steps=5, guidance_scale=cfg, target_fps=8,
                      latents=x0.cuda(), prompt_embeds=c["pe"].cuda(), negative_prompt_embeds=c["ne"].cuda(),
                      image_embeddings=c["ie"].cuda(), image_latents=c["il"].cuda(), return_dict=False, output_dir=out_dir)
    assert inv.shape == (1, 5, 4, f, h, w)

    def unet_fn(inp, t):
        b = inp.shape[0]
        pe = torch.cat([c["ne"], c["pe"]]) if b == 2 else c["pe"]
        ie = torch.cat([torch.zeros_like(c["ie"]), c["ie"]]) if b == 2 else c["ie"]
        il = torch.cat([c["il"]] * b)
        return o(inp.float(), int(t), torch.tensor([8] * b), il.float(), ie.float(), pe.float())[0].half()

    saved, ref_seq = loops_ref.invert_loop(unet_fn, sched_ref.DDIMInverseSchedulerRef(), x0, 5, cfg)
    assert (inv.cpu().float() - ref_seq.float()).abs().max() < 3e-2
    # files: one per inverse timestep, holding the latent AT that noise level, fp16 [1,4,F,h,w]
    ts = sorted(saved)
    assert ts == [1, 201, 401, 601, 801]
    for i, t in enumerate(ts):
        lat = torch.load(os.path.join(out_dir, f"ddim_latents_{t}.pt"))
        assert lat.dtype == torch.float16 and tuple(lat.shape) == (1, 4, f, h, w)
        assert torch.equal(lat, inv[0, len(ts) - 1 - i][None].cpu())
    # reconstruction sampling from the noisiest latent with the forward scheduler
    pipe.scheduler = DDIMScheduler()
    rec = pipe(height=h * 8, width=w * 8, num_frames=f, num_inference_steps=5, guidance_scale=cfg, target_fps=8,
               latents=inv[:, 0], prompt_embeds=c["pe"].cuda(), negative_prompt_embeds=c["ne"].cuda(),
               image_embeddings=c["ie"].cuda(), image_latents=c["il"].cuda(), output_type="latent", ddim_init_latents_t_idx=1).frames
    ref = loops_ref.sample_loop(unet_fn, sched_ref.DDIMSchedulerRef(), inv[:, 0].cpu(), 5, cfg, ddim_init_latents_t_idx=1)
    assert (rec.cpu().float() - ref.float()).abs().max() < 3e-2


def test_graph_replay_equals_eager():
    from mvoc_amd.pipeline import I2VGenXLPipeline
    from mvoc_amd.schedulers import DDIMInverseScheduler
    _, eng = _pair()
    g = torch.Generator().manual_seed(4)
    f, h, w = 2, 8, 8
    c = _cond(g, 1, f, h, w)
    x0 = torch.randn(1, 4, f, h, w, generator=g).half().cuda()
    outs = []
    for graphs in (False, True):
        pipe = I2VGenXLPipeline(eng, DDIMInverseScheduler(), use_graphs=graphs)
        outs.append(pipe.invert(height=64, width=64, num_frames=f, num_inference_steps=4, guidance_scale=1.0, latents=x0,
                                prompt_embeds=c["pe"].cuda(), negative_prompt_embeds=c["ne"].cuda(), image_embeddings=c["ie"].cuda(),
                                image_latents=c["il"].cuda(), return_dict=False, output_dir=None))
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("graphs", [False, True])
def test_composition_vs_oracle(graphs):
    from oracle import loops_ref, sched_ref
    from oracle.pnp_model_ref import PnPState, install_pnp
    from mvoc_amd import pnp_utils
    from mvoc_amd.pipeline import I2VGenXLPipeline
    from mvoc_amd.schedulers import DDIMScheduler
    o, eng = _pair()
    g = torch.Generator().manual_seed(5)
    f, h, w, cd, n = 3, 8, 8, 64, 5
    cond = dict(encoder_hidden_states=torch.randn(5, 7, cd, generator=g).half(), image_embeddings=torch.randn(5, f, cd, generator=g).half(),
                image_latents_first=torch.randn(5, 4, f, h, w, generator=g).half(), image_latents=torch.randn(5, 4, f, h, w, generator=g).half())
    u8 = torch.randint(0, 256, (2, f, h, w), generator=g)
    masks = [((u8[j].float() / 255).half()[None, None].repeat(1, 4, 1, 1, 1), (u8[j] > 10)[None, None].repeat(1, 4, 1, 1, 1)) for j in range(2)]
    s = DDIMScheduler()
    s.set_timesteps(n)
    src = {(k, int(t)): torch.randn(1, 4, f, h, w, generator=g).half() for k in range(3) for t in s.timesteps}
    x0 = torch.randn(1, 4, f, h, w, generator=g).half()
    # ---- oracle: schedules as composite.init_pnp builds them (prefixes of the full timestep list)
    rs = sched_ref.DDIMSchedulerRef()
    rs.set_timesteps(n)
    st = PnPState(conv_schedule=rs.timesteps[:1], spatial_schedule=rs.timesteps[:3], temporal_schedule=rs.timesteps[:4])
    install_pnp(o, st)
    st.masks = masks

    def unet_fn(inp, t):
        st.t = int(t)
        return o.forward_ext(inp.float(), int(t), torch.tensor([8] * 5), cond["image_latents_first"].float(), cond["image_latents"].float(),
                             cond["image_embeddings"].float(), cond["encoder_hidden_states"].float())[0].half()

    ref = loops_ref.composition_loop(unet_fn, sched_ref.DDIMSchedulerRef(), x0, lambda t: src[(0, t)], lambda j, t: src[(1 + j, t)],
                                     [m[0] for m in masks], n, guidance_scale=9.0, ddim_init_latents_t_idx=1, fusion_steps=(0, 2),
                                     random_noise_ratio=0.3, obj_random_noise_fusion=True)
    # ---- HIP pipeline
    pipe = I2VGenXLPipeline(eng, DDIMScheduler(), use_graphs=graphs)
    pnp_utils.register_temp_attention_pnp(pipe, s.timesteps[:4], False)
    pnp_utils.register_spatial_attention_pnp(pipe, s.timesteps[:3], False)
    pnp_utils.register_temp_conv_injection(pipe, s.timesteps[:1])
    pnp_utils.register_out_conv_injection(pipe, s.timesteps[:1])
    pnp_utils.register_resnet_injection(pipe, s.timesteps[:1])
    for (k, t), v in src.items():
        pipe.latent_cache.put(f"/virtual/src{k}", t, v.cuda())
    pipe.latent_cache.write_files = False

    class Cond:  # conditioner returning the prepared tensors in the reference's assembly order
        def __init__(self):
            self.k = {"p": 0}

        def encode_prompt(self, prompt, negative_prompt=None):
            if prompt == "edit":
                return cond["encoder_hidden_states"][4:5].cuda(), cond["encoder_hidden_states"][3:4].cuda()
            return cond["encoder_hidden_states"][0:1].cuda(), None

        def image_latents(self, image, num_frames, height, width):
            idx, fr, first = image
            return cond["image_latents_first" if first else "image_latents"][idx:idx + 1].cuda()

        def encode_image(self, image):
            idx, fr, first = image
            return cond["image_embeddings"][idx:idx + 1, fr:fr + 1].cuda()

    pipe.conditioner = Cond()
    pipe.latent_cache.write_files = False
    # the reference assembles [bg, obj1, obj2, uncond(neg), cond]; inv prompts are identical for bg/objects
    cond["encoder_hidden_states"][1] = cond["encoder_hidden_states"][0]
    cond["encoder_hidden_states"][2] = cond["encoder_hidden_states"][0]
    cond["image_embeddings"][3] = 0
    cond["image_latents_first"][3] = cond["image_latents_first"][4]
    cond["image_latents"][4] = cond["image_latents_first"][4]  # main branch: both built from the edited first frame (:1392-1410)
    cond["image_latents"][3] = cond["image_latents"][4]
    ref = loops_ref.composition_loop(unet_fn, sched_ref.DDIMSchedulerRef(), x0, lambda t: src[(0, t)], lambda j, t: src[(1 + j, t)],
                                     [m[0] for m in masks], n, guidance_scale=9.0, ddim_init_latents_t_idx=1, fusion_steps=(0, 2),
                                     random_noise_ratio=0.3, obj_random_noise_fusion=True)
    out = pipe.sample_with_pnp_pipeline_with_edit_prompt_extraction_with_attn_injection(
        prompt="edit", main_first_image=(4, 0, True), main_image_list=[(4, k, False) for k in range(f)],
        background_first_image=(0, 0, True), background_image_list=[(0, k, False) for k in range(f)],
        objs_first_image=[(1, 0, True), (2, 0, True)],
        objs_image_list=[[(1, k, False) for k in range(f)], [(2, k, False) for k in range(f)]],
        height=h * 8, width=w * 8, num_frames=f, num_inference_steps=n, guidance_scale=9.0, negative_prompt="neg", target_fps=8,
        latents=x0.cuda(), output_type="latent", ddim_init_latents_t_idx=1, ddim_inv_prompt="", fusion_steps=(0, 2),
        random_noise_ratio=0.3, obj_random_noise_fusion=True, bg_inv_latents_path="/virtual/src0",
        obj_ddim_latents_path=["/virtual/src1", "/virtual/src2"], obj_ddim_latents_idx_offset=[0, 0], obj_masks_tensors=masks).frames
    d = (out.cpu().float() - ref.float()).abs().max()
    print(f"composition after 4 steps (fusion + all injections, cfg 9): max-abs {float(d):.2e}")
    assert d < 3e-2, float(d)  # composition after <= 5 steps


def test_dropin_drivers_end_to_end(tmp_path):
    """inverse.py then composite.py of this repo on a tiny hand-made group (3 videos), synthetic weights/conditioning"""
    from PIL import Image
    sys.path[:0] = [os.path.join(REPO, "i2vgen-xl"), REPO]
    for m in ("utils", "pnp_utils", "inverse", "composite", "pipelines", "pipelines.pipeline_i2vgen_xl"):
        sys.modules.pop(m, None)
    import composite
    import inverse
    from mvoc_amd.config import OmegaConf
    data = tmp_path
    rng = np.random.default_rng(0)
    for name in ("clipA", "clipB"):
        d = data / "demo" / name / name
        d.mkdir(parents=True)
        for i in range(4):
            Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)).save(d / f"{i:05d}.png")
        (d / "edited_first_frame").mkdir()
        Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)).save(d / "edited_first_frame" / "00000.png")
        for mname in ("m1", "m2"):
            md = data / "demo" / name / mname
            md.mkdir()
            for i in range(4):
                m = np.zeros((64, 64), np.uint8)
                m[8 + 4 * i:40 + 4 * i, 16:48] = 255
                Image.fromarray(m).save(md / f"{i:05d}.png")
    tmpl = OmegaConf.load(os.path.join(REPO, "tests", "data", "inversion_template.yaml"))
    tmpl.data_dir = str(data)
    entries = [{"active": True, "force_recompute_latents": True, "video_name": n, "video_dir": str(data / "demo" / n), "image_size": [64, 64],
                "n_frames": 4, "recon_config": {"enable_recon": n == "clipA", "ddim_init_latents_t_idx": 1}} for n in ("clipA", "clipB")]
    entries.append({"active": False, "video_name": "skipped"})
    inverse.main(tmpl, entries, torch.device("cuda:0"), synthetic=True)
    for n in ("clipA", "clipB"):
        files = sorted(os.listdir(data / "inversions" / "i2vgen-xl" / n / "ddim_latents"))
        assert files == sorted(f"ddim_latents_{t}.pt" for t in (1, 201, 401, 601, 801))
    # --batch_entries 2: both clips inverted in one batched loop into a second tree; same latents (batch-size noise only)
    tmpl2 = OmegaConf.load(os.path.join(REPO, "tests", "data", "inversion_template.yaml"))
    tmpl2.data_dir = str(data)
    tmpl2.inv_dir = "inversions_batched"
    entries2 = [dict(e, force_recompute_latents=False, recon_config={"enable_recon": False}) if e.get("active") else e for e in entries]
    inverse.main(tmpl2, entries2, torch.device("cuda:0"), synthetic=True, batch_entries=2)
    for n in ("clipA", "clipB"):
        for t in (1, 201, 401, 601, 801):
            a = torch.load(data / "inversions" / "i2vgen-xl" / n / "ddim_latents" / f"ddim_latents_{t}.pt")
            b = torch.load(data / "inversions_batched" / "i2vgen-xl" / n / "ddim_latents" / f"ddim_latents_{t}.pt")
            assert a.shape == b.shape and (a.float() - b.float()).abs().max() < 3e-2
    assert (data / "inversions" / "i2vgen-xl" / "clipA" / "ddim_reconstruction_latents.pt").exists()
    ct = OmegaConf.load(os.path.join(REPO, "tests", "data", "composite_template.yaml"))
    ct.data_dir = str(data)
    centry = {"active": True, "task_name": "T", "video_name": "clipA", "image_size": [64, 64],
              "edited_first_frame_path": "demo/clipA/clipA/edited_first_frame/00000.png", "editing_prompt": "a b", "edited_video_name": "out",
              "ddim_init_latents_t_idx": 0, "pnp_f_t": 0.2, "pnp_spatial_attn_t": 1.0, "pnp_temp_attn_t": 1.0, "random_noise_ratio": 0.0,
              "obj_mask_path": ["demo/clipA/m1", "demo/clipA/m2"], "obj_width_height": [[64, 64], [64, 64]],
              "obj_ddim_latents_path": ["inversions/i2vgen-xl/clipA/ddim_latents", "inversions/i2vgen-xl/clipB/ddim_latents"],
              "bg_ddim_latents_path": "inversions/i2vgen-xl/clipB/ddim_latents", "edited_contorl_frame_path_main": "demo/clipA/clipA",
              "edited_contorl_frame_path_background": "demo/clipB/clipB", "edited_contorl_frame_path": ["demo/clipA/clipA", "demo/clipB/clipB"],
              "fusion_step": [0, 1]}
    composite.main(ct, [centry], torch.device("cuda:0"), synthetic=True)
    out_root = data / "Results" / "T" / "i2vgen-xl" / "clipA" / "out"
    sub = os.listdir(out_root)
    assert sub == ["ddim_init_latents_t_idx_0_nsteps_5_cfg_9.0_pnpf0.2_pnps1.0_pnpt1.0_ratio0.0noise_fusion_step0-1"]
    lat = torch.load(out_root / sub[0] / "video_latents.pt")
    assert tuple(lat.shape) == (1, 4, 4, 8, 8) and torch.isfinite(lat.float()).all()
    # with a VAE (seeded weights of the real architecture) the drivers go frames -> latents -> frames like the reference:
    # the source clips are VAE-encoded (encode_vae_video), results are decoded and written as gif + png (composite.py:217-224)
    # ... and with CLIP towers (seeded ViT-H/14 weights) the conditioning frames go through the batched vision tower
    os.environ["MVOC_SYNTHETIC_VAE"] = os.environ["MVOC_SYNTHETIC_CLIP"] = "1"
    try:
        tmpl3 = OmegaConf.load(os.path.join(REPO, "tests", "data", "inversion_template.yaml"))
        tmpl3.data_dir = str(data)
        tmpl3.inv_dir = "inversions_vae"
        e3 = [dict(entries[0], recon_config={"enable_recon": True, "ddim_init_latents_t_idx": 1})]
        inverse.main(tmpl3, e3, torch.device("cuda:0"), synthetic=True)
        assert (data / "inversions_vae" / "i2vgen-xl" / "clipA" / "ddim_reconstruction.gif").exists()
        composite.main(ct, [dict(centry, edited_video_name="out_frames")], torch.device("cuda:0"), synthetic=True)
    finally:
        del os.environ["MVOC_SYNTHETIC_VAE"], os.environ["MVOC_SYNTHETIC_CLIP"]
    fr = data / "Results" / "T" / "i2vgen-xl" / "clipA" / "out_frames" / sub[0]
    assert sorted(os.listdir(fr)) == ["video.gif"] + [f"video_{i:05d}.png" for i in range(4)]
    assert Image.open(fr / "video_00000.png").size == (64, 64)


def test_composite_py_under_torchrun_two_ranks(tmp_path):
    """`scripts/run_group_composition.sh NGPU=2`'s launch: composite.py under torch.distributed.run, two ranks, the group's
    entries dealt round-robin (BASELINE configs[4]: one composition per GPU, no collectives).  On this one-GPU box both ranks
    share the device (MVOC_ALLOW_GPU_SHARING=1); each must write exactly its own entry's result"""
    import subprocess
    from PIL import Image
    sys.path[:0] = [os.path.join(REPO, "i2vgen-xl"), REPO]
    for m in ("utils", "pnp_utils", "inverse", "composite", "pipelines", "pipelines.pipeline_i2vgen_xl"):
        sys.modules.pop(m, None)
    import inverse
    from mvoc_amd.config import OmegaConf
    data = tmp_path
    rng = np.random.default_rng(1)
    for name in ("clipA", "clipB"):
        d = data / "demo" / name / name
        d.mkdir(parents=True)
        for i in range(4):
            Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)).save(d / f"{i:05d}.png")
        (d / "edited_first_frame").mkdir()
        Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)).save(d / "edited_first_frame" / "00000.png")
        for mname in ("m1", "m2"):
            md = data / "demo" / name / mname
            md.mkdir()
            for i in range(4):
                m = np.zeros((64, 64), np.uint8)
                m[8 + 4 * i:40 + 4 * i, 16:48] = 255
                Image.fromarray(m).save(md / f"{i:05d}.png")
    tmpl = OmegaConf.load(os.path.join(REPO, "tests", "data", "inversion_template.yaml"))
    tmpl.data_dir = str(data)
    entries = [{"active": True, "force_recompute_latents": True, "video_name": n, "video_dir": str(data / "demo" / n), "image_size": [64, 64],
                "n_frames": 4, "recon_config": {"enable_recon": False}} for n in ("clipA", "clipB")]
    inverse.main(tmpl, entries, torch.device("cuda:0"), synthetic=True)
    torch.cuda.empty_cache()
    tpl = open(os.path.join(REPO, "tests", "data", "composite_template.yaml")).read().replace("REPLACED_BY_TEST", str(data))
    (data / "composite_template.yaml").write_text(tpl)
    base = {"active": True, "task_name": "T", "image_size": [64, 64], "editing_prompt": "a b", "ddim_init_latents_t_idx": 0,
            "pnp_f_t": 0.2, "pnp_spatial_attn_t": 1.0, "pnp_temp_attn_t": 1.0, "random_noise_ratio": 0.0,
            "obj_mask_path": ["demo/clipA/m1", "demo/clipA/m2"], "obj_width_height": [[64, 64], [64, 64]],
            "obj_ddim_latents_path": ["inversions/i2vgen-xl/clipA/ddim_latents", "inversions/i2vgen-xl/clipB/ddim_latents"],
            "edited_contorl_frame_path": ["demo/clipA/clipA", "demo/clipB/clipB"], "fusion_step": [0, 1]}
    group = []
    for k, (main, bg) in enumerate((("clipA", "clipB"), ("clipB", "clipA"))):
        group.append(dict(base, video_name=main, edited_video_name=f"out{k}",
                          edited_first_frame_path=f"demo/{main}/{main}/edited_first_frame/00000.png",
                          bg_ddim_latents_path=f"inversions/i2vgen-xl/{bg}/ddim_latents",
                          edited_contorl_frame_path_main=f"demo/{main}/{main}", edited_contorl_frame_path_background=f"demo/{bg}/{bg}"))
    (data / "group.json").write_text(json.dumps(group))
    env = dict(os.environ, MVOC_ALLOW_GPU_SHARING="1", PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(REPO, "i2vgen-xl", "composite.py"), "--template_config",
           str(data / "composite_template.yaml"), "--configs_json", str(data / "group.json"), "--synthetic"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=os.path.join(REPO, "i2vgen-xl"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    for k, main in enumerate(("clipA", "clipB")):
        root = data / "Results" / "T" / "i2vgen-xl" / main / f"out{k}"
        sub = os.listdir(root)
        assert len(sub) == 1
        lat = torch.load(root / sub[0] / "video_latents.pt")
        assert tuple(lat.shape) == (1, 4, 4, 8, 8) and torch.isfinite(lat.float()).all()
    # each entry was handled by exactly one rank (round-robin: entry k -> rank k)
    log = r.stdout + r.stderr
    assert log.count("out0") >= 1 and log.count("out1") >= 1


def test_invert_many_matches_separate_inversions(tmp_path):
    """batched inversion of three clips (UNet batch 3) == three inversions at batch 1: same files, same return values up to
    the GEMM-tile / accumulation-order noise of a different batch size (5 steps: <= 3e-2 like the loop-vs-oracle tests);
    with the elementwise stand-in UNet of the G8 tests the two are bit-identical"""
    from mvoc_amd.pipeline import I2VGenXLPipeline
    from mvoc_amd.schedulers import DDIMInverseScheduler
    _, eng = _pair()
    g = torch.Generator().manual_seed(5)
    f, h, w = 3, 8, 8
    x0 = [torch.randn(1, 4, f, h, w, generator=g).half().cuda() for _ in range(3)]
    prompts, images = ["a", "", "c d"], ["img0", "img1", "img2"]

    class Cond:
        vae_scale_factor, cross_attention_dim = 8, 64

        def _t(self, key, shape):
            return torch.randn(shape, generator=torch.Generator().manual_seed(sum(str(key).encode()))).half().cuda()

        def encode_prompt(self, prompt, negative_prompt=None):
            return self._t("p" + str(prompt), (1, 7, 64)), self._t("n" + str(negative_prompt), (1, 7, 64))

        def encode_image(self, image):
            return self._t("i" + str(image), (1, 1, 64))

        def image_latents(self, image, num_frames, height, width):
            return self._t("l" + str(image), (1, 4, num_frames, height // 8, width // 8))

    pipe = I2VGenXLPipeline(eng, DDIMInverseScheduler(), conditioner=Cond(), use_graphs=True)
    kw = dict(height=h * 8, width=w * 8, target_fps=8, num_frames=f, num_inference_steps=5, guidance_scale=1.0)
    single = [pipe.invert(prompt=p, image=im, latents=x, return_dict=False, output_dir=str(tmp_path / f"s{j}"), **kw)
              for j, (p, im, x) in enumerate(zip(prompts, images, x0))]
    many = pipe.invert_many(prompts, images, x0, [str(tmp_path / f"m{j}") for j in range(3)], **kw)
    for j in range(3):
        assert many[j].shape == single[j].shape == (1, 5, 4, f, h, w)
        assert (many[j].float() - single[j].float()).abs().max() < 3e-2
        for t in (1, 201, 401, 601, 801):
            a = torch.load(tmp_path / f"m{j}" / f"ddim_latents_{t}.pt")
            assert a.dtype == torch.float16 and tuple(a.shape) == (1, 4, f, h, w)
            assert (a.float() - torch.load(tmp_path / f"s{j}" / f"ddim_latents_{t}.pt").float()).abs().max() < 3e-2
    with pytest.raises(NotImplementedError):
        pipe.invert_many(prompts, images, x0, ["a", "b", "c"], **dict(kw, guidance_scale=7.5))
    # the three clips as three concurrent batch-1 loops on three HIP streams: each clip replays exactly the launches of invert()
    # -> return values and files BIT-IDENTICAL to the one-by-one pass (twice: the second call reuses the captured iterations)
    for rep in range(2):  # (concurrency_hint=False: exactly invert()'s launches; the hint only changes split-K / tile choices of
        # GEMMs far larger than this network's)
        conc = pipe.invert_concurrent(prompts, images, x0, [str(tmp_path / f"c{rep}{j}") for j in range(3)], concurrency_hint=bool(rep), **kw)
        for j in range(3):
            assert torch.equal(conc[j], single[j]), (rep, j)
            for t in (1, 201, 401, 601, 801):
                assert torch.equal(torch.load(tmp_path / f"c{rep}{j}" / f"ddim_latents_{t}.pt"), torch.load(tmp_path / f"s{j}" / f"ddim_latents_{t}.pt"))
    # ... with classifier-free guidance too (the stock iteration duplicates the sample), and a single clip
    g75 = dict(kw, guidance_scale=7.5)
    s75 = pipe.invert(prompt=prompts[0], image=images[0], latents=x0[0], return_dict=False, output_dir=None, **g75)
    c75 = pipe.invert_concurrent(prompts[:1], images[:1], x0[:1], [None], **g75)
    assert torch.equal(c75[0], s75)


# ---- G8: the HIP pipeline's loops against the REFERENCE's own loops (tests/golden/g8_loops.npz) -----------------------
def _g8_pipe(scheduler):
    """the tiny HIP engine (attribute tree for the hooks) with its forward replaced by the fixture's elementwise stand-in
    UNet -- what remains under test is everything else of a loop iteration: conditioning assembly, batch order, latent
    cache files, fusion / CFG / (inverse-)DDIM kernels"""
    from g8_common import fake_unet, seeded, prompt_key
    from mvoc_amd.pipeline import I2VGenXLPipeline
    _, eng = _pair()
    D = 64
    calls = []

    def fwd(sample, timestep, fps, image_latents, image_embeddings=None, encoder_hidden_states=None, **kw):
        calls.append(sample.clone())
        return (fake_unet(sample, timestep, encoder_hidden_states, fps, image_latents, image_latents, image_embeddings),)

    def fwd_ext(sample, timestep, fps, image_latents_first, image_latents, image_embeddings=None, encoder_hidden_states=None, **kw):
        calls.append(sample.clone())
        return (fake_unet(sample, timestep, encoder_hidden_states, fps, image_latents_first, image_latents, image_embeddings),)

    eng.forward, eng.forward_ext = fwd, fwd_ext
    eng.prepare_conditioning = lambda *a, **k: None

    class Cond:
        vae_scale_factor, cross_attention_dim = 8, D

        def encode_prompt(self, prompt, negative_prompt=None):
            return seeded(prompt_key(prompt), (1, 7, D)).cuda(), seeded(1000 + prompt_key(negative_prompt), (1, 7, D)).cuda()

        def encode_image(self, image):
            return seeded(300 + int(image), (1, D))[None].cuda()

        def image_latents(self, image, num_frames, height, width):
            h, w = height // 8, width // 8
            first = (seeded(500 + int(image), (1, 4, h, w)) * 0.18215)[:, :, None]
            ramp = [torch.full((1, 4, 1, h, w), k / (num_frames - 1)).half() for k in range(1, num_frames)]
            return torch.cat([first] + ramp, 2).cuda()

    pipe = I2VGenXLPipeline(eng, scheduler, conditioner=Cond(), use_graphs=False)
    return pipe, calls


@pytest.mark.parametrize("tag,gs", [("inv_cfg1", 1.0), ("inv_cfg75", 7.5)])
def test_g8_hip_invert_matches_reference_loop(golden_dir, tmp_path, tag, gs):
    from mvoc_amd.schedulers import DDIMInverseScheduler
    g = np.load(os.path.join(golden_dir, "g8_loops.npz"))
    Fr, h, w = int(g["frames"]), int(g["h"]), int(g["w"])
    pipe, calls = _g8_pipe(DDIMInverseScheduler())
    out_dir = str(tmp_path / "lat")
    seq = pipe.invert(prompt="a boat", image=3, height=h * 8, width=w * 8, target_fps=8, num_frames=Fr, num_inference_steps=4,
                      guidance_scale=gs, negative_prompt="bad", latents=torch.from_numpy(g[f"{tag}_x0"]).cuda(), return_dict=False,
                      output_dir=out_dir)
    for i, x in enumerate(calls):  # the batch fed to the UNet at every step, bit for bit
        assert torch.equal(x.cpu(), torch.from_numpy(g[f"{tag}_x"][i]))
    assert torch.equal(seq.cpu(), torch.from_numpy(g[f"{tag}_out"]))
    for f, lat in zip(g[f"{tag}_files"], g[f"{tag}_file_latents"]):
        assert torch.equal(torch.load(os.path.join(out_dir, str(f))), torch.from_numpy(lat))


def test_g8_hip_call_matches_reference_loop(golden_dir):
    from mvoc_amd.schedulers import DDIMScheduler
    g = np.load(os.path.join(golden_dir, "g8_loops.npz"))
    Fr, h, w = int(g["frames"]), int(g["h"]), int(g["w"])
    pipe, calls = _g8_pipe(DDIMScheduler())
    out = pipe(prompt="a boat", image=3, height=h * 8, width=w * 8, target_fps=8, num_frames=Fr, num_inference_steps=4,
               guidance_scale=9.0, negative_prompt="bad", latents=torch.from_numpy(g["call_xT"]).cuda(), output_type="latent",
               ddim_init_latents_t_idx=1).frames
    assert len(calls) == 3 and all(torch.equal(x.cpu(), torch.from_numpy(g["call_x"][i])) for i, x in enumerate(calls))
    assert torch.equal(out.cpu(), torch.from_numpy(g["call_out"]))


@pytest.mark.parametrize("tag,kw", [("comp", dict(random_noise_ratio=0.0, obj_random_noise_fusion=False, fusion_steps=(0, 1))),
                                    ("comp_rnf", dict(random_noise_ratio=0.3, obj_random_noise_fusion=True, fusion_steps=(0, 2)))])
def test_g8_hip_composition_matches_reference_loop(golden_dir, tmp_path, tag, kw):
    """``sample_with_pnp_...`` of this repo on the GPU (fusion / CFG+DDIM kernels, latent cache, hook pushes, conditioning
    assembly) against the reference's own method run on the same inputs: every UNet input batch and the final latents,
    bit for bit"""
    from g8_common import seeded
    from mvoc_amd import pnp_utils
    from mvoc_amd.schedulers import DDIMScheduler
    g = np.load(os.path.join(golden_dir, "g8_loops.npz"))
    Fr, h, w = int(g["frames"]), int(g["h"]), int(g["w"])
    pipe, calls = _g8_pipe(DDIMScheduler())
    full = DDIMScheduler()
    full.set_timesteps(5)
    pnp_utils.modify_diffuser_attention_forward(pipe.unet)
    pnp_utils.register_temp_attention_pnp(pipe, full.timesteps[:5], False)
    pnp_utils.register_spatial_attention_pnp(pipe, full.timesteps[:5], False)
    pnp_utils.register_temp_conv_injection(pipe, full.timesteps[:2])
    pnp_utils.register_out_conv_injection(pipe, full.timesteps[:2])
    pnp_utils.register_resnet_injection(pipe, full.timesteps[:2])
    dirs = {}
    for name, key in (("bg", 20), ("obj0", 30), ("obj1", 40)):
        d_ = tmp_path / name
        d_.mkdir()
        for t in full.timesteps:
            torch.save(seeded(key * 1000 + int(t), (1, 4, Fr, h, w)), str(d_ / f"ddim_latents_{int(t)}.pt"))
        dirs[name] = str(d_)
    masks = [(torch.from_numpy(g[f"{tag}_mask_float"][j]).cuda(), torch.from_numpy(g[f"{tag}_mask_bool"][j]).cuda()) for j in range(2)]
    out = pipe.sample_with_pnp_pipeline_with_edit_prompt_extraction_with_attn_injection(
        prompt="windsurf", main_first_image=1, main_image_list=[10 + i for i in range(Fr)], background_first_image=2,
        background_image_list=[20 + i for i in range(Fr)], objs_first_image=[4, 5],
        objs_image_list=[[40 + i for i in range(Fr)], [50 + i for i in range(Fr)]], height=h * 8, width=w * 8, target_fps=8,
        num_frames=Fr, num_inference_steps=5, guidance_scale=9.0, negative_prompt="chaotic",
        latents=torch.from_numpy(g[f"{tag}_xT"]).cuda(), output_type="latent", ddim_init_latents_t_idx=1, ddim_inv_prompt="",
        obj_mask=["0", "1"], obj_width_height=[(w * 8, h * 8)] * 2, obj_ddim_latents_idx_offset=[0, 1],
        bg_inv_latents_path=dirs["bg"], obj_ddim_latents_path=[dirs["obj0"], dirs["obj1"]], obj_masks_tensors=masks, **kw).frames
    assert len(calls) == 4
    for i, x in enumerate(calls):
        assert torch.equal(x.cpu(), torch.from_numpy(g[f"{tag}_x"][i])), f"step {i}"
    assert torch.equal(out.cpu(), torch.from_numpy(g[f"{tag}_out"]))
    assert pipe.unet.up_blocks[-1].resnets[0].t == 1  # the last register_time_all push


def test_bench_two_rank_protocol(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, env rendezvous on 127.0.0.1), on this one-GPU
    box with the gloo backend standing in for RCCL: both ranks build their shard, barrier, time, MAX-reduce, rank 0 prints
    ONE JSON line whose value is the whole-job aggregate (2 shards) with "scaling": "weak" """
    import subprocess
    env = dict(os.environ, MVOC_BENCH_BACKEND="gloo", MVOC_BENCH_OVERSUBSCRIBE="1")  # (two ranks on this box's one GPU: otherwise refused)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29561", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--frames", "4", "--latent", "32", "--no-roofline", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 4 and j["unit"] == "steps/s"
    assert abs(j["value"] - 2 * 4 / (j["ms_per_step"] * 4 / 1e3)) / j["value"] < 1e-3  # aggregate = world * steps / time


def test_bench_gpus2_launched_plainly_on_the_gpu(tmp_path):
    """the driver's own command form -- `python bench.py --gpus 2 ...`, no torchrun around it -- with the real workload: bench.py
    starts its two ranks as child processes before touching the GPU and relays rank 0's line (this one-GPU box: both ranks on the
    device, MVOC_BENCH_OVERSUBSCRIBE=1, gloo standing in for RCCL)"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MVOC_BENCH_BACKEND="gloo", MVOC_BENCH_OVERSUBSCRIBE="1")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--frames", "4", "--latent", "32",
           "--no-roofline", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and "2 independent shards" in j["config"]["parallelism"]
    assert "3 concurrent source clip(s) per rank" in j["config"]["parallelism"]
    # the line carries the evidence that N ranks ran: one row per rank (device identity, the rank's own clock) and the world size
    # the collective backend saw; without MVOC_BENCH_OVERSUBSCRIBE two ranks on one device are refused before anything is timed
    ranks = j["config"]["ranks"]
    assert [r_["rank"] for r_ in ranks] == [0, 1] and len({r_["pid"] for r_ in ranks}) == 2
    assert all(r_["device_name"] and r_["ms_per_step"] > 0 and r_["ms_per_step"] <= j["ms_per_step"] * 1.001 for r_ in ranks)
    assert j["config"]["collective_backend"]["world_size"] == 2
    env.pop("MVOC_BENCH_OVERSUBSCRIBE")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode != 0 and ("visible" in r.stderr + r.stdout or "distinct device" in r.stderr + r.stdout)


def test_fifty_step_inversion_drift_vs_oracle():
    """SURVEY 8d tolerance proposal: latents after a full 50-step schedule within rel-L2 2e-2 of the fp32 oracle loop
    (per-step UNet noise of ~2e-3 accumulated through the inverse-DDIM recurrence), toy UNet, cfg 1.0"""
    from oracle import loops_ref, sched_ref
    from mvoc_amd.pipeline import I2VGenXLPipeline
    from mvoc_amd.schedulers import DDIMInverseScheduler
    o, eng = _pair()
    g = torch.Generator().manual_seed(50)
    f, h, w = 3, 8, 8
    c = _cond(g, 1, f, h, w)
    x0 = torch.randn(1, 4, f, h, w, generator=g).half()
    pipe = I2VGenXLPipeline(eng, DDIMInverseScheduler(), use_graphs=True)
    inv = pipe.invert(height=h * 8, width=w * 8, num_frames=f, num_inference_steps=50, guidance_scale=1.0, target_fps=8,
                      latents=x0.cuda(), prompt_embeds=c["pe"].cuda(), negative_prompt_embeds=c["ne"].cuda(),
                      image_embeddings=c["ie"].cuda(), image_latents=c["il"].cuda(), return_dict=False, output_dir=None)

    def unet_fn(inp, t):
        return o(inp.float(), int(t), torch.tensor([8]), c["il"].float(), c["ie"].float(), c["pe"].float())[0].half()

    _, ref = loops_ref.invert_loop(unet_fn, sched_ref.DDIMInverseSchedulerRef(), x0, 50, 1.0)
    assert inv.shape == ref.shape == (1, 50, 4, f, h, w)
    got, want = inv[0, 0].float().cpu(), ref[0, 0].float()  # the noisiest latent: end of the recurrence
    rel = float((got - want).norm() / want.norm())
    assert rel <= 2e-2, rel
    first = float((inv[0, -1].float().cpu() - ref[0, -1].float()).norm() / ref[0, -1].float().norm())
    assert first <= 2e-3, first  # after one step


# ---- captured-iteration cache behaviour (round-1 advisor findings) ----------------------------------------------------
def test_stock_graph_is_reused_across_calls_with_new_conditioning(tmp_path):
    """invert() builds new conditioning tensors on every call: the captured iteration must be reused (keyed by shapes, its
    conditioning refreshed in place) -- one graph after three calls -- and replaying it with the refreshed conditioning must
    equal an eager run with that conditioning, bit for bit"""
    from mvoc_amd.pipeline import I2VGenXLPipeline
    from mvoc_amd.schedulers import DDIMInverseScheduler
    _, eng = _pair()
    g = torch.Generator().manual_seed(21)
    f, h, w = 2, 8, 8
    x0 = torch.randn(1, 4, f, h, w, generator=g).half().cuda()
    conds = [_cond(g, 1, f, h, w) for _ in range(3)]

    def run(pipe, c):
        return pipe.invert(height=64, width=64, num_frames=f, num_inference_steps=3, guidance_scale=1.0, latents=x0,
                           prompt_embeds=c["pe"].cuda(), negative_prompt_embeds=c["ne"].cuda(), image_embeddings=c["ie"].cuda(),
                           image_latents=c["il"].cuda(), return_dict=False, output_dir=None)

    graphed = I2VGenXLPipeline(eng, DDIMInverseScheduler(), use_graphs=True)
    outs = [run(graphed, c) for c in conds]
    assert len(graphed._graphs) == 1
    eager = I2VGenXLPipeline(eng, DDIMInverseScheduler(), use_graphs=False)
    for c, got in zip(conds, outs):
        assert torch.equal(run(eager, c), got)
    assert not torch.equal(outs[0], outs[1])  # the conditioning does matter
    # the cache is bounded: other shapes evict the oldest entry instead of pinning one UNet graph each forever
    graphed.max_cached_graphs = 2
    for ff in (1, 3, 4):
        c = _cond(g, 1, ff, h, w)
        graphed.invert(height=64, width=64, num_frames=ff, num_inference_steps=1, guidance_scale=1.0,
                       latents=torch.randn(1, 4, ff, h, w, generator=g).half().cuda(), prompt_embeds=c["pe"].cuda(),
                       negative_prompt_embeds=c["ne"].cuda(), image_embeddings=c["ie"].cuda(), image_latents=c["il"].cuda(),
                       return_dict=False, output_dir=None)
    assert len(graphed._graphs) == 2


def test_composition_graph_variants_follow_every_site_and_the_masks():
    """the reference's API allows one schedule PER SITE and in-place edits of the mask tensors; a captured composition
    iteration bakes both in, so the variant key must cover every site's injecting() bit and the masks' versions: graph
    replays == eager iterations bit for bit through a sequence that changes one site's schedule and then the masks"""
    from mvoc_amd import pnp_utils
    from mvoc_amd.pipeline import I2VGenXLPipeline
    from mvoc_amd.schedulers import DDIMScheduler
    _, eng = _pair()
    g = torch.Generator().manual_seed(22)
    f, h, w, cd = 2, 8, 8, 64
    r = lambda *s: torch.randn(*s, generator=g).half().cuda()
    cond = dict(encoder_hidden_states=r(5, 7, cd), image_embeddings=r(5, f, cd), image_latents_first=r(5, 4, f, h, w),
                image_latents=r(5, 4, f, h, w), fps=torch.full((5,), 8.0, device="cuda"))
    u8 = torch.randint(0, 256, (2, f, h, w), generator=g)
    mk = lambda: [((u8[j].float() / 255).half()[None, None].repeat(1, 4, 1, 1, 1).cuda(),
                   (u8[j] > 10)[None, None].repeat(1, 4, 1, 1, 1).cuda()) for j in range(2)]
    s = DDIMScheduler()
    s.set_timesteps(5, device="cuda")
    ts = s.timesteps
    x0, src = r(1, 4, f, h, w), [r(1, 4, f, h, w) for _ in range(3)]
    results = {}
    for graphs in (False, True):
        pipe = I2VGenXLPipeline(eng, s, use_graphs=graphs)
        for site in eng.hook_sites():
            site.injection_schedule = None
        pnp_utils.register_spatial_attention_pnp(pipe, ts[:5], False)
        pnp_utils.register_temp_attention_pnp(pipe, ts[:5], False)
        pnp_utils.register_resnet_injection(pipe, ts[:5])
        masks = mk()
        st = pipe.make_composition_state(x0, cond, masks, 9.0)
        table, index = s.coef_table(eng.device, 9.0)
        seq = []
        t0 = int(ts[0])
        step = lambda: (pipe.composition_step(st, t0, src[0], src[1:], table[index[t0]], None), seq.append(st["latents"].clone()))
        step()
        # one SITE leaves the schedule (the three representative sites of round 1's key do not change)
        eng.up_blocks[3].resnets[1].injection_schedule = ts[1:2]
        step()
        eng.up_blocks[2].attentions[1].transformer_blocks[0].attn1.processor.injection_schedule = None
        step()
        # in-place edit of a mask tensor (same object, same address)
        masks[0][1][:, :, :, :4].fill_(False)
        masks[0][0][:, :, :, :4].fill_(0)
        step()
        results[graphs] = seq
        if graphs:
            assert len(st["variants"]) == 4
    for a, b in zip(results[False], results[True]):
        assert torch.equal(a, b)
    for site in eng.hook_sites():
        site.injection_schedule, site.t, site.mask = None, None, None


def test_mask_cache_sees_new_and_edited_masks():
    """device masks are cached per mask list: a NEW list whose tensors landed on the old addresses, or an in-place edit,
    must not return the stale device copy"""
    _, eng = _pair()
    f, h, w = 2, 8, 8
    mk = lambda v: [(torch.full((1, 4, f, h, w), v).half().cuda(), torch.full((1, 4, f, h, w), v > 0.5).cuda())]
    a = mk(1.0)
    soft, hard = eng.device_masks(a)
    assert float(soft.min()) == 1.0 and float(hard.min()) == 1.0
    assert eng.device_masks(a)[0] is soft  # cached
    ptrs = (a[0][0].data_ptr(), a[0][1].data_ptr())
    del a
    b = mk(0.0)  # the caching allocator may hand back the same blocks
    soft_b, hard_b = eng.device_masks(b)
    assert float(soft_b.max()) == 0.0 and float(hard_b.max()) == 0.0, ptrs
    b[0][0].fill_(0.5)
    b[0][1].fill_(True)
    soft_c, hard_c = eng.device_masks(b)
    assert float(soft_c.min()) == 0.5 and float(hard_c.min()) == 1.0


def test_longclip_workload_at_cfg4_size():
    """BASELINE configs[3]'s clip (32 frames, 768x768 -> 96x96 latents, 1.42 B network) through bench.py --workload longclip
    on this one GPU: the size itself is exercised (294 912 rows at L0), one JSON line comes back"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "longclip", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["config"]["frames"] == 32 and j["config"]["height"] == 768 and j["n_gpus"] == 1 and j["value"] > 0
    print("longclip 32x768x768, 1 GPU:", j["ms_per_step"], "ms/step")


def test_composition_cfg_off_vs_oracle():
    """composition with classifier-free guidance OFF (guidance_scale 1.0): UNet batch [bg, obj1, obj2, cond],
    every injection family writes the single trailing chunk, the DDIM update takes the conditional prediction -- against the
    oracle's generalisation of the loop and of the hooks (the reference raises on this layout: `// 5` is hard-coded)"""
    from oracle import loops_ref, sched_ref
    from oracle.pnp_model_ref import PnPState, install_pnp
    from mvoc_amd import pnp_utils
    from mvoc_amd.pipeline import I2VGenXLPipeline
    from mvoc_amd.schedulers import DDIMScheduler
    o, eng = _pair()
    g = torch.Generator().manual_seed(15)
    f, h, w, cd, n = 3, 8, 8, 64, 4
    r = lambda *s_: torch.randn(*s_, generator=g).half()
    cond = dict(encoder_hidden_states=r(4, 7, cd), image_embeddings=r(4, f, cd), image_latents_first=r(4, 4, f, h, w),
                image_latents=r(4, 4, f, h, w))
    cond["encoder_hidden_states"][1] = cond["encoder_hidden_states"][0]
    cond["encoder_hidden_states"][2] = cond["encoder_hidden_states"][0]
    cond["image_latents"][3] = cond["image_latents_first"][3]
    u8 = torch.randint(0, 256, (2, f, h, w), generator=g)
    masks = [((u8[j].float() / 255).half()[None, None].repeat(1, 4, 1, 1, 1), (u8[j] > 10)[None, None].repeat(1, 4, 1, 1, 1)) for j in range(2)]
    s = DDIMScheduler()
    s.set_timesteps(n)
    src = {(k, int(t)): r(1, 4, f, h, w) for k in range(3) for t in s.timesteps}
    x0 = r(1, 4, f, h, w)
    rs = sched_ref.DDIMSchedulerRef()
    rs.set_timesteps(n)
    st = PnPState(conv_schedule=rs.timesteps[:2], spatial_schedule=rs.timesteps[:3], temporal_schedule=rs.timesteps[:4])
    st.ndst = 1
    install_pnp(o, st)
    st.masks = masks

    def unet_fn(inp, t):
        st.t = int(t)
        assert inp.shape[0] == 4
        return o.forward_ext(inp.float(), int(t), torch.tensor([8] * 4), cond["image_latents_first"].float(), cond["image_latents"].float(),
                             cond["image_embeddings"].float(), cond["encoder_hidden_states"].float())[0].half()

    ref = loops_ref.composition_loop(unet_fn, sched_ref.DDIMSchedulerRef(), x0, lambda t: src[(0, t)], lambda j, t: src[(1 + j, t)],
                                     [m[0] for m in masks], n, guidance_scale=1.0, ddim_init_latents_t_idx=0, fusion_steps=(0, 1),
                                     random_noise_ratio=0.0)
    results = []
    for graphs in (False, True):
        pipe = I2VGenXLPipeline(eng, DDIMScheduler(), use_graphs=graphs)
        pnp_utils.register_temp_attention_pnp(pipe, s.timesteps[:4], False)
        pnp_utils.register_spatial_attention_pnp(pipe, s.timesteps[:3], False)
        pnp_utils.register_temp_conv_injection(pipe, s.timesteps[:2])
        pnp_utils.register_out_conv_injection(pipe, s.timesteps[:2])
        pnp_utils.register_resnet_injection(pipe, s.timesteps[:2])
        for (k, t), v in src.items():
            pipe.latent_cache.put(f"/virtual/src{k}", t, v.cuda())
        pipe.latent_cache.write_files = False

        class Cond:
            def encode_prompt(self, prompt, negative_prompt=None):
                if prompt == "edit":
                    return cond["encoder_hidden_states"][3:4].cuda(), cond["encoder_hidden_states"][3:4].cuda()
                return cond["encoder_hidden_states"][0:1].cuda(), None

            def image_latents(self, image, num_frames, height, width):
                idx, fr, first = image
                return cond["image_latents_first" if first else "image_latents"][idx:idx + 1].cuda()

            def encode_image(self, image):
                idx, fr, first = image
                return cond["image_embeddings"][idx:idx + 1, fr:fr + 1].cuda()

        pipe.conditioner = Cond()
        out = pipe.sample_with_pnp_pipeline_with_edit_prompt_extraction_with_attn_injection(
            prompt="edit", main_first_image=(3, 0, True), main_image_list=[(3, k, False) for k in range(f)],
            background_first_image=(0, 0, True), background_image_list=[(0, k, False) for k in range(f)],
            objs_first_image=[(1, 0, True), (2, 0, True)],
            objs_image_list=[[(1, k, False) for k in range(f)], [(2, k, False) for k in range(f)]],
            height=h * 8, width=w * 8, num_frames=f, num_inference_steps=n, guidance_scale=1.0, negative_prompt="neg", target_fps=8,
            latents=x0.cuda(), output_type="latent", ddim_init_latents_t_idx=0, ddim_inv_prompt="", fusion_steps=(0, 1),
            random_noise_ratio=0.0, bg_inv_latents_path="/virtual/src0", obj_ddim_latents_path=["/virtual/src1", "/virtual/src2"],
            obj_ddim_latents_idx_offset=[0, 0], obj_masks_tensors=masks).frames
        d = float((out.cpu().float() - ref.float()).abs().max())
        print(f"CFG-off composition after {n} steps (graphs={graphs}): max-abs {d:.2e}")
        assert d < 3e-2, d
        results.append(out)
    assert torch.equal(results[0], results[1])
    for site in eng.hook_sites():
        site.injection_schedule, site.t, site.mask = None, None, None
