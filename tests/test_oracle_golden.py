"""CPU: the oracle (oracle/) against the golden vectors produced by the reference's own code
(tools/gen_golden.py).  This is what pins the oracle's in-repo half."""
import os

import numpy as np
import pytest
import torch

from oracle import pnp_ref, unet_ref as U

torch.set_grad_enabled(False)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _masks(g, prefix=""):
    mf = torch.from_numpy(g[prefix + "mask_float"])
    mb = torch.from_numpy(g[prefix + "mask_bool"])
    return [(mf[j], mb[j]) for j in range(mf.shape[0])]


def _attn_from_golden(g):
    attn = U.Attention(64, heads=int(g["heads"]), dim_head=64).half()
    attn.to_q.weight.copy_(torch.from_numpy(g["to_q"]))
    attn.to_k.weight.copy_(torch.from_numpy(g["to_k"]))
    attn.to_v.weight.copy_(torch.from_numpy(g["to_v"]))
    attn.to_out[0].weight.copy_(torch.from_numpy(g["to_out_w"]))
    attn.to_out[0].bias.copy_(torch.from_numpy(g["to_out_b"]))
    return attn


@pytest.mark.parametrize("bg", [0, 1])
def test_g1_spatial_qk_injection_bit_exact(golden_dir, bg):
    g = _load(golden_dir, f"g1_spatial_proc_bg{bg}.npz")
    attn = _attn_from_golden(g)
    hs = torch.from_numpy(g["hidden_states"])
    Fr, H, W = int(g["frames"]), int(g["height"]), int(g["width"])
    q, k = attn.to_q(hs), attn.to_k(hs)
    # off-schedule: Q/K untouched
    assert np.array_equal(q.numpy().view(np.uint16), g["q_off"][:, 0].view(np.uint16))
    masks = [m[1] for m in _masks(g)]
    qi, ki = pnp_ref.inject_qk_spatial(q, k, masks, Fr, H, W, inject_background=bool(bg))
    assert np.array_equal(qi.numpy().view(np.uint16), g["q_on"][:, 0].view(np.uint16))
    assert np.array_equal(ki.numpy().view(np.uint16), g["k_on"][:, 0].view(np.uint16))
    # chunks 3 and 4 identical, chunks 0..2 untouched
    cs = Fr
    assert torch.equal(qi[3 * cs:4 * cs], qi[4 * cs:])
    assert torch.equal(qi[:3 * cs], q[:3 * cs])


@pytest.mark.parametrize("bg", [0, 1])
def test_g2_temporal_qk_injection_bit_exact(golden_dir, bg):
    g = _load(golden_dir, f"g2_temporal_proc_bg{bg}.npz")
    attn = _attn_from_golden(g)
    hs = torch.from_numpy(g["hidden_states"])
    H, W = int(g["height"]), int(g["width"])
    q, k = attn.to_q(hs), attn.to_k(hs)
    masks = [m[0] for m in _masks(g)]
    qi, ki = pnp_ref.inject_qk_temporal(q, k, masks, H, W, inject_background=bool(bg))
    assert np.array_equal(qi.numpy().view(np.uint16), g["q_on"][:, 0].view(np.uint16))
    assert np.array_equal(ki.numpy().view(np.uint16), g["k_on"][:, 0].view(np.uint16))
    assert np.array_equal(q.numpy().view(np.uint16), g["q_off"][:, 0].view(np.uint16))


def _small4_half(seed):
    unet = U.I2VGenXLUNet(U.UNetConfig.small4()).half()
    U.init_weights_(unet, seed=seed)
    return unet


def test_g3_g4_g5_feature_injection_bit_exact(golden_dir):
    g = _load(golden_dir, "g3_g4_g5_feature_injection.npz")
    unet = _small4_half(5)
    rn, tc, co = unet.up_blocks[3].resnets[0], unet.up_blocks[3].temp_convs[0], unet.conv_out
    for prefix, mod in (("resnet.", rn), ("tconv.", tc), ("convout.", co)):
        sd = {k[len("w:" + prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:" + prefix)}
        mod.load_state_dict(sd)
    Fr = int(g["frames"])
    bmasks = [m[1] for m in _masks(g)]
    x, temb = torch.from_numpy(g["x_resnet"]), torch.from_numpy(g["temb"])
    # resnet: the injection sits between conv2 and the shortcut add (pnp_utils.py:968-1018)
    h = rn.conv1(rn.nonlinearity(rn.norm1(x)))
    h = h + rn.time_emb_proj(rn.nonlinearity(temb))[:, :, None, None]
    h = rn.conv2(rn.nonlinearity(rn.norm2(h)))
    sc = rn.conv_shortcut(x)
    assert np.array_equal((sc + h).numpy().view(np.uint16), g["resnet_off"].view(np.uint16))
    h_inj = pnp_ref.inject_feature_nchw(h, bmasks)
    assert np.array_equal((sc + h_inj).numpy().view(np.uint16), g["resnet_on"].view(np.uint16))
    # temporal conv: injection after the residual add
    xt = torch.from_numpy(g["x_tconv"])
    y = tc(xt, num_frames=Fr)
    assert np.array_equal(y.numpy().view(np.uint16), g["tconv_off"].view(np.uint16))
    assert np.array_equal(pnp_ref.inject_feature_nchw(y, bmasks).numpy().view(np.uint16), g["tconv_on"].view(np.uint16))
    # conv_out
    xc = torch.from_numpy(g["x_convout"])
    y = co(xc)
    assert np.array_equal(y.numpy().view(np.uint16), g["convout_off"].view(np.uint16))
    on = pnp_ref.inject_feature_nchw(y, bmasks)
    assert np.array_equal(on.numpy().view(np.uint16), g["convout_on"].view(np.uint16))
    assert torch.equal(on[3 * Fr:4 * Fr], on[4 * Fr:])


def test_g6_transformer_forwards(golden_dir):
    g = _load(golden_dir, "g6_transformer_forwards.npz")
    holder = torch.nn.Module()
    holder.spa = U.Transformer2DModel(1, 64, 64, 64, 8)
    holder.tmp = U.TransformerTemporalModel(1, 64, 64, 8)
    holder.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:")})
    x, enc, Fr = torch.from_numpy(g["x"]), torch.from_numpy(g["enc"]), int(g["frames"])
    o_spa = holder.spa(x, encoder_hidden_states=enc)[0]
    o_tmp = holder.tmp(x, num_frames=Fr)[0]
    assert torch.allclose(o_spa, torch.from_numpy(g["out_spatial"]), atol=1e-5, rtol=1e-5)
    assert torch.allclose(o_tmp, torch.from_numpy(g["out_temporal"]), atol=1e-5, rtol=1e-5)


def _g7_unet(g):
    unet = U.I2VGenXLUNet(U.UNetConfig.small4())
    U.init_weights_(unet, seed=9)
    for p in unet.parameters():
        p.copy_(p.half().float())
    s = sum(float(v.double().abs().sum()) for v in unet.state_dict().values())
    assert abs(s - float(g["weights_abs_sum"])) < 1e-6 * s, "seeded init drifted: regenerate tests/golden"
    return unet


def test_g7_unet_ext_plain(golden_dir):
    g = _load(golden_dir, "g7_unet_ext.npz")
    unet = _g7_unet(g)
    t = lambda k: torch.from_numpy(g["plain_" + k])
    out = unet.forward_ext(t("sample"), int(g["plain_t"]), t("fps"), t("image_latents_first"), t("image_latents"),
                           t("image_embeddings"), t("encoder_hidden_states"))[0]
    ref = torch.from_numpy(g["plain_out"])
    assert out.shape == ref.shape
    assert (out - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max())


def test_g7_unet_ext_pnp(golden_dir):
    """the oracle's own PnP-enabled forward (oracle/pnp_model_ref.py) against the reference's hooks"""
    from oracle.pnp_model_ref import PnPState, install_pnp
    g = _load(golden_dir, "g7_unet_ext.npz")
    unet = _g7_unet(g)
    t = lambda k: torch.from_numpy(g["pnp_" + k])
    masks = _masks(g, "pnp_")
    from oracle.sched_ref import DDIMSchedulerRef
    s = DDIMSchedulerRef()
    s.set_timesteps(50)
    state = PnPState(conv_schedule=s.timesteps[:5], spatial_schedule=s.timesteps[:50],
                     temporal_schedule=s.timesteps[:50], inject_background=False)
    install_pnp(unet, state)
    for tag, tt in (("t981", 981), ("t861", 861), ("t1", 1)):
        state.t, state.masks = tt, masks
        out = unet.forward_ext(t("sample"), tt, t("fps"), t("image_latents_first"), t("image_latents"),
                               t("image_embeddings"), t("encoder_hidden_states"))[0]
        ref = torch.from_numpy(g["pnp_out_" + tag])
        assert (out - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max()), tag


def test_g9_mask_fixture_shapes(golden_dir):
    g = _load(golden_dir, "g9_boat_surf_masks.npz")
    for name in ("boat_mask", "surf_mask"):
        assert g[f"{name}_90x160_float_u8"].shape == (16, 90, 160)
        assert g[f"{name}_64x64_bool"].shape == (16, 64, 64)
        # bool = (u8 > 10) as the reference's cv2.threshold(…, 10, 255) does
        assert np.array_equal(g[f"{name}_64x64_bool"], g[f"{name}_64x64_float_u8"] > 10)
        assert g[f"{name}_64x64_bool"].any()


# ---- G8: the three loops as run by the reference's own pipeline methods ---------------------------------------------
def _g8_unet_fn(calls, seen):
    """the loop restatements call ``unet_fn(x, t)``: conditioning as the reference assembled it (identical at every step),
    the UNet = the fixture's fake; every input batch is checked against what the reference's loop fed at that step"""
    from g8_common import fake_unet

    def fn(x, t):
        i = len(seen)
        assert int(t) == calls.t[i]
        assert torch.equal(x, calls.x[i]), f"step {i}: UNet input differs from the reference loop's"
        seen.append(int(t))
        return fake_unet(x, t, calls.ehs[i], calls.fps[i], calls.ilf[i], calls.il[i], calls.ie[i])
    return fn


@pytest.mark.parametrize("tag,gs", [("inv_cfg1", 1.0), ("inv_cfg75", 7.5)])
def test_g8_invert_loop_matches_reference(golden_dir, tag, gs):
    from g8_common import Calls
    from oracle import loops_ref, sched_ref
    g = _load(golden_dir, "g8_loops.npz")
    calls, seen = Calls(g, tag), []
    x0 = torch.from_numpy(g[f"{tag}_x0"])
    saved, seq = loops_ref.invert_loop(_g8_unet_fn(calls, seen), sched_ref.DDIMInverseSchedulerRef(), x0, 4, gs)
    assert len(seen) == calls.n == 4
    assert torch.equal(seq, torch.from_numpy(g[f"{tag}_out"]))  # [1, steps, 4, F, h, w], noisiest first
    # files: ddim_latents_{t}.pt holds the latent AT noise level t
    files = [str(f) for f in g[f"{tag}_files"]]
    assert files == sorted(f"ddim_latents_{t}.pt" for t in saved)
    for f, lat in zip(files, g[f"{tag}_file_latents"]):
        assert torch.equal(saved[int(f.split("_")[-1][:-3])], torch.from_numpy(lat))


def test_g8_sample_loop_matches_reference(golden_dir):
    from g8_common import Calls
    from oracle import loops_ref, sched_ref
    g = _load(golden_dir, "g8_loops.npz")
    calls, seen = Calls(g, "call"), []
    out = loops_ref.sample_loop(_g8_unet_fn(calls, seen), sched_ref.DDIMSchedulerRef(), torch.from_numpy(g["call_xT"]), 4, 9.0,
                                ddim_init_latents_t_idx=1)
    assert seen == [501, 251, 1]
    assert torch.equal(out, torch.from_numpy(g["call_out"]))


@pytest.mark.parametrize("tag,kw", [("comp", dict(random_noise_ratio=0.0, obj_random_noise_fusion=False, fusion_steps=(0, 1))),
                                    ("comp_rnf", dict(random_noise_ratio=0.3, obj_random_noise_fusion=True, fusion_steps=(0, 2)))])
def test_g8_composition_loop_matches_reference(golden_dir, tag, kw):
    """fusion arithmetic, [bg, obj1, obj2, latents, latents] assembly, CFG on the last two chunks, DDIM update, the
    never-incremented fusion counter and the per-object timestep offsets -- against the reference's own loop"""
    from g8_common import Calls, seeded
    from oracle import loops_ref, sched_ref
    g = _load(golden_dir, "g8_loops.npz")
    calls, seen = Calls(g, tag), []
    Fr, h, w = int(g["frames"]), int(g["h"]), int(g["w"])
    lat = lambda key, t: seeded(key * 1000 + int(t), (1, 4, Fr, h, w))  # the ddim_latents_{t}.pt files the generator wrote
    masks = [torch.from_numpy(m) for m in g[f"{tag}_mask_float"]]
    hook_ts = []
    out = loops_ref.composition_loop(_g8_unet_fn(calls, seen), sched_ref.DDIMSchedulerRef(), torch.from_numpy(g[f"{tag}_xT"]),
                                     lambda t: lat(20, t), lambda j, t: lat(30 + 10 * j, t), masks, 5, guidance_scale=9.0,
                                     ddim_init_latents_t_idx=1, obj_ddim_latents_idx_offset=[0, 1],
                                     on_step=lambda i, t: hook_ts.append(t), **kw)
    assert seen == [601, 401, 201, 1] and hook_ts == calls.hook_t  # register_time_all(t) precedes every UNet call
    assert torch.equal(out, torch.from_numpy(g[f"{tag}_out"]))


def test_g8_reference_conditioning_layout(golden_dir):
    """what the reference's own prepare_image_latents / _encode_image / batch assembly produce (the layout
    mvoc_amd.pipeline reproduces): frame-position ramp k/(F-1), zero image embedding for the uncond chunk, and the
    composition batch order [bg, obj_1, obj_2, main(uncond), main(cond)] for every conditioning tensor"""
    from g8_common import seeded, prompt_key
    g = _load(golden_dir, "g8_loops.npz")
    Fr, h, w, D = int(g["frames"]), int(g["h"]), int(g["w"]), int(g["dim"])
    il = torch.from_numpy(g["inv_cfg75_il"][0])  # [2,4,F,h,w]: CFG duplicates the image latents
    first = (seeded(503, (1, 4, h, w)) * 0.18215)
    assert torch.equal(il[0], il[1]) and torch.equal(il[0, :, 0], first[0])
    for k in range(1, Fr):
        assert torch.equal(il[0, :, k], torch.full((4, h, w), k / (Fr - 1)).half())
    ie = torch.from_numpy(g["inv_cfg75_ie"][0])  # [2,1,D]: zeros for uncond
    assert not ie[0].any() and torch.equal(ie[1, 0], seeded(303, (1, D))[0])
    ehs = torch.from_numpy(g["inv_cfg75_ehs"][0])
    assert torch.equal(ehs[0], seeded(1000 + prompt_key("bad"), (1, 7, D))[0]) and torch.equal(ehs[1], seeded(prompt_key("a boat"), (1, 7, D))[0])
    # composition: batch of 5
    ehs = torch.from_numpy(g["comp_ehs"][0])
    inv = seeded(prompt_key(""), (1, 7, D))[0]
    assert all(torch.equal(ehs[j], inv) for j in range(3))
    assert torch.equal(ehs[3], seeded(1000 + prompt_key("chaotic"), (1, 7, D))[0]) and torch.equal(ehs[4], seeded(prompt_key("windsurf"), (1, 7, D))[0])
    ilf = torch.from_numpy(g["comp_ilf"][0])  # first-frame latents: bg(2), obj(4), obj(5), main(1), main(1)
    for b, img in enumerate((2, 4, 5, 1, 1)):
        assert torch.equal(ilf[b, :, 0], (seeded(500 + img, (1, 4, h, w)) * 0.18215)[0])
    il = torch.from_numpy(g["comp_il"][0])  # image_latents: frame 0 of each list: bg 20, obj 40, obj 50; main = main_first_image
    for b, img in enumerate((20, 40, 50, 1, 1)):
        assert torch.equal(il[b, :, 0], (seeded(500 + img, (1, 4, h, w)) * 0.18215)[0])
    ie = torch.from_numpy(g["comp_ie"][0])  # [5,F,D]: per-frame CLIP embeddings; uncond chunk zero
    for b, base in ((0, 20), (1, 40), (2, 50), (4, 10)):
        for f in range(Fr):
            assert torch.equal(ie[b, f], seeded(300 + base + f, (1, D))[0])
    assert not ie[3].any()
    assert torch.equal(torch.from_numpy(g["comp_fps"][0]), torch.tensor([8] * 5))


def test_committed_fixtures_are_what_the_reference_produces(tmp_path):
    """Build container only (skipped wherever /root/reference is absent, e.g. on the GPU box): tools/gen_golden.py -- which IMPORTS the
    reference's own pnp_utils.py / pipeline_i2vgen_xl.py / utils.py -- regenerates every fixture under tests/golden/ into a scratch
    directory, and each file must come out byte for byte as committed: the golden vectors are outputs of the reference's code, not of
    the oracle's, and nobody edited them."""
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/i2vgen-xl"):
        pytest.skip("the reference checkout is not present here")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MVOC_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "gen_golden.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith((".npz", ".json")))
    assert made, "gen_golden.py wrote nothing"
    gold = os.path.join(repo, "tests", "golden")
    committed = sorted(f for f in os.listdir(gold) if f.endswith((".npz", ".json")))
    assert made == committed, (made, committed)
    for f in made:
        assert open(os.path.join(tmp_path, f), "rb").read() == open(os.path.join(gold, f), "rb").read(), f"{f} differs from the committed fixture"
