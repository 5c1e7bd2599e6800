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
