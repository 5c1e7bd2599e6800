"""GPU: the HIP CLIP towers (mvoc_amd.clip) against transformers' own CLIPVisionModelWithProjection /
CLIPTextModel on CPU in fp32 (oracle/clip_ref.py: the library the reference calls, installed in this image) on identical
fp16-rounded weights, plus the attention kernel's head_dim-96 / causal forms against torch SDPA.

Tolerance: tower outputs rel-L2 <= 4e-3, max-abs <= 2e-2 * max|ref| (fp16 kernels with fp32 accumulation vs fp32, up to 32
pre-LN layers deep); embedding kernels bit-exact."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

TINY_V = dict(hidden_size=320, intermediate_size=640, num_hidden_layers=2, num_attention_heads=4, image_size=56, patch_size=14,
              projection_dim=128)
TINY_V160 = dict(TINY_V, hidden_size=160, num_attention_heads=2, intermediate_size=320)  # K % 64 != 0: explicit LayerNorm path
TINY_T = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, vocab_size=1000)


def dev(t):
    return t.to("cuda", torch.float16).contiguous()


def _metrics(out, ref):
    out, ref = out.float().cpu(), ref.float().cpu()
    assert out.shape == ref.shape and torch.isfinite(out).all()
    return float((out - ref).norm() / ref.norm()), float((out - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("d,dp,t,causal", [(80, 96, 257, False), (80, 96, 17, False), (64, 64, 77, True), (64, 64, 200, True),
                                           (80, 96, 77, True)])
def test_flash_attn_clip_forms(d, dp, t, causal):
    """head_dim 96 (80 real + 16 zero columns per head), explicit scale 1/sqrt(80), causal mask: against torch SDPA"""
    from mvoc_amd import ops
    g = torch.Generator().manual_seed(d + t)
    nb, heads = 3, 4
    q, k, v = (torch.randn(nb, t, heads, d, generator=g).half() for _ in range(3))

    def pad(x):
        return F.pad(x, (0, dp - d)).reshape(nb * t, heads * dp)

    out = ops.flash_attn(dev(pad(q)), dev(pad(k)), dev(pad(v)), nbatch=nb, heads=heads, tq=t, tk=t, head_dim=dp, causal=causal,
                         scale=1.0 / math.sqrt(d))
    ref = F.scaled_dot_product_attention(q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2),
                                         is_causal=causal).transpose(1, 2)
    o = out.float().cpu().view(nb, t, heads, dp)
    assert (o[..., d:] == 0).all()
    assert (o[..., :d] - ref).abs().max() < 1e-2
    assert float((o[..., :d] - ref).norm() / ref.norm()) < 2e-3


def test_clip_embedding_kernels_bit_exact():
    """patch im2col and the (class | patch | token) + position assembly against torch indexing"""
    import ctypes as C
    from mvoc_amd._ffi import check, lib
    g = torch.Generator().manual_seed(1)
    b, size, patch, c = 3, 56, 14, 64
    gsz, k = size // patch, 3 * patch * patch
    kpad = (k + 63) // 64 * 64
    x = torch.randn(b, 3, size, size, generator=g).half()
    cols = torch.full((b * gsz * gsz, kpad), 9.0, dtype=torch.float16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    xd = dev(x)
    check(lib.mvoc_clip_patches_f16(xd.data_ptr(), cols.data_ptr(), b, size, patch, kpad, st), "patches")
    ref = F.unfold(x.float(), patch, stride=patch).transpose(1, 2).reshape(b * gsz * gsz, k).half()  # (c, py, px) order
    assert torch.equal(cols[:, :k].cpu(), ref) and (cols[:, k:] == 0).all()
    t = gsz * gsz + 1
    pe, cls, pos = torch.randn(b * (t - 1), c, generator=g).half(), torch.randn(c, generator=g).half(), torch.randn(t, c, generator=g).half()
    out = torch.empty((b * t, c), dtype=torch.float16, device="cuda")
    ped, clsd, posd = dev(pe), dev(cls), dev(pos)
    check(lib.mvoc_clip_embed_f16(ped.data_ptr(), None, clsd.data_ptr(), posd.data_ptr(), out.data_ptr(), b * t, t, c, st), "embed")
    ref = torch.cat([cls.expand(b, 1, c), pe.view(b, t - 1, c)], 1) + pos[None]
    assert torch.equal(out.cpu(), ref.reshape(b * t, c))
    tab, ids = torch.randn(50, c, generator=g).half(), torch.randint(0, 50, (b, t), generator=g).int()
    tabd, idsd = dev(tab), ids.cuda()
    check(lib.mvoc_clip_embed_f16(tabd.data_ptr(), idsd.data_ptr(), None, posd.data_ptr(), out.data_ptr(), b * t, t, c, st), "embed")
    assert torch.equal(out.cpu(), (tab[ids.long()] + pos[None]).reshape(b * t, c))


def _vision_pair(cfg):
    """(transformers module in fp32 on CPU, HIP tower) holding the same fp16-rounded seeded weights"""
    from mvoc_amd.clip import CLIPVisionModelWithProjection
    from oracle import clip_ref
    eng = CLIPVisionModelWithProjection(cfg)
    captured = {}
    orig = eng._build
    eng._build = lambda s: (captured.update(s), orig(s))[1]
    eng.init_random(7)
    o = clip_ref.load(clip_ref.vision(**cfg), captured)
    return o, eng


@pytest.mark.parametrize("cfg,b", [(TINY_V, 3), (TINY_V160, 2), ({}, 2)], ids=["tiny320", "tiny160", "ViT-H"])
def test_vision_tower_vs_transformers(cfg, b):
    o, eng = _vision_pair(cfg)
    size = eng.config.image_size
    px = (torch.randn(b, 3, size, size, generator=torch.Generator().manual_seed(b)) * 1.2).half()
    out = eng(px)
    ref = o(px.float()).image_embeds
    rel, mx = _metrics(out, ref)
    assert rel < 4e-3 and mx < 2e-2, (rel, mx)


def _text_pair(cfg):
    from mvoc_amd.clip import CLIPTextModel
    from oracle import clip_ref
    eng = CLIPTextModel(cfg)
    captured = {}
    orig = eng._build
    eng._build = lambda s: (captured.update(s), orig(s))[1]
    eng.init_random(9)
    o = clip_ref.load(clip_ref.text(**cfg), captured)
    return o, eng


@pytest.mark.parametrize("cfg,b,t", [(TINY_T, 3, 77), (TINY_T, 2, 20), ({}, 2, 77)], ids=["tiny77", "tiny20", "ViT-H-text"])
def test_text_tower_vs_transformers(cfg, b, t):
    o, eng = _text_pair(cfg)
    ids = torch.randint(3, eng.config.vocab_size, (b, t), generator=torch.Generator().manual_seed(t))
    out = eng(ids)
    ref = o(ids)[0]
    rel, mx = _metrics(out, ref)
    assert rel < 4e-3 and mx < 2e-2, (rel, mx)
    # causality: a changed later token leaves earlier positions bit-identical
    ids2 = ids.clone()
    ids2[:, t // 2] = (ids2[:, t // 2] + 1) % eng.config.vocab_size
    out2 = eng(ids2)
    assert torch.equal(out2[:, :t // 2], out[:, :t // 2]) and not torch.equal(out2[:, t // 2:], out[:, t // 2:])


def test_conditioner_batched_images_and_prompts():
    """ClipCodec behind the pipeline's conditioner: PIL frames -> one batched ViT pass == the oracle on the reference's host
    preprocessing (PIL BILINEAR resize + CLIPImageProcessor normalisation), batch-size independent; token ids -> hidden states"""
    from PIL import Image
    from mvoc_amd.clip import ClipCodec
    from mvoc_amd.pipeline import SyntheticConditioner
    from oracle import clip_ref
    o, eng = _vision_pair(TINY_V)
    ot, engt = _text_pair(TINY_T)
    rng = np.random.default_rng(0)
    frames = [Image.fromarray(rng.integers(0, 256, (90, 160, 3), dtype=np.uint8)) for _ in range(5)]
    cond = SyntheticConditioner("cuda:0", clip=ClipCodec(eng, engt, batch=64))
    emb = cond.encode_images(frames)
    assert emb.shape == (5, 1, TINY_V["projection_dim"])
    ref = o(clip_ref.pixel_values(frames, TINY_V["image_size"])).image_embeds
    rel, mx = _metrics(emb[:, 0], ref)
    assert rel < 4e-3 and mx < 2e-2, (rel, mx)
    cond.clip.batch = 2  # 3 passes instead of 1
    # (the GEMM chooses tiles / split-K by row count, so another batch size may differ in accumulation order, not in value)
    assert _metrics(cond.encode_images(frames), emb)[0] < 1e-3
    assert _metrics(cond.encode_image(frames[3]), emb[3:4])[0] < 1e-3
    ids = torch.randint(3, 1000, (2, 77), generator=torch.Generator().manual_seed(3))
    pe, ne = cond.encode_prompt(ids[0], ids[1])
    ref = ot(ids)[0]
    assert _metrics(torch.cat([pe, ne]), ref)[0] < 4e-3
    with pytest.raises(RuntimeError, match="tokenizer"):
        cond.clip.encode_prompt("a boat", "")
