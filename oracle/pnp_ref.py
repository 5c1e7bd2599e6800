"""ORACLE (test infrastructure, not product): the PnP injection arithmetic of MVOC.

Restates, with plain PyTorch CPU ops in the tensor's own dtype (so fp16 inputs are
rounded to fp16 after *every* op exactly as the reference's eager GPU ops are):

* spatial Q/K injection   ``i2vgen-xl/pnp_utils.py:624-672``  (bool mask, nearest-resized)
* temporal Q/K injection  ``i2vgen-xl/pnp_utils.py:778-850``  (float mask channel 0, nearest-resized)
* feature injection       ``i2vgen-xl/pnp_utils.py:970-1004`` (resnet), ``:1059-1082`` (temporal conv),
                          ``:1114-1146`` (conv_out): bool mask at latent resolution, base = chunk 0

Batch layout is positional ``[bg, obj_1 .. obj_n, uncond, cond]`` (``pipeline_i2vgen_xl.py:1676``);
destination chunks are the last two; base is the last chunk (``inject_background=False``) or chunk 0.
``ndst=1`` is the generalisation to classifier-free guidance OFF (the reference hard-codes the
batch of 5 and cannot run it): layout ``[bg, obj_1 .. obj_n, cond]``, one destination chunk.
The blend is the arithmetic form ``x*(1-m) + y*m`` -- NOT a select: with m=1, ``y=-0.0`` becomes
``+0.0`` and ``x=inf`` becomes NaN.  Bit-exactness is defined against this form.

Pinned by ``tests/golden/g1..g5*.npz`` (reference code run by ``tools/gen_golden.py``).
"""
import torch
import torch.nn.functional as F


def nearest_resize(mask_fhw: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """``F.interpolate(..., mode='nearest')`` of a ``[F,h,w]`` mask (``pnp_utils.py:650, 807``)."""
    return F.interpolate(mask_fhw[None], size=(height, width), mode="nearest")[0]


def _blend(base, obj, m):
    # three rounded ops per element, in this order: (1-m), base*(1-m), obj*m, then the add
    return base * (1 - m) + obj * m


def inject_qk_spatial(query, key, masks, num_frames, height, width, inject_background=False, ndst=2):
    """query/key ``[(n_obj+3)*F, H*W, C]``; masks: list of bool ``[1,4,F,h,w]`` (or ``[F,h,w]``) tensors.
    Returns new (query, key) with the two trailing chunks overwritten."""
    nchunk = len(masks) + 1 + ndst
    cs = query.shape[0] // nchunk
    c = query.shape[-1]
    outs = []
    for t in (query, key):
        t = t.clone().reshape(t.shape[0], height, width, c)
        inj = t[:cs] if inject_background else t[(nchunk - 1) * cs:]
        for j, bm in enumerate(masks):
            m = bm.reshape(-1, *bm.shape[-3:])[0] if bm.ndim == 5 else bm
            m = nearest_resize(m.to(t.dtype), height, width)  # [F,H,W], exact {0,1}
            m = m.unsqueeze(-1)
            inj = _blend(inj, t[cs * (j + 1):cs * (j + 2)], m)
        t[(nchunk - ndst) * cs:] = torch.cat([inj] * ndst)
        outs.append(t.reshape(t.shape[0], height * width, c))
    return outs[0], outs[1]


def inject_qk_temporal(query, key, masks, height, width, inject_background=False, ndst=2):
    """query/key ``[(n_obj+3)*H*W, F, C]``; masks: list of float ``[1,4,F,h,w]`` (or ``[F,h,w]``) tensors
    (values k/255 in the tensor dtype); channel 0 is used (``pnp_utils.py:805-809``)."""
    nchunk = len(masks) + 1 + ndst
    f, c = query.shape[1], query.shape[2]
    outs = []
    for t in (query, key):
        t = t.clone().reshape(nchunk, height, width, f, c)
        inj = t[:1] if inject_background else t[nchunk - 1:]
        for j, fm in enumerate(masks):
            m = fm[0, 0] if fm.ndim == 5 else fm
            m = nearest_resize(m.to(t.dtype), height, width)  # [F,H,W]
            m = m.permute(1, 2, 0)[None, :, :, :, None]  # [1,H,W,F,1]
            inj = _blend(inj, t[j + 1:j + 2], m)
        t[nchunk - ndst:] = torch.cat([inj] * ndst)
        outs.append(t.reshape(nchunk * height * width, f, c))
    return outs[0], outs[1]


def inject_feature_nchw(x, masks, ndst=2):
    """x ``[(n_obj+3)*F, C, H, W]``; masks: list of bool ``[1,4,F,h,w]`` (or ``[F,h,w]``) at latent resolution
    (no resize -- ``pnp_utils.py:986-1000``).  Base is chunk 0."""
    nchunk = len(masks) + 1 + ndst
    cs = x.shape[0] // nchunk
    x = x.clone()
    inj = x[:cs]
    for j, bm in enumerate(masks):
        m = bm.reshape(-1, *bm.shape[-3:])[0] if bm.ndim == 5 else bm
        m = m.to(x.dtype).unsqueeze(1)  # [F,1,H,W]
        inj = _blend(inj, x[cs * (j + 1):cs * (j + 2)], m)
    x[(nchunk - ndst) * cs:] = torch.cat([inj] * ndst)
    return x
