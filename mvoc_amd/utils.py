"""Host helpers of the path with the reference's names (``i2vgen-xl/utils.py``): seeding, latent ``.pt`` loading,
mask preprocessing.  Pure PIL / numpy / torch -- cv2, torchvision and einops are not needed."""
import os
import os.path as osp
import random
from glob import glob

import numpy as np
import torch
from PIL import Image


def seed_everything(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    random.seed(seed)
    np.random.seed(seed)


def load_ddim_latents_at_t(t, ddim_latents_path):
    """``utils.py:31-36``"""
    p = os.path.join(ddim_latents_path, f"ddim_latents_{t}.pt")
    assert os.path.exists(p), f"Missing latents at t {t} path {p}"
    return torch.load(p, map_location="cpu")


_PRECISION_BITS = 32 - 8 - 2  # Pillow's 8-bit resampling: fixed-point weights with 22 fractional bits


def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_tables(in_size, out_size):
    """Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the BICUBIC filter (the default of ``Image.resize`` on
    an "L" image, ``utils.py:95, 126``): bounds int32 [out, 2] = (xmin, n), weights int32 [out, ksize].  Same double
    arithmetic in the same order as the C code, so a device pass over these tables equals PIL bit for bit (checked against
    PIL itself in tests/test_host_cpu.py)."""
    import math
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x, w in enumerate(k):
            if ww != 0.0:
                w = w / ww
            kk[xx, x] = int(-0.5 + w * (1 << _PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def resize8_reference(a, out_hw):
    """numpy evaluation of the same two passes (host check of the tables; the device runs ``mvoc_mask_resize_u8``)"""
    def one(a, bounds, kk, axis):
        a = np.moveaxis(a, axis, -1).astype(np.int64)
        out = np.zeros(a.shape[:-1] + (len(bounds),), np.int64)
        for xx, (xmin, n) in enumerate(bounds):
            acc = (1 << (_PRECISION_BITS - 1)) + (a[..., xmin:xmin + n] * kk[xx, :n].astype(np.int64)).sum(-1)
            out[..., xx] = np.clip(acc >> _PRECISION_BITS, 0, 255)
        return np.moveaxis(out.astype(np.uint8), -1, axis)
    h, w = out_hw
    return one(one(a, *pil_bicubic_tables(a.shape[-1], w), axis=a.ndim - 1), *pil_bicubic_tables(a.shape[-2], h), axis=a.ndim - 2)


def mask_preprocess_device(u8_frames, device, dtype, batch_size, channel, downscale=8):
    """the resize / float / bool part of ``mask_preprocess`` on the GPU: ``u8_frames`` uint8 [F, H, W] ("L" images, decoded on
    the host) -> (float [b,c,F,h,w] in ``dtype``, bool [b,c,F,h,w]) on ``device``"""
    import ctypes as C
    from ._ffi import check, lib
    if dtype != torch.float16:
        raise RuntimeError("mask_preprocess_device produces fp16 float masks (the dtype of the path)")
    a = torch.from_numpy(np.ascontiguousarray(u8_frames)).to(device)
    F_, H, W = a.shape
    h, w = H // downscale, W // downscale
    bh, kh = pil_bicubic_tables(W, w)
    bv, kv = pil_bicubic_tables(H, h)
    tb = [torch.from_numpy(t).to(device) for t in (bh, kh, bv, kv)]
    tmp = torch.empty((F_, H, w), dtype=torch.uint8, device=device)
    out = torch.empty((F_, h, w), dtype=torch.uint8, device=device)
    st = torch.cuda.current_stream().cuda_stream
    check(lib.mvoc_mask_resize_u8(a.data_ptr(), tmp.data_ptr(), out.data_ptr(), F_, H, W, h, w, tb[0].data_ptr(), tb[1].data_ptr(),
                                  kh.shape[1], tb[2].data_ptr(), tb[3].data_ptr(), kv.shape[1], st), "mask_resize")
    fl = torch.empty((F_, h, w), dtype=torch.float16, device=device)
    bl = torch.empty((F_, h, w), dtype=torch.uint8, device=device)
    check(lib.mvoc_mask_finish(out.data_ptr(), fl.data_ptr(), bl.data_ptr(), out.numel(), st), "mask_finish")
    fl = fl[None, None].repeat(batch_size, channel, 1, 1, 1)
    bl = bl.bool()[None, None].repeat(batch_size, channel, 1, 1, 1)
    return fl, bl


def _one_mask(path, downscale):
    """``utils.py:92-110 / 121-137``: 'L' image -> PIL default-resample resize to (W//d, H//d) -> float = v/255,
    bool = v > 10 (cv2.threshold(v, 10, 255, THRESH_BINARY) then /255 -> bool)"""
    m = Image.open(path).convert("L")
    w, h = m.size
    m = m.resize((w // downscale, h // downscale))
    u8 = torch.from_numpy(np.asarray(m).copy())
    return u8.to(torch.float32).div_(255.0), u8 > 10


def mask_preprocess(mask, device, dtype, batch_size, channel, frames, downscale=8):
    """-> (float [b,c,F,h,w] in ``dtype``, bool [b,c,F,h,w]); a directory gives one mask per frame
    (``utils.py:113-154``), a single file is repeated over frames.  On a GPU device the resize / threshold / scaling run
    there (``mask_preprocess_device``: bit-identical to the PIL path below); PNG decoding stays on the host."""
    if torch.device(device).type == "cuda" and dtype == torch.float16:
        if osp.isdir(mask):
            paths = glob(osp.join(mask, "*.png"))
            paths.sort(key=lambda p: int(osp.basename(p).split(".")[0]))
            imgs = [np.asarray(Image.open(p).convert("L")) for p in paths[:frames]]
        else:
            imgs = [np.asarray(Image.open(mask).convert("L"))] * frames
        if len({im.shape for im in imgs}) == 1:
            return mask_preprocess_device(np.stack(imgs), device, dtype, batch_size, channel, downscale)
    if osp.isdir(mask):
        paths = glob(osp.join(mask, "*.png"))
        paths.sort(key=lambda p: int(osp.basename(p).split(".")[0]))
        paths = paths[:frames]
        fl, bl = zip(*[_one_mask(p, downscale) for p in paths])
        fl, bl = torch.stack(fl), torch.stack(bl)
    else:
        f1, b1 = _one_mask(mask, downscale)
        fl, bl = f1[None].repeat(frames, 1, 1), b1[None].repeat(frames, 1, 1)
    fl = fl.to(device, dtype)[None, None].repeat(batch_size, channel, 1, 1, 1)
    bl = bl.to(device)[None, None].repeat(batch_size, channel, 1, 1, 1)
    return fl, bl
