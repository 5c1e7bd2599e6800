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
    (``utils.py:113-154``), a single file is repeated over frames."""
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
