"""Drop-in for the reference's ``i2vgen-xl/utils.py`` (the helpers the denoising path uses)."""
import os

from PIL import Image

from common.filesystem import scan_dir
from mvoc_amd.utils import load_ddim_latents_at_t, mask_preprocess, seed_everything  # noqa: F401


def load_image(path):
    """diffusers.utils.load_image: open, EXIF-transpose, RGB"""
    from PIL import ImageOps
    return ImageOps.exif_transpose(Image.open(path)).convert("RGB")


def load_video_frames(frames_path, n_frames, image_size=(512, 512)):
    """``inverse.py:32-45``: numerically sorted frames, first n, LANCZOS-resized to image_size (W, H)"""
    _, paths = scan_dir(frames_path)
    paths.sort(key=lambda p: int(os.path.basename(p).split(".")[0]))
    paths = paths[:n_frames]
    frames = [load_image(p) for p in paths]
    frames = [f if f.size == tuple(image_size) else f.resize(tuple(image_size), resample=Image.Resampling.LANCZOS) for f in frames]
    return paths, frames


def convert_video_to_frames(video_path, img_size=(512, 512), save_frames=True):
    raise NotImplementedError("mp4 decoding needs torchvision/av, which this environment lacks: extract PNG frames first")


def export_to_gif(frames, path, fps=10):
    frames[0].save(path, save_all=True, append_images=frames[1:], optimize=False, duration=1000 // fps, loop=0)
    return path
