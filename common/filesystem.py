"""glob helper the drivers import (reference ``common/filesystem.py:20-30``)"""
import os.path as osp
from glob import glob


def scan_dir(path, recursive=False, exts=("jpg", "jpeg", "png")):
    files = []
    for ext in exts:
        pattern = osp.join(path, "**", f"*.{ext}") if recursive else osp.join(path, f"*.{ext}")
        files.extend(glob(pattern, recursive=recursive))
    return len(files), files
