"""One process per GPU, no collectives: the reference iterates its group config sequentially on one device
(``inverse.py:136``, ``composite.py:87``); per-video inversions and per-entry compositions share nothing but
read-only weights, so entry k goes to process k % n.

    torchrun --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 inverse.py ...      (RANK / WORLD_SIZE / LOCAL_RANK)
or  python inverse.py --shard 2/4 ...
"""
import os

import torch


def shard_of(shard=None):
    if shard:
        i, n = shard.split("/")
        return int(i), int(n)
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def my_entries(configs_list, shard=None):
    """active entries are dealt round-robin; inactive ones keep their slot so logs match the reference"""
    i, n = shard_of(shard)
    out, k = [], 0
    for e in configs_list:
        if not e.get("active", True):
            continue
        if k % n == i:
            out.append(e)
        k += 1
    return out


def pick_device(template_device, shard=None):
    """template says e.g. "cuda:0"; under a multi-process launch each rank takes its LOCAL_RANK / shard index"""
    i, n = shard_of(shard)
    if n == 1 or not str(template_device).startswith("cuda"):
        return torch.device(template_device)
    idx = int(os.environ.get("LOCAL_RANK", i))
    ndev = torch.cuda.device_count()
    if idx >= max(ndev, 1):
        # more ranks than GPUs: refuse rather than silently stacking two shards on one device (one process per GPU);
        # the one-GPU test boxes opt in explicitly
        if os.environ.get("MVOC_ALLOW_GPU_SHARING") != "1":
            raise RuntimeError(f"rank with LOCAL_RANK / shard index {idx} has no GPU of its own ({ndev} visible); "
                               "launch at most one process per GPU (or set MVOC_ALLOW_GPU_SHARING=1 for tests)")
        idx %= max(ndev, 1)
    return torch.device("cuda", idx)
