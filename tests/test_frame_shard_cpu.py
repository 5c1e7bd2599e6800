"""CPU, world_size 2 over gloo: the frame <-> pixel exchanges of the frame-axis shard (cfg 4) against plain
slicing of the whole clip, both exchange forms, batch 1 / 2 / 5."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(nproc, port, *args, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "tests", "shard_worker.py"), *args]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("world", [2])
def test_exchanges_world2(tmp_path, world):
    r = launch(world, 29541, "exchange", str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    for rank in range(world):
        rep = json.load(open(tmp_path / f"exchange_r{rank}.json"))
        assert rep["cases"] == 8


def test_frame_shard_needs_process_group():
    from mvoc_amd.frame_shard import FrameShard
    with pytest.raises(RuntimeError):
        FrameShard()
