"""Shared scaffolding of the G8 loop tests (tests/golden/g8_loops.npz, produced by tools/gen_golden.py: gen_loops from
the REFERENCE's own ``invert`` / ``__call__`` / ``sample_with_pnp_...`` methods).  ``fake_unet`` is the stand-in UNet the
fixture was recorded with: elementwise fp16 arithmetic only, bit-reproducible on CPU and GPU."""
import numpy as np
import torch


def fake_unet(x, t, ehs, fps, ilf, il, ie):
    s = (ehs[:, 0, 0].float() * 0.01 + ie[:, 0, 0].float() * 0.02 + fps.float() * 0.001 + float(t) * 1e-4).to(x.dtype)
    return x * 0.5 + il * 0.25 - ilf * 0.125 + s[:, None, None, None, None]


def seeded(key, shape, scale=1.0):
    g = torch.Generator().manual_seed(int(key) % (2 ** 31))
    return (torch.randn(shape, generator=g) * scale).half()


def prompt_key(s):
    return 7 + sum(str(s).encode())


class Calls:
    """what the reference handed to the UNet at every step of one recorded loop"""

    def __init__(self, g, tag):
        self.n = int(g[f"{tag}_ncalls"])
        self.t = [int(v) for v in g[f"{tag}_t"]]
        self.hook_t = [int(v) for v in g[f"{tag}_hook_t"]]
        for k in ("x", "ehs", "fps", "ilf", "il", "ie"):
            setattr(self, k, torch.from_numpy(np.asarray(g[f"{tag}_{k}"])))
