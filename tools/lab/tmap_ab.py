"""A/B of the temporal convs' tile order (gemm8.hip: tmap_t): MVOC_G8_TMAP=0 (row order) against the default (frames of a 256-pixel
patch first), one subprocess per setting, same box, min of 5 x 10 launches."""
import os, subprocess, sys
sys.path.insert(0, '.')
if len(sys.argv) == 1:
    for rep in range(2):
        for v in ("0", "1"):
            print(f"== MVOC_G8_TMAP={v} (pass {rep})", flush=True)
            subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, MVOC_G8_TMAP=v))
    sys.exit(0)
import torch
from mvoc_amd import ops
from mvoc_amd.unet import pack_tconv
torch.manual_seed(0)
for nvid, c, hw in [(5, 320, 4096), (5, 640, 1024), (1, 320, 4096), (1, 640, 1024), (3, 320, 4096)]:
    frames = 16
    m = nvid * frames * hw
    x = (torch.randn(m, c) * 0.7).half().cuda()
    w = pack_tconv((torch.randn(c, c, 3, 1, 1) / (3 * c) ** 0.5).half().cuda())
    b = torch.randn(c).half().cuda()
    for resid in (None, x):
        for _ in range(3): ops.tconv3(x, w, b, nvid=nvid, frames=frames, hw=hw, resid=resid, sums=True)
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.tconv3(x, w, b, nvid=nvid, frames=frames, hw=hw, resid=resid, sums=True)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        print(f"tconv M={m} C={c} hw={hw} resid={int(resid is not None)}: {best:7.1f} us {2*m*c*3*c/best/1e6:6.0f} TF", flush=True)
