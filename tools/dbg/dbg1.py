import torch, sys
sys.path.insert(0,'.')
from mvoc_amd import ops
from oracle import loops_ref
g = torch.Generator().manual_seed(32)
shp = (1, 4, 6, 9, 10)
lat, bgl = torch.randn(shp, generator=g).half(), torch.randn(shp, generator=g).half()
objs = torch.randn((2,) + shp, generator=g).half()
masks = (torch.randint(0, 256, (2,) + shp, generator=g).float() / 255).half()
ref = loops_ref.latent_fusion(lat, bgl, [objs[0], objs[1]], [masks[0], masks[1]], 0.01, False)
out = ops.latent_fusion(lat.cuda(), bgl.cuda(), objs.cuda(), masks.cuda(), 0.01, False).cpu()
bad = (out.view(torch.int16) != ref.view(torch.int16)).flatten().nonzero().flatten()
print(len(bad))
for i in bad[:8].tolist():
    print(i, float(out.flatten()[i]), float(ref.flatten()[i]), float(lat.flatten()[i]), float(bgl.flatten()[i]), float(objs[0].flatten()[i]), float(masks[0].flatten()[i]), float(objs[1].flatten()[i]), float(masks[1].flatten()[i]))
# subnormal conversion check
x = torch.tensor([1e-5, 3e-6, 6e-5, -2e-5], dtype=torch.float32)
one = torch.ones(4, dtype=torch.float16)
o = ops.latent_fusion((x*100).half().cuda(), torch.zeros(4).half().cuda(), torch.zeros(1,4).half().cuda(), torch.zeros(1,4).half().cuda(), 0.01, False).cpu()
print(o, (x*100).half().float()*0.01)
