import sys, torch, time
sys.path.insert(0, '.')
from mvoc_amd.unet import I2VGenXLUNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = I2VGenXLUNet(device="cuda:0").init_random(8888)
torch.cuda.synchronize(); print("weights ok", torch.cuda.memory_allocated()/1e9, flush=True)
F,h,w=16,64,64
g=torch.Generator().manual_seed(0)
x=torch.randn(B,4,F,h,w,generator=g).half().cuda()
il=torch.randn(B,4,F,h,w,generator=g).half().cuda()
ie=torch.randn(B,F,1024,generator=g).half().cuda()
eh=torch.randn(B,77,1024,generator=g).half().cuda()
fps=torch.full((B,),8.0).cuda()
t=torch.tensor([981.0]).cuda()
out=eng.forward_ext(x,t,fps,il,il,ie,eh)[0]
torch.cuda.synchronize()
print(out.shape, float(out.float().std()), bool(torch.isfinite(out).all()))
t0=time.time()
for _ in range(3): out=eng.forward_ext(x,t,fps,il,il,ie,eh)[0]
torch.cuda.synchronize(); print("eager ms/fwd", (time.time()-t0)/3*1e3)
