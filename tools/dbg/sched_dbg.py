import math, hashlib, torch, numpy as np
def H(t): return hashlib.md5(t.numpy().tobytes()).hexdigest()[:10]
n=1000
def alpha_bar(t): return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
b64=[min(1 - alpha_bar((i+1)/n) / alpha_bar(i/n), 0.999) for i in range(n)]
print("betas64", hashlib.md5(np.array(b64).tobytes()).hexdigest()[:10])
betas=torch.tensor(b64, dtype=torch.float32); print("betas32", H(betas))
cp=torch.cumprod(1.0-betas, dim=0); print("cumprod1", H(cp), "np seq", hashlib.md5(np.cumprod((1.0-betas).numpy()).tobytes()).hexdigest()[:10])
r=cp.sqrt(); print("sqrt", H(r))
s0,sT=r[0].clone(), r[-1].clone()
r2=(r-sT)*(s0/(s0-sT)); print("rescale", H(r2))
ab=r2**2; print("sq", H(ab))
al=torch.cat([ab[0:1], ab[1:]/ab[:-1]]); print("alphas", H(al))
b2=1-al; print("betas2", H(b2))
f=torch.cumprod(1.0-b2, dim=0); print("final", H(f), float(f[1]))
print(torch.__version__, torch.backends.cpu.get_cpu_capability() if hasattr(torch.backends,"cpu") else None)
