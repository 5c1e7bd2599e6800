import sys, torch
sys.path.insert(0, ".")
from mvoc_amd import ops
c, hw, nb = 320, 4096, 80
qkv = torch.randn(nb * hw, 3 * c, device="cuda").half()
for _ in range(4):
    ops.flash_attn(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], nbatch=nb, heads=5, tq=hw, tk=hw)
torch.cuda.synchronize()
