import cProfile, pstats, time, sys, os
sys.path[:0] = [os.path.join(os.getcwd(), "i2vgen-xl"), os.getcwd()]
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
os.environ["MVOC_SYNTHETIC_VAE"] = os.environ["MVOC_SYNTHETIC_CLIP"] = "1"
import inverse
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
p = inverse.build_pipeline(torch.device("cuda:0"), True)
torch.cuda.synchronize()
pr.disable()
print("build_pipeline()", round(time.time() - t0, 2))
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
