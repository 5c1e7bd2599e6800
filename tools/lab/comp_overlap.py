"""How much would a batch-5 composition step gain from running beside independent work?  Two independent composition states (two
jobs) replayed (a) one after the other on one stream, (b) at the same time on two streams; and the split the reference's batch
allows on ONE job -- nothing here changes the product."""
import sys, time
sys.path.insert(0, '.')
import torch
import bench
job = bench.Job(torch.device("cuda:0"), 16, 64, True)
job.mix = "job"
pipe = job.pipe
import copy
st1 = job.comp_state
cond = st1["cond"]
st2 = pipe.make_composition_state(job.comp_latents.clone(), cond, st1["masks"], 9.0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def run(st, i):
    t = int(job.sched.timesteps[5 + i % 40])
    bg, o1, o2 = job.src[(0, t)], job.src[(1, t)], job.src[(2, t)]
    pipe.composition_step(st, t, bg, [o1, o2], job.comp_table[job.comp_index[t]], None)

for i in range(3):
    run(st1, i); run(st2, i)
torch.cuda.synchronize()

def timed(fn, n=4):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n): fn(i)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best

def seq(i):
    run(st1, i); run(st2, i)

def conc(i):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): run(st1, i)
    with torch.cuda.stream(s2): run(st2, i)
    cur.wait_stream(s1); cur.wait_stream(s2)

a = timed(lambda i: run(st1, i))
b = timed(seq)
c = timed(conc)
print(f"one composition step {a:.2f} ms; two states one after the other {b:.2f} ms; two states on two streams {c:.2f} ms "
      f"({c / b:.3f} of sequential)")
