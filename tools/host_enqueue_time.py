import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
sys.argv = [sys.argv[0]]
import bench, cfgs
from ldm3d.schedulers import DDPMScheduler
dev = torch.device("cuda:0")
unet = bench.make_unet(dev)
sch = DDPMScheduler(**cfgs.SCHED)
gen = torch.Generator(device=dev).manual_seed(1)
x = torch.randn((1, 4, 24, 24, 24), device=dev)
tbuf = torch.empty((1,), device=dev)
def step(i, x):
    t = 999 - i % 1000
    tbuf.fill_(float(t))
    eps = unet(x=x, timesteps=tbuf)
    return sch.step(eps, t, x, generator=gen)[0]
with torch.no_grad():
    for i in range(20): x = step(i, x)
    torch.cuda.synchronize()
    # CPU enqueue time: run 50 steps, measure host time until all enqueued (GPU lags behind)
    t0 = time.perf_counter()
    for i in range(50): x = step(i, x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/50:.3f} ms/step; total {1e3*(t2-t0)/50:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
with torch.no_grad():
    for i in range(50): x = step(i, x)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
