"""Does HIP-graph replay of one UNet forward beat eager launches?  (diagnostic; torch.cuda.CUDAGraph captures the
library's launches because they go to torch's current stream.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
sys.argv = [sys.argv[0]]
import bench
dev = torch.device("cuda:0")
unet = bench.make_unet(dev)
x = torch.randn((1, 4, 24, 24, 24), device=dev)
t = torch.full((1,), 500.0, device=dev)
with torch.no_grad():
    for _ in range(5):
        y = unet(x=x, timesteps=t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        y = unet(x=x, timesteps=t)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 200
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            y = unet(x=x, timesteps=t)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        yg = unet(x=x, timesteps=t)
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 200
    ref = unet(x=x, timesteps=t)
    print(f"eager {eager*1e3:.3f} ms  graph replay {graph*1e3:.3f} ms  same result: {torch.equal(ref, yg)}")
