#!/usr/bin/env python3
"""Throughput of K INDEPENDENT denoising chains sharing one GPU (what `inference.py -n N` runs per rank): every chain has its
own stream, module instance (weights + workspace) and HIP graph, so launches of different chains overlap and fill the
per-launch floor and the partially filled rounds of the single-chain step.  The headline metric (bench.py) stays ONE chain
per GPU; this is the serving-throughput figure next to it.

    python tools/bench_chains.py --chains 1 2 3 4 --steps 100"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, nargs="+", default=[1, 2, 3, 4])
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1, help="latents per chain (one forward of this batch per step)")
    args = ap.parse_args()
    import torch
    import bench
    import cfgs
    from ldm3d.schedulers import DDPMScheduler
    dev = torch.device("cuda:0")
    K = max(args.chains)
    nets = [bench.make_unet(dev, seed=i).enable_graph_replay(True) for i in range(K)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
    sch = DDPMScheduler(**cfgs.SCHED)
    T = sch.num_train_timesteps
    gens = [torch.Generator(device=dev).manual_seed(100 + i) for i in range(K)]
    xs = [torch.randn((args.batch, 4, 24, 24, 24), device=dev, generator=gens[i]) for i in range(K)]
    tbufs = [torch.empty((args.batch,), dtype=torch.float32, device=dev) for _ in range(K)]
    torch.cuda.synchronize()

    def run(k, nsteps, base):
        for i in range(nsteps):
            t = (T - 1 - base - i) % T
            for c in range(k):
                with torch.cuda.stream(streams[c]):
                    tbufs[c].fill_(float(t))
                    eps = nets[c](x=xs[c], timesteps=tbufs[c])
                    xs[c] = sch.step(eps, t, xs[c], generator=gens[c])[0]

    out = {}
    with torch.no_grad():
        for k in args.chains:
            run(k, args.warmup, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(k, args.steps, args.warmup)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[k] = k * args.batch * args.steps / dt
            print(f"chains {k} x batch {args.batch}: {out[k]:8.1f} latent-steps/s total  ({dt / args.steps * 1e3:.3f} ms per round of {k})", flush=True)
    print(json.dumps({"chains_steps_per_s": out, "finite": bool(all(torch.isfinite(x).all() for x in xs))}))


if __name__ == "__main__":
    main()
