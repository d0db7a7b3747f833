#!/usr/bin/env python3
"""Per-step trace of bench.py's headline region at the DRIVER's flags (--steps 20 --warmup 5 by default): a HIP event after every
step of warm-up and timed region plus the host clock at every enqueue, so that a one-off stall (graph instantiate / upload, derived
weights, clock ramp) can be told from a per-step cost.  Prints one row per step and a summary.

    python tools/short_run.py [--steps 20] [--warmup 5] [--repeat 3]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeat", type=int, default=3, help="further timed regions of the same length, back to back")
    ap.add_argument("--eager", action="store_true")
    args = ap.parse_args()
    import torch
    import bench
    import cfgs
    from ldm3d.schedulers import DDPMScheduler
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    t_setup = time.perf_counter()
    unet = bench.make_unet(dev, seed=0)
    if not args.eager:
        unet.enable_graph_replay(True)
    sch = DDPMScheduler(**cfgs.SCHED)
    gen = torch.Generator(device=dev).manual_seed(1234)
    x = torch.randn((1, 4, 24, 24, 24), device=dev, generator=gen)
    tbuf = torch.empty((1,), dtype=torch.float32, device=dev)
    sampler = sch.device_sampler(seed=1234)
    T = sch.num_train_timesteps
    print(f"set-up {time.perf_counter() - t_setup:.2f} s")

    def step(i, x):
        if i % T == 0:
            sampler.reset(tbuf)
        return unet.denoise_step(x, tbuf, sampler)

    n_total = args.warmup + args.steps * (1 + args.repeat)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_total + 1)]
    host = []
    with torch.no_grad():
        torch.cuda.synchronize()
        ev[0].record()
        h0 = time.perf_counter()
        k = 0
        regions = []
        for i in range(args.warmup):
            t_in = time.perf_counter()
            x = step(k, x)
            host.append((t_in - h0, time.perf_counter() - t_in))
            k += 1
            ev[k].record()
        for r in range(1 + args.repeat):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            first = k
            for i in range(args.steps):
                t_in = time.perf_counter()
                x = step(k, x)
                host.append((t_in - h0, time.perf_counter() - t_in))
                k += 1
                ev[k].record()
            torch.cuda.synchronize()
            regions.append((first, k, (time.perf_counter() - t0) * 1e3))
    print("step  phase     gpu_end_ms  gpu_delta_ms  host_enqueue_at_ms  host_call_ms")
    for j in range(1, k + 1):
        end = ev[0].elapsed_time(ev[j])
        d = ev[j - 1].elapsed_time(ev[j])
        phase = "warmup" if j <= args.warmup else f"timed{(j - 1 - args.warmup) // args.steps}"
        print(f"{j - 1:4d}  {phase:8s}  {end:10.3f}  {d:12.3f}  {host[j - 1][0] * 1e3:18.3f}  {host[j - 1][1] * 1e3:12.3f}")
    for first, last, wall in regions:
        ds = sorted(ev[j].elapsed_time(ev[j + 1]) for j in range(first, last))
        n = len(ds)
        print(f"region steps {first}-{last - 1}: wall {wall:.3f} ms = {wall / n:.4f} ms/step = {n / wall * 1e3:.1f} steps/s; "
              f"per-step events: min {ds[0]:.4f} median {ds[n // 2]:.4f} max {ds[-1]:.4f} sum {sum(ds):.3f}")


if __name__ == "__main__":
    main()
