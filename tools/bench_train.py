#!/usr/bin/env python3
"""Training-step benchmark of the diffusion UNet (BASELINE config 4 shape family; reported beside, never instead of,
bench.py's steps/s):  python tools/bench_train.py [--dims 24 24 24] [--batch 1] [--steps 20] [--cond 4]

One step = UNet forward (training plan) + MSE + backward plan + [all-reduce of the flat gradients when launched under
torchrun] + gradient-norm clip + fused Adam + bf16 re-pack of the weights.  Phases are timed with torch.cuda.Event on
the current stream (the library launches on torch's current stream).  Prints one JSON line."""
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
    ap.add_argument("--dims", type=int, nargs=3, default=[24, 24, 24])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cond", type=int, default=0, help="extra concat-conditioning channels (mode='concat')")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"], help="fp32: the reference's own training arithmetic (csrc/f32_train.h)")
    args = ap.parse_args()

    import torch
    import torch.nn.functional as F
    import cfgs
    from ldm3d.networks import DiffusionModelUNet
    from ldm3d.optim import FlatAdam
    from ldm3d.trainer import GradSync

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        from ldm3d import parallel
        parallel.setup_ddp(rank, world)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    cfg = dict(cfgs.UNET_FULL, in_channels=4 + args.cond)
    m = DiffusionModelUNet(**cfg)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / p[0].numel() ** 0.5))
    m = m.to(dev).train().set_precision(args.precision)
    opt = FlatAdam(m, lr=5e-6, max_grad_norm=1.0)
    sync = GradSync()
    sync.broadcast(m.flat_params)
    B, dims = args.batch, tuple(args.dims)
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    x = torch.randn((B, 4, *dims), device=dev, generator=gen)
    cond = torch.randn((B, args.cond, *dims), device=dev, generator=gen) if args.cond else None
    noise = torch.randn((B, 4, *dims), device=dev, generator=gen)
    t = torch.randint(0, 1000, (B,), device=dev, generator=gen).float()
    names = ["forward", "loss", "backward", "allreduce", "optimizer"]
    acc = {n: 0.0 for n in names}

    def step(timed):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        ev[0].record()
        pred = m(x=x, timesteps=t, cond=cond)            # includes the bf16 re-pack of the updated weights
        ev[1].record()
        loss = F.mse_loss(pred.float(), noise)
        ev[2].record()
        loss.backward()
        ev[3].record()
        sync.mean_(m.flat_grads)
        ev[4].record()
        opt.step()
        ev[5].record()
        if timed:
            torch.cuda.synchronize()
            for i, n in enumerate(names):
                acc[n] += ev[i].elapsed_time(ev[i + 1])
        return loss

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(False)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    for _ in range(5):
        step(True)
    assert torch.isfinite(loss)
    if rank == 0:
        print(json.dumps({"metric": "UNet train steps/sec (fwd + bwd + grad all-reduce + clip + Adam)", "value": world * args.steps / dt,
                          "unit": "steps/s", "n_gpus": world, "ms_per_step": dt / args.steps * 1e3, "batch_per_gpu": B, "dims": dims,
                          "in_channels": 4 + args.cond, "phase_ms": {n: acc[n] / 5 for n in names},
                          "fwd_bwd_tflops": 3 * 889.1 * (dims[0] * dims[1] * dims[2] / 13824.0) * B / (dt / args.steps * 1e3) if not args.cond else None,
                          "loss": float(loss)}), flush=True)
    if world > 1:
        parallel.cleanup_ddp()


if __name__ == "__main__":
    main()
