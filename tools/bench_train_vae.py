#!/usr/bin/env python3
"""Training-step benchmark of the stage-1 AutoencoderKL (3d_ldm/train_autoencoder.py:352-451, generator step in the warm-up
regime: reconstruction L1 + KL, clip 0.5, AdamW) on the HIP forward / backward plans:

    python tools/bench_train_vae.py [--dims 64 64 64] [--batch 1] [--steps 10]"""
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
    ap.add_argument("--dims", type=int, nargs=3, default=[64, 64, 64])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    import torch
    import cfgs
    from ldm3d.networks import AutoencoderKL
    from ldm3d.trainer import AutoencoderTrainer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    vae = AutoencoderKL(**cfgs.VAE_FULL)
    with torch.no_grad():
        for p in vae.parameters():
            if p.dim() > 1:
                p.normal_(0.0, 0.5 / p[0].numel() ** 0.5)
    vae = vae.to(dev).train()
    tr = AutoencoderTrainer(vae, lr=1e-5, kl_weight=1e-6)
    x = torch.rand((args.batch, 1, *args.dims), device=dev)
    for _ in range(args.warmup):
        losses, skipped = tr.train_step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses, skipped = tr.train_step(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"metric": "AutoencoderKL train step (fwd + bwd + clip + AdamW)", "ms_per_step": dt * 1e3, "steps_per_s": 1.0 / dt,
                      "dims": args.dims, "batch": args.batch, "skipped": bool(skipped),
                      "loss_g": float(losses.get("loss_g", float("nan")))}))


if __name__ == "__main__":
    main()
