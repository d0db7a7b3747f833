#!/usr/bin/env python3
"""Sampling harness on the MI355X-native path: the command line and JSON schema of the reference's 3d_ldm/inference.py
(flags :32-52, config merge :61-67, networks / scheduler / inferer :71-85, sampling loop :88-102).

    python inference.py -e config/environment.json -c config/config_train_16g.json -n 1 [--steps 1000] [--random-init]

Extensions, all opt-in: --steps N switches to an N-step DDIM schedule (the reference always runs the full DDPM chain);
--random-init skips the checkpoints (synthetic smoke runs); under torchrun with -g > 1 the -n samples are dealt to the
ranks round-robin (independent chains, no collective: the reference is single process).  Volumes are written as
NIfTI-1 by this package's own writer (nibabel is not a dependency)."""
import argparse
import json
import logging
import os
import sys
import time
from pathlib import Path

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
log = logging.getLogger("inference")


def parse_cli():
    ap = argparse.ArgumentParser(description="3D latent diffusion sampling (MI355X-native)")
    ap.add_argument("-e", "--environment-file", default="./config/environment.json", help="JSON with data / model / output paths")
    ap.add_argument("-c", "--config-file", default="./config/config_train_32g.json", help="JSON with network and training hyper-parameters")
    ap.add_argument("-n", "--num", type=int, default=1, help="volumes to generate")
    ap.add_argument("-g", "--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0 = all training timesteps with DDPM (reference behaviour); N = N-step DDIM")
    ap.add_argument("--random-init", action="store_true", help="no checkpoints: random weights")
    ap.add_argument("--seed", type=int, default=42)
    ns = ap.parse_args()
    for path in (ns.environment_file, ns.config_file):           # both JSON files land on the namespace, config last
        with open(path) as fh:
            vars(ns).update(json.load(fh))
    return ns


def load_networks(ns, device):
    import torch
    from ldm3d.config import define_instance
    nets = {}
    for key, ckpt in (("autoencoder_def", "autoencoder.pt"), ("diffusion_def", "diffusion_unet.pt")):
        net = define_instance(ns, key)
        if ns.random_init:
            with torch.no_grad():                                  # MONAI zero-initialises some convs: give them values
                for p in net.parameters():
                    if p.dim() > 1 and not bool(p.any()):
                        p.normal_(0.0, 0.02)
        else:
            net.load_state_dict(torch.load(os.path.join(ns.model_dir, ckpt), weights_only=True))
        nets[key] = net.to(device).eval()
    return nets["autoencoder_def"], nets["diffusion_def"]


def make_scheduler(ns):
    from ldm3d.schedulers import DDIMScheduler, DDPMScheduler
    cfg = ns.NoiseScheduler
    kw = dict(num_train_timesteps=cfg["num_train_timesteps"], schedule="scaled_linear_beta", beta_start=cfg["beta_start"],
              beta_end=cfg["beta_end"])
    if 0 < ns.steps < cfg["num_train_timesteps"]:
        sch = DDIMScheduler(**kw)
        sch.set_timesteps(ns.steps)
        return sch
    return DDPMScheduler(**kw)


def main():
    ns = parse_cli()
    import torch
    from ldm3d import parallel
    from ldm3d.inferer import LatentDiffusionInferer
    from ldm3d.nifti import save_nifti

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        parallel.setup_ddp(rank, world)
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    torch.manual_seed(ns.seed + rank)

    autoencoder, unet = load_networks(ns, device)
    scheduler = make_scheduler(ns)
    inferer = LatentDiffusionInferer(scheduler, scale_factor=1.0)
    out_dir = Path(ns.output_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    shape = [1, unet.in_channels] + [int(p) // autoencoder.factor for p in ns.diffusion_train["patch_size"]]
    for idx in parallel.shard_indices(ns.num, rank, world):
        z = torch.randn(shape, dtype=torch.float32).to(device)    # host draw then move, as the reference does
        t0 = time.perf_counter()
        with torch.no_grad():
            vol = inferer.sample(input_noise=z, autoencoder_model=autoencoder, diffusion_model=unet, scheduler=scheduler)
        torch.cuda.synchronize()
        stem = out_dir / time.strftime(f"synimg_%Y%m%d_%H%M%S_r{rank}_{idx}")
        written = save_nifti(vol[0, 0].unsqueeze(-1).cpu().numpy(), str(stem))
        log.info("rank %d: %s %s in %.2f s", rank, written, tuple(vol.shape), time.perf_counter() - t0)
    if world > 1:
        parallel.cleanup_ddp()


if __name__ == "__main__":
    logging.basicConfig(stream=sys.stdout, level=logging.INFO, format="%(asctime)s %(levelname)s %(name)s: %(message)s")
    main()
