#!/usr/bin/env python3
"""inference.py - same command line and JSON schema as the reference's 3d_ldm/inference.py (:32-52 flags, :61-67 config
merge, :71-85 model / scheduler / inferer set-up, :88-102 sampling loop), running on the MI355X-native path.

    python inference.py -e config/environment.json -c config/config_train_16g.json -n 1 [--steps 1000] [--random-init]

Differences, all opt-in: --steps N uses an N-step DDIM schedule instead of the reference's 1000-step DDPM;
--random-init runs without checkpoints (synthetic smoke runs); with -g > 1 under torchrun the -n samples are sharded
over ranks (independent chains, no collective) - the reference is single process."""
import argparse
import json
import logging
import os
import sys
from datetime import datetime
from pathlib import Path

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    parser = argparse.ArgumentParser(description="3D latent diffusion inference (MI355X-native)")
    parser.add_argument("-e", "--environment-file", default="./config/environment.json")
    parser.add_argument("-c", "--config-file", default="./config/config_train_32g.json")
    parser.add_argument("-n", "--num", type=int, default=1, help="number of generated images")
    parser.add_argument("-g", "--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=0, help="0 = reference behaviour (DDPM, all train timesteps)")
    parser.add_argument("--random-init", action="store_true", help="no checkpoints: random weights (smoke runs)")
    parser.add_argument("--seed", type=int, default=42)
    args = parser.parse_args()

    import torch
    from ldm3d import parallel
    from ldm3d.config import define_instance
    from ldm3d.inferer import LatentDiffusionInferer
    from ldm3d.nifti import save_nifti
    from ldm3d.schedulers import DDIMScheduler, DDPMScheduler

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        parallel.setup_ddp(rank, world)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    for path in (args.environment_file, args.config_file):        # JSON -> Namespace merge (inference.py:61-67)
        for k, v in json.load(open(path)).items():
            setattr(args, k, v)
    torch.manual_seed(args.seed + rank)

    autoencoder = define_instance(args, "autoencoder_def")
    diffusion_model = define_instance(args, "diffusion_def")
    if not args.random_init:
        autoencoder.load_state_dict(torch.load(os.path.join(args.model_dir, "autoencoder.pt"), weights_only=True))
        diffusion_model.load_state_dict(torch.load(os.path.join(args.model_dir, "diffusion_unet.pt"), weights_only=True))
    else:
        with torch.no_grad():
            for m in (autoencoder, diffusion_model):
                for p in m.parameters():
                    if float(p.abs().max()) == 0.0 and p.dim() > 1:
                        p.normal_(0.0, 0.02)
    autoencoder, diffusion_model = autoencoder.to(device).eval(), diffusion_model.to(device).eval()

    ns = args.NoiseScheduler
    kw = dict(num_train_timesteps=ns["num_train_timesteps"], schedule="scaled_linear_beta",
              beta_start=ns["beta_start"], beta_end=ns["beta_end"])
    if args.steps and args.steps < ns["num_train_timesteps"]:
        scheduler = DDIMScheduler(**kw)
        scheduler.set_timesteps(args.steps)
    else:
        scheduler = DDPMScheduler(**kw)
    inferer = LatentDiffusionInferer(scheduler, scale_factor=1.0)

    Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    latent_shape = [p // 4 for p in args.diffusion_train["patch_size"]]
    noise_shape = [1, diffusion_model.in_channels] + latent_shape
    for i in parallel.shard_indices(args.num, rank, world):
        noise = torch.randn(noise_shape, dtype=torch.float32).to(device)
        with torch.no_grad():
            img = inferer.sample(input_noise=noise, autoencoder_model=autoencoder, diffusion_model=diffusion_model,
                                 scheduler=scheduler)
        name = os.path.join(args.output_dir, datetime.now().strftime(f"synimg_%Y%m%d_%H%M%S_r{rank}_{i}"))
        out = save_nifti(img[0, 0, ...].unsqueeze(-1).cpu().numpy(), name)
        logging.info("rank %d wrote %s  shape %s", rank, out, tuple(img.shape))
    if world > 1:
        parallel.cleanup_ddp()


if __name__ == "__main__":
    logging.basicConfig(stream=sys.stdout, level=logging.INFO,
                        format="[%(asctime)s.%(msecs)03d][%(levelname)5s](%(name)s) - %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
    main()
