#!/usr/bin/env python3
"""Sampling harness on the MI355X-native path: the command line and JSON schema of the reference's 3d_ldm/inference.py
(flags :32-52, config merge :61-67, networks / scheduler / inferer :71-85, sampling loop :88-102).

    python inference.py -e config/environment.json -c config/config_train_16g.json -n 1 [--steps 1000] [--random-init]

Extensions, all opt-in: --steps N switches to an N-step DDIM schedule (the reference always runs the full DDPM chain);
--random-init skips the checkpoints (synthetic smoke runs); under torchrun with -g > 1 the -n samples are dealt to the
ranks round-robin (independent chains, no collective: the reference is single process); --batch B denoises B volumes
per chain in one forward and --chains K advances K independent chains concurrently, each on its own stream
(tools/bench_chains.py: 1.6x the latent-steps/s of one-at-a-time at B = 4, 1.9x with 2 chains x B = 4); --condition FILE
runs the conditional sampling the trained model is for: the low-count volume of an NPZ pair is
cropped / percentile-scaled like the training data (3d_ldm/utils.py:94-143), encoded by the autoencoder and concatenated
to the noisy latent at every step (mode="concat", 3d_ldm/train_diffusion.py:326-333), with the scale factor that
train_diffusion.py saved (model_dir/scale_factor.json, or --scale-factor).  Volumes are written as NIfTI-1 by this
package's own writer (nibabel is not a dependency)."""
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
    ap.add_argument("--batch", type=int, default=1, help="volumes denoised together in one chain")
    ap.add_argument("--chains", type=int, default=1, help="independent chains advanced concurrently on this GPU (own stream + module instance each)")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host instead of replaying the UNet's HIP graph")
    ap.add_argument("--condition", default=None, help="NPZ pair whose low-count volume conditions the sampling (mode='concat')")
    ap.add_argument("--precision", default=None, choices=["bf16", "fp32"],
                        help="arithmetic of the networks: bf16 (default, the fast path) or fp32 (the reference's own arithmetic, 1e-5 from its CPU path; also LDM_PRECISION)")
    ap.add_argument("--scale-factor", type=float, default=None, help="latent scale (default: model_dir/scale_factor.json, else 1.0)")
    ns = ap.parse_args()
    if ns.precision:
        os.environ["LDM_PRECISION"] = ns.precision     # read by every network at construction (networks.py)
    for path in (ns.environment_file, ns.config_file):           # both JSON files land on the namespace, config last
        with open(path) as fh:
            vars(ns).update(json.load(fh))
    return ns


def load_networks(ns, device, only_unet=False, like=None):
    import torch
    from ldm3d.config import define_instance
    if only_unet:                                                  # another instance with the same weights
        net = define_instance(ns, "diffusion_def")
        net.load_state_dict(like.state_dict())
        return net.to(device).eval()
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


def resolve_scale_factor(ns) -> float:
    if ns.scale_factor is not None:
        return float(ns.scale_factor)
    path = os.path.join(getattr(ns, "model_dir", "."), "scale_factor.json")
    if os.path.exists(path):
        with open(path) as fh:
            return float(json.load(fh)["scale_factor"])
    return 1.0                                                   # the reference's hard-coded value (inference.py:85)


def condition_latent(path, patch, autoencoder, scale_factor, device):
    """Low-count volume of an NPZ pair -> centre crop, 0..99.5 percentile scaling -> scaled image latent [1, C, d, h, w]."""
    import numpy as np
    import torch
    from ldm3d.data import crop, crop_start, load_pair, scale_percentiles
    image, _ = load_pair(path)
    f = autoencoder.factor                                       # a volume smaller than the patch is used whole (the loaders clip the
    roi = [min(int(p), int(d)) // f * f for p, d in zip(patch, image.shape)]      # roi the same way), cut to a multiple of the VAE factor
    image = scale_percentiles(crop(image, crop_start(image.shape, roi, None), roi))
    x = torch.from_numpy(np.ascontiguousarray(image))[None, None].to(device)
    return autoencoder.encode_stage_2_inputs(x) * scale_factor


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
    unets = [unet]
    for _ in range(1, max(1, ns.chains)):                          # every chain owns its module instance (workspace + launch graph)
        unets.append(load_networks(ns, device, only_unet=True, like=unet))
    if not ns.eager:
        for u in unets:
            u.enable_graph_replay(True)
    scheduler = make_scheduler(ns)
    inferer = LatentDiffusionInferer(scheduler, scale_factor=resolve_scale_factor(ns))
    out_dir = Path(ns.output_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    patch = [int(p) for p in ns.diffusion_train["patch_size"]]
    cond = None
    if ns.condition:
        with torch.no_grad():
            cond = condition_latent(ns.condition, patch, autoencoder, inferer.scale_factor, device)
        if unet.in_channels != 2 * autoencoder.latent_channels:
            raise SystemExit(f"--condition needs a concat-conditioned UNet (in_channels {2 * autoencoder.latent_channels}), got {unet.in_channels}")
    elif unet.in_channels != unet.out_channels:
        raise SystemExit(f"this UNet is concat-conditioned (in_channels {unet.in_channels}, out_channels {unet.out_channels}): pass --condition FILE")
    lat_ch = autoencoder.latent_channels if cond is not None else unet.in_channels
    todo = list(parallel.shard_indices(ns.num, rank, world))
    bsz, nch = max(1, ns.batch), max(1, ns.chains)
    for lo in range(0, len(todo), bsz * nch):                     # one round = up to `chains` batches of up to `batch` volumes
        groups = [todo[g:g + bsz] for g in range(lo, min(lo + bsz * nch, len(todo)), bsz)]
        spatial = list(cond.shape[2:]) if cond is not None else [p // autoencoder.factor for p in patch]
        zs = [torch.randn([len(ids), lat_ch] + spatial, dtype=torch.float32).to(device) for ids in groups]   # host draw then move, as the reference does
        cs = [None if cond is None else cond.expand(len(ids), -1, -1, -1, -1).contiguous() for ids in groups]
        t0 = time.perf_counter()
        with torch.no_grad():
            if len(groups) == 1:
                kw = {} if cond is None else dict(conditioning=cs[0], mode="concat")
                vols = [inferer.sample(input_noise=zs[0], autoencoder_model=autoencoder, diffusion_model=unet, scheduler=scheduler, **kw)]
            else:
                vols = inferer.sample_concurrent(zs, autoencoder, unets, scheduler=scheduler, conditionings=cs,
                                                 mode="concat" if cond is not None else "crossattn")
        torch.cuda.synchronize()
        for ids, vol in zip(groups, vols):
            for j, idx in enumerate(ids):
                stem = out_dir / time.strftime(f"synimg_%Y%m%d_%H%M%S_r{rank}_{idx}")
                written = save_nifti(vol[j, 0].unsqueeze(-1).cpu().numpy(), str(stem))
                log.info("rank %d: %s %s", rank, written, tuple(vol.shape[1:]))
        log.info("rank %d: %d volume(s) in %.2f s", rank, sum(len(g) for g in groups), time.perf_counter() - t0)
    if world > 1:
        parallel.cleanup_ddp()


if __name__ == "__main__":
    logging.basicConfig(stream=sys.stdout, level=logging.INFO, format="%(asctime)s %(levelname)s %(name)s: %(message)s")
    main()
