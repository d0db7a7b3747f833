#!/usr/bin/env python3
"""BASELINE.json configs[3] and configs[4] on ONE MI355X (per-GPU share of the 8-GPU jobs; the multi-GPU runs are the driver's).

  config 4: train_diffusion step on a 144x176x112 patch (latent 36x44x28), batch 1 per GPU: label/image VAE encodes (no grad)
            + UNet forward/backward + clip + Adam.  The gradient all-reduce is not in this number (one GPU).
  config 5: 50-step DDIM on 4x40x56x40 latents + VAE decode to 160x224x160, B volumes per GPU.

    python tools/bench_configs.py [--skip-train] [--batch 1 2]"""
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
    ap.add_argument("--skip-train", action="store_true")
    ap.add_argument("--skip-sample", action="store_true")
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--train-steps", type=int, default=6)
    args = ap.parse_args()
    import torch
    import torch.nn.functional as F
    import bench
    import cfgs
    from ldm3d.inferer import LatentDiffusionInferer
    from ldm3d.networks import AutoencoderKL, DiffusionModelUNet
    from ldm3d.optim import FlatAdam
    from ldm3d.schedulers import DDIMScheduler, DDPMScheduler
    dev = torch.device("cuda:0")
    out = {}
    vae_cfg = dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=4, channels=[64, 128, 256], num_res_blocks=2,
                   norm_num_groups=32, norm_eps=1e-6, attention_levels=[False, False, False],
                   with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False)
    torch.manual_seed(0)
    vae = AutoencoderKL(**vae_cfg)
    with torch.no_grad():
        for p in vae.parameters():
            if p.dim() > 1:
                p.normal_(0.0, 0.02)
    vae = vae.to(dev).eval()

    if not args.skip_train:
        unet = bench.make_unet(dev, seed=0, in_channels=8)        # concat-conditioned: noisy label latent | image latent
        unet.train()
        opt = FlatAdam(unet, lr=1e-5, max_grad_norm=1.0)
        sch = DDPMScheduler(**cfgs.SCHED)
        inferer = LatentDiffusionInferer(sch, scale_factor=1.0)
        images, labels = torch.rand((1, 1, 144, 176, 112), device=dev), torch.rand((1, 1, 144, 176, 112), device=dev)

        def step():
            with torch.no_grad():
                cond = vae.encode_stage_2_inputs(images)
            noise = torch.randn((1, 4, 36, 44, 28)).to(dev)
            t = torch.randint(0, 1000, (1,), device=dev).long()
            pred = inferer(inputs=labels, autoencoder_model=vae, diffusion_model=unet, noise=noise, timesteps=t, condition=cond, mode="concat")
            loss = F.mse_loss(pred.float(), noise.float())
            loss.backward()
            opt.step()
            return loss
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.train_steps):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.train_steps
        out["config4_train_step_ms_per_gpu"] = dt * 1e3
        out["config4_train_steps_per_s_per_gpu"] = 1.0 / dt
        out["config4_loss_finite"] = bool(torch.isfinite(loss))
        print(f"config 4 (one GPU, batch 1, 144x176x112 patch): {dt * 1e3:.1f} ms per train step", flush=True)
        del unet, opt
        torch.cuda.empty_cache()

    if not args.skip_sample:
        unet = bench.make_unet(dev, seed=1).eval()
        sch = DDIMScheduler(**cfgs.SCHED)
        sch.set_timesteps(50)
        inferer = LatentDiffusionInferer(sch, scale_factor=1.0)
        for b in args.batch:
            z = torch.randn((b, 4, 40, 56, 40), device=dev)
            with torch.no_grad():
                for rep in range(2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    lat = inferer.sample(input_noise=z, autoencoder_model=None, diffusion_model=unet, scheduler=sch)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    vol = vae.decode_stage_2_outputs(lat)
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
            out[f"config5_batch{b}"] = {"ddim50_s": t1 - t0, "decode_s": t2 - t1, "volumes_per_s": b / (t2 - t0),
                                        "shape": list(vol.shape), "finite": bool(torch.isfinite(vol).all())}
            print(f"config 5 (one GPU, batch {b}): 50 DDIM steps {t1 - t0:.3f} s + decode to {tuple(vol.shape[2:])} {t2 - t1:.3f} s "
                  f"= {b / (t2 - t0):.2f} volumes/s", flush=True)
            del vol, lat
            torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
