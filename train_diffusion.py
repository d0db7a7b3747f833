#!/usr/bin/env python3
"""train_diffusion.py - same command line, JSON schema, checkpoint names and per-step order as the reference's
3d_ldm/train_diffusion.py (:24-38 flags, :58-64 config merge, :90-124 autoencoder + scale factor, :127-156 UNet /
scheduler / inferer / Adam / MultiStepLR, :166-305 epoch loop with validation and rank-0 checkpoints), on the
MI355X-native path.

    python train_diffusion.py -e config/environment.json -c config/config_train_16g.json -g 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train_diffusion.py ... -g 8

Opt-in extras (never on by default): --random-init (no autoencoder checkpoint: smoke / benchmark runs),
--synthetic N (write N synthetic NPZ pairs into npz_dir first), --max-steps K (stop after K optimizer steps),
--reference-rng-order (also run the label encode the reference uses only to learn the latent shape),
--sample-steps N (reverse-diffusion steps of the periodic validation sample; 0 = the scheduler's full chain as in the reference).
Scalars go to <tfevent_path>/diffusion/scalars.jsonl (tensorboard is not a dependency here); the periodic conditional sample of
:308-359 (every 2 * val_interval epochs on rank 0) goes to <tfevent_path>/diffusion/samples/epoch_<n>.npz as centre slices."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def centre_slices(vol):
    """The three centre slices of a [D, H, W] volume (visualize_one_slice_in_3d_image of 3d_ldm/utils.py at the centre of each axis)."""
    d, h, w = vol.shape
    return [vol[d // 2].float().cpu().numpy(), vol[:, h // 2].float().cpu().numpy(), vol[:, :, w // 2].float().cpu().numpy()]


def sample_validation_volume(trainer, val_loader, device, out_dir, epoch, sample_steps, scalar, schedule_args=None):
    """3d_ldm/train_diffusion.py:306-359: noise in the shape of the label latents (:309-310), image latents of the first sample of the
    last validation batch (:324), ``inferer.sample(input_noise, autoencoder, unet, scheduler, conditioning=image_latents,
    mode="concat")`` (:326-333), then the centre slices of low-count input, high-count ground truth and the conditional sample
    (:335-359) -- written as one NPZ per epoch instead of TensorBoard images.  The chain runs on the device-resident sampler (one HIP
    graph launch per step) seeded from torch's RNG; ``--sample-steps N`` (an extra) replaces the full DDPM chain by N DDIM steps
    over the same beta schedule."""
    import numpy as np
    import torch
    from ldm3d.schedulers import DDIMScheduler
    batch = None
    for batch in val_loader:                                  # the reference uses whatever batch the validation loop ended on
        pass
    if batch is None:
        return None
    images, labels = batch["image"].to(device).float(), batch["label"].to(device).float()
    unet, autoencoder, inferer = trainer.unet, trainer.autoencoder, trainer.inferer
    was_training = unet.training
    unet.eval()
    with torch.no_grad():
        shape = autoencoder.encode_stage_2_inputs(labels[0:1]).shape
        test_noise = torch.randn(shape, dtype=torch.float32).to(device)
        image_latents = autoencoder.encode_stage_2_inputs(images[0:1])
        sch = inferer.scheduler
        if sample_steps and sample_steps < sch.num_train_timesteps and schedule_args:
            sch = DDIMScheduler(**schedule_args)
            sch.set_timesteps(sample_steps)
        t0 = time.perf_counter()
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        sample = inferer.sample(input_noise=test_noise, autoencoder_model=autoencoder, diffusion_model=unet, scheduler=sch,
                                conditioning=image_latents, mode="concat", fused_seed=seed)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    unet.train(was_training)
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"epoch_{epoch}.npz")
    arrays = {}
    for name, vol in (("val_lowcount_input", images[0, 0]), ("val_highcount_gt", labels[0, 0]), ("val_denoised_cond", sample[0, 0])):
        for axis, sl in enumerate(centre_slices(vol)):
            arrays[f"{name}_{axis}"] = sl
    np.savez_compressed(path, **arrays)
    err = float((sample[0, 0] - labels[0, 0]).abs().mean())
    scalar("val_denoised_cond_l1", err, epoch)
    print(f"Epoch {epoch}: conditional sample ({len(sch.timesteps)} steps, {dt:.2f} s) -> {path}, L1 vs ground truth {err:.4f}")
    return path


def main():
    parser = argparse.ArgumentParser(description="Latent diffusion model training (MI355X-native)")
    parser.add_argument("-e", "--environment-file", default="./config/environment.json")
    parser.add_argument("-c", "--config-file", default="./config/config_train_32g.json")
    parser.add_argument("-g", "--gpus", default=1, type=int, help="number of gpus per node")
    parser.add_argument("--random-init", action="store_true")
    parser.add_argument("--synthetic", type=int, default=0)
    parser.add_argument("--max-steps", type=int, default=0)
    parser.add_argument("--reference-rng-order", action="store_true")
    parser.add_argument("--gpu-transforms", action="store_true",
                        help="percentile intensity scaling on the GPU (ldm_op_scale_intensity_percentiles) instead of in the host loader")
    parser.add_argument("--precision", default=None, choices=["bf16", "fp32"],
                        help="arithmetic of the networks: bf16 (default, the fast path) or fp32 (the reference's own arithmetic, 1e-5 from its CPU path; also LDM_PRECISION)")
    parser.add_argument("--sample-steps", type=int, default=0,
                        help="steps of the periodic validation sample (0 = all num_train_timesteps, as 3d_ldm/train_diffusion.py:326-333)")
    parser.add_argument("--grad-allreduce-dtype", default="fp32", choices=["fp32", "bf16"],
                        help="wire format of the data-parallel gradient all-reduce (the reference's DDP uses fp32)")
    args = parser.parse_args()
    if args.precision:
        os.environ["LDM_PRECISION"] = args.precision     # read by every network at construction (networks.py)

    import torch
    from ldm3d import parallel
    from ldm3d.config import define_instance
    from ldm3d.data import prepare_dataloader, write_synthetic_pairs
    from ldm3d.inferer import LatentDiffusionInferer
    from ldm3d.schedulers import DDPMScheduler
    from ldm3d.trainer import DiffusionTrainer, GradSync, compute_scale_factor

    ddp = args.gpus > 1
    rank = int(os.environ.get("LOCAL_RANK", "0")) if ddp else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if ddp else 1
    if ddp:
        parallel.setup_ddp(int(os.environ.get("RANK", rank)), world)
    device = torch.device("cuda", rank)
    torch.cuda.set_device(device)
    torch.set_num_threads(4)

    for path in (args.environment_file, args.config_file):
        for k, v in json.load(open(path)).items():
            setattr(args, k, v)
    torch.manual_seed(42)                               # set_determinism(42), train_diffusion.py:66

    tcfg = args.diffusion_train
    if args.synthetic and rank == 0:
        write_synthetic_pairs(args.npz_dir, args.synthetic, tcfg["patch_size"], seed=int(getattr(args, "seed", 0)))
    if ddp:
        torch.distributed.barrier()
    train_loader, val_loader = prepare_dataloader(args, tcfg["batch_size"], tcfg["patch_size"], randcrop=False, rank=rank,
                                                  world_size=world, scale_on_host=not args.gpu_transforms)
    if args.gpu_transforms:                                # raw crops from the loader; the scaling runs on the device, batch by batch
        from ldm3d.data import gpu_scale_batch

        class _OnDevice:
            def __init__(self, loader):
                self.loader, self.sampler = loader, getattr(loader, "sampler", None)

            def __iter__(self):
                return (gpu_scale_batch(b, device) for b in self.loader)

            def __len__(self):
                return len(self.loader)
        train_loader, val_loader = _OnDevice(train_loader), _OnDevice(val_loader)
    log = None
    if rank == 0:
        tb = os.path.join(getattr(args, "tfevent_path", os.path.join(args.model_dir, "tfevent")), "diffusion")
        Path(tb).mkdir(parents=True, exist_ok=True)
        Path(args.model_dir).mkdir(parents=True, exist_ok=True)
        log = open(os.path.join(tb, "scalars.jsonl"), "a")

    def scalar(tag, value, step):
        if log:
            log.write(json.dumps({"tag": tag, "value": float(value), "step": int(step), "time": time.time()}) + "\n")
            log.flush()

    autoencoder = define_instance(args, "autoencoder_def")
    if not args.random_init:
        autoencoder.load_state_dict(torch.load(os.path.join(args.model_dir, "autoencoder.pt"), map_location="cpu", weights_only=True))
    else:
        with torch.no_grad():
            for p in autoencoder.parameters():
                if p.dim() > 1 and float(p.abs().max()) == 0.0:
                    p.normal_(0.0, 0.02)
    autoencoder = autoencoder.to(device).eval()

    first = next(iter(train_loader))
    scale_factor = compute_scale_factor(autoencoder, first["label"].to(device).float(), GradSync())
    print(f"Rank {rank}: scale_factor -> {float(scale_factor):.6f}")
    if rank == 0:                                   # kept next to the checkpoints: inference.py reads it back (the reference
        # hard-codes 1.0 there, inference.py:85)
        os.makedirs(args.model_dir, exist_ok=True)
        with open(os.path.join(args.model_dir, "scale_factor.json"), "w") as fh:
            json.dump({"scale_factor": float(scale_factor)}, fh)

    unet = define_instance(args, "diffusion_def")
    best_path = os.path.join(args.model_dir, "diffusion_unet.pt")
    last_path = os.path.join(args.model_dir, "diffusion_unet_last.pt")
    if getattr(args, "resume_ckpt", False) and os.path.exists(best_path):
        unet.load_state_dict(torch.load(best_path, map_location="cpu", weights_only=True))
        print(f"Rank {rank}: loaded {best_path}")
    elif args.random_init:
        with torch.no_grad():
            for p in unet.parameters():
                if p.dim() > 1 and float(p.abs().max()) == 0.0:
                    p.normal_(0.0, 0.02)
    unet = unet.to(device)
    ns = args.NoiseScheduler
    scheduler = DDPMScheduler(num_train_timesteps=ns["num_train_timesteps"], schedule="scaled_linear_beta",
                              beta_start=ns["beta_start"], beta_end=ns["beta_end"])
    inferer = LatentDiffusionInferer(scheduler, scale_factor=float(scale_factor))
    trainer = DiffusionTrainer(unet, autoencoder, inferer, lr=tcfg["lr"], reference_rng_order=args.reference_rng_order,
                               grad_dtype=torch.bfloat16 if args.grad_allreduce_dtype == "bf16" else torch.float32)

    total_step, best_val, done = 0, float("inf"), False
    for epoch in range(tcfg["max_epochs"]):
        if ddp:
            train_loader.sampler.set_epoch(epoch)
            val_loader.sampler.set_epoch(epoch)
        t0, n_steps = time.perf_counter(), 0
        for step, batch in enumerate(train_loader):
            loss, skipped = trainer.train_step(batch["image"].to(device), batch["label"].to(device))
            if skipped:
                print(f"NaN loss detected at epoch {epoch}, step {step}: skipped on every rank")
                continue
            total_step += 1
            n_steps += 1
            scalar("train_diffusion_loss_iter", loss, total_step)
            if args.max_steps and total_step >= args.max_steps:
                done = True
                break
        trainer.end_epoch()
        torch.cuda.synchronize()
        if rank == 0:
            print(f"Epoch {epoch}: {n_steps} steps in {time.perf_counter() - t0:.2f} s, lr {trainer.optimizer.param_groups[0]['lr']:.3g}")
        if epoch % tcfg["val_interval"] == 0 or done:
            val = trainer.validate(val_loader, device)
            if rank == 0:
                scalar("val_diffusion_loss", val, epoch)
                print(f"Epoch {epoch} val_diffusion_loss: {val}")
                torch.save(unet.state_dict(), last_path)
                if val < best_val:
                    best_val = val
                    torch.save(unet.state_dict(), best_path)
                    print("Got best val noise pred loss. Saved", best_path)
                # "Test denoising capability" (3d_ldm/train_diffusion.py:306-359): every 2 * val_interval epochs rank 0 samples one
                # high-count volume conditioned on the low-count latents of the last validation batch (mode="concat") and logs the
                # centre slices of input / ground truth / sample along the three axes
                if epoch % (2 * tcfg["val_interval"]) == 0:
                    sample_validation_volume(trainer, val_loader, device, os.path.join(tb, "samples"), epoch, args.sample_steps, scalar,
                                             dict(num_train_timesteps=ns["num_train_timesteps"], schedule="scaled_linear_beta",
                                                  beta_start=ns["beta_start"], beta_end=ns["beta_end"]))
        if done:
            break
    if log:
        log.close()
    if ddp:
        parallel.cleanup_ddp()


if __name__ == "__main__":
    main()
