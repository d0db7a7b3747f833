#!/usr/bin/env python3
"""train_autoencoder.py - command line, JSON schema and checkpoint names of the reference's 3d_ldm/train_autoencoder.py
(:64-84 flags, :123-129 config merge, :133-145 loader with random crops, :149 define_instance, :226-279 losses / lr scaling
/ AdamW, :330-451 generator step, :565-647 validation + rank-0 checkpoints) on the MI355X-native path.

    python train_autoencoder.py -e config/environment.json -c config/config_train_16g.json -g 1

The PatchDiscriminator / LSGAN adversarial phase after 5 warm-up epochs (:150-158,407-424,454-494) runs on the same HIP kernels
(ldm3d/discriminator.py); checkpoints discriminator.pt / discriminator_last.pt as the reference (:183-186).  NOT reproduced: the
perceptual loss as MONAI builds it (it downloads a pretrained LPIPS / SqueezeNet): `--perceptual-weights file.pt` supplies those weights
(ldm3d/perceptual.py); without the flag a non-zero `autoencoder_train.perceptual_weight` (every shipped config has one) is reported once,
recorded in the scalars log and the checkpoint directory (perceptual_term.json), and the term is left out.
--profile (:81,312-329) traces the launch plans of a few steps with the reference's schedule (wait 1, warm-up 1, active 3, repeat 2)
into ./profiler_logs (ldm3d/profiling.py on ldm_set_plan_trace).  --amp / --compile / --no-images are accepted and ignored (compute
is bf16 on fp32 master weights, or fp32 with --precision fp32; there is no tracing compiler on this path).
Opt-in extras: --random-init, --synthetic N, --max-steps K (as train_diffusion.py)."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    parser = argparse.ArgumentParser(description="VAE (stage 1) training, MI355X-native")
    parser.add_argument("-e", "--environment-file", default="./config/environment.json")
    parser.add_argument("-c", "--config-file", default="./config/config_train_32g.json")
    parser.add_argument("-g", "--gpus", default=1, type=int)
    parser.add_argument("--amp", action="store_true")
    parser.add_argument("--compile", action="store_true")
    parser.add_argument("--profile", action="store_true")
    parser.add_argument("--no-images", action="store_true")
    parser.add_argument("--local_rank", type=int, default=0)
    parser.add_argument("--random-init", action="store_true")
    parser.add_argument("--synthetic", type=int, default=0)
    parser.add_argument("--precision", default=None, choices=["bf16", "fp32"],
                        help="arithmetic of the networks: bf16 (default, the fast path) or fp32 (the reference's own arithmetic, 1e-5 from its CPU path; also LDM_PRECISION)")
    parser.add_argument("--max-steps", type=int, default=0)
    parser.add_argument("--perceptual-weights", default=None,
                        help="state_dict of lpips.LPIPS(net='squeeze') (torch.save): enables the perceptual term of :236,406; "
                             "without it the term is dropped and recorded (no download is possible offline)")
    args = parser.parse_args()
    if args.precision:
        os.environ["LDM_PRECISION"] = args.precision     # read by every network at construction (networks.py)

    import torch
    from ldm3d import parallel
    from ldm3d.config import define_instance
    from ldm3d.data import prepare_dataloader, write_synthetic_pairs
    from ldm3d.trainer import AutoencoderTrainer

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
    torch.manual_seed(42)
    tcfg = args.autoencoder_train
    if args.synthetic and rank == 0:
        write_synthetic_pairs(args.npz_dir, args.synthetic, [int(1.25 * p) for p in tcfg["patch_size"]], seed=int(getattr(args, "seed", 0)))
    if ddp:
        torch.distributed.barrier()
    train_loader, val_loader = prepare_dataloader(args, tcfg["batch_size"], tcfg["patch_size"], randcrop=True, rank=rank, world_size=world,
                                                  size_divisible=2 ** (len(args.autoencoder_def["channels"]) - 1))

    autoencoder = define_instance(args, "autoencoder_def")
    best_path = os.path.join(args.model_dir, "autoencoder.pt")
    last_path = os.path.join(args.model_dir, "autoencoder_last.pt")
    if getattr(args, "resume_ckpt", False) and os.path.exists(best_path):
        autoencoder.load_state_dict(torch.load(best_path, map_location="cpu", weights_only=True))
        print(f"Rank {rank}: loaded {best_path}")
    autoencoder = autoencoder.to(device)
    trainer = AutoencoderTrainer(autoencoder, lr=tcfg["lr"], kl_weight=tcfg["kl_weight"], recon_loss=tcfg.get("recon_loss", "l1"),
                                 perceptual_weight=tcfg.get("perceptual_weight", 0.0), perceptual_weights=args.perceptual_weights,
                                 warm_up_epochs=int(tcfg.get("warm_up_epochs", 5)))      # 5 in the reference (:304); the key is an extension
    d_best = os.path.join(args.model_dir, "discriminator.pt")
    if getattr(args, "resume_ckpt", False) and os.path.exists(d_best):
        trainer.discriminator.load_state_dict(torch.load(d_best, map_location=device, weights_only=True))
    log = None
    if rank == 0:
        tb = os.path.join(getattr(args, "tfevent_path", os.path.join(args.model_dir, "tfevent")), "autoencoder")
        Path(tb).mkdir(parents=True, exist_ok=True)
        Path(args.model_dir).mkdir(parents=True, exist_ok=True)
        log = open(os.path.join(tb, "scalars.jsonl"), "a")

    def scalar(tag, value, step):
        if log:
            log.write(json.dumps({"tag": tag, "value": float(value), "step": int(step), "time": time.time()}) + "\n")
            log.flush()

    profiler = None
    if args.profile and rank == 0:                          # 3d_ldm/train_autoencoder.py:312-329
        from ldm3d.profiling import PlanProfiler
        profiler = PlanProfiler("./profiler_logs")
        print("Profiler started")
    if rank == 0 and trainer.perceptual_dropped:
        scalar("perceptual_term_dropped", 1.0, 0)
        with open(os.path.join(args.model_dir, "perceptual_term.json"), "w") as fh:
            json.dump({"perceptual_term": "dropped", "perceptual_weight": tcfg.get("perceptual_weight", 0.0),
                       "reason": "pretrained SqueezeNet / LPIPS weights are not available offline"}, fh)
        print(f"perceptual_term: dropped (autoencoder_train.perceptual_weight = {tcfg.get('perceptual_weight', 0.0)}; no pretrained weights offline)")
    total_step, best_val, done = 0, 100.0, False
    for epoch in range(tcfg["max_epochs"]):
        if ddp:
            train_loader.sampler.set_epoch(epoch)
            val_loader.sampler.set_epoch(epoch)
        t0, sums, nb = time.perf_counter(), {}, 0
        for step, batch in enumerate(train_loader):
            losses, skipped = trainer.train_step(batch["image"].to(device), epoch)
            if profiler is not None:
                profiler.step()
            if skipped:
                print(f"Warning: non-finite input or loss at epoch {epoch}, step {step}: skipped on every rank")
                continue
            total_step += 1
            nb += 1
            for k, v in losses.items():
                sums[k] = sums.get(k, 0.0) + v
            scalar("train_recon_loss_iter", losses["recons"], total_step)
            if args.max_steps and total_step >= args.max_steps:
                done = True
                break
        torch.cuda.synchronize()
        if rank == 0 and nb:
            print(f"Epoch {epoch}: {nb} steps in {time.perf_counter() - t0:.2f} s; " +
                  ", ".join(f"{k} {float(v) / nb:.5f}" for k, v in sums.items()))
            for k, v in sums.items():
                scalar(f"train_{k}_epoch", float(v) / nb, epoch)
        if epoch % tcfg["val_interval"] == 0 or done:
            val = trainer.validate(val_loader, device)
            if rank == 0:
                scalar("val_recon_loss", val, epoch)
                print(f"Epoch {epoch} val_recon_loss: {val:.4f}")
                torch.save(autoencoder.state_dict(), last_path)
                torch.save(trainer.discriminator.state_dict(), os.path.join(args.model_dir, "discriminator_last.pt"))
                if val < best_val:                       # the reference saves "best" unconditionally: fixed
                    best_val = val
                    torch.save(autoencoder.state_dict(), best_path)
                    torch.save(trainer.discriminator.state_dict(), os.path.join(args.model_dir, "discriminator.pt"))
                    print("Got best val recon loss. Saved", best_path)
        if done:
            break
    if profiler is not None:
        profiler.stop()
    if log:
        log.close()
    if ddp:
        parallel.cleanup_ddp()


if __name__ == "__main__":
    main()
