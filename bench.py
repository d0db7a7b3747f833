#!/usr/bin/env python3
"""bench.py - UNet denoising steps/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one DiffusionModelUNet forward + one DDPM scheduler step (fresh N(0,1) noise drawn on the device) on a
synthetic 1x4x24x24x24 fp32 latent: BASELINE.json configs[2] (benchmark UNet = diffusion_def of
3d_ldm/config/config_train_16g.json:39-48 with 4 latent channels, DDPM schedule of :56-60), random-init weights.
With N > 1 every rank drives its own independent reverse chain on its own GPU (the sampling path has no exchange
step: inference.py has no distributed code) -> weak scaling, no data-path collective; the only collectives are the
timing barrier and the MAX over ranks.

Also reported on the same JSON line:
  roofline     - the dominant kernel (conv3_halo_kernel: implicit-GEMM 3x3x3 conv, 126x128 tile, K 64 per step) timed live
                 with HIP events on the launch stream: algorithmic FLOPs / measured kernel time vs the dense bf16 MFMA peak.
  cpu_baseline - the CPU oracle (fp32 torch ops, all host cores) timed on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0         # /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
UNET_STEP_GFLOP = 889.1                      # SURVEY.md section 8d / BASELINE.md section 3 (2*MAC, real channel counts)
# what the launch plan executes: the two Upsample layers (nearest x2 + 3^3 conv: 73.4 GFLOP as the reference computes them) run as
# eight 2^3 phase convolutions on the source grid with pre-summed taps (21.7 GFLOP): DESIGN.md section 3.1c
UNET_STEP_EXECUTED_GFLOP = 889.1 - 73.4 + 21.7
DOMINANT_TILE = (2, 2, 64 | 256)             # conv3_halo_kernel (bk | 256 selects it): 126 voxels x 128 couts x K 64 per step


def make_unet(dev, seed=0, in_channels=None):
    import torch
    import cfgs
    from ldm3d.networks import DiffusionModelUNet
    cfg = dict(cfgs.UNET_FULL)
    if in_channels is not None:               # concat-conditioned variant (tools/bench_configs.py)
        cfg["in_channels"] = in_channels
    m = DiffusionModelUNet(**cfg)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():                     # random init everywhere (MONAI zero-inits conv2/out: zeros would flatter DVFS)
        for name, p in m.named_parameters():
            if p.dim() > 1:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) / fan_in ** 0.5)
            elif name.endswith(".weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return m.to(dev).eval()


CPU_BASELINE_THREAD_CAP = 16                 # threads the CPU baseline uses at most (a 1-GPU box's CPU share on the pool this was tuned on)


def host_cores():
    """(threads the CPU baseline will use, what was discovered): the affinity mask and the cgroup CPU quota of THIS process are read,
    not assumed; the cap is reported separately in the bench line."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    found = {"os_cpu_count": os.cpu_count(), "sched_affinity": aff, "cgroup_quota_cores": None}
    n = aff
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            found["cgroup_quota_cores"] = int(quota) / int(period)
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                found["cgroup_quota_cores"] = q / per
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    found["available"] = max(1, n)
    return max(1, min(n, CPU_BASELINE_THREAD_CAP)), found


def pmc_traffic():
    """(HBM bytes per launch of the dominant kernel, where that number comes from): PMC counters need rocprofv3 around the process,
    so the bench line carries the figure of the newest committed PMC summary under profiles/ and NAMES it (file, the commit the
    profiled library was built from, the kernel it was taken on) -- a stale figure is then visible as such."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
            t = d.get("conv_hbm_traffic_bytes_per_launch")
            if t:
                return t["total"], {"file": os.path.relpath(f, ROOT), "commit": d.get("commit"), "kernel": t.get("kernel"),
                                    "launches_profiled": t.get("launches"), "note": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) "
                                    "+ WRITE_SIZE passes of `bench.py --no-cpu-baseline`, NOT a measurement of this run"}
        except (OSError, ValueError):
            pass
    return None, None


def cpu_baseline(steps=3):
    """Oracle (test infrastructure) as the CPU baseline: fp32 torch ops, all host cores, same UNet, same latent."""
    import torch
    import cfgs
    from oracle import unet as ou
    from oracle.schedulers import OracleDDPM
    cores, found = host_cores()
    torch.set_num_threads(cores)
    sd = ou.init_state_dict(ou.unet_param_shapes(cfgs.UNET_FULL), 0)
    sch = OracleDDPM(**cfgs.SCHED)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, 4, 24, 24, 24), generator=g)

    def one(t, x):
        eps = ou.unet_forward(sd, cfgs.UNET_FULL, x, torch.tensor([float(t)]))
        return sch.step(eps, t, x, torch.randn(x.shape, generator=g))[0]
    with torch.no_grad():
        tw = time.perf_counter()
        x = one(999, x)                       # warm-up (oneDNN primitive creation)
        if time.perf_counter() - tw > 15.0:   # keep the default run within minutes on a slow host
            steps = 1
        t0 = time.perf_counter()
        for i in range(steps):
            x = one(998 - i, x)
        dt = time.perf_counter() - t0
        # the reference pins torch to 4 threads (3d_ldm/train_diffusion.py:56, inference.py:58): the same step at that setting
        four = None
        if cores > 4 and dt / steps < 5.0:
            torch.set_num_threads(4)
            x = one(994, x)
            t4 = time.perf_counter()
            x = one(993, x)
            four = 1.0 / (time.perf_counter() - t4)
            torch.set_num_threads(cores)
    return {"value": steps / dt, "unit": "steps/s", "cores": cores, "kind": "port",
            "cores_discovered": found, "cores_cap": CPU_BASELINE_THREAD_CAP,
            "sample": f"{steps} DDPM steps (UNet fwd + scheduler step) on 1x4x24^3 after 1 warm-up step, fp32 torch-CPU oracle",
            "value_at_the_reference_4_threads": four}


def other_paths_leg(dev):
    """AutoencoderKL encode / decode of one 1x1x96^3 volume (BASELINE configs[1]; 3d_ldm/train_diffusion.py:104, inference.py:94-99) and the
    UNet training step at 1x4x24^3 (forward + MSE + backward + clip + Adam + bf16 re-pack: train_diffusion.py:207-223), bf16, wall clock
    around back-to-back calls with a device synchronisation on both sides."""
    import torch
    from ldm3d.networks import AutoencoderKL
    from ldm3d.optim import FlatAdam, mse_loss
    rec = {}
    vae = AutoencoderKL(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=4, channels=[64, 128, 256], num_res_blocks=2,
                        norm_num_groups=32, norm_eps=1e-6, attention_levels=[False, False, False],
                        with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for prm in vae.parameters():
            if prm.dim() > 1:
                prm.copy_(torch.randn(prm.shape, generator=g) / prm[0].numel() ** 0.5)
    vae = vae.to(dev).eval()
    img = torch.rand((1, 1, 96, 96, 96), generator=g).to(dev)
    lat = torch.randn((1, 4, 24, 24, 24), generator=g).to(dev)

    def clock(fn, n=10, warm=2):
        with torch.no_grad():
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    rec["vae_96cube"] = {"encode_ms": clock(lambda: vae.encode(img)), "decode_ms": clock(lambda: vae.decode(lat)),
                         "what": "AutoencoderKL (64/128/256, 2 res blocks per level) on 1x1x96^3, bf16, random init"}
    del vae
    torch.cuda.empty_cache()
    unet = make_unet(dev, seed=0).train()
    opt = FlatAdam(unet, lr=5e-6, max_grad_norm=1.0)
    x, noise = torch.randn((1, 4, 24, 24, 24), device=dev), torch.randn((1, 4, 24, 24, 24), device=dev)
    t = torch.tensor([500.0], device=dev)

    def step():
        loss = mse_loss(unet(x=x, timesteps=t), noise)
        loss.backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    rec["train_step_24cube"] = {"ms_per_step": (time.perf_counter() - t0) / n * 1e3, "skipped_steps": int(opt.skipped_steps()),
                                "what": "benchmark UNet, batch 1, 1x4x24^3: training-plan forward + fused MSE + backward plan + gradient-norm "
                                        "clip + fused Adam + bf16 re-pack, no host read inside the step"}
    del unet, opt
    torch.cuda.empty_cache()
    return rec


def ddp_train_leg(dev, rank, world, dist, rehearsal, steps=8, warmup=3, latent=(36, 44, 28)):
    """BASELINE configs[3] beside the headline: the data-parallel training step of 3d_ldm/train_diffusion.py:172-223 (two no-grad VAE
    encodes of the 144x176x112 patch pair, concat-conditioned benchmark UNet forward at the 36x44x28 latent, MSE, backward, clip,
    Adam), batch 1 per GPU, through DiffusionTrainer -> GradSync.attach -> ldm_model_set_grad_sync: the gradients are averaged over the
    ranks by the library's bucketed RCCL all-reduce while backward is still running (DDP's bucket hooks, :147-149,214).  World 1 runs
    the same path on a world-size-1 RCCL communicator (the N = 1 anchor).  Every rank trains on its OWN synthetic batch, so equal
    parameter checksums on all ranks after the steps are evidence that every rank applied the same (mean) gradient."""
    import ctypes as C
    import torch
    import cfgs
    from ldm3d import _lib
    from ldm3d.inferer import LatentDiffusionInferer
    from ldm3d.networks import AutoencoderKL, DiffusionModelUNet
    from ldm3d.schedulers import DDPMScheduler
    from ldm3d.trainer import DiffusionTrainer
    L = _lib.lib()
    torch.manual_seed(100)                                  # identical initial weights on every rank (and a broadcast at wrap time)
    vae = AutoencoderKL(**cfgs.VAE_FULL)
    unet = DiffusionModelUNet(**dict(cfgs.UNET_FULL, in_channels=8))
    with torch.no_grad():
        for mod, gain in ((vae, 1.0), (unet, 0.5)):
            for name, p in mod.named_parameters():
                if p.dim() > 1:
                    p.copy_(torch.randn(p.shape) * (gain / p[0].numel() ** 0.5))
    vae, unet = vae.to(dev).eval(), unet.to(dev)
    tr = DiffusionTrainer(unet, vae, LatentDiffusionInferer(DDPMScheduler(**cfgs.SCHED), scale_factor=1.0), lr=1e-5)
    if world == 1 and not tr.overlap:                       # N = 1 anchor: the same in-library path on a world-size-1 communicator
        tr.overlap = tr.sync.attach(unet, force_single=True)
    patch = tuple(4 * v for v in latent)
    gen = torch.Generator(device=dev).manual_seed(7 + rank)  # a different batch on every rank
    images = torch.rand((1, 1, *patch), device=dev, generator=gen)
    labels = torch.rand((1, 1, *patch), device=dev, generator=gen)

    def fence():                                            # through GradSync: its quiescence assertion covers this leg's collectives too
        tr.sync.barrier()
        torch.cuda.synchronize()

    def checksum():
        f = unet.flat_params.double()
        return [float(f.sum()), float(f.abs().sum()), float((f * torch.arange(f.numel(), device=dev, dtype=torch.float64) % 8191).sum())]
    for _ in range(warmup):
        loss, skipped = tr.train_step(images, labels)
    sums = checksum()                                       # after `warmup` (3) optimizer steps
    equal = True
    if dist is not None:
        allt = tr.sync.gather_objects(sums)
        equal = all(a == allt[0] for a in allt)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, skipped = tr.train_step(images, labels)
    fence()
    dt_local = dt = time.perf_counter() - t0
    if dist is not None:
        dt = float(tr.sync.max_scalar(torch.tensor([dt], dtype=torch.float64, device=dev)).item())
    rec = {"what": "BASELINE configs[3]: train_diffusion step (2 x VAE encode of a 144x176x112 patch, UNet in 8 / out 4 fwd + bwd at "
                   "the 36x44x28 latent, MSE, clip 1.0, Adam), bf16 compute / fp32 master weights and gradients, batch 1 per GPU, "
                   "gradients averaged over ranks inside backward",
           "world": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
           "train_steps_per_s_per_gpu": steps / dt, "train_steps_per_s": world * steps / dt, "global_batch": world,
           "loss": float(loss), "skipped": bool(skipped),
           "exchange": ("in-library bucketed RCCL all-reduce overlapped with backward (ldm_model_set_grad_sync)" if tr.overlap else
                        "torch.distributed all-reduce after backward (GradSync.mean_)" + (" [rehearsal: gloo]" if rehearsal else "")),
           "param_checksum_after_3_steps": sums, "param_checksums_equal_on_all_ranks": equal}
    rec["exchange_path"] = "in-library" if tr.overlap else "fallback"          # machine-readable form of "exchange"
    rec["rccl_version"] = L.ldm_comm_rccl_version()                              # ncclGetVersion() of the librccl this process loaded
    # what THIS rank saw: phases, its own wall clock, the bucket timeline of its last step.  Rank 0's copy also sits at the top level
    # (the r03 / r04 field names); "per_rank" carries every rank's, so the first N > 1 run says by itself WHICH rank waited where
    mine = {"rank": rank, "ms_per_step_local": dt_local / steps * 1e3, "phases_ms": ddp_phase_trace(tr, images, labels, dev)}
    if tr.overlap:
        n_max = 64
        issue, done, elems = (C.c_double * n_max)(), (C.c_double * n_max)(), (C.c_int64 * n_max)()
        n = L.ldm_model_grad_sync_trace(unet._h, issue, done, elems, n_max)
        comm = getattr(unet, "_grad_comm", None)
        mine["ldm_comm_world"] = L.ldm_comm_world(comm) if comm else None
        if comm:
            st = (C.c_int64 * 4)()
            L.ldm_comm_stats(comm, st)
            counted = steps + warmup + 3                 # warm-up + timed + the three steps of the phase trace
            mine["ldm_comm"] = {"transport_is_rccl": bool(L.ldm_comm_is_rccl(comm) == 1), "allreduce_calls": int(st[0]),
                                "allreduce_bytes": int(st[1]), "optimizer_steps_counted": counted,
                                "bytes_per_step": int(st[1]) // counted,
                                "note": "counted inside the library where each bucket is handed to ncclAllReduce"}
        if 0 < n < n_max:
            end = issue[n]
            mine.update({"backward_ms": end, "buckets": n, "bucket_mb": [round(elems[k] * 4 / 2 ** 20, 1) for k in range(n)],
                         "bucket_issue_ms": [round(issue[k], 3) for k in range(n)], "bucket_done_ms": [round(done[k], 3) for k in range(n)],
                         # what the launch stream has to wait for after its own last backward kernel (the join): 0 = fully hidden
                         "allreduce_exposed_ms": max(0.0, done[n - 1] - issue[n - 1] if n else 0.0),
                         "allreduce_busy_ms": sum(done[k] - max(issue[k], done[k - 1] if k else 0.0) for k in range(n)),
                         "trace_note": "times of the LAST step of this rank, ms since its backward began; backward_ms includes the join"})
    mine["rccl_debug"] = rccl_debug_digest()
    rec.update({k: v for k, v in mine.items() if k not in ("rank", "ms_per_step_local")})
    everyone = tr.sync.gather_objects(mine) if dist is not None else [mine]
    rec["per_rank"] = everyone
    exposed = [r.get("allreduce_exposed_ms") for r in everyone if r.get("allreduce_exposed_ms") is not None]
    if exposed:
        rec["allreduce_exposed_ms_over_ranks"] = {"min": min(exposed), "max": max(exposed)}
    rec["ms_per_step_over_ranks"] = {"min": min(r["ms_per_step_local"] for r in everyone), "max": max(r["ms_per_step_local"] for r in everyone)}
    return rec


RCCL_DEBUG_LOG = None                        # this process's RCCL debug file (set_rccl_debug)
RCCL_DEBUG_PREV = None


def set_rccl_debug(rank):
    """First-contact instrumentation of the N > 1 runs (nobody has seen this code's collectives on real xGMI): RCCL's own INFO log
    of the INIT (+ TUNING on rank 0: algorithm / protocol / channels per collective size) subsystems goes to a per-process file that
    ddp_train digests into the JSON line and echoes to stderr.  Must run before the first communicator of the process exists;
    whatever NCCL_DEBUG* the environment had is reported in the digest.  LDM_BENCH_RCCL_DEBUG=0 switches it off."""
    global RCCL_DEBUG_LOG, RCCL_DEBUG_PREV
    if os.environ.get("LDM_BENCH_RCCL_DEBUG", "1") == "0":
        return
    RCCL_DEBUG_PREV = {k: os.environ.get(k) for k in ("NCCL_DEBUG", "NCCL_DEBUG_SUBSYS", "NCCL_DEBUG_FILE")}   # what the environment had (reported)
    import tempfile
    RCCL_DEBUG_LOG = os.path.join(tempfile.gettempdir(), f"ldm_bench_rccl_{os.getpid()}.log")
    os.environ["NCCL_DEBUG"] = "INFO"
    os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,TUNING" if rank == 0 else "INIT"
    os.environ["NCCL_DEBUG_FILE"] = RCCL_DEBUG_LOG


def rccl_debug_digest(max_lines=40):
    """The lines of this process's RCCL debug log that say what the library chose: per communicator its size / channel counts, and
    (rank 0) the DISTINCT (bytes -> algorithm, protocol, channels) decisions of the collectives it ran.  The whole log goes to stderr."""
    if not RCCL_DEBUG_LOG:
        return None
    if not os.path.exists(RCCL_DEBUG_LOG):
        return {"log_lines": 0, "digest": [], "note": "RCCL wrote no debug file (debug state initialised before bench.py set NCCL_DEBUG?)",
                "env_before": RCCL_DEBUG_PREV}
    import re
    keep, seen = [], set()
    try:
        lines = open(RCCL_DEBUG_LOG, errors="replace").read().splitlines()
    except OSError:
        return None
    for ln in lines:
        body = ln.split("NCCL INFO", 1)[-1].strip()
        if re.search(r"Init COMPLETE|nranks|coll channels|Channel \d+/\d+ *:|Trees|Connected all (rings|trees)|threadThresholds|NET/|P2P|xGMI|XGMI", body) \
                or re.search(r"Bytes -> Algo|-> algorithm", body, re.I):
            key = re.sub(r"0x[0-9a-f]+|\b\d{5,}\b", "#", body)      # pointers / opCounts / big byte counts do not make a line new
            if "Bytes ->" in body:
                key = body.split("Bytes ->", 1)[0].split()[-1] + body.split("Bytes ->", 1)[1]
            elif re.match(r"Channel \d+/\d+", body):                 # one line per channel (128 of them at world 1): keep the first, count the rest
                key = "Channel " + re.sub(r"^Channel \d+", "", body).split(":")[0]
            if key not in seen:
                seen.add(key)
                keep.append(body[:240])
    sys.stderr.write("".join(f"[rccl] {ln}\n" for ln in lines[-400:]))
    return {"log_lines": len(lines), "digest": keep[:max_lines], "truncated": len(keep) > max_lines, "env_before": RCCL_DEBUG_PREV}


def ddp_phase_trace(tr, images, labels, dev, n=3):
    """Where a configs[3] train step goes (outside every timed region): the calls of DiffusionTrainer.train_step in the same order
    (3d_ldm/train_diffusion.py:194-219) with a HIP event between the phases, mean over n steps.  Steps through the real optimizer."""
    import torch
    from ldm3d.optim import mse_loss
    names = ["vae_encode_condition", "vae_encode_labels", "add_noise_unet_forward", "mse_backward_incl_exchange", "clip_adam_repack"]
    acc = [0.0] * len(names)
    vae, unet, inf = tr.autoencoder, tr.unet, tr.inferer
    for _ in range(n):
        noise, timesteps = tr._draw(labels)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        unet.train()
        e[0].record()
        with torch.no_grad():
            z_img = vae.encode_stage_2_inputs(images)
        e[1].record()
        with torch.no_grad():
            z = vae.encode_stage_2_inputs(labels)
            if inf.scale_factor != 1.0:
                z = z * inf.scale_factor
        e[2].record()
        noisy = inf.scheduler.add_noise(original_samples=z, noise=noise, timesteps=timesteps)
        pred = unet(x=noisy, timesteps=timesteps, context=None, cond=z_img)
        e[3].record()
        mse_loss(pred, noise).backward()
        if not tr.overlap:
            tr.sync.mean_(unet.flat_grads)
        e[4].record()
        tr.optimizer.step()
        e[5].record()
        torch.cuda.synchronize()
        for k in range(len(names)):
            acc[k] += e[k].elapsed_time(e[k + 1]) / n
    out = {k: round(v, 3) for k, v in zip(names, acc)}
    out["sum"] = round(sum(acc), 3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host instead of replaying the HIP graph")
    ap.add_argument("--host-step", action="store_true", help="scheduler step driven from the host (torch.randn + coefficient lookup) instead of the fused device sampler")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the fp32 precision mode figure reported beside the bf16 one")
    ap.add_argument("--no-other-paths", action="store_true", help="skip the AutoencoderKL 96^3 and 24^3 training-step figures (BASELINE configs[1], a6) reported beside the headline")
    ap.add_argument("--no-ddp-train", action="store_true", help="skip the data-parallel training leg (configs[3]) reported beside the headline")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  One fresh child process per GPU through torch.distributed.run,
        # started BEFORE this process makes any GPU call (device_count() does not initialise the runtime); the children print the
        # JSON line, this process only forwards their exit code.  (The reference's launchers do the same with torchrun:
        # 3d_ldm/train_LDM.sh:71-76.)
        import socket
        import subprocess
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus and os.environ.get("LDM_BENCH_REHEARSAL", "0") != "1":
            print(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible", file=sys.stderr)
            sys.exit(2)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    # RCCL prints a version banner on stdout whenever a communicator is created (torch's process group, the library's own): stdout is
    # kept for the ONE JSON line, everything else of this process goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE {world} (launch with --nproc-per-node {args.gpus})", file=sys.stderr)
        sys.exit(2)
    if not args.no_ddp_train:
        set_rccl_debug(rank)                                # before torch's process group / the library's communicator exist
    dist = None
    # rehearsal of the multi-rank control flow on a one-GPU box: LDM_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo
    # (NCCL/RCCL refuses two ranks on one device); the numbers of such a run mean nothing
    rehearsal = os.environ.get("LDM_BENCH_REHEARSAL", "0") == "1"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            local_rank = 0
            dist.init_process_group(backend="gloo", init_method="env://")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", init_method="env://", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    import cfgs
    from ldm3d import _lib
    from ldm3d.schedulers import DDPMScheduler
    L = _lib.lib()
    unet = make_unet(dev, seed=rank)
    if not args.eager:
        unet.enable_graph_replay(True)       # same kernels; one hipGraphLaunch per forward keeps the host far off the critical path
    sch = DDPMScheduler(**cfgs.SCHED)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn((1, 4, 24, 24, 24), device=dev, generator=gen)
    tbuf = torch.empty((1,), dtype=torch.float32, device=dev)
    T = sch.num_train_timesteps

    # the loop body of 3d_ldm/inference.py:94-99 on the device-resident sampler: UNet forward + DDPM step with the noise drawn inside
    # the step kernel (Philox keyed by a seed) and the timestep advanced on the device: one HIP graph launch per step, no torch
    # kernels (fill_ / normal_) and no host-side coefficient lookups in the timed region.  --host-step: DDPMScheduler.step from the host.
    sampler = sch.device_sampler(seed=1234 + rank)

    def step(i, x):
        if args.host_step:
            t = (T - 1 - i) % T
            tbuf.fill_(float(t))
            eps = unet(x=x, timesteps=tbuf)
            return sch.step(eps, t, x, generator=gen)[0]
        if i % T == 0:
            sampler.reset(tbuf)                  # a new 1000-step chain starts (never inside the default timed region)
        return unet.denoise_step(x, tbuf, sampler)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Host hygiene for a short timed region (the driver runs 20 steps = 42 ms): a step is ONE graph launch that costs the host
    # ~0.06 ms against ~2.1 ms on the GPU, so the host runs ahead of the device -- except at the first steps after the fence, where a
    # host stall (a garbage-collector pass over torch's heap, a page fault on a cold code path) is paid in full by the wall clock.
    # The collector is therefore parked for the region and the stream has one event per step, which costs nothing measurable and
    # shows in the JSON line whether a low figure was one stalled step, a ramp, or every step (ms_per_step_median / min / max).
    import gc
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host_at = [0.0] * args.steps
    with torch.no_grad():
        for i in range(args.warmup):
            x = step(i, x)
        gc.collect()
        gc.disable()
        try:
            fence()
            t0 = time.perf_counter()
            ev[0].record()
            for i in range(args.steps):
                host_at[i] = time.perf_counter()
                x = step(args.warmup + i, x)
                ev[i + 1].record()
            fence()
            dt = time.perf_counter() - t0
        finally:
            gc.enable()
        dt_local = dt
        per_step = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]           # ms, on the launch stream
        host_gap = [(host_at[i + 1] - host_at[i]) * 1e3 for i in range(args.steps - 1)]  # ms between two enqueues on the host
        # the headline is final HERE (MAX over ranks), before any of the legs reported beside it runs: nothing below can change or lose it
        per_rank_sps = [args.steps / dt_local]
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            allt = [torch.empty_like(tt) for _ in range(world)]
            dist.all_gather(allt, tt)                    # every replica's own wall clock: a slow rank is then visible by number
            per_rank_sps = [args.steps / float(a.item()) for a in allt]
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        # Roofline leg: the SAME K steps once more with HIP events recorded on the launch stream around every launch
        # of the dominant kernel (kept out of the timed region above: 2 events x ~60 launches per step cost ~25 %).
        profile = (not args.no_roofline) and rank == 0
        prof = (C.c_double * 5)()
        if profile:
            unet.enable_graph_replay(False)     # HIP events around individual launches need eager launches
            _lib.check(L.ldm_profile_start(*DOMINANT_TILE, 64 * args.steps + 64))
            for i in range(args.steps):
                x = step(args.warmup + args.steps + i, x)
            torch.cuda.synchronize()
            nmax = 64 * args.steps + 64
            dfl, dms = (C.c_double * nmax)(), (C.c_double * nmax)()
            nl = L.ldm_profile_detail(dfl, dms, nmax)
            by_shape = {}
            for k in range(max(0, min(nl, nmax))):                # launches of one shape have the same algorithmic FLOP count
                e = by_shape.setdefault(round(dfl[k]), [0, 0.0])
                e[0] += 1
                e[1] += dms[k]
            _lib.check(L.ldm_profile_stop(prof))
        # fp32 precision mode (the reference's own arithmetic; 1e-3 parity bar met at ~1e-5, tests/test_gpu_fp32.py): the same
        # step, reported BESIDE the headline bf16 figure, never instead of it
        fp32_leg = None
        if rank == 0 and world == 1 and not args.no_fp32_leg:
            unet.set_precision("fp32")
            if not args.eager:
                unet.enable_graph_replay(True)
            n32 = max(10, min(50, args.steps // 4))
            for i in range(5):
                x = step(i, x)
            torch.cuda.synchronize()
            t32 = time.perf_counter()
            for i in range(n32):
                x = step(5 + i, x)
            torch.cuda.synchronize()
            d32 = time.perf_counter() - t32
            fp32_leg = {"steps_per_s": n32 / d32, "ms_per_step": d32 / n32 * 1e3, "steps": n32,
                        "unet_step_tflops": UNET_STEP_GFLOP / (d32 / n32 * 1e3),
                        "what": "set_precision('fp32'): fp32 activations / weights; convolutions as 3 x bf16 MFMA (fp32 operands split into "
                                "hi + lo bf16, hi*hi + hi*lo + lo*hi accumulated in fp32: the 3^3 convs of the ResBlocks on conv3_halo_kernel over "
                                "the split the GroupNorm in front wrote, the rest on csrc/f32_path.h conv_x3_kernel), "
                                "attention / GroupNorm / linears in fp32; same step, same graph replay; rel-L2 vs the fp32 CPU oracle ~5e-5 "
                                "(bf16 path: ~3e-2; LDM_F32_X3=0 = exact fp32 MFMA everywhere: ~1e-5 at 84 steps/s)"}
            unet.set_precision("bf16")
    assert torch.isfinite(x).all()
    # the other section-8 paths on rank 0 at N = 1, after the headline's timed region: reported beside the metric, never instead of it
    other = None
    if not args.no_other_paths and world == 1:
        other = other_paths_leg(dev)
    # BASELINE configs[3] (DDP training over RCCL) on every rank, after the headline's timed region: its own record in the same line.
    # It is the one leg that exchanges data between ranks, on a fabric this code has never run on with N > 1: a watchdog on every rank
    # bounds it, and if it expires rank 0 still prints the line (headline and the other legs, ddp_train = {"error": ...}) before
    # the ranks exit -- a stuck collective must not cost the run its headline.
    import threading
    emitted = threading.Lock()
    state = {"out": None, "done": False}

    def emit(extra):
        with emitted:
            if state["done"] or state["out"] is None:
                return
            state["done"] = True
            state["out"].update(extra)
            print(json.dumps(state["out"]), file=json_out, flush=True)

    def run_ddp():
        budget = float(os.environ.get("LDM_BENCH_DDP_TIMEOUT_S", "300"))

        def give_up():
            if rank == 0:
                emit({"ddp_train": {"error": f"the data-parallel training leg did not finish within {budget:.0f} s (stuck collective?); "
                                             "the headline above was final before it started"}})
            os._exit(3)                                    # the line is out; the RUN failed and every launcher above must see that
        timer = threading.Timer(budget, give_up)
        timer.daemon = True
        timer.start()
        try:
            rec = ddp_train_leg(dev, rank, world, dist, rehearsal)
        except Exception as e:                               # a failing leg is reported, never fatal for the line
            rec = {"error": f"{type(e).__name__}: {e}"}
        timer.cancel()
        return rec

    if rank != 0:
        failed = False
        if not args.no_ddp_train:
            del unet
            torch.cuda.empty_cache()
            failed = "error" in run_ddp()
        if dist is not None:
            dist.destroy_process_group()
        if failed:
            sys.exit(3)
        return

    ms_per_step = dt / args.steps * 1e3
    out = {
        "metric": "UNet denoising steps/sec (bf16, 1x4x24x24x24 latent; whole job = sum over GPUs)",
        "value": world * args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        # the same region per step (rank 0): HIP events between the graph launches; value / ms_per_step above stay the wall-clock mean
        "ms_per_step_median": sorted(per_step)[len(per_step) // 2], "ms_per_step_min": min(per_step), "ms_per_step_max": max(per_step),
        "timed_region": {"sum_of_step_events_ms": sum(per_step), "wall_ms": dt * 1e3,
                         "slowest_step_index": per_step.index(max(per_step)),
                         "host_enqueue_gap_ms_max": max(host_gap) if host_gap else 0.0,
                         "host_hygiene": "python gc parked (collect + disable) for the region and one HIP event recorded per step inside it "
                                         "(rounds 4+; rounds 1-3 timed the same loop without either)",
                         "note": "wall_ms - sum_of_step_events_ms = time before the first / after the last step; a host stall shows as a "
                                 "large host_enqueue_gap with a long step behind it, a device-side one as a long step alone"},
        "config": {"workload": "DiffusionModelUNet (channels 256/256/512, attn at 12^3 and 6^3, 191.18 M params) fwd + DDPM "
                               "step on 1x4x24^3, 1000-step schedule scaled_linear_beta 0.0015-0.0195 (BASELINE configs[2])",
                   "per_gpu_batch": 1, "parallelism": f"replicas x{world} (independent chains, no collective)",
                   "weights": "random init, seeded",
                   "launch": ("eager (one C-ABI call per forward, ~150 kernel launches)" if args.eager else
                              "one HIP graph launch per denoising step: forward plan (~150 kernels) + fused DDPM step with in-kernel "
                              "Philox noise and device-resident timestep (ldm_unet_denoise_step)" if not args.host_step else
                              "HIP graph replay of the forward plan; scheduler step driven from the host (torch.randn)")},
        "steps_per_s_per_gpu": args.steps / dt,
        # every replica's own rate (value uses the MAX of the ranks' wall clocks): min << max = one slow GPU / host share, not the code
        "replica_steps_per_s": {"min": min(per_rank_sps), "max": max(per_rank_sps), "per_rank": [round(v, 2) for v in per_rank_sps]},
        "unet_step_tflops": UNET_STEP_GFLOP / ms_per_step,
        # unet_step_tflops and whole_step_fractions price the step at the REFERENCE's FLOP count (889.1 G); the plan executes fewer
        # (executed_gflop), so matrix-pipe utilisation of the step is executed_tflops / peak, not the fraction below
        "executed_gflop": UNET_STEP_EXECUTED_GFLOP, "executed_tflops": UNET_STEP_EXECUTED_GFLOP / ms_per_step,
        "executed_frac_of_mfma_peak": UNET_STEP_EXECUTED_GFLOP / ms_per_step / MFMA_BF16_DENSE_PEAK_TFLOPS,
        # SURVEY.md section 8d asks for the three fractions side by side (per GPU): whole step vs the dense bf16 MFMA peak,
        # vs the vector-fp32 peak north_star's wording implies (157 TFLOP/s), and algorithmic bytes (0.992 GB/step) vs HBM
        "whole_step_fractions": {"mfma_bf16_dense": UNET_STEP_GFLOP / ms_per_step / MFMA_BF16_DENSE_PEAK_TFLOPS,
                                 "vector_fp32_157TF": UNET_STEP_GFLOP / ms_per_step / 157.0,
                                 "hbm_8TBps": 0.992 / (ms_per_step * 1e-3) / 8000.0},
    }
    if profile and prof[0] > 0:
        achieved = prof[2] / (prof[1] * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic()
        out["roofline"] = {
            "bound": "mfma", "achieved": achieved, "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "conv3_halo_kernel<6> (implicit-GEMM 3x3x3 stride-1 conv3d, 126x128 tile, W-halo reuse, bf16 MFMA 16x16x32)",
            "launches": int(prof[0]), "avg_launch_us": prof[1] * 1e3 / prof[0],
            "algorithmic_gflop_per_launch": prof[2] / prof[0] / 1e9,
            "share_of_conv_flops": prof[2] / prof[4] if prof[4] else None,
            # the same measurement split by launch shape (identified by its algorithmic FLOP count): the 24^3 launches carry the
            # FLOPs, the split-K 12^3 / 6^3 launches are bound by the per-launch floor
            "by_shape": [{"gflop_per_launch": fl / 1e9, "launches_per_step": n / args.steps, "avg_launch_us": ms * 1e3 / n,
                          "tflops": fl * n / (ms * 1e-3) / 1e12, "frac": fl * n / (ms * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS}
                         for fl, (n, ms) in sorted(by_shape.items(), reverse=True)],
            "whole_step_frac_of_mfma_peak": UNET_STEP_GFLOP / ms_per_step / MFMA_BF16_DENSE_PEAK_TFLOPS,
            "how": "HIP events on the launch stream around every launch of this kernel, second pass of the same K steps; "
                   "traffic = HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), read side x2 per "
                   "the gfx950 FETCH_SIZE correction",
            # two per-launch figures exist and differ by method: THIS one (events around eager launches: each carries ~3 us of event
            # and launch-gap overhead, so frac reads ~8 % low) and the rocprofv3 kernel trace of the graph-replayed step in profiles/
            "authoritative": "profiles/<newest r*_summary.md>: rocprofv3 --kernel-trace avg duration of this kernel under graph replay "
                             "(no per-launch events); 'achieved' / 'frac' in this line are the live HIP-event measurement the contract asks "
                             "for and read ~8 % lower by method; by_shape has the same bias per shape",
        }
    if fp32_leg is not None:
        out["fp32_mode"] = fp32_leg
    if other is not None:
        out["other_paths"] = other
    state["out"] = out
    extra = {}
    if not args.no_ddp_train:
        del unet
        torch.cuda.empty_cache()
        extra["ddp_train"] = run_ddp()
    if not args.no_cpu_baseline and world == 1:
        extra["cpu_baseline"] = cpu_baseline()
    emit(extra)
    if dist is not None:
        dist.destroy_process_group()
    if "error" in (extra.get("ddp_train") or {}):          # the headline line is printed; a failed leg still fails the run (exit code 3)
        sys.exit(3)


if __name__ == "__main__":
    main()
