"""The three entry points end to end on synthetic pairs (-m gpu): SURVEY.md section 8 row H and 8f-4.

train_autoencoder.py -> train_diffusion.py (writes scale_factor.json next to the checkpoints) -> inference.py with
--condition (concat-conditioned sampling, batched): same flags / JSON schema as 3d_ldm/train_autoencoder.py,
3d_ldm/train_diffusion.py, 3d_ldm/inference.py, each run as its own process like the reference's launch scripts do."""
import glob
import json
import os
import struct
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(script, env_file, *extra):
    cmd = [sys.executable, os.path.join(ROOT, script), "-e", env_file, "-c", os.path.join(ROOT, "config", "config_synthetic_train.json"), *extra]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout + r.stderr


def test_train_then_conditional_sampling(tmp_path):
    env = {"npz_dir": str(tmp_path / "pairs"), "val_fraction": 0.25, "model_dir": str(tmp_path / "ckpt"),
           "tfevent_path": str(tmp_path / "tfevent"), "output_dir": str(tmp_path / "out"), "resume_ckpt": False, "seed": 0}
    env_file = str(tmp_path / "environment.json")
    with open(env_file, "w") as fh:
        json.dump(env, fh)
    log_ae = _run("train_autoencoder.py", env_file, "--random-init", "--synthetic", "8", "--max-steps", "4", "--profile")
    assert os.path.exists(tmp_path / "ckpt" / "autoencoder.pt")
    # every shipped config sets a perceptual weight; without --perceptual-weights the term is dropped AND recorded (not only warned)
    assert "perceptual_term: dropped" in log_ae and json.load(open(tmp_path / "ckpt" / "perceptual_term.json"))["perceptual_term"] == "dropped"
    # --profile (3d_ldm/train_autoencoder.py:81,312-329): per-op timelines of the traced steps
    assert "Profiler started" in log_ae and "[profile]" in log_ae and glob.glob(os.path.join(ROOT, "profiler_logs", "plan_trace_cycle*.csv"))
    log = _run("train_diffusion.py", env_file, "--random-init", "--max-steps", "6", "--gpu-transforms", "--sample-steps", "5")
    assert "scale_factor" in log and os.path.exists(tmp_path / "ckpt" / "diffusion_unet.pt")
    # the periodic rank-0 conditional sample of 3d_ldm/train_diffusion.py:306-359 (epoch 0 is a multiple of 2 * val_interval)
    import numpy as np
    smp = np.load(tmp_path / "tfevent" / "diffusion" / "samples" / "epoch_0.npz")
    assert sorted(smp.files) == sorted(f"{n}_{a}" for n in ("val_lowcount_input", "val_highcount_gt", "val_denoised_cond") for a in range(3))
    assert all(np.isfinite(smp[k]).all() and smp[k].ndim == 2 for k in smp.files)
    assert smp["val_denoised_cond_0"].shape == smp["val_highcount_gt_0"].shape and "conditional sample (5 steps" in log
    sf = json.load(open(tmp_path / "ckpt" / "scale_factor.json"))["scale_factor"]
    assert sf > 0
    pair = sorted(glob.glob(str(tmp_path / "pairs" / "*.npz")))[0]
    # the UNet of this config is concat-conditioned: unconditional sampling must refuse, conditional sampling must run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "inference.py"), "-e", env_file, "-c",
                          os.path.join(ROOT, "config", "config_synthetic_train.json"), "-n", "1", "--steps", "3"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "--condition" in (bad.stdout + bad.stderr)
    _run("inference.py", env_file, "-n", "3", "--batch", "2", "--steps", "4", "--condition", pair)
    vols = sorted(glob.glob(str(tmp_path / "out" / "*.nii")))
    assert len(vols) == 3
    _run("inference.py", env_file, "-n", "4", "--batch", "1", "--chains", "2", "--steps", "4", "--condition", pair)   # two concurrent chains
    assert len(glob.glob(str(tmp_path / "out" / "*.nii"))) == 7
    with open(vols[0], "rb") as fh:                      # NIfTI-1 header: sizeof_hdr 348, dim[1..3] = the decoded volume
        hdr = fh.read(348)
    assert struct.unpack("<i", hdr[:4])[0] == 348
    dim = struct.unpack("<8h", hdr[40:56])
    assert dim[1] == dim[2] == dim[3] and dim[1] % 4 == 0 and dim[1] >= 64


def test_entry_points_in_the_fp32_precision_mode(tmp_path):
    """--precision fp32 (the reference's arithmetic, train_autoencoder.py / train_diffusion.py without --amp): stage 1 (the warm-up
    phase: reconstruction + KL), stage 2 and sampling run end to end on the fp32 kernels."""
    env = {"npz_dir": str(tmp_path / "pairs"), "val_fraction": 0.25, "model_dir": str(tmp_path / "ckpt"),
           "tfevent_path": str(tmp_path / "tfevent"), "output_dir": str(tmp_path / "out"), "resume_ckpt": False, "seed": 0}
    env_file = str(tmp_path / "environment.json")
    with open(env_file, "w") as fh:
        json.dump(env, fh)
    log = _run("train_autoencoder.py", env_file, "--random-init", "--synthetic", "8", "--max-steps", "3", "--precision", "fp32")
    assert os.path.exists(tmp_path / "ckpt" / "autoencoder.pt") and "nan" not in log.lower()
    log = _run("train_diffusion.py", env_file, "--random-init", "--max-steps", "3", "--precision", "fp32")
    assert os.path.exists(tmp_path / "ckpt" / "diffusion_unet.pt") and "nan" not in log.lower()
    pair = sorted(glob.glob(str(tmp_path / "pairs" / "*.npz")))[0]
    _run("inference.py", env_file, "-n", "1", "--steps", "3", "--condition", pair, "--precision", "fp32")
    assert len(glob.glob(str(tmp_path / "out" / "*.nii"))) == 1


@pytest.mark.parametrize("shape,b", [((37, 41, 29), 2), ((64, 64, 64), 1), ((144, 176, 112), 1)])
def test_percentile_scaling_on_device_matches_the_host_transform(cuda, shape, b):
    """ScaleIntensityRangePercentiles(0, 99.5 -> 0, 1) (3d_ldm/utils.py:94-107) on the device against the numpy restatement of
    ldm3d.data: the order statistics are exact (radix select); the interpolated percentile differs from numpy's by its float32 index
    arithmetic (<= 2e-5 relative), so the scaled volumes agree to ~1e-5."""
    import numpy as np
    from ldm3d import data
    rng = np.random.RandomState(sum(shape))
    vols = rng.gamma(2.0, 1.0, size=(b, *shape)).astype(np.float32)
    vols[0, 0, 0, :5] = [-3.0, 0.0, -0.0, 1e-30, 250.0]                     # negatives, signed zeros, a denormal-ish value, an outlier
    got = data.scale_percentiles_gpu(torch.from_numpy(vols).to(cuda)).cpu().numpy()
    for i in range(b):
        ref = data.scale_percentiles(vols[i])
        assert np.abs(got[i] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
        s = np.sort(vols[i].reshape(-1))
        n = s.size
        pos = (n - 1) * 0.995
        lo = int(np.floor(pos))
        a_max = float(s[lo]) + (float(s[min(lo + 1, n - 1)]) - float(s[lo])) * (pos - lo)          # exact float64 statement of the percentile
        a_min = float(s[0])
        exact = ((vols[i].astype(np.float64) - a_min) / (np.float32(a_max) - np.float32(a_min))).astype(np.float32)
        assert np.abs(got[i] - exact).max() <= 2e-6 * max(1.0, np.abs(exact).max())
    const = torch.full((1, 4, 4, 4), 3.0, device=cuda)
    assert float(data.scale_percentiles_gpu(const).abs().max()) == 0.0          # constant volume -> b_min (MONAI)
    other = data.scale_percentiles_gpu(torch.from_numpy(vols).to(cuda), 5.0, 50.0, -1.0, 1.0).cpu().numpy()
    ref = data.scale_percentiles(vols[0], 5.0, 50.0, -1.0, 1.0)
    assert np.abs(other[0] - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())
