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
    _run("train_autoencoder.py", env_file, "--random-init", "--synthetic", "8", "--max-steps", "4")
    assert os.path.exists(tmp_path / "ckpt" / "autoencoder.pt")
    log = _run("train_diffusion.py", env_file, "--random-init", "--max-steps", "6")
    assert "scale_factor" in log and os.path.exists(tmp_path / "ckpt" / "diffusion_unet.pt")
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
