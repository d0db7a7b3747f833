"""Paired low-count / high-count volume pipeline of the diffusion trainer.

What the reference builds in ``prepare_dataloader`` (3d_ldm/utils.py:66-240) out of MONAI dictionary transforms, restated
on numpy/torch only (MONAI is not a dependency here):

* NPZ pair files: array ``arr0`` or ``arr_0`` (else the first key) of shape ``(2, D, H, W)``; index 0 = low-count
  "image", index 1 = high-count "label" (3d_ldm/utils.py:124-143);
* centre crop (``randcrop=False``, what train_diffusion.py:70-80 asks for) or random crop to ``patch_size``
  (3d_ldm/utils.py:85-92);
* ``ScaleIntensityRangePercentiles(lower=0, upper=99.5, b_min=0, b_max=1)`` per volume, no clipping
  (3d_ldm/utils.py:94-107);
* train/val split: either two directories or one directory shuffled with ``RandomState(seed)`` and cut at
  ``val_fraction`` (3d_ldm/utils.py:157-186);
* ``DistributedSampler`` sharding, ``drop_last`` under DDP (3d_ldm/utils.py:188-193,214).
"""
from __future__ import annotations

import os
from glob import glob
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler


def load_pair(path: str) -> Tuple[np.ndarray, np.ndarray]:
    with np.load(path) as data:
        keys = list(data.keys())
        key = "arr0" if "arr0" in keys else ("arr_0" if "arr_0" in keys else (keys[0] if keys else None))
        if key is None:
            raise RuntimeError(f"NPZ {path} holds no array")
        arr = data[key]
    if arr.ndim < 4 or arr.shape[0] < 2:
        raise RuntimeError(f"NPZ {path}: expected shape (2, D, H, W), got {arr.shape}")
    return np.asarray(arr[0], dtype=np.float32), np.asarray(arr[1], dtype=np.float32)


def crop_start(shape: Sequence[int], roi: Sequence[int], rng: np.random.RandomState | None) -> List[int]:
    """Centre crop start (MONAI CenterSpatialCrop: (dim - roi) // 2, roi clipped to the volume) or a random one."""
    out = []
    for d, r in zip(shape, roi):
        r = min(int(r), int(d))
        out.append(int(rng.randint(0, d - r + 1)) if rng is not None else (int(d) - r) // 2)
    return out


def crop(vol: np.ndarray, start: Sequence[int], roi: Sequence[int]) -> np.ndarray:
    sl = tuple(slice(s, s + min(int(r), vol.shape[-3 + i])) for i, (s, r) in enumerate(zip(start, roi)))
    return vol[(..., *sl)]


def scale_percentiles(vol: np.ndarray, lower: float = 0.0, upper: float = 99.5, b_min: float = 0.0, b_max: float = 1.0) -> np.ndarray:
    a_min, a_max = np.percentile(vol, lower), np.percentile(vol, upper)
    if a_max - a_min == 0.0:                           # MONAI's ScaleIntensityRange: constant image -> b_min
        return np.full_like(vol, b_min, dtype=np.float32)
    return ((vol - a_min) / (a_max - a_min) * (b_max - b_min) + b_min).astype(np.float32)


def scale_percentiles_gpu(vol: torch.Tensor, lower: float = 0.0, upper: float = 99.5, b_min: float = 0.0, b_max: float = 1.0) -> torch.Tensor:
    """``scale_percentiles`` for a batch of volumes that already live on the GPU: ``vol`` [B, ...] fp32 CUDA, every ``vol[b]`` scaled by
    its OWN percentiles (``ldm_op_scale_intensity_percentiles``: exact order statistics by a radix select, one pass to apply).  Lets the
    trainers keep raw crops on the device and skip the host-side sort of ~3 M voxels per volume per step."""
    from . import _lib
    if not vol.is_cuda:
        raise _lib.LdmError("scale_percentiles_gpu: CUDA tensors only (the host path is scale_percentiles)")
    x = vol.detach().to(torch.float32).contiguous()
    B = x.shape[0]
    n = x[0].numel()
    out = torch.empty_like(x)
    L = _lib.lib()
    scratch = torch.empty((L.ldm_op_scale_intensity_percentiles_scratch_bytes(B),), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.ldm_op_scale_intensity_percentiles(x.data_ptr(), out.data_ptr(), B, n, lower, upper, b_min, b_max,
                                                        scratch.data_ptr(), scratch.numel(), _lib.current_stream()))
    return out


class PairVolumes(Dataset):
    """{'image': low-count [1,D,H,W], 'label': high-count [1,D,H,W]} float tensors."""

    def __init__(self, files: Sequence[str], patch_size: Sequence[int], randcrop: bool = False, seed: int = 0,
                 dtype: torch.dtype = torch.float32, scale_on_host: bool = True):
        """``scale_on_host=False`` returns the raw crops: the caller applies ``scale_percentiles_gpu`` after moving the batch to the GPU."""
        self.files, self.patch, self.randcrop, self.seed, self.dtype = list(files), list(patch_size), randcrop, seed, dtype
        self.scale_on_host = scale_on_host
        self._draws = {}                                   # per sample: how often it has been cropped (RandSpatialCropd draws afresh every time)
        self.epoch = 0

    def __len__(self):
        return len(self.files)

    def set_epoch(self, epoch: int) -> None:
        """Optional: pins the crop stream to the epoch (workers that re-create the dataset each epoch stay reproducible)."""
        self.epoch = int(epoch)
        self._draws = {}

    def __getitem__(self, idx) -> Dict[str, torch.Tensor]:
        low, high = load_pair(self.files[idx])
        rng = None
        if self.randcrop:
            # a fresh crop on EVERY access, as MONAI's RandSpatialCropd (3d_ldm/utils.py:87): the stream is keyed by (seed, sample,
            # epoch, access count), so two epochs see different crops of a sample and a re-run with the same seed repeats them
            k = self._draws.get(idx, 0)
            self._draws[idx] = k + 1
            rng = np.random.RandomState(((self.seed * 1000003 + idx) * 7919 + self.epoch * 104729 + k) % (2 ** 32))
        st = crop_start(low.shape[-3:], self.patch, rng)           # one start for both volumes of the pair
        out = {}
        for key, vol in (("image", low), ("label", high)):
            v = crop(vol, st, self.patch)
            if self.scale_on_host:
                v = scale_percentiles(v)
            out[key] = torch.from_numpy(np.ascontiguousarray(v)).reshape(1, *v.shape[-3:]).to(self.dtype)
        return out


def split_files(args) -> Tuple[List[str], List[str]]:
    tr, va, one = getattr(args, "npz_dir_train", None), getattr(args, "npz_dir_val", None), getattr(args, "npz_dir", None)
    if tr and va and os.path.isdir(tr) and os.path.isdir(va):
        train, val = sorted(glob(os.path.join(tr, "*.npz"))), sorted(glob(os.path.join(va, "*.npz")))
        if not train or not val:
            raise ValueError(f"no .npz files under {tr} / {va}")
        return train, val
    if not one or not os.path.isdir(one):
        raise ValueError("provide (npz_dir_train and npz_dir_val) or npz_dir with .npz pair files")
    files = sorted(glob(os.path.join(one, "*.npz")))
    if not files:
        raise ValueError(f"no .npz files under {one}")
    idx = np.arange(len(files))
    np.random.RandomState(int(getattr(args, "seed", 0))).shuffle(idx)
    n_val = int(len(idx) * float(getattr(args, "val_fraction", 0.1)))
    val = [files[i] for i in idx[:n_val]] if n_val > 0 else [files[idx[0]]]
    return [files[i] for i in idx[n_val:]], val


def prepare_dataloader(args, batch_size: int, patch_size: Sequence[int], randcrop: bool = False, rank: int = 0,
                       world_size: int = 1, num_workers: int = 0, size_divisible: int = 16, scale_on_host: bool = True):
    """``scale_on_host=False``: batches carry raw crops, ``gpu_scale_batch`` finishes the transform chain on the device."""
    train_files, val_files = split_files(args)
    seed = int(getattr(args, "seed", 0))
    train_ds = PairVolumes(train_files, patch_size, randcrop, seed, scale_on_host=scale_on_host)
    # validation is centre-cropped; when training uses random crops the reference validates on a 1.5x patch rounded up to a
    # multiple of size_divisible (3d_ldm/utils.py:75,88; train_autoencoder.py:131 passes 2^(levels - 1), train_diffusion.py:69 16)
    val_patch = [int(np.ceil(1.5 * p / float(size_divisible)) * size_divisible) for p in patch_size] if randcrop else list(patch_size)
    val_ds = PairVolumes(val_files, val_patch, False, seed, scale_on_host=scale_on_host)
    ddp = world_size > 1
    ts = DistributedSampler(train_ds, num_replicas=world_size, rank=rank, shuffle=True) if ddp else None
    vs = DistributedSampler(val_ds, num_replicas=world_size, rank=rank, shuffle=False) if ddp else None
    kw = dict(batch_size=batch_size, num_workers=num_workers, pin_memory=torch.cuda.is_available(), drop_last=ddp)
    return (DataLoader(train_ds, shuffle=(ts is None), sampler=ts, **kw), DataLoader(val_ds, shuffle=False, sampler=vs, **kw))


def gpu_scale_batch(batch: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    """Moves a raw-crop batch to the GPU and applies the percentile scaling there ("image" and "label" each by its own percentiles)."""
    return {k: scale_percentiles_gpu(v.to(device)) for k, v in batch.items()}


def write_synthetic_pairs(directory: str, n: int, shape: Sequence[int], seed: int = 0) -> List[str]:
    """Synthetic stand-in for the PET pairs (no dataset can be downloaded here): smooth blobs = "high count",
    the same field Poisson-thinned = "low count".  Returns the file list."""
    os.makedirs(directory, exist_ok=True)
    rng = np.random.RandomState(seed)
    zz, yy, xx = np.meshgrid(*[np.linspace(-1, 1, s, dtype=np.float32) for s in shape], indexing="ij")
    files = []
    for i in range(n):
        high = np.zeros(shape, dtype=np.float32)
        for _ in range(6):
            c, w, a = rng.uniform(-0.6, 0.6, 3), rng.uniform(0.15, 0.5), rng.uniform(0.3, 1.0)
            high += a * np.exp(-((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) / (2 * w * w))
        low = rng.poisson(np.clip(high, 0, None) * 8.0).astype(np.float32) / 8.0
        path = os.path.join(directory, f"pair_{i:04d}.npz")
        np.savez(path, arr0=np.stack([low, high]))
        files.append(path)
    return files
