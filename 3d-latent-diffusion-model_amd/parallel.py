"""Process-group plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

Mirrors the reference's helpers (3d_ldm/utils.py:55-63 ``setup_ddp``; scalar reductions at
3d_ldm/train_diffusion.py:121-123,281-283 and 3d_ldm/train_autoencoder.py:29-39).  The sampling / denoising path
shards by independent chains (no exchange step); collectives only synchronise and reduce scalars.  Unlike the
reference's launch scripts (3d_ldm/train_LDM.sh:40-42) nothing here disables peer-to-peer: xGMI is the fabric.
"""
from __future__ import annotations

import datetime
import os
from typing import List

import torch
import torch.distributed as dist


def setup_ddp(rank: int, world_size: int, backend: str | None = None, timeout_s: int = 36000):
    """init_process_group(env://) + barrier, as 3d_ldm/utils.py:55-63 (backend defaults to nccl when a GPU exists)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", rank))
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world_size,
                            timeout=datetime.timedelta(seconds=timeout_s), **kw)
    dist.barrier()
    return dist.group.WORLD


def cleanup_ddp():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def shard_indices(n: int, rank: int, world_size: int) -> List[int]:
    """Independent units (samples / chains) owned by this rank: i with i % world == rank."""
    return list(range(rank, n, world_size))


def _dev():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def max_over_ranks(value: float) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_dev())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_mean(t: torch.Tensor) -> torch.Tensor:
    """scale_factor / validation-loss averaging (3d_ldm/train_diffusion.py:121-123,281-283)."""
    if not (dist.is_available() and dist.is_initialized()):
        return t
    t = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t / dist.get_world_size()
