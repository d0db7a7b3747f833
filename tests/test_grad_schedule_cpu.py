"""The library's own bucket bookkeeping for the data-parallel gradient exchange, checked without a GPU.

Reference behaviour: DistributedDataParallel all-reduces gradient buckets while ``backward`` is still running and the optimizer only
sees fully reduced gradients (3d_ldm/train_diffusion.py:147-149 wrap, :214 backward, :217-219 clip + step).  Here the backward launch
plan carries OP_BUCKET / OP_BUCKET_JOIN entries (csrc/ldm3d.hip, ``Builder::close_bucket``); plans are built on the host, so
``ldm_model_grad_schedule`` can list, in launch order, every op that leaves final values in the flat gradient buffer and every bucket
hand-over.  The invariants a wrong schedule would break on N > 1 ranks (and that a world-size-1 run cannot see):

  * the buckets tile [0, numel) exactly once                       (an element reduced twice is divided by world twice; a gap is never reduced)
  * no op writes into a range after that range's bucket was issued  (the peers would get a stale value, the local copy a fresh one)
  * every element is written before its bucket is issued            (the whole buffer is overwritten by backward: nothing is "left as is")
  * the join comes after the last bucket and is the last event      (clip + Adam follow it in stream order)
"""
import ctypes as C
import os

import numpy as np
import pytest

import cfgs
from ldm3d import _lib
from ldm3d.networks import AutoencoderKL, DiffusionModelUNet


def schedule(model, shape):
    L = _lib.lib()
    B, D, H, W = shape
    cnt = L.ldm_model_grad_schedule(model._h, B, D, H, W, None, None, None, None, 0)
    assert cnt > 0, _lib.lib().ldm_last_error()
    kind, op = (C.c_int * cnt)(), (C.c_int * cnt)()
    lo, n = (C.c_int64 * cnt)(), (C.c_int64 * cnt)()
    assert L.ldm_model_grad_schedule(model._h, B, D, H, W, kind, lo, n, op, cnt) == cnt
    return [(kind[k], lo[k], n[k], op[k]) for k in range(cnt)], int(L.ldm_model_param_numel_total(model._h))


def check(events, total):
    assert all(k != 3 for k, *_ in events), "an op writes the gradient buffer in a way the schedule dump does not classify"
    ops = [e[3] for e in events]
    assert ops == sorted(ops)
    written = np.zeros(total, dtype=np.uint8)       # 1 = final value present
    issued = np.zeros(total, dtype=np.uint8)        # 1 = handed to the communicator
    buckets, joined = [], False
    for kind, lo, n, _ in events:
        assert not joined, "events after the join"
        if kind == 0:
            assert 0 <= lo and lo + n <= total and n > 0
            assert not issued[lo:lo + n].any(), f"write into [{lo}, {lo + n}) after its bucket was issued"
            written[lo:lo + n] = 1
        elif kind == 1:
            assert 0 <= lo and lo + n <= total and n > 0
            assert not issued[lo:lo + n].any(), f"bucket [{lo}, {lo + n}) overlaps an earlier bucket"
            assert written[lo:lo + n].all(), f"bucket [{lo}, {lo + n}) issued before all of its gradients were final"
            issued[lo:lo + n] = 1
            buckets.append((lo, n))
        elif kind == 2:
            joined = True
    assert joined and events[-1][0] == 2, "no join behind the last bucket"
    assert issued.all(), "part of the gradient buffer is never reduced"
    return buckets


@pytest.mark.parametrize("cfg,shape", [(cfgs.UNET_TINY, (1, 8, 8, 8)), (cfgs.UNET_TINY_COND, (2, 8, 8, 8)),
                                       (cfgs.UNET_TINY_ALT, (1, 8, 8, 8)), (cfgs.UNET_FULL, (1, 24, 24, 24))])
def test_unet_bucket_schedule(cfg, shape):
    m = DiffusionModelUNet(**cfg)
    events, total = schedule(m, shape)
    buckets = check(events, total)
    # the walk finishes the buffer from its end towards its front: bucket k ends where bucket k-1 began
    for (lo0, _), (lo1, n1) in zip(buckets, buckets[1:]):
        assert lo1 + n1 == lo0
    if cfg is cfgs.UNET_FULL:
        assert total == 191_175_172
        # default 48 MB buckets: every bucket but the last (time embedding + whatever is left) holds at least that much, none is
        # absurdly larger (a single conv's 28 MB weight is the granularity)
        mb = [n * 4 / 2 ** 20 for _, n in buckets]
        assert len(buckets) >= 8 and min(mb[:-1]) >= 48 and max(mb) < 48 + 64, mb
        # overlap: the first bucket is on the wire long before backward ends
        first_bucket_op = next(e[3] for e in events if e[0] == 1)
        assert first_bucket_op < events[0][3] + 0.35 * (events[-1][3] - events[0][3])


@pytest.mark.parametrize("cfg,shape", [(cfgs.VAE_TINY, (1, 16, 16, 16)), (cfgs.VAE_TINY_ATTN, (2, 16, 16, 16))])
def test_autoencoder_bucket_schedule(cfg, shape):
    m = AutoencoderKL(**cfg)
    events, total = schedule(m, shape)
    check(events, total)


def test_bucket_size_follows_the_environment(monkeypatch):
    """LDM_GRAD_BUCKET_MB is read when the plan is built: a smaller value gives more buckets with the same invariants (ranks must
    agree on it: GradSync.attach broadcasts rank 0's value before any plan exists)."""
    monkeypatch.setenv("LDM_GRAD_BUCKET_MB", "1")
    m = DiffusionModelUNet(**cfgs.UNET_TINY)
    events, total = schedule(m, (1, 8, 8, 8))
    small = check(events, total)
    monkeypatch.setenv("LDM_GRAD_BUCKET_MB", "48")
    m2 = DiffusionModelUNet(**cfgs.UNET_TINY)
    events2, _ = schedule(m2, (1, 8, 8, 8))
    assert len(small) > len(check(events2, total))
