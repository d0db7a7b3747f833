"""RCCL through the library's C ABI and the bucketed, backward-overlapped gradient exchange (-m gpu, one GPU).

A one-GPU box can only run a world-size-1 communicator, which still exercises every entry point (unique id, init, all-reduce
SUM / AVG in fp32 and bf16, broadcast, barrier, destroy), the bucket plumbing of the backward plans (events, comm stream, join) and
its timeline.  The N > 1 arithmetic (mean of shard gradients == full-batch gradient) is covered by the gloo tests in
tests/test_dist_cpu.py; the 8-GPU run is the driver's.
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

import cfgs

pytestmark = pytest.mark.gpu


def test_rccl_c_abi_world_size_one(cuda, built_lib):
    from ldm3d import _lib
    L = built_lib
    uid = C.create_string_buffer(128)
    _lib.check(L.ldm_comm_unique_id(uid))
    comm = C.c_void_p()
    _lib.check(L.ldm_comm_init(0, 1, uid, C.byref(comm)))
    assert L.ldm_comm_rank(comm) == 0 and L.ldm_comm_world(comm) == 1
    st = torch.cuda.current_stream().cuda_stream
    for dtype, code in ((torch.float32, 0), (torch.bfloat16, 1)):
        for op in (0, 1):                                  # sum, avg
            x = torch.randn(100003, device=cuda).to(dtype)
            ref = x.clone()
            _lib.check(L.ldm_comm_allreduce(comm, x.data_ptr(), x.numel(), code, op, st))
            torch.cuda.synchronize()
            assert torch.equal(x, ref)                     # one rank: sum = mean = identity
        y = torch.randn(4099, device=cuda).to(dtype)
        ref = y.clone()
        _lib.check(L.ldm_comm_broadcast(comm, y.data_ptr(), y.numel(), code, 0, st))
        torch.cuda.synchronize()
        assert torch.equal(y, ref)
    _lib.check(L.ldm_comm_barrier(comm, st))
    with pytest.raises(_lib.LdmError):
        _lib.check(L.ldm_comm_allreduce(comm, None, 4, 0, 0, st))
    with pytest.raises(_lib.LdmError):
        _lib.check(L.ldm_comm_broadcast(comm, x.data_ptr(), 4, 0, 3, st))      # root outside the world
    L.ldm_comm_destroy(comm)


def _train_once(m, x, t, target):
    out = m(x=x, timesteps=t)
    F.mse_loss(out.float(), target).backward()
    torch.cuda.synchronize()
    return m.flat_grads.clone()


def test_bucketed_gradient_exchange_overlaps_backward(cuda, monkeypatch):
    """Benchmark UNet, 1x4x16^3: with a communicator attached the backward plan hands tail ranges of the flat gradient buffer to
    the comm stream while it is still running.  The buckets tile the buffer exactly, back to front; the gradients equal the
    un-synchronised ones bit for bit (world size 1: mean = identity); the first bucket is issued in the first part of backward."""
    from ldm3d import _lib
    from ldm3d.networks import DiffusionModelUNet
    from ldm3d.trainer import GradSync
    from oracle import unet as ou
    monkeypatch.setenv("LDM_GRAD_BUCKET_MB", "32")
    cfg = cfgs.UNET_FULL
    m = DiffusionModelUNet(**cfg)
    m.load_state_dict(ou.init_state_dict(ou.unet_param_shapes(cfg), 3, gain=0.5))
    m = m.to(cuda).train()
    m.flatten_parameters()
    g = torch.Generator().manual_seed(4)
    x = torch.randn((1, 4, 16, 16, 16), generator=g).to(cuda)
    target = torch.randn((1, 4, 16, 16, 16), generator=g).to(cuda)
    t = torch.tensor([211.0], device=cuda)
    plain = _train_once(m, x, t, target)
    sync = GradSync()
    assert sync.attach(m, force_single=True) and sync.attached(m)
    m.flat_grads.fill_(float("nan"))
    synced = _train_once(m, x, t, target)
    assert torch.equal(plain, synced)
    L = _lib.lib()
    n_max = 256
    issue, done, elems = (C.c_double * n_max)(), (C.c_double * n_max)(), (C.c_int64 * n_max)()
    n = L.ldm_model_grad_sync_trace(m._h, issue, done, elems, n_max)
    total = int(L.ldm_model_param_numel_total(m._h))
    assert n >= 10 and sum(elems[k] for k in range(n)) == total            # 765 MB in 32 MB buckets, tiling the buffer exactly
    end = issue[n]
    assert all(issue[k] <= issue[k + 1] for k in range(n - 1)) and all(done[k] >= issue[k] for k in range(n))
    print("bucket timeline (ms since backward start; backward ends at %.2f): " % end +
          ", ".join(f"{issue[k]:.2f}" for k in range(n)))
    assert issue[0] <= 0.35 * end and issue[n // 2] <= 0.8 * end            # the exchange starts early and is spread over backward
    _lib.check(L.ldm_model_set_grad_sync(m._h, None))                        # detach: plain backward again
    again = _train_once(m, x, t, target)
    assert torch.equal(plain, again)


def test_autoencoder_backward_buckets_tile_the_buffer(cuda, monkeypatch):
    from ldm3d import _lib
    from ldm3d.networks import AutoencoderKL
    from ldm3d.trainer import GradSync
    from oracle import autoencoder as oa
    from oracle.unet import init_state_dict
    monkeypatch.setenv("LDM_GRAD_BUCKET_MB", "8")
    cfg = cfgs.VAE_FULL_ATTN
    m = AutoencoderKL(**cfg)
    m.load_state_dict(init_state_dict(oa.ae_param_shapes(cfg), 3))
    m = m.to(cuda).train()
    m.flatten_parameters()
    x = torch.rand((1, 1, 32, 32, 32), device=cuda)
    eps = torch.randn((1, 16, 8, 8, 8), device=cuda)

    def once():
        rec, mu, sigma = m(x, eps=eps)
        (F.l1_loss(rec, x) + 1e-6 * oa.kl_loss(mu, sigma).mean()).backward()
        torch.cuda.synchronize()
        return m.flat_grads.clone()
    plain = once()
    assert GradSync().attach(m, force_single=True)
    m.flat_grads.fill_(float("nan"))
    assert torch.equal(plain, once())
    L = _lib.lib()
    issue, done, elems = (C.c_double * 64)(), (C.c_double * 64)(), (C.c_int64 * 64)()
    n = L.ldm_model_grad_sync_trace(m._h, issue, done, elems, 64)
    assert n >= 5 and sum(elems[k] for k in range(n)) == int(L.ldm_model_param_numel_total(m._h))
