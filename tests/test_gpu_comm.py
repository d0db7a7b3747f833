"""RCCL through the library's C ABI and the bucketed, backward-overlapped gradient exchange (-m gpu, one GPU).

A one-GPU box can only run a world-size-1 RCCL communicator, which still exercises every entry point (unique id, init, all-reduce
SUM / AVG in fp32 and bf16, broadcast, barrier, destroy), the bucket plumbing of the backward plans (events, comm stream, join) and
its timeline.  The N > 1 behaviour of the in-library path runs here against a fake second rank (ldm_comm_init_custom: the
transport is the only thing replaced); tests/test_grad_schedule_cpu.py checks the bucket schedule itself without a GPU; the gloo tests in
tests/test_dist_cpu.py cover the torch.distributed fallback; the 8-GPU run is the driver's.
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


class FakePeer:
    """Stands in for rank 1 of a 2-rank job on one GPU: the transport handed to ``ldm_comm_init_custom``.  It receives exactly what
    ncclAllReduce would (buffer, count, dtype, op, stream) and leaves op(own, peer) in the buffer ON THAT STREAM, with the peer's
    gradients taken from a tensor computed beforehand.  Everything else -- which ranges are handed over, when, with which op, the
    join in front of the optimizer -- is the library's production code."""

    def __init__(self, flat: torch.Tensor, peer: torch.Tensor):
        self.flat, self.peer, self.calls = flat, peer, []

    def __call__(self, buf, count, dtype, op, stream):
        off = (buf - self.flat.data_ptr()) // 4
        self.calls.append((off, count, dtype, op))
        if not (0 <= off and off + count <= self.flat.numel()):
            return 1
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            view = self.flat[off:off + count]
            view.add_(self.peer[off:off + count])
            if op == 1:
                view.mul_(0.5)
        return 0


def test_bucketed_exchange_with_a_fake_second_rank_gives_the_mean(cuda, monkeypatch):
    """World size 2 through the in-library path (3d_ldm/train_diffusion.py:147-149,214: DDP averages the ranks' gradients inside
    backward).  Rank 0 (this process) differentiates sample A, the fake peer contributes the gradients of sample B: the buffer must end
    up as (gA + gB) / 2 bit for bit -- a bucket issued before its gradients were final, a range reduced twice or never, a sum instead
    of a mean, or an optimizer step that does not wait for the comm stream would all show here (none of them can at world size 1)."""
    from ldm3d import _lib
    from ldm3d.networks import DiffusionModelUNet
    from ldm3d.optim import FlatAdam
    from ldm3d.trainer import GradSync
    from oracle import unet as ou
    monkeypatch.setenv("LDM_GRAD_BUCKET_MB", "1")
    cfg = cfgs.UNET_TINY
    sd = ou.init_state_dict(ou.unet_param_shapes(cfg), 3, gain=0.5)

    def fresh():
        m = DiffusionModelUNet(**cfg)
        m.load_state_dict(sd)
        m = m.to(cuda).train()
        m.flatten_parameters()
        return m
    m = fresh()
    g = torch.Generator().manual_seed(7)
    xs = torch.randn((2, 4, 8, 8, 8), generator=g).to(cuda)
    tg = torch.randn((2, 4, 8, 8, 8), generator=g).to(cuda)
    ts = torch.tensor([17.0, 803.0], device=cuda)
    gA = _train_once(m, xs[:1], ts[:1], tg[:1])
    gB = _train_once(m, xs[1:], ts[1:], tg[1:])
    assert not torch.equal(gA, gB)
    peer = FakePeer(m.flat_grads, gB)
    sync = GradSync()
    assert sync.attach(m, transport=peer, world=2, rank=0) and sync.attached(m)
    L = _lib.lib()
    assert L.ldm_comm_world(m._grad_comm) == 2 and L.ldm_comm_rank(m._grad_comm) == 0
    m.flat_grads.fill_(float("nan"))
    synced = _train_once(m, xs[:1], ts[:1], tg[:1])
    total = m.flat_grads.numel()
    assert torch.equal(synced, (gA + gB) * 0.5)
    # what the transport saw: fp32, op = mean, ranges that tile the buffer back to front exactly once
    assert len(peer.calls) >= 4 and all(d == 0 and op == 1 for _, _, d, op in peer.calls)
    assert peer.calls[0][0] + peer.calls[0][1] == total and peer.calls[-1][0] == 0
    assert all(a[0] == b[0] + b[1] for a, b in zip(peer.calls, peer.calls[1:]))
    # the B = 2 gradient of the same two samples (mse mean over the batch) is that mean, up to bf16 summation order
    gAB = _train_once(fresh(), xs, ts, tg)
    r = float((gAB - synced).norm() / gAB.norm())
    print(f"mean of the two ranks' gradients vs the B = 2 gradient: rel-L2 {r:.2e}")
    assert r <= 2e-2
    # and the optimizer waits for the exchange: the same step with a peer that is slow to answer updates the parameters identically
    def step_with(peer_delay):
        mm = fresh()
        opt = FlatAdam(mm, lr=1e-3, max_grad_norm=1.0)
        pr = FakePeer(mm.flat_grads, gB)
        if peer_delay:
            inner = pr.__call__

            def slow(buf, count, dtype, op, stream):
                with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                    torch.cuda._sleep(20_000_000)              # ~10 ms on the comm stream in front of every bucket
                return inner(buf, count, dtype, op, stream)
            tr = slow
        else:
            tr = pr
        assert GradSync().attach(mm, transport=tr, world=2, rank=0)
        out = mm(x=xs[:1], timesteps=ts[:1])
        F.mse_loss(out.float(), tg[:1]).backward()
        opt.step()
        torch.cuda.synchronize()
        return mm.flat_params.clone()
    assert torch.equal(step_with(False), step_with(True))


def test_fake_second_rank_on_the_benchmark_unet_and_the_autoencoder(cuda, monkeypatch):
    """The same check at full width (191 M gradients, default 48 MB buckets) and for the AutoencoderKL backward plan."""
    from ldm3d.networks import AutoencoderKL, DiffusionModelUNet
    from ldm3d.trainer import GradSync
    from oracle import autoencoder as oa
    from oracle import unet as ou
    cfg = cfgs.UNET_FULL
    m = DiffusionModelUNet(**cfg)
    m.load_state_dict(ou.init_state_dict(ou.unet_param_shapes(cfg), 3, gain=0.5))
    m = m.to(cuda).train()
    m.flatten_parameters()
    g = torch.Generator().manual_seed(5)
    xs = torch.randn((2, 4, 16, 16, 16), generator=g).to(cuda)
    tg = torch.randn((2, 4, 16, 16, 16), generator=g).to(cuda)
    ts = torch.tensor([400.0, 91.0], device=cuda)
    gA = _train_once(m, xs[:1], ts[:1], tg[:1])
    gB = _train_once(m, xs[1:], ts[1:], tg[1:])
    peer = FakePeer(m.flat_grads, gB)
    assert GradSync().attach(m, transport=peer, world=2)
    m.flat_grads.fill_(float("nan"))
    assert torch.equal(_train_once(m, xs[:1], ts[:1], tg[:1]), (gA + gB) * 0.5)
    assert sum(c for _, c, _, _ in peer.calls) == m.flat_grads.numel()
    del m, gA, gB, peer
    monkeypatch.setenv("LDM_GRAD_BUCKET_MB", "2")
    vcfg = cfgs.VAE_TINY_ATTN
    v = AutoencoderKL(**vcfg)
    v.load_state_dict(ou.init_state_dict(oa.ae_param_shapes(vcfg), 3))
    v = v.to(cuda).train()
    v.flatten_parameters()
    imgs = torch.rand((2, 1, 16, 16, 16), device=cuda)
    eps = torch.randn((2, 8, 4, 4, 4), device=cuda)

    def once(k):
        rec, mu, sigma = v(imgs[k:k + 1], eps=eps[k:k + 1])
        (F.l1_loss(rec, imgs[k:k + 1]) + 1e-6 * oa.kl_loss(mu, sigma).mean()).backward()
        torch.cuda.synchronize()
        return v.flat_grads.clone()
    vA, vB = once(0), once(1)
    vpeer = FakePeer(v.flat_grads, vB)
    assert GradSync().attach(v, transport=vpeer, world=2)
    v.flat_grads.fill_(float("nan"))
    assert torch.equal(once(0), (vA + vB) * 0.5)


def test_two_real_processes_share_the_gradient_exchange(cuda):
    """World size 2 with two real PROCESSES on this GPU (tests/two_rank_gpu_worker.py; gloo carries the bytes, the library's bucketed
    path does everything else: 3d_ldm/train_diffusion.py:121-123,147-149,214).  The fake-peer test above plays rank 1 inside one process;
    here rendezvous, two independent backward plans, two comm streams and two optimizers meet: the averaged gradient must be
    (g_0 + g_1) / 2 bit for bit in both ranks and the parameters must stay identical across the processes after two steps."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "two_rank_gpu_worker.py")], env=env, cwd=root,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    recs = []
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
        recs.append(json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1]))
    print("two processes:", recs)
    assert sorted(r["rank"] for r in recs) == [0, 1]
    for r in recs:
        assert r["mean_exact"] and r["tiled"] and r["buckets"] >= 4 and r["checksums_equal"] and r["moved"], r
    assert recs[0]["checksum"] == recs[1]["checksum"]

