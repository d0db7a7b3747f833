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


def _device_view_i16(ptr, count, device):
    """An int16 torch view of `count` 2-byte elements at a raw device pointer (the library's bf16 staging slice)."""
    class _Arr:
        __cuda_array_interface__ = {"shape": (count,), "typestr": "<i2", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(_Arr(), device=device)


def test_bf16_wire_format_inside_the_bucketed_exchange(cuda, monkeypatch):
    """GradSync(grad_dtype=torch.bfloat16).attach -> ldm_model_set_grad_wire(1): every bucket crosses the wire as bf16 (cast into the
    library's staging slice on the comm stream, all-reduce(avg) on that slice, cast back into the fp32 gradient buffer): half the
    bytes of the fp32 exchange (a link-bound ring exposes 2-3 ms at the 24^3 shape).
    The fake second rank sees dtype = bf16, op = avg, counts that tile the buffer back to front, and pointers OUTSIDE the gradient
    buffer; the result equals bf16(mean(bf16 gA, bf16 gB)) bit for bit, i.e. the fp32 mean to bf16 precision (the tolerances of the
    torch-fallback wire test, tests/test_dist_cpu.py:73-78)."""
    from ldm3d import _lib
    from ldm3d.networks import DiffusionModelUNet
    from ldm3d.trainer import GradSync
    from oracle import unet as ou
    monkeypatch.setenv("LDM_GRAD_BUCKET_MB", "1")
    cfg = cfgs.UNET_TINY
    m = DiffusionModelUNet(**cfg)
    m.load_state_dict(ou.init_state_dict(ou.unet_param_shapes(cfg), 3, gain=0.5))
    m = m.to(cuda).train()
    m.flatten_parameters()
    g = torch.Generator().manual_seed(7)
    xs = torch.randn((2, 4, 8, 8, 8), generator=g).to(cuda)
    tg = torch.randn((2, 4, 8, 8, 8), generator=g).to(cuda)
    ts = torch.tensor([17.0, 803.0], device=cuda)
    gA = _train_once(m, xs[:1], ts[:1], tg[:1])
    gB = _train_once(m, xs[1:], ts[1:], tg[1:])
    total = m.flat_grads.numel()
    calls, pos = [], [total]

    def peer(buf, count, dtype, op, stream):
        inside = m.flat_grads.data_ptr() <= buf < m.flat_grads.data_ptr() + 4 * total
        calls.append((count, dtype, op, inside))
        pos[0] -= count
        off = pos[0]
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            mine = _device_view_i16(buf, count, cuda).view(torch.bfloat16)
            other = gB[off:off + count].to(torch.bfloat16)
            mine.copy_(((mine.float() + other.float()) * (0.5 if op == 1 else 1.0)).to(torch.bfloat16))
        return 0
    sync = GradSync(grad_dtype=torch.bfloat16)
    assert sync.attach(m, transport=peer, world=2, rank=0)
    m.flat_grads.fill_(float("nan"))
    got = _train_once(m, xs[:1], ts[:1], tg[:1])
    assert pos[0] == 0 and len(calls) >= 4
    assert all(d == 1 and op == 1 and not inside for _, d, op, inside in calls)          # bf16, avg, staging slice
    want = ((gA.to(torch.bfloat16).float() + gB.to(torch.bfloat16).float()) * 0.5).to(torch.bfloat16).float()
    assert torch.equal(got, want)
    mean = (gA + gB) * 0.5
    rel = float((got - mean).norm() / mean.norm())
    print(f"bf16 wire vs fp32 mean: rel-L2 {rel:.2e}")
    # three bf16 roundings (both operands, the mean): every element within 2^-8 of the operand magnitudes (a cancelling pair may lose
    # all RELATIVE accuracy of a small mean, which is why the default wire format stays fp32)
    assert rel <= 1e-2 and bool(((got - mean).abs() <= 2.0 ** -8 * (gA.abs() + gB.abs()) + 1e-30).all())
    st = (C.c_int64 * 4)()
    assert _lib.lib().ldm_comm_stats(m._grad_comm, st) == 0
    assert st[0] == len(calls) and st[1] == 2 * total and _lib.lib().ldm_comm_is_rccl(m._grad_comm) == 0   # half of 4 * total bytes


def test_no_torch_collective_while_a_bucket_is_in_flight(cuda, monkeypatch):
    """Two communicators live in a training process (torch.distributed's group for scalars / barriers, the library's own for the
    buckets: trainer.py GradSync).  They must never have collectives in flight at the same time in rank-dependent order.  Driven
    through DiffusionTrainer with the fake second rank: train -> validate -> train.  Every torch.distributed collective the trainer
    issues (patched to record) finds ldm_model_grad_sync_pending == 0; inside backward (from the transport callback, while buckets
    are un-joined) GradSync refuses to issue one."""
    import torch.distributed as dist
    from ldm3d import _lib
    from ldm3d.inferer import LatentDiffusionInferer
    from ldm3d.networks import AutoencoderKL, DiffusionModelUNet
    from ldm3d.schedulers import DDPMScheduler
    from ldm3d.trainer import DiffusionTrainer
    from oracle import autoencoder as oa
    from oracle import unet as ou
    monkeypatch.setenv("LDM_GRAD_BUCKET_MB", "1")
    L = _lib.lib()
    vcfg = dict(cfgs.VAE_TINY, in_channels=1, out_channels=1, latent_channels=4)
    vae = AutoencoderKL(**vcfg)
    vae.load_state_dict(ou.init_state_dict(oa.ae_param_shapes(vcfg), 2))
    ucfg = dict(cfgs.UNET_TINY, in_channels=8)
    unet = DiffusionModelUNet(**ucfg)
    unet.load_state_dict(ou.init_state_dict(ou.unet_param_shapes(ucfg), 3, gain=0.5))
    vae, unet = vae.to(cuda).eval(), unet.to(cuda)
    tr = DiffusionTrainer(unet, vae, LatentDiffusionInferer(DDPMScheduler(**cfgs.SCHED), scale_factor=1.0), lr=1e-4)
    assert not tr.overlap                                       # no process group: nothing attached yet
    seen = {"in_backward": [], "refused": 0, "torch": []}

    def peer(buf, count, dtype, op, stream):                    # rank 1 contributes the same gradients: mean == own
        seen["in_backward"].append(L.ldm_model_grad_sync_pending(unet._h))
        try:
            tr.sync._quiescent()
        except RuntimeError:
            seen["refused"] += 1
        return 0
    tr.overlap = tr.sync.attach(unet, transport=peer, world=2, rank=0)
    assert tr.overlap
    tr.sync.on, tr.sync.world = True, 2                         # the trainer now believes it is rank 0 of 2

    def recorded(name):
        def fn(t, *a, **k):
            seen["torch"].append((name, L.ldm_model_grad_sync_pending(unet._h)))
            return None
        return fn
    monkeypatch.setattr(dist, "all_reduce", recorded("all_reduce"))
    monkeypatch.setattr(dist, "broadcast", recorded("broadcast"))
    monkeypatch.setattr(dist, "barrier", lambda *a, **k: seen["torch"].append(("barrier", L.ldm_model_grad_sync_pending(unet._h))))
    g = torch.Generator().manual_seed(3)
    images = torch.rand((1, 1, 32, 32, 32), generator=g).to(cuda)      # latent 8^3: the 3-level UNet needs a multiple of 4
    labels = torch.rand((1, 1, 32, 32, 32), generator=g).to(cuda)
    loss, skipped = tr.train_step(images, labels)
    assert not bool(skipped) and L.ldm_model_grad_sync_pending(unet._h) == 0
    val = tr.validate([{"image": images, "label": labels}], cuda)      # mean_scalar -> torch all_reduce
    tr.sync.barrier()
    loss2, skipped = tr.train_step(images, labels)
    torch.cuda.synchronize()
    assert not bool(skipped) and val == val
    assert len(seen["in_backward"]) >= 8 and max(seen["in_backward"]) >= 1       # buckets were in flight while backward ran
    assert seen["refused"] == sum(1 for n in seen["in_backward"] if n > 0)         # ... and a torch collective was refused there
    assert [n for n, _ in seen["torch"]] == ["all_reduce", "barrier"]
    assert all(p == 0 for _, p in seen["torch"])                                   # the trainer's own collectives: quiescent


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
    try:
        for p in procs:
            out, err = p.communicate(timeout=600)
            assert p.returncode == 0, err[-2000:]
            recs.append(json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1]))
    finally:                                                 # never leave a rank behind on the GPU (it would sit in the rendezvous)
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    print("two processes:", recs)
    assert sorted(r["rank"] for r in recs) == [0, 1]
    for r in recs:
        assert r["mean_exact"] and r["tiled"] and r["buckets"] >= 4 and r["checksums_equal"] and r["moved"], r
    assert recs[0]["checksum"] == recs[1]["checksum"]

