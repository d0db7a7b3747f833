"""N > 1 path on CPU (gloo, world_size 2): the sampling path shards by independent chains, so the only collectives
are the timing barrier and the MAX over ranks that bench.py uses; setup_ddp mirrors 3d_ldm/utils.py:55-63."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from ldm3d import parallel
    parallel.setup_ddp(rank, world, backend="gloo")
    try:
        # each rank owns the chains i with i % world == rank: a partition with no overlap and no gap
        mine = parallel.shard_indices(10, rank, world)
        counts = torch.zeros(10)
        counts[mine] = 1
        dist.all_reduce(counts)
        # bench timing contract: MAX over ranks of the local elapsed time
        t = parallel.max_over_ranks(float(rank + 1))
        avg = parallel.all_reduce_mean(torch.tensor([float(rank)]))
        q.put((rank, mine, counts.tolist(), t, float(avg)))
    finally:
        parallel.cleanup_ddp()


def test_two_rank_sharding_and_timing_reduction():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert res[0][1] == [0, 2, 4, 6, 8] and res[1][1] == [1, 3, 5, 7, 9]
    assert res[0][2] == [1.0] * 10
    assert res[0][3] == res[1][3] == 2.0
    assert res[0][4] == res[1][4] == 0.5


# ------------------------------------------------------------------------------------------------ data-parallel training (row a7)
def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from ldm3d import parallel
    from ldm3d.trainer import GradSync
    parallel.setup_ddp(rank, world, backend="gloo")
    try:
        sync = GradSync(chunk_elems=1000)                      # force several chunks: 4097 elements -> 5 collectives
        torch.manual_seed(100 + rank)                          # ranks start from DIFFERENT parameters ...
        flat = torch.randn(4097)
        sync.broadcast(flat, 0)                                # ... and leave the wrap with rank 0's
        w = flat[:12].view(3, 4).clone().requires_grad_(True)
        g = torch.Generator().manual_seed(7)
        x, y = torch.randn((8, 4), generator=g), torch.randn((8, 3), generator=g)
        xs, ys = x[rank::world], y[rank::world]                # DistributedSampler-style shard of the global batch
        loss = torch.nn.functional.mse_loss(xs @ w.t(), ys)
        loss.backward()
        grads = torch.zeros(4097)
        grads[:12] = w.grad.reshape(-1)
        grads[12:] = float(rank + 1)
        sync.mean_(grads)
        low = GradSync(chunk_elems=1000, grad_dtype=torch.bfloat16)     # opt-in bf16 wire format: same mean to bf16 precision
        g16 = torch.zeros(4097)
        g16[:12] = w.grad.reshape(-1)
        g16[12:] = float(rank + 1)
        low.mean_(g16)
        assert torch.allclose(g16, grads, rtol=2e-2, atol=1e-3) and g16.dtype == torch.float32
        nan_flag = sync.any(torch.tensor(1.0 if rank == 1 else 0.0))     # one rank saw a NaN loss -> everyone skips
        sf = sync.mean_scalar(torch.tensor(2.0 + rank))
        q.put((rank, flat[:12].tolist(), grads[:12].tolist(), float(grads[-1]), float(nan_flag), float(sf)))
    finally:
        parallel.cleanup_ddp()


def test_two_rank_gradient_mean_equals_full_batch_gradient():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert res[0][1] == res[1][1]                              # broadcast parameters agree
    assert res[0][2] == res[1][2]                              # reduced gradients agree bit for bit
    # ... and equal the single-process gradient of the whole batch (equal shard sizes: mean of means)
    w = torch.tensor(res[0][1]).view(3, 4).requires_grad_(True)
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn((8, 4), generator=g), torch.randn((8, 3), generator=g)
    torch.nn.functional.mse_loss(x @ w.t(), y).backward()
    assert torch.allclose(torch.tensor(res[0][2]), w.grad.reshape(-1), rtol=1e-5, atol=1e-7)
    assert res[0][3] == res[1][3] == 1.5                       # chunk tail reduced too
    assert res[0][4] == res[1][4] == 1.0                       # agreed NaN-skip
    assert res[0][5] == res[1][5] == 2.5                       # scale-factor average (train_diffusion.py:121-123)
