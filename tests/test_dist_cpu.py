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
