"""Worker of tests/test_gpu_comm.py::test_two_real_processes_share_the_gradient_exchange (not a test module).

Two of these run as separate processes on ONE GPU (RANK / WORLD_SIZE / MASTER_* in the environment, gloo for the wire): each is a real
rank of a world-size-2 data-parallel job whose gradient exchange runs through the LIBRARY's bucketed path (`GradSync.attach(transport=)`
-> `ldm_comm_init_custom`): the backward plan hands tail ranges of the flat gradient buffer to the transport while it is still running,
the transport averages them with the other PROCESS over gloo, the fused clip + Adam waits for the join.  Checked in every rank: the
averaged gradient equals (g_0 + g_1) / 2 bit for bit (both single-sample gradients are recomputed locally without any exchange), and
after two optimizer steps the parameters of the two processes have the same checksum.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    import torch.nn.functional as F
    import cfgs
    from ldm3d.networks import DiffusionModelUNet
    from ldm3d.optim import FlatAdam
    from ldm3d.trainer import GradSync
    from oracle import unet as ou
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))   # a lost peer must not park this rank for 30 min
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ["LDM_GRAD_BUCKET_MB"] = "1"
    cfg = cfgs.UNET_TINY
    sd = ou.init_state_dict(ou.unet_param_shapes(cfg), 3, gain=0.5)

    def fresh():
        m = DiffusionModelUNet(**cfg)
        m.load_state_dict(sd)
        m = m.to(dev).train()
        m.flatten_parameters()
        return m

    def grads(m, k):
        out = m(x=xs[k:k + 1], timesteps=ts[k:k + 1])
        F.mse_loss(out.float(), tg[k:k + 1]).backward()
        torch.cuda.synchronize()
        return m.flat_grads.clone()
    g = torch.Generator().manual_seed(7)
    xs = torch.randn((world, 4, 8, 8, 8), generator=g).to(dev)
    tg = torch.randn((world, 4, 8, 8, 8), generator=g).to(dev)
    ts = torch.tensor([17.0, 803.0, 411.0, 90.0][:world], device=dev)
    ref = fresh()
    g_all = [grads(ref, k) for k in range(world)]                      # no exchange: every rank's gradient, computed locally
    want = g_all[0].clone()
    for k in range(1, world):
        want += g_all[k]
    want *= 1.0 / world

    m = fresh()
    calls = []

    def transport(buf, count, dtype, op, stream):                      # what ncclAllReduce would get; the wire is gloo on host copies
        off = (buf - m.flat_grads.data_ptr()) // 4
        calls.append((off, count, dtype, op))
        st = torch.cuda.ExternalStream(stream)
        st.synchronize()                                               # the bucket's gradients are final on the comm stream
        host = m.flat_grads[off:off + count].cpu()
        dist.all_reduce(host)
        if op == 1:
            host *= 1.0 / world
        with torch.cuda.stream(st):
            m.flat_grads[off:off + count].copy_(host.to(dev))
        st.synchronize()
        return 0
    sync = GradSync()
    assert sync.attach(m, transport=transport, world=world, rank=rank) and sync.attached(m)
    m.flat_grads.fill_(float("nan"))
    got = grads(m, rank)
    exact = bool(torch.equal(got, want))
    total = m.flat_grads.numel()
    spans = sorted((o, o + c) for o, c, _, _ in calls)
    tiled = spans[0][0] == 0 and spans[-1][1] == total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    opt = FlatAdam(m, lr=1e-3, max_grad_norm=1.0)
    for _ in range(2):
        out = m(x=xs[rank:rank + 1], timesteps=ts[rank:rank + 1])
        F.mse_loss(out.float(), tg[rank:rank + 1]).backward()
        opt.step()
    torch.cuda.synchronize()
    cks = float(m.flat_params.double().sum().item())
    all_cks = [None] * world
    dist.all_gather_object(all_cks, cks)
    moved = bool((m.flat_params - ref.flat_params).abs().max().item() > 0)
    print(json.dumps({"rank": rank, "world": world, "mean_exact": exact, "buckets": len(calls), "tiled": bool(tiled),
                      "checksums_equal": all(c == all_cks[0] for c in all_cks), "moved": moved, "checksum": cks}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
