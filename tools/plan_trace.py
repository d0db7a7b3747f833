#!/usr/bin/env python3
"""Per-op timeline of the UNet forward plan (LDM_PLAN_TRACE: a HIP event before every op of the launch plan).

    LDM_PLAN_TRACE=/tmp/trace.csv python tools/plan_trace.py [--steps 30]

Prints the average duration of every op of the headline step (1x4x24^3) in plan order and the totals per op kind /
conv configuration.  Event-to-event time includes the launch gap, so the sum is a little above the graph-replayed step."""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KINDS = ("PACK CONV FINALIZE GN_STATS GN_FINALIZE GN_PREP GN_APPLY ATTN SINUSOID GEMV VAE_HEADS GN_FUSED WT WT_BATCH WGRAD EXPORT "
         "EXPORT_BATCH COLSUM GNB ATTN_BWD ADD SUMPOOL LIN_DX LIN_DW VAE_HEADS_BWD GEMM_LIGHT COLSUM_BATCH IM2COL "
         "PACK32 CONV32 FIN32 GN_STATS32 GN_APPLY32 ATTN32 GEMV32 TAP BUCKET BUCKET_JOIN UPS_SPLIT32 CONV_THIN FIN_GN GEMM_LIGHT32 CONV_BLOCK TEMB_ROW").split()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--what", default="unet", choices=["unet", "enc", "dec"], help="plan: UNet forward at 24^3, or the VAE encode / decode at 96^3")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    args = ap.parse_args()
    path = os.environ.get("LDM_PLAN_TRACE")
    if not path:
        sys.exit("set LDM_PLAN_TRACE=<file>")
    if os.path.exists(path):
        os.remove(path)
    import torch
    import bench
    dev = torch.device("cuda:0")
    if args.what == "unet":
        unet = bench.make_unet(dev, seed=0)
        if args.precision == "fp32":
            unet.set_precision("fp32")
        x = torch.randn((1, 4, 24, 24, 24), device=dev)
        t = torch.tensor([500.0], device=dev)
        run = lambda: unet(x=x, timesteps=t)
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import cfgs
        from ldm3d.networks import AutoencoderKL
        vae = AutoencoderKL(**cfgs.VAE_FULL)
        with torch.no_grad():
            for p in vae.parameters():
                if p.dim() > 1:
                    p.normal_(0.0, 0.02)
        vae = vae.to(dev).eval()
        img, lat = torch.rand((1, 1, 96, 96, 96), device=dev), torch.randn((1, 4, 24, 24, 24), device=dev)
        run = (lambda: vae.encode(img)) if args.what == "enc" else (lambda: vae.decode(lat))
    with torch.no_grad():
        for _ in range(args.warmup + args.steps):
            run()
    torch.cuda.synchronize()
    rows = [ln.rstrip("\n").split(",", 4) for ln in open(path)]
    nops = int(rows[0][0])
    rows = rows[args.warmup * nops:]
    acc = collections.defaultdict(list)
    desc = {}
    for n, oi, kind, us, d in rows:
        acc[int(oi)].append(float(us))
        desc[int(oi)] = (KINDS[int(kind)], d)
    total = 0.0
    groups = collections.defaultdict(lambda: [0, 0.0])
    for oi in sorted(acc):
        us = sum(acc[oi]) / len(acc[oi])
        total += us
        k, d = desc[oi]
        print(f"{oi:4d} {k:12s} {us:8.1f} us  {d}")
        key = k
        if k == "CONV":
            f = dict(kv.split("=") for kv in d.split() if "=" in kv)
            key = f"CONV M={f['M']} k={f['k']} halo={f['halo']} cfg={f['cfg']} splitk={f['splitk']}"
        groups[key][0] += 1
        groups[key][1] += us
    print(f"\ntotal {total:.1f} us over {len(acc)} ops")
    for key, (n, us) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        print(f"{us:9.1f} us  {100 * us / total:5.1f}%  x{n:3d}  {key}")


if __name__ == "__main__":
    main()
