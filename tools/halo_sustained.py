"""conv3_halo_kernel on the three 24^3 shapes of the headline step under SUSTAINED load: what splits the K step's
0.37 us against its 0.27 us of MFMA issue into clock and stall.

    python tools/halo_sustained.py [--seconds 2.5] [--shapes plain,wide,skip] [--stamps]

For each shape the operator is launched back to back on random data for >= `seconds` (DVFS settles in the first few hundred ms:
MI355X_MICROARCH.md, DVFS items 5 - 7); reported per shape: launch time from the sustained rate, and -- with --stamps, which needs an
experiments build of the library (make EXTRA=-DLDM_EXPERIMENTS OUT=../libldm3d_exp.so; LDM3D_LIB=.../libldm3d_exp.so) -- from the
in-kernel stamps of the LAST launch (s_memrealtime at 100 MHz and s_memtime in shader clocks around the K loop of every workgroup): K loop
time, shader clock held, cycles per K step.  The MFMA-busy counters of the same loop come from tools/halo_sustained_pmc.sh.
shapes: plain = 256 -> 256, wide = 512 -> 256, skip = 256 -> 256 with the fused 1x1 skip over cat(256, 256) (up_blocks.2 conv2)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {"plain": (256, 256, 0), "wide": (512, 256, 0), "skip": (256, 256, 256)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--shapes", default="plain,wide,skip")
    ap.add_argument("--stamps", action="store_true")
    ap.add_argument("--iters", type=int, default=0, help="fixed launch count instead of a duration (PMC passes)")
    ap.add_argument("--dbg", type=int, default=0, help="extra LDM_CONV_DBG bits (experiments build; results are wrong): 1 = voxel copies from 1024 "
                    "L2-resident rows, 2 = weight copies out of range (no bytes move), 4 / 8 / 16 / 32 / 64 = conv_halo.h's compile-time ablations")
    args = ap.parse_args()
    if args.stamps or args.dbg:
        os.environ["LDM_CONV_DBG"] = str((512 if args.stamps else 0) | args.dbg)
    import torch
    from ldm3d import _lib
    dev = torch.device("cuda:0")
    L = _lib.lib()
    D = H = W = 24
    M = D * H * W
    st = torch.cuda.current_stream().cuda_stream
    for name in args.shapes.split(","):
        cin, cout, cskip = SHAPES[name]
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn((1, D, H, W, cin), device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn((27, cout, cin), device=dev, generator=g) / (27 * cin) ** 0.5).to(torch.bfloat16)
        b = torch.zeros((cout,), device=dev)
        out = torch.empty((1, D, H, W, cout), dtype=torch.bfloat16, device=dev)
        xa = xb = w1 = b1 = None
        if cskip:
            xa = torch.randn((1, D, H, W, cskip), device=dev, generator=g).to(torch.bfloat16)
            xb = torch.randn((1, D, H, W, cskip), device=dev, generator=g).to(torch.bfloat16)
            w1 = (torch.randn((cout, 2 * cskip), device=dev, generator=g) / (2 * cskip) ** 0.5).to(torch.bfloat16)
            b1 = torch.zeros((cout,), device=dev)
        scratch = torch.zeros((4 << 20,), dtype=torch.uint8, device=dev)

        def call():
            _lib.check(L.ldm_op_conv3d(x.data_ptr(), cin, None, 0, w.data_ptr(), b.data_ptr(),
                                       _lib.ptr(xa), cskip, _lib.ptr(xb), cskip, _lib.ptr(w1), _lib.ptr(b1), None, 0,
                                       None, out.data_ptr(), None, 1, D, H, W, 3, 1, 1, 0, cout, cout, 2, 1, scratch.data_ptr(), scratch.numel(), st))
        for _ in range(20):
            call()
        torch.cuda.synchronize()
        n = 0
        t0 = time.perf_counter()
        if args.iters:
            for _ in range(args.iters):
                call()
            n = args.iters
        else:
            while time.perf_counter() - t0 < args.seconds:
                for _ in range(500):
                    call()
                n += 500
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gf = 2.0 * M * cout * (27 * cin + 2 * cskip) / 1e9
        us = dt / n * 1e6
        line = (f"{name:5s} {cin}->{cout}{' +skip ' + str(2 * cskip) if cskip else ''} @24^3: {n} launches in {dt:.2f} s = {us:.2f} us per launch, "
                f"{gf / us:.3f} PFLOP/s = {gf / us / 2.5:.3f} of the bf16 MFMA peak")
        if args.stamps:
            s8 = scratch.view(torch.int64).view(-1, 8).cpu().double()
            s8 = s8[s8[:, 0] > 0]
            if len(s8):
                kus = (s8[:, 2] - s8[:, 0]) * 0.01
                clk = (s8[:, 3] - s8[:, 1]) / kus / 1e3
                steps = 27 * cin // 64 + (2 * cskip) // 64
                line += (f" | last launch, {len(s8)} workgroups: K loop {kus.mean():.2f} us (min {kus.min():.2f} max {kus.max():.2f}), shader clock "
                         f"{clk.mean():.3f} GHz (min {clk.min():.3f} max {clk.max():.3f}), {(s8[:, 3] - s8[:, 1]).mean() / steps:.0f} cycles / K step "
                         f"({steps} steps; 512 = its MFMA issue), K step {kus.mean() / steps * 1e3:.0f} ns")
            else:
                line += " | no stamps (product library: build with EXTRA=-DLDM_EXPERIMENTS)"
        print(line, flush=True)


if __name__ == "__main__":
    main()
