#!/usr/bin/env python3
"""conv3_block_kernel alone (csrc/conv_block.h): the AutoencoderKL's full-resolution 3^3 conv, Cin -> 64 channels, against the 254 x 64 halo tile.

    python tools/bench_conv_block.py [D,H,W] [cin] [th] [cout = 64 | 128]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ldm3d import _lib  # noqa: E402


def main():
    dims = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "96,96,96").split(",")]
    cin = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    th = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    cout = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    L = _lib.lib()
    dev = torch.device("cuda:0")
    m = dims[0] * dims[1] * dims[2]
    x = torch.randn((1, *dims, cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((27, cout, cin), device=dev) / (27 * cin) ** 0.5).to(torch.bfloat16)
    b = torch.randn((cout,), device=dev)
    out = torch.empty((1, *dims, cout), dtype=torch.bfloat16, device=dev)
    rows = L.ldm_op_conv3d_block_stats_rows(*dims, th)
    stats = torch.empty((rows, cout, 2), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    gflop = 2.0 * m * cout * cin * 27 / 1e9

    def block():
        if cout == 128:
            _lib.check(L.ldm_op_conv3d_block128(x.data_ptr(), cin, w.data_ptr(), b.data_ptr(), None, 0, None, out.data_ptr(), stats.data_ptr(), 1, *dims, st))
            return
        _lib.check(L.ldm_op_conv3d_block(x.data_ptr(), cin, w.data_ptr(), b.data_ptr(), None, 0, None, out.data_ptr(), stats.data_ptr(), 1, *dims, th, st))

    scratch = torch.empty((64 * m * cout * 4 + 256,), dtype=torch.uint8, device=dev)

    def halo():
        _lib.check(L.ldm_op_conv3d(x.data_ptr(), cin, None, 0, w.data_ptr(), b.data_ptr(), None, 0, None, 0, None, None, None, cout, None, out.data_ptr(), None,
                                   1, *dims, 3, 1, 1, 0, cout, cout, 0, 0, scratch.data_ptr(), scratch.numel(), st))

    for name, fn in (("conv3_block_kernel", block), ("planner's choice through ldm_op_conv3d", halo)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"{name}: {us:.1f} us  {gflop / us:.3f} PFLOP/s  ({gflop / us / 2.5:.3f} of the bf16 MFMA peak)")


if __name__ == "__main__":
    main()
