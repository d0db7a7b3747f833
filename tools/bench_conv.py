"""Micro-benchmark of the implicit-GEMM conv kernel through the operator-level ABI (ldm_op_conv3d).
usage: python tools/bench_conv.py [shape ...]   shapes: cin,cout,D,H,W[,wgn[,splitk]]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldm3d import _lib  # noqa: E402

DEFAULT = ["256,256,24,24,24", "512,256,24,24,24", "256,256,12,12,12", "512,512,12,12,12", "1024,512,6,6,6", "512,512,6,6,6",
           "64,64,96,96,96", "128,128,48,48,48"]


def run(spec, iters=20):
    v = [int(a) for a in spec.split(",")]
    cin, cout, D, H, W = v[:5]
    wgn = v[5] if len(v) > 5 else 0
    splitk = v[6] if len(v) > 6 else 0
    dev = torch.device("cuda:0")
    L = _lib.lib()
    cout_pad = (cout + 63) // 64 * 64
    x = torch.randn((1, D, H, W, cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((27, cout_pad, cin), device=dev) / (27 * cin) ** 0.5).to(torch.bfloat16)
    b = torch.zeros((cout_pad,), device=dev)
    out = torch.empty((1, D, H, W, cout), dtype=torch.bfloat16, device=dev)
    M = D * H * W
    scratch = torch.empty((64 * M * cout_pad * 4 if M < 4096 else 4 * M * cout_pad * 4,), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def call():
        _lib.check(L.ldm_op_conv3d(x.data_ptr(), cin, None, 0, w.data_ptr(), b.data_ptr(), None, 0, None, 0, None, None, None, 0,
                                   None, out.data_ptr(), None, 1, D, H, W, 3, 1, 1, 0, cout, cout_pad, wgn, splitk,
                                   scratch.data_ptr(), scratch.numel(), st))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    gf = 2.0 * M * cin * cout * 27 / 1e9
    print(f"{spec:28s} {us:9.1f} us  {gf / us * 1e-3 * 1e3:8.1f} TFLOP/s  ({gf:.1f} GF)", flush=True)


if __name__ == "__main__":
    for s in (sys.argv[1:] or DEFAULT):
        run(s)
