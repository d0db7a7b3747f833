"""Micro-benchmark of the conv weight-gradient kernels through ldm_op_conv3d_wgrad.
usage: python tools/bench_wgrad.py [cin,cout,D,H,W[,ksplit] ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldm3d import _lib
for spec in sys.argv[1:] or ["256,256,24,24,24,5", "512,256,24,24,24,3", "256,256,12,12,12,2", "512,512,6,6,6,1"]:
    v = [int(a) for a in spec.split(",")]
    cin, cout, D, H, W = v[:5]
    ks = v[5] if len(v) > 5 else 1
    dev = torch.device("cuda:0"); L = _lib.lib()
    x = torch.randn((1, D, H, W, cin), device=dev).to(torch.bfloat16)
    dy = torch.randn((1, D, H, W, cout), device=dev).to(torch.bfloat16)
    dw = torch.empty((ks, 27, cout, cin), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def call():
        _lib.check(L.ldm_op_conv3d_wgrad(dy.data_ptr(), cout, x.data_ptr(), cin, dw.data_ptr(), cout, cin, 1, D, H, W, 3, 1, 1, 0, ks, st))
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    gf = 2.0 * D * H * W * cin * cout * 27 / 1e9
    print(f"{spec:28s} {us:9.1f} us  {gf / us:8.1f} GFLOP/ms = TFLOP/s {gf / us / 1e3 * 1e3:7.1f}", flush=True)
