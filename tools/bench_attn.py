"""Micro-benchmark of attn_fwd_kernel (ldm_op_attention): shapes B,N,C."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldm3d import _lib
for spec in sys.argv[1:] or ["1,1728,256", "1,216,512", "1,5544,256"]:
    B, N, C = [int(a) for a in spec.split(",")]
    dev = torch.device("cuda:0"); L = _lib.lib()
    qkv = torch.randn((B, N, 3 * C), device=dev).to(torch.bfloat16)
    out = torch.empty((B, N, C), dtype=torch.bfloat16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.check(L.ldm_op_attention(qkv.data_ptr(), out.data_ptr(), B, N, C, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        _lib.check(L.ldm_op_attention(qkv.data_ptr(), out.data_ptr(), B, N, C, st))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    gf = 4.0 * B * N * N * C / 1e9
    print(f"{spec:16s} {us:8.1f} us  {gf / us * 1e3:7.1f} TFLOP/s ({gf:.2f} GF)")
