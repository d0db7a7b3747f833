"""Diagnostic: phase timeline of conv_igemm_kernel for 1x1 convolutions (light GEMMs) from in-kernel stamps (LDM_CONV_DBG=512).
spec = cin,cout,rows"""
import os, sys
os.environ["LDM_CONV_DBG"] = "512"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldm3d import _lib
for spec in sys.argv[1:] or ["256,768,1728", "256,256,1728", "512,1536,216", "512,512,216"]:
    cin, cout, M = [int(a) for a in spec.split(",")]
    dev = torch.device("cuda:0"); L = _lib.lib()
    x = torch.randn((1, 1, 1, M, cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((1, cout, cin), device=dev) / cin ** 0.5).to(torch.bfloat16)
    b = torch.zeros((cout,), device=dev); out = torch.empty((1, 1, 1, M, cout), dtype=torch.bfloat16, device=dev)
    scratch = torch.zeros((1 << 20,), dtype=torch.uint8, device=dev)
    st_ = torch.cuda.current_stream().cuda_stream
    def run(wgn=2, splitk=1):       # forced tile = conv_igemm_kernel; (0, 0) = automatic = gemm_light_kernel where it applies
        _lib.check(L.ldm_op_conv3d(x.data_ptr(), cin, None, 0, w.data_ptr(), b.data_ptr(), None, 0, None, 0, None, None, None, 0,
                                   None, out.data_ptr(), None, 1, 1, 1, M, 1, 1, 0, 0, cout, cout, wgn, splitk, scratch.data_ptr(), scratch.numel(), st_))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    nwg = ((M + 127) // 128) * (cout // 128)
    st = scratch[: nwg * 64].view(torch.int64).view(nwg, 8).cpu().double()
    t0 = st[:, 0].min()
    rel = (st[:, :6] - t0) * 0.01          # 100 MHz ticks -> us
    names = ["entry", "setup done", "prologue done", "loop done", "reduced", "end"]
    print(spec, "nwg", nwg)
    for i, n in enumerate(names):
        print(f"  {n:14s} mean {rel[:, i].mean():7.2f} us  min {rel[:, i].min():7.2f}  max {rel[:, i].max():7.2f}")
    os.environ["LDM_CONV_DBG"] = "0"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        run()
    e0.record()
    for _ in range(50):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"  back-to-back launches: {e0.elapsed_time(e1) * 20:.2f} us each (conv_igemm_kernel)")
    for _ in range(5):
        run(0, 0)
    e0.record()
    for _ in range(50):
        run(0, 0)
    e1.record(); torch.cuda.synchronize()
    print(f"  back-to-back launches: {e0.elapsed_time(e1) * 20:.2f} us each (automatic: gemm_light_kernel)")
    os.environ["LDM_CONV_DBG"] = "512"
