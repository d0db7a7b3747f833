"""Diagnostic: effective shader clock and K-loop time of conv3_halo_kernel from in-kernel stamps (LDM_CONV_DBG=512[+ablation bits])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ldm3d import _lib
spec = sys.argv[1] if len(sys.argv) > 1 else "256,256,24,24,24"     # cin,cout,D,H,W[,wgn]  (wgn 2 = 126 x 128 tiles, 0 = the planner's choice)
cin, cout, D, H, W = [int(a) for a in spec.split(",")[:5]]
wgn = int(spec.split(",")[5]) if spec.count(",") >= 5 else 2
dev = torch.device("cuda:0"); L = _lib.lib()
x = torch.randn((1, D, H, W, cin), device=dev).to(torch.bfloat16)
w = (torch.randn((27, cout, cin), device=dev) / (27 * cin) ** 0.5).to(torch.bfloat16)
b = torch.zeros((cout,), device=dev); out = torch.empty((1, D, H, W, cout), dtype=torch.bfloat16, device=dev)
for abl in [int(a) for a in os.environ.get('ABLS', '0,4,8,20,24,32').split(',')]:
    os.environ["LDM_CONV_DBG"] = str(512 + abl)
    scratch = torch.zeros((4 << 20,), dtype=torch.uint8, device=dev)    # the stamps land at the start of the scratch area
    for _ in range(5):
        _lib.check(L.ldm_op_conv3d(x.data_ptr(), cin, None, 0, w.data_ptr(), b.data_ptr(), None, 0, None, 0, None, None, None, 0,
                                   None, out.data_ptr(), None, 1, D, H, W, 3, 1, 1, 0, cout, cout, wgn, 1, scratch.data_ptr(), scratch.numel(),
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    st = scratch.view(torch.int64).view(-1, 8).cpu().double()
    st = st[st[:, 0] > 0]                                    # one row per workgroup that ran
    us = (st[:, 2] - st[:, 0]) * 0.01
    clk = (st[:, 3] - st[:, 1]) / us / 1e3
    print(f"{spec} abl {abl:2d} ({len(st)} workgroups): K loop {us.mean():6.2f} us (min {us.min():.2f} max {us.max():.2f}); shader clock {clk.mean():.3f} GHz "
          f"(min {clk.min():.3f} max {clk.max():.3f}); cycles/step {(st[:, 3] - st[:, 1]).mean() / (27 * cin // 64):.0f}")
