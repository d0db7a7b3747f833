"""Diagnostic: phase timeline of conv_igemm_kernel from in-kernel s_memrealtime stamps (LDM_CONV_DBG=512)."""
import os, sys
os.environ["LDM_CONV_DBG"] = "512"
os.environ["LDM_CONV_HALO"] = "0"        # this tool reads conv_igemm_kernel's stamps (tools/stamp_halo.py has the halo kernel's)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldm3d import _lib
for spec in sys.argv[1:] or ["64,256,24,24,24", "256,256,24,24,24"]:
    cin, cout, D, H, W = [int(a) for a in spec.split(",")]
    dev = torch.device("cuda:0"); L = _lib.lib()
    x = torch.randn((1, D, H, W, cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((27, cout, cin), device=dev) / (27 * cin) ** 0.5).to(torch.bfloat16)
    b = torch.zeros((cout,), device=dev); out = torch.empty((1, D, H, W, cout), dtype=torch.bfloat16, device=dev)
    scratch = torch.zeros((1 << 20,), dtype=torch.uint8, device=dev)
    for _ in range(5):
        _lib.check(L.ldm_op_conv3d(x.data_ptr(), cin, None, 0, w.data_ptr(), b.data_ptr(), None, 0, None, 0, None, None, None, 0,
                                   None, out.data_ptr(), None, 1, D, H, W, 3, 1, 1, 0, cout, cout, 2, 1, scratch.data_ptr(), scratch.numel(),
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    nwg = ((D * H * W + 127) // 128) * (cout // 128)
    st = scratch[: nwg * 64].view(torch.int64).view(nwg, 8).cpu().double()
    t0 = st[:, 0].min()
    rel = (st[:, :6] - t0) * 0.01          # 100 MHz ticks -> us
    names = ["entry", "setup done", "prologue done", "loop done", "reduced", "end"]
    print(spec, "nwg", nwg)
    for i, n in enumerate(names):
        print(f"  {n:14s} mean {rel[:, i].mean():7.2f} us  min {rel[:, i].min():7.2f}  max {rel[:, i].max():7.2f}")
    if int(os.environ.get("STEPS", "0")):
        os.environ["LDM_CONV_DBG"] = "1536"
        scratch.zero_()
        for _ in range(3):
            _lib.check(L.ldm_op_conv3d(x.data_ptr(), cin, None, 0, w.data_ptr(), b.data_ptr(), None, 0, None, 0, None, None, None, 0,
                                       None, out.data_ptr(), None, 1, D, H, W, 3, 1, 1, 0, cout, cout, 2, 1, scratch.data_ptr(), scratch.numel(),
                                       torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        os.environ["LDM_CONV_DBG"] = "512"
        ns = 27 * cin // 64
        base = nwg * 64
        ts = scratch[base: base + nwg * 4096].view(torch.int64).view(nwg, 512)[:, :ns].cpu().double() * 0.01
        d = ts[:, 1:] - ts[:, :-1]
        dm = d.mean(0)
        print("  per-step us (mean over WGs):", " ".join(f"{v:.2f}" for v in dm.tolist()))
