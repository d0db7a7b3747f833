"""Split-K sweep of the low-resolution 3^3 convs of the benchmark UNet through the operator-level ABI (conv + split-K finalize, as the
plans launch them), with the weights COLD: the whole UNet's 382 MB of weights stream from HBM once per step, so a loop over one
weight buffer (resident in the 256 MiB Infinity Cache) would flatter the weight-bound 6^3 shapes; the loop below cycles through enough
copies of the weights to exceed it.  Prints, per shape, the time of every candidate split next to the planner's own choice.

    python tools/sweep_splitk.py [cin,cout,D[,wgn]] ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ldm3d import _lib  # noqa: E402

DEFAULT = ["256,256,12", "512,256,12", "768,256,12", "256,512,6", "512,512,6", "768,512,6", "1024,512,6"]
SPLITS = [0, 1, 2, 3, 4, 6, 8, 9, 12, 16, 18, 24, 27, 32, 48]


def run(spec, iters=24):
    v = [int(a) for a in spec.split(",")]
    cin, cout, D = v[:3]
    wgn = v[3] if len(v) > 3 else 2
    dev = torch.device("cuda:0")
    L = _lib.lib()
    cout_pad = (cout + 63) // 64 * 64
    M = D * D * D
    wbytes = 27 * cout_pad * cin * 2
    ncopy = max(2, (320 << 20) // wbytes + 1)
    x = torch.randn((1, D, D, D, cin), device=dev).to(torch.bfloat16)
    ws = [(torch.randn((27, cout_pad, cin), device=dev) / (27 * cin) ** 0.5).to(torch.bfloat16) for _ in range(ncopy)]
    b = torch.zeros((cout_pad,), device=dev)
    out = torch.empty((1, D, D, D, cout), dtype=torch.bfloat16, device=dev)
    scratch = torch.empty((64 * M * cout_pad * 4,), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for sk in SPLITS:
        def call(i):
            return L.ldm_op_conv3d(x.data_ptr(), cin, None, 0, ws[i % ncopy].data_ptr(), b.data_ptr(), None, 0, None, 0, None, None, None, 0,
                                   None, out.data_ptr(), None, 1, D, D, D, 3, 1, 1, 0, cout, cout_pad, wgn if sk else 0, sk,
                                   scratch.data_ptr(), scratch.numel(), st)
        if call(0) != 0:
            continue
        for i in range(3):
            call(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        res.append((sk, e0.elapsed_time(e1) * 1e3 / iters))
    gf = 2.0 * M * cin * cout * 27 / 1e9
    best = min(res[1:], key=lambda r: r[1])
    print(f"{spec:16s} {gf:6.2f} GF  planner {res[0][1]:6.1f} us | best splitk {best[0]:2d}: {best[1]:6.1f} us | " +
          " ".join(f"{sk}:{us:.1f}" for sk, us in res[1:]), flush=True)


if __name__ == "__main__":
    for s in (sys.argv[1:] or DEFAULT):
        run(s)
