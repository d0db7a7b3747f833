"""BASELINE configs[1]: AutoencoderKL 3D encode -> decode on one 1x1x96^3 synthetic volume (bf16 compute), timing."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cfgs  # noqa: E402
from ldm3d.networks import AutoencoderKL  # noqa: E402


def main():
    dims = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "96,96,96").split(",")]
    dev = torch.device("cuda:0")
    m = AutoencoderKL(**cfgs.VAE_FULL)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) / p[0].numel() ** 0.5)
    m = m.to(dev).eval()
    zz, yy, xx = torch.meshgrid(*[torch.linspace(-1, 1, d) for d in dims], indexing="ij")
    img = sum(torch.exp(-((zz - c) ** 2 + (yy + c) ** 2 + (xx - 0.3 * c) ** 2) / 0.1) for c in (-0.5, 0.0, 0.4)).clamp(0, 1)
    x = img[None, None].to(dev)
    with torch.no_grad():
        for _ in range(2):
            mu, sig = m.encode(x)
            rec = m.decode(mu)
        torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            mu, sig = m.encode(x)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            rec = m.decode(mu)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    vox = dims[0] * dims[1] * dims[2] / 96 ** 3
    enc_ms, dec_ms = (t1 - t0) / n * 1e3, (t2 - t1) / n * 1e3
    print(f"VAE {dims}: encode {enc_ms:.2f} ms ({1.342 * vox / enc_ms * 1e3:.0f} TFLOP/s), decode {dec_ms:.2f} ms "
          f"({2.867 * vox / dec_ms * 1e3:.0f} TFLOP/s), finite {bool(torch.isfinite(rec).all())}")


if __name__ == "__main__":
    main()
