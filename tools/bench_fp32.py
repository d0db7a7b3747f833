"""fp32 precision mode of the headline step (UNet forward on 1x4x24^3 + DDPM step, graph replay), alone: ms per step and the share
of the fp32 MFMA peak.  Used for the sweeps of csrc/f32_path.h (LDM_F32_X3, LDM_F32_BN, LDM_F32_WGS, LDM_GN32_FOLD are read when the library plans).

    python tools/bench_fp32.py [--steps 30]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    args = ap.parse_args()
    import torch
    import bench
    dev = torch.device("cuda:0")
    unet = bench.make_unet(dev, seed=0)
    unet.set_precision("fp32")
    unet.enable_graph_replay(True)
    x = torch.randn((1, 4, 24, 24, 24), device=dev)
    t = torch.tensor([500.0], device=dev)
    with torch.no_grad():
        for _ in range(args.warmup):
            y = unet(x=x, timesteps=t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = unet(x=x, timesteps=t)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
    assert torch.isfinite(y).all()
    print(json.dumps({"fp32_forward_ms": dt * 1e3, "forwards_per_s": 1.0 / dt, "tflops": bench.UNET_STEP_GFLOP / (dt * 1e3),
                      "frac_of_157TF": bench.UNET_STEP_GFLOP / (dt * 1e3) / 157.3,
                      "knobs": {k: os.environ.get(k) for k in ("LDM_F32_X3", "LDM_F32_BN", "LDM_F32_WGS", "LDM_GN32_FOLD")}}))


if __name__ == "__main__":
    main()
