"""Generates the committed golden vectors with the CPU oracle (run in the authoring container, no GPU needed):

    python tests/golden/make_golden.py

unet_full_24.pt : benchmark UNet (tests/cfgs.py UNET_FULL) on the headline shape 1x4x24^3, t = 500.
                  Weights and input are regenerated from the stored seeds (oracle.unet.init_state_dict /
                  torch.Generator on CPU), only eps_hat is stored (bf16-emulating and pure fp32 oracle), as fp16-safe
                  fp32 tensors (2 x 221 KB).
sched_tables.pt : DDPM/DDIM known values of the (T=1000, scaled_linear_beta, 0.0015->0.0195) schedule.
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import cfgs  # noqa: E402
from oracle import unet as ou  # noqa: E402
from oracle.schedulers import OracleDDPM  # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    wseed, iseed, t = 0, 0, 500.0
    sd = ou.init_state_dict(ou.unet_param_shapes(cfgs.UNET_FULL), wseed)
    g = torch.Generator().manual_seed(iseed)
    x = torch.randn((1, 4, 24, 24, 24), generator=g)
    t0 = time.time()
    e_bf = ou.unet_forward(sd, cfgs.UNET_FULL, x, torch.tensor([t]), emulate_bf16=True)
    t1 = time.time()
    e_32 = ou.unet_forward(sd, cfgs.UNET_FULL, x, torch.tensor([t]), emulate_bf16=False)
    t2 = time.time()
    print(f"oracle 24^3: bf16-emulated {t1 - t0:.1f}s, fp32 {t2 - t1:.1f}s, rel-L2 between them "
          f"{float((e_bf - e_32).norm() / e_32.norm()):.3e}")
    torch.save(dict(weight_seed=wseed, input_seed=iseed, t=t, eps_bf16_oracle=e_bf, eps_fp32_oracle=e_32,
                    torch_version=torch.__version__), os.path.join(HERE, "unet_full_24.pt"))
    s = OracleDDPM(**cfgs.SCHED)
    torch.save(dict(betas=s.betas, alphas_cumprod=s.alphas_cumprod), os.path.join(HERE, "sched_tables.pt"))


if __name__ == "__main__":
    main()
