"""Generates the committed golden vectors with the CPU oracle (run in the authoring container, no GPU needed):

    python tests/golden/make_golden.py

unet_full_24.pt : benchmark UNet (tests/cfgs.py UNET_FULL) on the headline shape 1x4x24^3, t = 500.
                  Weights and input are regenerated from the stored seeds (oracle.unet.init_state_dict /
                  torch.Generator on CPU), only eps_hat is stored (bf16-emulating and pure fp32 oracle), as fp16-safe
                  fp32 tensors (2 x 221 KB).
vae_full_96.pt  : AutoencoderKL (VAE_FULL) on the BASELINE configs[1] volume 1x1x96^3: mu and the stride-4 sub-lattice of decode(mu),
                  fp32 and bf16-emulating oracle (python tests/golden/make_golden.py vae regenerates only this one).
sched_tables.pt : DDPM/DDIM known values of the (T=1000, scaled_linear_beta, 0.0015->0.0195) schedule.
train_step_tiny.pt : one training step of tests/cfgs.py UNET_TINY (train_diffusion.py:197-219 in miniature): seeds, the MSE
                  loss, per-parameter gradient norms and projections on seeded +-1 directions (fp32 and bf16-emulating
                  oracle, torch autograd), and the parameter checksum after one Adam step with clip 1.0.
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


def train_case():
    """Seeded (weights, x, t, target) of the golden training step; shared with tests/test_gpu_train.py."""
    cfg = cfgs.UNET_TINY
    sd = ou.init_state_dict(ou.unet_param_shapes(cfg), 21, gain=0.5)
    g = torch.Generator().manual_seed(22)
    x = torch.randn((2, 4, 8, 8, 8), generator=g)
    target = torch.randn((2, 4, 8, 8, 8), generator=g)
    t = torch.tensor([123.0, 877.0])
    return cfg, sd, x, t, target


def directions(sd, k=4, seed=23):
    g = torch.Generator().manual_seed(seed)
    return {n: (torch.randint(0, 2, (k, v.numel()), generator=g).float() * 2 - 1) for n, v in sd.items()}


def train_golden():
    import torch.nn.functional as F
    cfg, sd, x, t, target = train_case()
    dirs = directions(sd)
    out = dict(torch_version=str(torch.__version__))
    for tag, bf in (("fp32", False), ("bf16", True)):
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        loss = F.mse_loss(ou.unet_forward(leaves, cfg, x, t, emulate_bf16=bf), target)
        loss.backward()
        out[f"loss_{tag}"] = float(loss.detach())
        out[f"grad_norm_{tag}"] = {k: float(v.grad.norm()) for k, v in leaves.items()}
        out[f"grad_proj_{tag}"] = {k: (dirs[k] @ v.grad.reshape(-1)).tolist() for k, v in leaves.items()}
        if not bf:                                   # one Adam step (lr 1e-3, clip 1.0) on the fp32 gradients
            params = [v for v in leaves.values()]
            opt = torch.optim.Adam(params, lr=1e-3)
            out["total_grad_norm_fp32"] = float(torch.nn.utils.clip_grad_norm_(params, 1.0))
            opt.step()
            out["param_sum_after_adam_fp32"] = float(sum(v.detach().double().sum() for v in params))
            out["param_abs_sum_after_adam_fp32"] = float(sum(v.detach().double().abs().sum() for v in params))
    torch.save(out, os.path.join(HERE, "train_step_tiny.pt"))
    print("train golden: loss fp32 %.6f bf16 %.6f, |g| %.4f" % (out["loss_fp32"], out["loss_bf16"], out["total_grad_norm_fp32"]))


def vae_case(seed=0):
    """BASELINE configs[1] input: a smooth synthetic "MRI" (three Gaussian blobs clipped to [0, 1]) on 1x1x96^3."""
    zz, yy, xx = torch.meshgrid(*[torch.linspace(-1, 1, 96) for _ in range(3)], indexing="ij")
    img = sum(torch.exp(-((zz - c) ** 2 + (yy + c) ** 2 + (xx - 0.3 * c) ** 2) / 0.1) for c in (-0.5, 0.0, 0.4)).clamp(0, 1)
    return img[None, None].contiguous()


def vae_golden():
    """vae_full_96.pt: AutoencoderKL (tests/cfgs.py VAE_FULL) encode (mu) and decode(mu) of the configs[1] volume, fp32 and
    bf16-emulating oracle.  mu is stored whole (4 x 24^3), the 96^3 reconstructions on the stride-4 sub-lattice (24^3 samples)
    plus their mean / mean square, which keeps the fixture at ~0.7 MB."""
    from oracle import autoencoder as oa
    cfg = cfgs.VAE_FULL
    wseed = 3
    sd = ou.init_state_dict(oa.ae_param_shapes(cfg), wseed)
    x = vae_case()
    out = dict(weight_seed=wseed, torch_version=str(torch.__version__))
    for tag, bf in (("fp32", False), ("bf16", True)):
        t0 = time.time()
        mu, _ = oa.encode(sd, cfg, x, emulate_bf16=bf)
        rec = oa.decode(sd, cfg, mu, emulate_bf16=bf)
        out[f"mu_{tag}"] = mu
        out[f"rec_sub_{tag}"] = rec[..., ::4, ::4, ::4].contiguous()
        out[f"rec_mean_{tag}"], out[f"rec_msq_{tag}"] = float(rec.double().mean()), float((rec.double() ** 2).mean())
        print(f"vae golden {tag}: {time.time() - t0:.1f}s")
    out["mu_for_decode"] = out["mu_fp32"]                 # the decoder is tested in isolation on the fp32 oracle's latent
    torch.save(out, os.path.join(HERE, "vae_full_96.pt"))


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    if len(sys.argv) > 1 and sys.argv[1] == "vae":
        return vae_golden()
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
                    torch_version=str(torch.__version__)), os.path.join(HERE, "unet_full_24.pt"))
    train_golden()
    vae_golden()
    s = OracleDDPM(**cfgs.SCHED)
    torch.save(dict(betas=s.betas, alphas_cumprod=s.alphas_cumprod), os.path.join(HERE, "sched_tables.pt"))


if __name__ == "__main__":
    main()
