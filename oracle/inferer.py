"""Oracle: LatentDiffusionInferer call / sample (test infrastructure only, see oracle/__init__.py).

Restates monai.inferers.LatentDiffusionInferer ([MONAI-ext]) as used at
3d_ldm/train_diffusion.py:152,197-205,260-268,326-333 and 3d_ldm/inference.py:85,94-99.
Every random draw (VAE sampling eps, DDPM z) is an explicit input.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from . import autoencoder as ae
from .unet import unet_forward


def inferer_call(unet_sd, unet_cfg, ae_sd, ae_cfg, scheduler, scale_factor: float,
                 inputs: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor,
                 vae_eps: torch.Tensor, condition: Optional[torch.Tensor] = None, mode: str = "crossattn",
                 emulate_bf16: bool = False) -> torch.Tensor:
    """Training forward: z = AE.encode_stage_2_inputs(inputs) * scale -> add_noise -> (concat cond) -> UNet."""
    with torch.no_grad():
        z = ae.encode_stage_2_inputs(ae_sd, ae_cfg, inputs, vae_eps, emulate_bf16) * scale_factor
    zt = scheduler.add_noise(z, noise, timesteps)
    if mode == "concat" and condition is not None:
        zt = torch.cat([zt, condition], dim=1)
    return unet_forward(unet_sd, unet_cfg, zt, timesteps, emulate_bf16)


def sample(unet_sd, unet_cfg, scheduler, input_noise: torch.Tensor, step_noise: Callable[[int], torch.Tensor],
           ae_sd=None, ae_cfg=None, scale_factor: float = 1.0, conditioning: Optional[torch.Tensor] = None,
           mode: str = "crossattn", emulate_bf16: bool = False, trace: Optional[List[torch.Tensor]] = None):
    """Reverse loop over scheduler.timesteps, then VAE decode of latent / scale_factor (if an AE is given).

    ``step_noise(t)`` supplies the DDPM z for step t (ignored by DDIM / at t == 0).
    """
    x = input_noise
    for t in scheduler.timesteps.tolist():
        xin = torch.cat([x, conditioning], dim=1) if (mode == "concat" and conditioning is not None) else x
        ts = torch.full((x.shape[0],), float(t))
        eps_hat = unet_forward(unet_sd, unet_cfg, xin, ts, emulate_bf16)
        x, _ = scheduler.step(eps_hat, t, x, step_noise(t) if t > 0 else None)
        if trace is not None:
            trace.append(x.clone())
    if ae_sd is None:
        return x
    return ae.decode_stage_2_outputs(ae_sd, ae_cfg, x / scale_factor, emulate_bf16)
