"""Oracle: DDPM / DDIM schedulers (test infrastructure only, see oracle/__init__.py).

Restates monai.networks.schedulers.{DDPMScheduler, DDIMScheduler} (MONAI >= 1.4, [MONAI-ext],
SURVEY.md section 8a rows a3, a3.1-a3.3) as constructed by the reference:
3d_ldm/train_diffusion.py:140-145, 3d_ldm/inference.py:79-84 with
3d_ldm/config/config_train_16g.json:56-60 (T=1000, scaled_linear_beta, 0.0015 -> 0.0195).
Defaults in force: variance_type="fixed_small", clip_sample=True ([-1, 1]), prediction_type="epsilon",
DDIM: eta=0, set_alpha_to_one=True, steps_offset=0.
Noise is always an explicit argument (never drawn inside) so goldens are RNG-independent.
"""
from __future__ import annotations

import numpy as np
import torch


class OracleScheduler:
    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "scaled_linear_beta",
                 beta_start: float = 1e-4, beta_end: float = 2e-2, clip_sample: bool = True):
        self.num_train_timesteps = num_train_timesteps
        if schedule == "scaled_linear_beta":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        elif schedule == "linear_beta":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise ValueError(schedule)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.clip_sample = clip_sample
        self.num_inference_steps = num_train_timesteps
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    def set_timesteps(self, num_inference_steps: int):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def add_noise(self, original: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor):
        """sqrt(abar_t) x0 + sqrt(1 - abar_t) eps, per-sample t (a3.1)."""
        ac = self.alphas_cumprod[timesteps.long()]
        shape = (-1,) + (1,) * (original.dim() - 1)
        return ac.sqrt().reshape(shape) * original + (1 - ac).sqrt().reshape(shape) * noise


class OracleDDPM(OracleScheduler):
    def step(self, model_output: torch.Tensor, t: int, sample: torch.Tensor, noise: torch.Tensor | None):
        """-> (x_{t-1}, x0_hat).  ``noise`` is z ~ N(0, I); ignored at t == 0 (a3.2)."""
        t = int(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[t - 1] if t > 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        if self.clip_sample:
            x0 = torch.clamp(x0, -1, 1)
        c0 = (a_prev ** 0.5 * self.betas[t]) / b_t
        c1 = self.alphas[t] ** 0.5 * b_prev / b_t
        prev = c0 * x0 + c1 * sample
        if t > 0:
            var = torch.clamp((1 - a_prev) / (1 - a_t) * self.betas[t], min=1e-20)
            prev = prev + var ** 0.5 * noise
        return prev, x0


class OracleDDIM(OracleScheduler):
    def step(self, model_output: torch.Tensor, t: int, sample: torch.Tensor, noise: torch.Tensor | None = None,
             eta: float = 0.0):
        """Deterministic (eta = 0) DDIM update; eps_hat is NOT recomputed after the clamp (a3.3)."""
        t = int(t)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one   # set_alpha_to_one=True
        b_t = 1 - a_t
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        if self.clip_sample:
            x0 = torch.clamp(x0, -1, 1)
        var = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
        std = eta * var ** 0.5
        direction = (1 - a_prev - std ** 2) ** 0.5 * model_output
        prev = a_prev ** 0.5 * x0 + direction
        if eta > 0:
            prev = prev + std * noise
        return prev, x0
