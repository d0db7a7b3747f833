"""DDPM / DDIM schedulers with MONAI's interface, element-wise math on the GPU through libldm3d.so.

Mirror of monai.networks.schedulers.{DDPMScheduler, DDIMScheduler} as the reference constructs them
(3d_ldm/train_diffusion.py:140-145, 3d_ldm/inference.py:79-84: T=1000, "scaled_linear_beta", 0.0015 -> 0.0195).
The beta / alpha-bar tables and the per-timestep scalar coefficients are computed on the host in fp32 with the
same torch op order MONAI uses; ``step`` / ``add_noise`` launch one fused element-wise kernel each
(ldm_ddpm_step / ldm_ddim_step / ldm_add_noise, include/ldm3d.h).  CUDA tensors only - no CPU fallback.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib


class _Scheduler:
    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta", beta_start: float = 1e-4,
                 beta_end: float = 2e-2, clip_sample: bool = True, prediction_type: str = "epsilon"):
        if prediction_type != "epsilon":
            raise NotImplementedError("only prediction_type='epsilon' is on the reference's path")
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.clip_sample = clip_sample
        if schedule == "scaled_linear_beta":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif schedule == "linear_beta":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(f"schedule '{schedule}' is not used by the reference")
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.num_inference_steps = num_train_timesteps
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())
        self._sqrt_ac = self.alphas_cumprod ** 0.5
        self._sqrt_1mac = (1 - self.alphas_cumprod) ** 0.5
        self._dev_tables = {}

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                             f"`self.num_train_timesteps`: {self.num_train_timesteps}")
        self.num_inference_steps = num_inference_steps
        step_ratio = self.num_train_timesteps // self.num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)
        self._on_set_timesteps()

    def _on_set_timesteps(self):
        pass

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """sqrt(abar_t) x0 + sqrt(1 - abar_t) eps with per-sample t (inside inferer.__call__,
        3d_ldm/train_diffusion.py:197-205)."""
        x0 = original_samples
        if not x0.is_cuda:
            raise _lib.LdmError("add_noise: CUDA tensors only (no CPU fallback)")
        dev = x0.device
        tab = self._dev_tables.get(dev)
        if tab is None:
            tab = (self._sqrt_ac.to(dev), self._sqrt_1mac.to(dev))
            self._dev_tables[dev] = tab
        t = timesteps.to(device=dev, dtype=torch.long).reshape(-1)
        sa = tab[0][t].contiguous()
        sb = tab[1][t].contiguous()
        x0c = x0.detach().to(torch.float32).contiguous()
        ec = noise.detach().to(device=dev, dtype=torch.float32).contiguous()
        out = torch.empty_like(x0c)
        B = x0c.shape[0]
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ldm_add_noise(x0c.data_ptr(), ec.data_ptr(), sa.data_ptr(), sb.data_ptr(),
                                                out.data_ptr(), B, x0c.numel() // B, _lib.current_stream()))
        return out

    def device_sampler(self, seed: int = 0, eta: float = 0.0) -> "DeviceSampler":
        """The fused, device-resident form of ``step`` over ``self.timesteps`` (see DeviceSampler)."""
        return DeviceSampler(self, seed, eta)

    @staticmethod
    def _draw(model_output: torch.Tensor, generator: Optional[torch.Generator]) -> torch.Tensor:
        # MONAI draws with the generator's device and moves to the sample; a None / CUDA generator draws in place.
        if generator is not None and generator.device.type == "cpu":
            return torch.randn(model_output.size(), dtype=torch.float32, generator=generator).to(model_output.device)
        return torch.randn(model_output.size(), dtype=torch.float32, device=model_output.device, generator=generator)


class DDPMScheduler(_Scheduler):
    """variance_type fixed_small / fixed_large, clip_sample=True (MONAI default, not overridden by the reference)."""

    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta", variance_type: str = "fixed_small",
                 clip_sample: bool = True, prediction_type: str = "epsilon", **schedule_args):
        super().__init__(num_train_timesteps, schedule, clip_sample=clip_sample, prediction_type=prediction_type,
                         **schedule_args)
        if variance_type not in ("fixed_small", "fixed_large"):
            raise NotImplementedError("learned variance is not on the reference's path")
        self.variance_type = variance_type
        ac = self.alphas_cumprod
        ac_prev = torch.cat([self.one.reshape(1), ac[:-1]])
        beta_prod = 1 - ac
        beta_prod_prev = 1 - ac_prev
        self._inv_sqrt_a = (1.0 / ac ** 0.5).tolist()
        self._sqrt_b = (beta_prod ** 0.5).tolist()
        self._c0 = ((ac_prev ** 0.5 * self.betas) / beta_prod).tolist()
        self._c1 = (self.alphas ** 0.5 * beta_prod_prev / beta_prod).tolist()
        var = (1 - ac_prev) / (1 - ac) * self.betas
        var = torch.clamp(var, min=1e-20) if variance_type == "fixed_small" else self.betas.clone()
        self._sigma = (var ** 0.5).tolist()

    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor,
             generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None
             ) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (x_{t-1}, x0_hat).  ``noise`` (extension) supplies z explicitly; otherwise torch.randn."""
        t = int(timestep)
        if not sample.is_cuda:
            raise _lib.LdmError("DDPMScheduler.step: CUDA tensors only (no CPU fallback)")
        eps = model_output.detach().to(torch.float32).contiguous()
        x = sample.detach().to(torch.float32).contiguous()
        z = None
        if t > 0:
            z = noise if noise is not None else self._draw(eps, generator)
            z = z.to(device=x.device, dtype=torch.float32).contiguous()
        prev = torch.empty_like(x)
        x0 = torch.empty_like(x)
        with torch.cuda.device(x.device):
            # x0 = (x - sqrt_b eps) / sqrt_a is evaluated as a multiply by 1/sqrt_a (<= 1 ulp from MONAI's divide)
            _lib.check(_lib.lib().ldm_ddpm_step(eps.data_ptr(), x.data_ptr(), _lib.ptr(z), prev.data_ptr(), x0.data_ptr(),
                                                x.numel(), self._inv_sqrt_a[t], self._sqrt_b[t], self._c0[t], self._c1[t],
                                                self._sigma[t] if t > 0 else 0.0, int(self.clip_sample),
                                                _lib.current_stream()))
        return prev, x0


class DDIMScheduler(_Scheduler):
    """eta = 0 by default, set_alpha_to_one=True, steps_offset=0 (MONAI defaults; BASELINE configs 1 and 5)."""

    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta", clip_sample: bool = True,
                 set_alpha_to_one: bool = True, steps_offset: int = 0, prediction_type: str = "epsilon", **schedule_args):
        super().__init__(num_train_timesteps, schedule, clip_sample=clip_sample, prediction_type=prediction_type,
                         **schedule_args)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.steps_offset = steps_offset

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        super().set_timesteps(num_inference_steps, device)
        if self.steps_offset:
            self.timesteps = self.timesteps + self.steps_offset

    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor, eta: float = 0.0,
             generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None
             ) -> Tuple[torch.Tensor, torch.Tensor]:
        t = int(timestep)
        if not sample.is_cuda:
            raise _lib.LdmError("DDIMScheduler.step: CUDA tensors only (no CPU fallback)")
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        var = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
        std = eta * var ** 0.5
        direction = (1 - a_prev - std ** 2) ** 0.5
        eps = model_output.detach().to(torch.float32).contiguous()
        x = sample.detach().to(torch.float32).contiguous()
        z = None
        if eta > 0:
            z = noise if noise is not None else self._draw(eps, generator)
            z = z.to(device=x.device, dtype=torch.float32).contiguous()
        prev = torch.empty_like(x)
        x0 = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ldm_ddim_step(eps.data_ptr(), x.data_ptr(), _lib.ptr(z), prev.data_ptr(), x0.data_ptr(),
                                                x.numel(), float(1.0 / a_t ** 0.5), float(b_t ** 0.5), float(a_prev ** 0.5),
                                                float(direction), float(std), int(self.clip_sample), _lib.current_stream()))
        return prev, x0


class DeviceSampler:
    """The scheduler step as ONE kernel with everything it needs on the device (``ldm_sampler_*``): the per-step coefficients
    (computed here exactly as ``DDPMScheduler.step`` / ``DDIMScheduler.step`` pass them by value) live in a device table, the
    current step index and the UNet's timestep input are device state advanced by the kernel itself, and the noise z is drawn
    inside the kernel (Philox4x32-10, counter = (element, step), key = seed) instead of ``torch.randn``.  One denoising step
    (``DiffusionModelUNet.denoise_step``) is then a fixed launch sequence with fixed arguments and replays as ONE HIP graph:
    no ``fill_`` / ``normal_`` launches and no host work per step (3d_ldm/inference.py:94-99's loop body).

    Not MONAI's RNG stream: a chain sampled this way is a different (equally distributed) draw than the same seed through
    ``torch.randn``; ``noise(step, shape)`` returns the exact z of a step for reproducibility checks."""

    def __init__(self, scheduler: _Scheduler, seed: int = 0, eta: float = 0.0):
        import ctypes as C
        self.scheduler = scheduler
        self.timesteps = [int(t) for t in scheduler.timesteps.tolist()]
        rows = []
        if isinstance(scheduler, DDIMScheduler):
            kind = 1
            ratio = scheduler.num_train_timesteps // scheduler.num_inference_steps
            for t in self.timesteps:
                prev_t = t - ratio
                a_t = scheduler.alphas_cumprod[t]
                a_prev = scheduler.alphas_cumprod[prev_t] if prev_t >= 0 else scheduler.final_alpha_cumprod
                var = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
                std = eta * var ** 0.5
                rows.append([float(1.0 / a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_prev ** 0.5),
                             float((1 - a_prev - std ** 2) ** 0.5), float(std), float(t)])
        elif isinstance(scheduler, DDPMScheduler):
            kind = 0
            for t in self.timesteps:
                rows.append([scheduler._inv_sqrt_a[t], scheduler._sqrt_b[t], scheduler._c0[t], scheduler._c1[t],
                             scheduler._sigma[t] if t > 0 else 0.0, float(t)])
        else:
            raise TypeError("DeviceSampler needs a DDPMScheduler or a DDIMScheduler")
        coef = torch.tensor(rows, dtype=torch.float32).contiguous()
        self._h = C.c_void_p()
        _lib.check(_lib.lib().ldm_sampler_create(coef.data_ptr(), len(rows), kind, int(scheduler.clip_sample), int(seed) & (2 ** 64 - 1),
                                                 C.byref(self._h)))
        self.n_steps, self.seed = len(rows), int(seed)

    def reset(self, tbuf: torch.Tensor) -> None:
        """Step counter := 0 and ``tbuf`` (the UNet's fp32 timestep input, one entry per sample) := the first timestep."""
        with torch.cuda.device(tbuf.device):
            _lib.check(_lib.lib().ldm_sampler_reset(self._h, tbuf.data_ptr(), tbuf.numel(), _lib.current_stream()))

    def step(self, eps: torch.Tensor, x: torch.Tensor, tbuf: torch.Tensor, x0_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x := step(x, eps) IN PLACE for the step the device counter points at; advances the counter and ``tbuf``."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and eps.is_contiguous() and eps.dtype == torch.float32):
            raise _lib.LdmError("DeviceSampler.step: contiguous fp32 CUDA tensors only")
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ldm_sampler_step(self._h, eps.data_ptr(), x.data_ptr(), _lib.ptr(x0_out), x.numel(), tbuf.data_ptr(),
                                                   tbuf.numel(), _lib.current_stream()))
        return x

    def noise(self, step: int, shape, device) -> torch.Tensor:
        out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
        with torch.cuda.device(out.device):
            _lib.check(_lib.lib().ldm_sampler_noise(self._h, int(step), out.data_ptr(), out.numel(), _lib.current_stream()))
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _lib.lib().ldm_sampler_destroy(self._h)
                self._h = None
        except Exception:
            pass
