"""LatentDiffusionInferer with MONAI's interface (3d_ldm/train_diffusion.py:152,197-205,260-268,326-333;
3d_ldm/inference.py:85,94-99).  Pure orchestration: every tensor op is a libldm3d.so launch."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Union

import torch

from . import _lib


class LatentDiffusionInferer:
    def __init__(self, scheduler, scale_factor: float = 1.0, ldm_latent_shape=None, autoencoder_latent_shape=None):
        if ldm_latent_shape is not None or autoencoder_latent_shape is not None:
            raise NotImplementedError("latent resizing is not used by the reference")
        self.scheduler = scheduler
        self.scale_factor = scale_factor

    def __call__(self, inputs: torch.Tensor, autoencoder_model, diffusion_model, noise: torch.Tensor,
                 timesteps: torch.Tensor, condition: Optional[torch.Tensor] = None, mode: str = "crossattn",
                 seg: Optional[torch.Tensor] = None, vae_eps: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Training-time forward: z = AE.encode_stage_2_inputs(inputs) * scale -> add_noise -> UNet."""
        if mode not in ("crossattn", "concat"):
            raise NotImplementedError(f"{mode} condition is not supported")
        with torch.no_grad():
            if vae_eps is not None:
                latent = autoencoder_model.encode_stage_2_inputs(inputs, vae_eps)
            else:
                latent = autoencoder_model.encode_stage_2_inputs(inputs)
            if self.scale_factor != 1.0:
                latent = latent * self.scale_factor
        noisy = self.scheduler.add_noise(original_samples=latent, noise=noise, timesteps=timesteps)
        if mode == "concat" and condition is not None:
            return diffusion_model(x=noisy, timesteps=timesteps, context=None, cond=condition)
        if condition is not None:
            raise NotImplementedError("cross-attention conditioning is not on the reference's path (mode='concat')")
        return diffusion_model(x=noisy, timesteps=timesteps, context=None)

    @torch.no_grad()
    def sample(self, input_noise: torch.Tensor, autoencoder_model, diffusion_model, scheduler=None,
               save_intermediates: bool = False, intermediate_steps: int = 100,
               conditioning: Optional[torch.Tensor] = None, mode: str = "crossattn", verbose: bool = False,
               seg: Optional[torch.Tensor] = None, step_noise: Optional[Callable[[int], torch.Tensor]] = None,
               fused_seed: Optional[int] = None) -> Union[torch.Tensor, tuple]:
        """Reverse diffusion over scheduler.timesteps then VAE decode of latent / scale_factor.  ``fused_seed`` (extension) runs
        the loop on the device-resident sampler: noise from Philox(seed) inside the step kernel, one HIP graph per step."""
        if mode not in ("crossattn", "concat"):
            raise NotImplementedError(f"{mode} condition is not supported")
        if conditioning is not None and mode != "concat":
            raise NotImplementedError("cross-attention conditioning is not on the reference's path (mode='concat')")
        scheduler = scheduler or self.scheduler
        image = input_noise
        ts = scheduler.timesteps.tolist()
        it = ts
        if verbose:
            try:
                from tqdm import tqdm
                it = tqdm(ts)
            except ImportError:
                pass
        intermediates: List[torch.Tensor] = []
        B = image.shape[0]
        tbuf = torch.empty((B,), dtype=torch.float32, device=image.device)
        if fused_seed is not None and step_noise is None and hasattr(diffusion_model, "denoise_step"):
            sampler = scheduler.device_sampler(fused_seed)
            image = image.detach().to(torch.float32).contiguous().clone()
            cond = None if conditioning is None else conditioning.detach().to(torch.float32).contiguous()
            sampler.reset(tbuf)
            for t in it:
                diffusion_model.denoise_step(image, tbuf, sampler, cond=cond)
                if save_intermediates and t % intermediate_steps == 0:
                    intermediates.append(image.clone())
            it = []
        for t in it:
            tbuf.fill_(float(t))
            if conditioning is not None:
                eps = diffusion_model(x=image, timesteps=tbuf, context=None, cond=conditioning)
            else:
                eps = diffusion_model(x=image, timesteps=tbuf, context=None)
            if step_noise is not None:
                image, _ = scheduler.step(eps, t, image, noise=step_noise(t) if t > 0 else None)
            else:
                image, _ = scheduler.step(eps, t, image)
            if save_intermediates and t % intermediate_steps == 0:
                intermediates.append(image)
        latent = image if self.scale_factor == 1.0 else image / self.scale_factor
        out = autoencoder_model.decode_stage_2_outputs(latent) if autoencoder_model is not None else latent
        if save_intermediates:
            return out, [autoencoder_model.decode_stage_2_outputs(x / self.scale_factor) for x in intermediates]
        return out

    @torch.no_grad()
    def sample_concurrent(self, input_noises: Sequence[torch.Tensor], autoencoder_model, diffusion_models: Sequence,
                          scheduler=None, conditionings: Optional[Sequence[Optional[torch.Tensor]]] = None,
                          mode: str = "crossattn") -> List[torch.Tensor]:
        """K independent reverse-diffusion chains on one GPU, one stream and one module instance per chain, advanced
        round-robin from this thread so that the launches of different chains overlap (a single B = 1 chain is bound by the
        per-launch floor: DESIGN.md sections 3.5 / 5c).  Same arithmetic per chain as ``sample``; returns the decoded volumes.
        Not in the reference (it samples one volume at a time, 3d_ldm/inference.py:88-102)."""
        if len(diffusion_models) < len(input_noises):
            raise ValueError("one diffusion model instance per chain is needed (each owns a workspace and a launch graph)")
        scheduler = scheduler or self.scheduler
        conds = list(conditionings) if conditionings is not None else [None] * len(input_noises)
        if any(c is not None for c in conds) and mode != "concat":
            raise NotImplementedError("cross-attention conditioning is not on the reference's path (mode='concat')")
        dev = input_noises[0].device
        main = torch.cuda.current_stream(dev)
        streams = [torch.cuda.Stream(device=dev) for _ in input_noises]
        images = list(input_noises)
        tbufs = []
        for st, img in zip(streams, images):
            st.wait_stream(main)
            tbufs.append(torch.empty((img.shape[0],), dtype=torch.float32, device=dev))
        for t in scheduler.timesteps.tolist():
            for k, st in enumerate(streams):
                with torch.cuda.stream(st):
                    tbufs[k].fill_(float(t))
                    if conds[k] is not None:
                        eps = diffusion_models[k](x=images[k], timesteps=tbufs[k], context=None, cond=conds[k])
                    else:
                        eps = diffusion_models[k](x=images[k], timesteps=tbufs[k], context=None)
                    images[k], _ = scheduler.step(eps, t, images[k])
        outs = []
        for k, st in enumerate(streams):                    # the autoencoder (one workspace) decodes the chains one after another
            main.wait_stream(st)
            images[k].record_stream(main)
            latent = images[k] if self.scale_factor == 1.0 else images[k] / self.scale_factor
            outs.append(autoencoder_model.decode_stage_2_outputs(latent) if autoencoder_model is not None else latent)
        return outs
