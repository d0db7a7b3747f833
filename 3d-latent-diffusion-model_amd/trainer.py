"""Data-parallel training step of the latent-diffusion UNet.

Reference behaviour being reproduced (3d_ldm/train_diffusion.py):
  :100-124  scale_factor = 1 / std(AE.encode_stage_2_inputs(first label batch)), averaged over ranks
  :147-149  DDP wrap: parameters broadcast from rank 0, gradients averaged over ranks during backward
  :155-156  Adam(lr) + MultiStepLR(milestones=[100, 1000], gamma=0.1), stepped once per epoch (:222-223)
  :172-223  per batch: noise drawn on the HOST then moved (:182), timesteps = randint on the device (:185-187),
            image latents (condition) under no_grad (:194-195), inferer(..., mode="concat") (:197-205), MSE (:207),
            NaN -> skip (:210-212), backward (:214), clip_grad_norm_(1.0) (:217), optimizer step (:219)

MI355X design: one process per GPU; the UNet's gradients live in ONE flat fp32 buffer written in place by the
hand-written backward plan, so the data-parallel exchange is a single (optionally chunked) all-reduce over RCCL/xGMI
issued right after backward, and clip + Adam are two fused launches over the flat buffers (``FlatAdam``).  Two
reference quirks are fixed on purpose: the NaN-skip is agreed across ranks (a single rank skipping
``backward`` dead-locks DDP in the reference) and decided ON THE DEVICE: a NaN loss makes every gradient NaN, the data-parallel
mean carries that to every rank, and the fused clip + Adam launch leaves parameters and moments untouched when the gradient norm is
not finite (``FlatAdam.skipped_steps``) -- no flag all-reduce and no host read inside the step; and the label encode that the
reference runs only to learn the latent shape (:179-181) is not executed (``reference_rng_order=True`` restores it).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .optim import mse_loss


class GradSync:
    """Collectives of the data-parallel trainer over torch.distributed ("nccl" = RCCL on ROCm; "gloo" in CPU tests)."""

    def __init__(self, chunk_elems: int = 1 << 26, grad_dtype: torch.dtype = torch.float32):
        """``grad_dtype=torch.bfloat16`` (opt-in) sends the gradients as bf16: 382 instead of 765 MB per step for the benchmark
        UNet, for rings that are xGMI-link bound; the reference's DDP reduces fp32."""
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if self.on else 1
        self.chunk = int(chunk_elems)                    # 64 Mi elements = 256 MB per collective
        self._avg = self.on and dist.get_backend() == "nccl"
        if grad_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("grad_dtype must be torch.float32 or torch.bfloat16")
        self.grad_dtype = grad_dtype
        self._modules = []                               # modules whose exchange runs inside backward on the library's communicator

    def _quiescent(self) -> None:
        """Two communicators live in a training process: torch.distributed's group (scalars, barriers, broadcasts, the fallback
        mean) and the library's own RCCL communicator on its comm stream (the buckets).  RCCL requires that two communicators never
        have collectives in flight in an order that differs between ranks; here that is excluded by construction -- every bucket is
        joined into the launch stream at the end of backward, and torch's collectives are ordered behind the launch stream -- and
        this check turns the construction into an assertion: a torch.distributed collective issued while a bucket is still un-joined
        is a programming error, not a race to be debugged on eight GPUs.

        Scope: the assertion guards the collectives issued THROUGH this class (every one the trainers and bench.py's ddp_train leg
        issue); code that calls torch.distributed directly is outside it.  The counter it reads is host bookkeeping: a backward that
        failed between a bucket and its join leaves it non-zero until the next backward begins (which orders itself behind the stray
        buckets and clears it: grad_sync_begin), so after such an error the next collective here raises rather than races."""
        from . import _lib
        L = _lib.lib()
        for m in self._modules:
            n = L.ldm_model_grad_sync_pending(m._h)
            if n:
                raise RuntimeError(f"torch.distributed collective while {n} gradient bucket(s) of the library's communicator are not joined")

    def attach(self, module, force_single: bool = False, transport=None, world: Optional[int] = None, rank: int = 0) -> bool:
        """Route the gradient exchange of ``module`` through the library's own RCCL communicator (``ldm_comm_*``) and overlap it
        with backward: ``loss.backward()`` then all-reduces (mean, fp32) the flat gradient buffer bucket by bucket on a comm
        stream while the backward plan is still running (``ldm_model_set_grad_sync``; DistributedDataParallel's bucketed hooks,
        3d_ldm/train_diffusion.py:147-149), and ``mean_`` becomes a no-op for that module.  The 128-byte RCCL unique id is made
        on rank 0 and broadcast over the existing torch.distributed group.  Only with the "nccl" backend on GPUs;
        ``grad_dtype=torch.bfloat16`` selects the library's bf16 wire format (``ldm_model_set_grad_wire``: cast, all-reduce, cast
        back on the comm stream); ``force_single`` builds a world-size-1 communicator without torch.distributed (tests, traces).

        ``transport(buf_ptr, count, dtype, op, stream_ptr) -> int`` (with ``world`` / ``rank``) replaces RCCL by a caller-supplied
        all-reduce (``ldm_comm_init_custom``): the test seam that lets one GPU play a rank of a world > 1 job.

        Every rank must build the same buckets, so rank 0's ``LDM_GRAD_BUCKET_MB`` is broadcast and written into this process's
        environment before any training plan exists.  If the communicator cannot be created (no librccl, a failing
        ``ncclCommInitRank``) the ranks agree on that and fall back together to ``mean_`` after backward instead of raising."""
        import ctypes as C
        import os
        import warnings
        from . import _lib
        L = _lib.lib()
        comm = C.c_void_p()
        if transport is not None:
            fn = _lib.ALLREDUCE_FN(lambda user, buf, count, dtype, op, stream: int(transport(buf, count, dtype, op, stream) or 0))
            _lib.check(L.ldm_comm_init_custom(int(rank), int(world or 1), C.cast(fn, C.c_void_p), None, C.byref(comm)))
            module._grad_transport = fn                      # the C function pointer must outlive the communicator
        else:
            single = force_single and not self.on
            if not single and not (self.on and dist.get_backend() == "nccl"):
                return False
            rank = 0 if single else dist.get_rank()
            world = 1 if single else self.world
            uid = C.create_string_buffer(128)
            # every rank probes librccl BEFORE any rank enters ncclCommInitRank (which blocks until all `world` ranks have joined):
            # a rank that cannot load it must make all of them fall back, not leave the others waiting inside the rendezvous
            ok = 1 if L.ldm_comm_rccl_version() > 0 else 0
            if ok and rank == 0 and L.ldm_comm_unique_id(uid) != 0:
                ok = 0
            if not single:
                flag = torch.tensor([float(ok)], device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if ok and not single:
                try:
                    mb = int(os.environ.get("LDM_GRAD_BUCKET_MB", "48") or 48)
                except ValueError:
                    mb = 48
                mb = min(max(mb, 1), 0xFFFF)
                t = torch.tensor(list(uid.raw) + [mb & 0xFF, (mb >> 8) & 0xFF], dtype=torch.uint8, device="cuda")
                dist.broadcast(t, src=0)
                raw = bytes(t.cpu().tolist())
                uid = C.create_string_buffer(raw[:128], 128)
                os.environ["LDM_GRAD_BUCKET_MB"] = str(raw[128] | (raw[129] << 8))
            if ok and L.ldm_comm_init(rank, world, uid, C.byref(comm)) != 0:
                ok = 0
            if not single:                                   # all ranks or none
                flag = torch.tensor([float(ok)], device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if not ok:
                msg = L.ldm_last_error()
                warnings.warn("in-library RCCL gradient exchange unavailable (%s): falling back to one all-reduce after backward "
                              "over torch.distributed" % (msg.decode() if msg else "?"))
                if comm:
                    L.ldm_comm_destroy(comm)
                return False
        _lib.check(L.ldm_model_set_grad_sync(module._h, comm))
        _lib.check(L.ldm_model_set_grad_wire(module._h, 1 if self.grad_dtype == torch.bfloat16 else 0))
        module._grad_comm = comm                         # keeps the handle alive as long as the module
        self._attached = getattr(self, "_attached", set()) | {id(module)}
        self._modules.append(module)
        return True

    def attached(self, module) -> bool:
        return id(module) in getattr(self, "_attached", set())

    def broadcast(self, flat: torch.Tensor, src: int = 0) -> None:
        """Parameter broadcast at wrap time (what DistributedDataParallel.__init__ does)."""
        if self.on:
            self._quiescent()
            dist.broadcast(flat, src=src)

    def mean_(self, flat: torch.Tensor) -> torch.Tensor:
        """In-place mean over ranks of a flat buffer; chunks are queued asynchronously and waited for together."""
        if not self.on:
            return flat
        self._quiescent()
        works, staged = [], []
        for lo in range(0, flat.numel(), self.chunk):
            part = flat[lo:lo + self.chunk]
            if self.grad_dtype != flat.dtype:              # reduced-precision wire format: cast, reduce, cast back
                if not self._avg:
                    part = part * (1.0 / self.world)       # pre-scale: the SUM of bf16 values then stays in range
                low = part.to(self.grad_dtype)
                staged.append((lo, low))
                part = low
            works.append(dist.all_reduce(part, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()
        for lo, low in staged:
            flat[lo:lo + low.numel()].copy_(low)
        if not self._avg and not staged:
            flat.mul_(1.0 / self.world)
        return flat

    def any(self, flag: torch.Tensor) -> torch.Tensor:
        """Logical OR over ranks of a 0/1 device scalar (the agreed NaN-skip)."""
        if self.on:
            self._quiescent()
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return flag

    def mean_scalar(self, t: torch.Tensor) -> torch.Tensor:
        if not self.on:
            return t
        self._quiescent()
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t / self.world

    def max_scalar(self, t: torch.Tensor) -> torch.Tensor:
        """MAX over ranks (wall clocks of a timed region: bench.py's ddp_train leg)."""
        if not self.on:
            return t
        self._quiescent()
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t

    def gather_objects(self, obj) -> list:
        """Every rank's (picklable) record on every rank, in rank order: per-rank timelines of the bench's ddp_train leg."""
        if not self.on:
            return [obj]
        self._quiescent()
        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    def barrier(self) -> None:
        if self.on:
            self._quiescent()
            dist.barrier()


@torch.no_grad()
def compute_scale_factor(autoencoder, labels: torch.Tensor, sync: Optional[GradSync] = None) -> torch.Tensor:
    """1 / std of the label latents of the first batch, averaged over ranks (train_diffusion.py:100-124)."""
    z = autoencoder.encode_stage_2_inputs(labels)
    sf = 1.0 / torch.std(z)
    return (sync or GradSync()).mean_scalar(sf)


class DiffusionTrainer:
    def __init__(self, unet, autoencoder, inferer, lr: float, max_grad_norm: float = 1.0, milestones=(100, 1000),
                 gamma: float = 0.1, reference_rng_order: bool = False, grad_dtype: torch.dtype = torch.float32):
        from .optim import FlatAdam
        self.unet, self.autoencoder, self.inferer = unet, autoencoder, inferer
        self.sync = GradSync(grad_dtype=grad_dtype)
        self.optimizer = FlatAdam(unet, lr=lr, max_grad_norm=max_grad_norm)      # flattens the parameters
        self.sync.broadcast(unet.flat_params, 0)
        unet.mark_weights_dirty()
        # RCCL through the library's C ABI, bucketed and overlapped with backward ("nccl" backend); else one all-reduce after it
        self.overlap = self.sync.attach(unet)
        self.lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=list(milestones), gamma=gamma)
        self.reference_rng_order = reference_rng_order
        self.factor = getattr(autoencoder, "factor", 4)

    def _draw(self, labels: torch.Tensor):
        B = labels.shape[0]
        lat = [s // self.factor for s in labels.shape[2:]]
        if self.reference_rng_order:
            with torch.no_grad():
                lat = list(self.autoencoder.encode_stage_2_inputs(labels).shape[2:])
        noise = torch.randn((B, self.autoencoder.latent_channels, *lat), dtype=torch.float32).to(labels.device)
        timesteps = torch.randint(0, self.inferer.scheduler.num_train_timesteps, (B,), device=labels.device).long()
        return noise, timesteps

    def _predict(self, images, labels, noise, timesteps):
        with torch.no_grad():
            image_latents = self.autoencoder.encode_stage_2_inputs(images)
        return self.inferer(inputs=labels, autoencoder_model=self.autoencoder, diffusion_model=self.unet, noise=noise,
                            timesteps=timesteps, condition=image_latents, mode="concat")

    def train_step(self, images: torch.Tensor, labels: torch.Tensor, noise: Optional[torch.Tensor] = None,
                   timesteps: Optional[torch.Tensor] = None):
        """One optimizer step.  Returns (loss [device scalar], skipped [0-dim device bool tensor: truthy if this step was a NaN-skip]).
        Nothing in here reads a device value on the host: ranks never wait for each other's host, only for the gradient exchange."""
        self.unet.train()
        images, labels = images.float(), labels.float()
        self.optimizer.zero_grad(set_to_none=True)
        if noise is None or timesteps is None:
            n2, t2 = self._draw(labels)
            noise = n2 if noise is None else noise
            timesteps = t2 if timesteps is None else timesteps
        noise_pred = self._predict(images, labels, noise, timesteps)
        loss = mse_loss(noise_pred, noise)               # F.mse_loss (:207) + its gradient in two HIP launches
        before = self.optimizer.skipped_steps().clone()
        loss.backward()                                  # with self.overlap the buckets are reduced inside this call
        if not self.overlap:
            self.sync.mean_(self.unet.flat_grads)
        self.optimizer.step()                            # a NaN / inf gradient norm (on any rank, after the mean) skips the update on the device
        return loss.detach(), self.optimizer.skipped_steps() > before

    def end_epoch(self):
        self.lr_scheduler.step()

    @torch.no_grad()
    def validate(self, loader, device) -> float:
        """Mean training-style loss over the loader, NaN batches skipped, averaged over ranks (:231-283)."""
        self.unet.eval()
        total, n = torch.zeros((), device=device), 0
        for batch in loader:
            images, labels = batch["image"].to(device).float(), batch["label"].to(device).float()
            noise, timesteps = self._draw(labels)
            v = F.mse_loss(self._predict(images, labels, noise, timesteps).float(), noise.float())
            if not bool(torch.isnan(v)):
                total += v
                n += 1
        val = total / n if n else torch.tensor(float("inf"), device=device)
        return float(self.sync.mean_scalar(val))


# ------------------------------------------------------------------------------------------------ stage 1: AutoencoderKL
class _GateGrad(torch.autograd.Function):
    """Identity whose backward passes the incoming gradient only where ``ok`` (a 0-dim device bool, filled in AFTER the branch behind
    this node has run forward) is true and zeros otherwise.  Selection, not multiplication: a NaN gradient coming back from a
    non-finite adversarial branch is replaced, not scaled (0 * NaN = NaN), so the branch is really dropped (train_autoencoder.py:417-422)."""

    @staticmethod
    def forward(ctx, x, ok):
        ctx.ok = ok
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return torch.where(ctx.ok, g, torch.zeros_like(g)), None


def kl_loss(z_mu: torch.Tensor, z_sigma: torch.Tensor) -> torch.Tensor:
    """The reference's KL term (3d_ldm/utils.py:249-262): 0.5 * sum(mu^2 + s^2 - log(s^2 + eps) - 1) over all but the batch
    dimension with s = clamp(sigma, min=1e-8), divided by the batch size, clamped to [0, 1000]."""
    eps = 1e-8
    s = torch.clamp(z_sigma, min=eps)
    kl = 0.5 * torch.sum(z_mu.pow(2) + s.pow(2) - torch.log(s.pow(2) + eps) - 1, dim=list(range(1, z_sigma.dim())))
    return torch.clamp(kl / kl.shape[0], 0.0, 1000.0)


class AutoencoderTrainer:
    """The stage-1 (VAE-GAN) training step (3d_ldm/train_autoencoder.py:352-494) on the HIP forward / backward plans:
    clamp the images to [0, 1] (:362), ``reconstruction, z_mu, z_sigma = autoencoder(images)`` (:366), reconstruction L1 / L2
    (:226-233,374), KL (:375,386) * kl_weight, and after ``warm_up_epochs`` (:304,409) the LSGAN generator term
    ``adv_weight * mse(D(reconstruction)[-1], 1)`` (:410-424); NaN-skip agreed over ranks, ``loss_g.backward()``,
    ``clip_grad_norm_(0.5)`` and AdamW(betas (0.5, 0.9), weight_decay 1e-5) (:274-279,440-451), lr scaled by sqrt(world) * 0.5
    under DDP (:246-259); then the discriminator step ``adv_weight * 0.5 * (mse(D(recon.detach())[-1], 0) + mse(D(images)[-1], 1))``
    with its own clip 0.5 + AdamW (:454-494).  The PatchDiscriminator runs on the same HIP kernels (``ldm3d.discriminator``).

    Not reproduced: the perceptual term needs a downloaded SqueezeNet (no network here): a non-zero ``perceptual_weight`` is
    reported once and the term is left out."""

    def __init__(self, autoencoder, lr: float, kl_weight: float, recon_loss: str = "l1", perceptual_weight: float = 0.0,
                 max_grad_norm: float = 0.5, weight_decay: float = 1e-5, warm_up_epochs: int = 5, adv_weight: float = 0.01,
                 discriminator=None, perceptual_weights: Optional[str] = None):
        from .discriminator import PatchAdversarialLoss, PatchDiscriminator
        from .optim import FlatAdam, FlatModuleAdam
        # the perceptual term (:236,386,406) needs LPIPS / SqueezeNet weights: from a user-supplied file, else dropped and recorded
        self.perceptual_weight, self.loss_perceptual = float(perceptual_weight or 0.0), None
        if perceptual_weight and perceptual_weights:
            from .perceptual import PerceptualLoss
            self.loss_perceptual = PerceptualLoss.from_file(perceptual_weights).to(autoencoder.flat_params.device
                                                                                  if getattr(autoencoder, "flat_params", None) is not None
                                                                                  else next(autoencoder.parameters()).device)
        self.perceptual_dropped = bool(perceptual_weight) and self.loss_perceptual is None   # logged by train_autoencoder.py
        if self.perceptual_dropped:
            # the reference's PerceptualLoss downloads a pretrained SqueezeNet (train_autoencoder.py:236): no weights, no network here.
            # Every shipped config sets a (small: 1e-5 .. 1e-3) weight, so warn once and train without the term instead of refusing.
            import warnings
            warnings.warn(f"autoencoder_train.perceptual_weight = {perceptual_weight}: the perceptual term needs pretrained SqueezeNet "
                          "weights that are not available offline; training continues WITHOUT it (reconstruction + KL + adversarial)")
        self.autoencoder = autoencoder
        self.sync = GradSync()
        world = self.sync.world
        lr = lr * (world ** 0.5 * 0.5 if world > 1 else 1.0)
        self.optimizer = FlatAdam(autoencoder, lr=lr, betas=(0.5, 0.9), eps=1e-8, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        self.sync.broadcast(autoencoder.flat_params, 0)
        autoencoder.mark_weights_dirty()
        self.overlap = self.sync.attach(autoencoder)
        self.kl_weight, self.l2 = kl_weight, recon_loss == "l2"
        self.warm_up_epochs, self.adv_weight = warm_up_epochs, adv_weight
        # the reference hard-codes in_channels=1 (:155); the image channel count of the autoencoder keeps 2-channel configs working
        dev = autoencoder.flat_params.device
        self.discriminator = discriminator if discriminator is not None else PatchDiscriminator(
            spatial_dims=3, num_layers_d=3, channels=32, in_channels=autoencoder.out_channels, out_channels=1, norm="INSTANCE")
        self.discriminator = self.discriminator.to(dev)
        self.adv_loss = PatchAdversarialLoss(criterion="least_squares")
        self.optimizer_d = FlatModuleAdam(self.discriminator, lr=lr, betas=(0.5, 0.9), eps=1e-8, weight_decay=weight_decay,
                                          max_grad_norm=max_grad_norm)
        self.sync.broadcast(self.optimizer_d.flat_params, 0)

    def intensity_loss(self, a, b):
        return F.mse_loss(a, b) if self.l2 else F.l1_loss(a, b)

    def train_step(self, images: torch.Tensor, epoch: int = 0, eps: Optional[torch.Tensor] = None):
        """-> (dict of detached scalar losses, skipped)."""
        self.autoencoder.train()
        images = torch.clamp(images.float(), 0.0, 1.0)
        # No host read of a device value in this step: non-finite inputs or losses (:362-365,426-437,470-484 `continue`) make the
        # gradient norm non-finite, the data-parallel mean carries that to every rank, and the fused clip + AdamW launches then leave
        # parameters and moments untouched (optim.FlatAdam.skipped_steps).  A non-finite adversarial term ALONE is dropped and the
        # step continues on reconstruction + KL (:417-422): its value is selected away in the forward and its gradient is selected
        # away where the branch leaves the reconstruction (_GateGrad), so no 0 * NaN reaches the autoencoder.  A skipped generator
        # step also skips the discriminator step of that batch (the reference `continue`s before it).
        adversarial = epoch > self.warm_up_epochs
        reconstruction, z_mu, z_sigma = self.autoencoder(images, eps=eps)
        recons = self.intensity_loss(reconstruction, images)
        kl = kl_loss(z_mu, z_sigma).mean()
        loss_g = recons + self.kl_weight * kl
        out = {"recons": recons.detach(), "kl": kl.detach()}
        if self.loss_perceptual is not None:                                       # :386,406
            p_loss = self.loss_perceptual(reconstruction.float(), images.float())
            loss_g = loss_g + self.perceptual_weight * p_loss
            out["perceptual"] = p_loss.detach()
        if adversarial:
            ok = torch.ones((), dtype=torch.bool, device=reconstruction.device)      # filled in below, read by _GateGrad.backward
            logits_fake = self.discriminator(_GateGrad.apply(reconstruction.contiguous().float(), ok))[-1]
            generator_loss = self.adv_loss(logits_fake, target_is_real=True, for_discriminator=False)
            ok.copy_(torch.isfinite(generator_loss.detach()))
            loss_g = loss_g + self.adv_weight * torch.where(ok, generator_loss, torch.zeros_like(generator_loss))
            out["adv_g"] = generator_loss.detach()
        before = self.optimizer.skipped_steps().clone()
        loss_g.backward()
        if not self.overlap:
            self.sync.mean_(self.autoencoder.flat_grads)
        self.optimizer.step()
        out["loss_g"] = loss_g.detach()
        skipped_g = self.optimizer.skipped_steps() > before
        if adversarial:
            # discriminator step (:454-494): fake = the detached reconstruction, real = the images
            self.optimizer_d.zero_grad()
            logits_fake = self.discriminator(reconstruction.contiguous().detach())[-1]
            loss_d_fake = self.adv_loss(logits_fake, target_is_real=False, for_discriminator=True)
            logits_real = self.discriminator(images.contiguous().detach())[-1]
            loss_d_real = self.adv_loss(logits_real, target_is_real=True, for_discriminator=True)
            discriminator_loss = (loss_d_fake + loss_d_real) * 0.5
            loss_d = self.adv_weight * discriminator_loss
            # a batch whose generator step was skipped gets no discriminator step either: a NaN factor makes every discriminator
            # gradient NaN and FlatModuleAdam skips itself on the device (still no host read)
            loss_d = loss_d * torch.where(skipped_g, torch.full_like(loss_d, float("nan")), torch.ones_like(loss_d))
            loss_d.backward()
            self.sync.mean_(self.optimizer_d.flat_grads)
            self.optimizer_d.step()                          # skips itself on a non-finite gradient norm
            out["adv_d"] = discriminator_loss.detach()
        return out, skipped_g

    @torch.no_grad()
    def validate(self, loader, device) -> float:
        """Mean clamped-reconstruction loss (:565-611), averaged over ranks."""
        self.autoencoder.eval()
        total, n = torch.zeros((), device=device), 0
        for batch in loader:
            images = torch.clamp(batch["image"].to(device).float(), 0.0, 1.0)
            reconstruction, _, _ = self.autoencoder(images)
            v = self.intensity_loss(torch.clamp(reconstruction, 0.0, 1.0), images)
            if bool(torch.isfinite(v)):
                total += v
                n += 1
        val = total / n if n else torch.tensor(float("inf"), device=device)
        return float(self.sync.mean_scalar(val))
