"""Optimizer tail of the diffusion trainer on flat buffers.

The reference does (3d_ldm/train_diffusion.py:155-156, 214-223)::

    optimizer = torch.optim.Adam(params=unet.parameters(), lr=args.diffusion_train["lr"])
    lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[100, 1000], gamma=0.1)
    loss.backward(); torch.nn.utils.clip_grad_norm_(unet.parameters(), 1.0); optimizer.step()

``FlatAdam`` is a ``torch.optim.Optimizer`` (so ``MultiStepLR`` and ``zero_grad`` keep working) whose single parameter
is the module's flat fp32 buffer (``module.flatten_parameters()``): gradient clipping + Adam are two HIP launches over
191 M elements instead of ~320 per-tensor kernel groups, and the data-parallel mean is one all-reduce of
``module.flat_grads``.  Same arithmetic as ``torch.optim.Adam`` (defaults betas (0.9, 0.999), eps 1e-8, no amsgrad) and,
with ``weight_decay > 0``, as the decoupled ``torch.optim.AdamW`` of the stage-1 trainer
(3d_ldm/train_autoencoder.py:274-279: betas (0.5, 0.9), weight_decay 1e-5), and as ``clip_grad_norm_`` (scale = max_norm /
(norm + 1e-6), clamped to 1).
"""
from __future__ import annotations

import torch

from . import _lib


class _MseLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        p = pred.detach().to(torch.float32).contiguous()
        t = target.detach().to(device=p.device, dtype=torch.float32).contiguous()
        if not p.is_cuda:
            raise _lib.LdmError("mse_loss: CUDA tensors only (no CPU fallback)")
        if p.shape != t.shape:
            raise ValueError(f"mse_loss: shapes differ: {tuple(p.shape)} vs {tuple(t.shape)}")
        loss = torch.empty((1,), dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().ldm_op_mse_loss(p.data_ptr(), t.data_ptr(), p.numel(), loss.data_ptr(), grad.data_ptr(), _lib.current_stream()))
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


def mse_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """``F.mse_loss(pred, target)`` (mean reduction; 3d_ldm/train_diffusion.py:207) in two HIP launches that also leave the gradient
    ``2 (pred - target) / n`` for ``backward`` (instead of ~6 element-wise / reduction kernels of the tensor library between the
    forward and the backward launch plan).  Differentiable w.r.t. ``pred`` only."""
    return _MseLossFn.apply(pred, target)


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, module, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, max_grad_norm: float | None = None,
                 weight_decay: float = 0.0):
        if getattr(module, "flat_params", None) is None:
            module.flatten_parameters()
        self.module = module
        flat = module.flat_params
        if not flat.is_cuda:
            raise _lib.LdmError("FlatAdam needs the module on the GPU (no CPU fallback)")
        self._flat = torch.nn.Parameter(flat, requires_grad=False)      # shares storage with module.flat_params
        super().__init__([self._flat], dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self.fuse_repack = True                        # Adam + bf16 re-pack in one kernel (ldm_model_adam_step)
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        # {sum g^2 of the current gradients, optimizer steps skipped on the device because that sum was not finite}: include/ldm3d.h
        self.sq_norm = torch.zeros((2,), dtype=torch.float32, device=flat.device)
        self.steps = 0

    def zero_grad(self, set_to_none: bool = True):     # gradients are overwritten by every backward: nothing to clear
        pass

    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the current gradients (device scalar), as clip_grad_norm_ returns it."""
        L = _lib.lib()
        g = self.module.flat_grads
        with torch.cuda.device(g.device):
            _lib.check(L.ldm_grad_sq_norm(g.data_ptr(), g.numel(), self.sq_norm.data_ptr(), _lib.current_stream()))
        return self.sq_norm.sqrt()[0]

    def skipped_steps(self) -> torch.Tensor:
        """Device scalar: how many ``step()`` calls left the parameters untouched because the gradient norm was NaN / inf (the agreed
        NaN-skip of the trainers, decided on the device: no host read inside the step)."""
        return self.sq_norm[1]

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not used by the reference's trainers")
        L = _lib.lib()
        grp = self.param_groups[0]
        p, g = self.module.flat_params, self.module.flat_grads
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        self.steps += 1
        with torch.cuda.device(p.device):
            # the squared norm is taken whether or not clipping is on: the device-side NaN-skip reads it (max_norm <= 0 only disables
            # the clip factor inside the kernel; 3d_ldm/train_diffusion.py:210-212 skips a NaN batch regardless of clipping)
            _lib.check(L.ldm_grad_sq_norm(g.data_ptr(), g.numel(), self.sq_norm.data_ptr(), _lib.current_stream()))
            args = (float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]),
                    float(grp.get("weight_decay", 0.0)), self.steps, self.sq_norm.data_ptr(),
                    float(self.max_grad_norm or 0.0) if clip else 0.0, _lib.current_stream())
            h = getattr(self.module, "_h", None)
            if self.fuse_repack and h is not None and not getattr(self.module, "_dirty", True):
                # one pass: Adam on the flat master weights + the bf16 re-pack of the library's arena from the new values
                _lib.check(L.ldm_model_adam_step(h, p.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), *args))
                return None
            _lib.check(L.ldm_adam_step(p.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), p.numel(), *args))
        self.module.mark_weights_dirty()               # the bf16 arena is re-packed before the next forward
        return None


class FlatModuleAdam:
    """clip_grad_norm_ + AdamW for a plain ``nn.Module`` (the PatchDiscriminator of the stage-1 trainer,
    3d_ldm/train_autoencoder.py:277-279,487-494) on the same two HIP launches as ``FlatAdam``: the module's parameters are re-homed
    as views of one flat fp32 buffer, their ``.grad`` as views of a second one that autograd accumulates into."""

    def __init__(self, module: torch.nn.Module, lr: float, betas=(0.5, 0.9), eps: float = 1e-8, weight_decay: float = 1e-5,
                 max_grad_norm: float | None = 0.5):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params or not params[0].is_cuda:
            raise _lib.LdmError("FlatModuleAdam needs the module on the GPU (no CPU fallback)")
        dev = params[0].device
        total = sum(p.numel() for p in params)
        self.flat_params = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                v = self.flat_params[off:off + n].view(p.shape)
                v.copy_(p.detach().float())
                p.data = v
                p.grad = self.flat_grads[off:off + n].view(p.shape)
                off += n
        self.params = params
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, tuple(betas), eps, weight_decay, max_grad_norm
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat_params), torch.zeros_like(self.flat_params)
        self.sq_norm = torch.zeros((2,), dtype=torch.float32, device=dev)
        self.steps = 0
        self.param_groups = [dict(lr=lr)]

    def zero_grad(self, set_to_none: bool = False):
        self.flat_grads.zero_()
        off = 0
        for p in self.params:                              # a caller may have dropped the views (set_to_none elsewhere): restore them
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_grads.data_ptr() + 4 * off:
                p.grad = self.flat_grads[off:off + n].view(p.shape)
            off += n

    @torch.no_grad()
    def step(self):
        L = _lib.lib()
        p, g = self.flat_params, self.flat_grads
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        self.steps += 1
        with torch.cuda.device(p.device):
            _lib.check(L.ldm_grad_sq_norm(g.data_ptr(), g.numel(), self.sq_norm.data_ptr(), _lib.current_stream()))   # NaN-skip needs it, clip or not
            _lib.check(L.ldm_adam_step(p.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), p.numel(),
                                       float(self.param_groups[0]["lr"]), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                       float(self.weight_decay), self.steps, self.sq_norm.data_ptr(),
                                       float(self.max_grad_norm or 0.0) if clip else 0.0, _lib.current_stream()))
