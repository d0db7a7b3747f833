"""nn.Module shells over libldm3d.so: ``DiffusionModelUNet`` and ``AutoencoderKL``.

Drop-in for the classes the reference instantiates from its JSON configs through ``define_instance``
(3d_ldm/utils.py:243-246; ``_target_`` strings at 3d_ldm/config/config_train_16g.json:8,40 and
config_train_32g.json:8,41) and then uses only through the nn.Module surface:
``__call__(x=, timesteps=, context=None)``, ``encode_stage_2_inputs``, ``decode_stage_2_outputs``,
``forward -> (recon, z_mu, z_sigma)``, ``state_dict`` / ``load_state_dict`` / ``parameters`` / ``to`` /
``train`` / ``eval``.  Parameter names and shapes (the MONAI state_dict layout) are enumerated from the
library, so there is one source of truth for them; parameters are ordinary fp32 nn.Parameters and are
re-packed into the library's bf16 weight arena whenever they change.

``DiffusionModelUNet`` is differentiable w.r.t. its parameters: under ``torch.enable_grad()`` with parameters that
require grad, forward runs the training plan (same kernels, activations kept) and ``loss.backward()`` runs the
hand-written backward plan (3d_ldm/train_diffusion.py:197-216); likewise ``AutoencoderKL.forward`` (stage-1 trainer,
3d_ldm/train_autoencoder.py:366-451).  ``encode`` / ``decode`` alone are inference entry points (the diffusion trainer
uses them under ``no_grad``: 3d_ldm/train_diffusion.py:104,180).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Sequence, Union

import torch
import torch.nn as nn

from . import _lib


class _Node(nn.Module):
    """Anonymous container used to give parameters their dotted MONAI names."""


def _per_level(v, n, name):
    if isinstance(v, (int, bool)):
        return [int(v)] * n
    v = [int(a) for a in v]
    if len(v) != n:
        raise ValueError(f"{name} must have one entry per level ({n}), got {len(v)}")
    return v


class _LdmModule(nn.Module):
    """Shared plumbing: C handle, parameter tree, weight upload, workspace cache."""

    def __init__(self):
        super().__init__()
        self._h = C.c_void_p()
        self._dirty = True
        self._uploaded_versions: Dict[str, int] = {}
        self._ws: Dict[tuple, torch.Tensor] = {}

    # -- parameter tree -------------------------------------------------------------------------------
    def _build_params(self):
        L = _lib.lib()
        n = L.ldm_model_num_params(self._h)
        for i in range(n):
            name = L.ldm_model_param_name(self._h, i).decode()
            nd = L.ldm_model_param_ndim(self._h, i)
            shp = L.ldm_model_param_shape(self._h, i)
            shape = tuple(int(shp[k]) for k in range(nd))
            parts = name.split(".")
            node = self
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            node.register_parameter(parts[-1], nn.Parameter(torch.empty(shape, dtype=torch.float32)))
        self.reset_parameters()

    def reset_parameters(self):
        """PyTorch-default init (what MONAI's Convolution / nn.Linear / nn.GroupNorm do) ..."""
        params = dict(self.named_parameters())
        for name, p in params.items():
            with torch.no_grad():
                if p.dim() == 1 and name.endswith(".weight"):
                    p.fill_(1.0)                                  # GroupNorm gamma
                elif p.dim() == 1:
                    w = params.get(name[:-len("bias")] + "weight")
                    if w is not None and w.dim() > 1:
                        fan_in = w[0].numel()
                        bound = 1.0 / math.sqrt(fan_in)
                        p.uniform_(-bound, bound)
                    else:
                        p.zero_()                                 # GroupNorm beta
                else:
                    nn.init.kaiming_uniform_(p, a=math.sqrt(5))
        self._zero_init()
        self._dirty = True

    def _zero_init(self):
        pass

    # -- keep the device arena in sync ------------------------------------------------------------------
    def mark_weights_dirty(self):
        self._dirty = True

    # legacy (MONAI GenerativeModels / MONAI < 1.4 checkpoints) -> MONAI core >= 1.4 key names, as MONAI's own
    # ``load_old_state_dict`` renames them: the attention block's q / k / v / output projections moved under ``.attn`` and the
    # upsampling conv became ``postconv``.  ``state_dict()`` always emits the MONAI >= 1.4 names (what the reference's
    # ``monai.networks.nets.*`` targets save: 3d_ldm/train_diffusion.py:92-95,129-136, inference.py:67-77).
    _LEGACY_KEY_RULES = (
        (r"\.upsampler\.conv\.conv\.", ".upsampler.postconv.conv."),                      # DiffusionModelUNet upsample conv
        (r"^(decoder\.blocks\.\d+)\.conv\.conv\.", r"\1.postconv.conv."),                 # AutoencoderKL decoder upsample conv
        (r"\.(to_q|to_k|to_v)\.(weight|bias)$", r".attn.\1.\2"),                         # attention projections
        (r"\.proj_attn\.(weight|bias)$", r".attn.out_proj.\1"),
    )

    @classmethod
    def remap_legacy_keys(cls, state_dict, own_keys):
        """Returns ``state_dict`` with legacy key names translated, only where the translated name is one of ``own_keys`` and the
        legacy name is not (so a current checkpoint passes through untouched)."""
        import re
        out = {}
        for k, v in state_dict.items():
            nk = k
            if k not in own_keys:
                for pat, rep in cls._LEGACY_KEY_RULES:
                    cand = re.sub(pat, rep, nk)
                    if cand != nk and (cand in own_keys or ".attn." not in cand):
                        nk = cand
                if nk not in own_keys:
                    nk = k
            out[nk] = v
        return out

    def load_state_dict(self, state_dict, *a, **k):
        own = set(super().state_dict().keys())
        r = super().load_state_dict(self.remap_legacy_keys(state_dict, own), *a, **k)
        self._dirty = True
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._dirty = True
        return r

    def train(self, mode: bool = True):
        if mode != self.training:                      # a mode switch re-checks the weights once; repeated .train() calls do not
            self._dirty = True
        return super().train(mode)

    def _param_list(self):
        """Parameters in the library's order (the order of the flat gradient buffer)."""
        pl = getattr(self, "_plist", None)
        if pl is None:
            L = _lib.lib()
            d = dict(self.named_parameters())
            pl = [d[L.ldm_model_param_name(self._h, i).decode()] for i in range(L.ldm_model_num_params(self._h))]
            self._plist = pl
        return pl

    def _sync_weights(self):
        if not (self._dirty or self.training):
            return
        L = _lib.lib()
        pl = self._param_list()
        if pl and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in pl):
            # device-resident master weights (training): one call re-packs everything on the GPU
            sig = tuple((p._version, p.data_ptr()) for p in pl)
            if self._dirty or sig != getattr(self, "_dev_sig", None):
                flat = getattr(self, "flat_params", None)
                with torch.cuda.device(pl[0].device):
                    if flat is not None and self._is_flat(pl, flat):
                        _lib.check(L.ldm_model_load_params_flat(self._h, flat.data_ptr(), _lib.current_stream()))
                    else:
                        arr = (C.c_void_p * len(pl))(*[p.data_ptr() for p in pl])
                        _lib.check(L.ldm_model_load_params_device(self._h, arr, len(pl), _lib.current_stream()))
                self._dev_sig = sig
            self._dirty = False
            return
        for name, p in self.named_parameters():
            v = p._version
            if self._uploaded_versions.get(name) == (v, p.data_ptr()):
                continue
            host = p.detach().to(device="cpu", dtype=torch.float32).contiguous()
            _lib.check(L.ldm_model_load_param(self._h, name.encode(), host.data_ptr(), host.numel()))
            self._uploaded_versions[name] = (v, p.data_ptr())
        self._dirty = False

    def _is_flat(self, pl, flat) -> bool:
        """Every parameter still is the view of ``flat`` that flatten_parameters made (a .to() or an optimizer that
        re-binds .data breaks it; the per-tensor path then takes over)."""
        offs = getattr(self, "_offs", None)
        if offs is None:
            L = _lib.lib()
            offs = self._offs = [int(L.ldm_model_param_offset(self._h, i)) for i in range(len(pl))]
        base = flat.data_ptr()
        return flat.is_cuda and all(p.data_ptr() == base + 4 * offs[i] for i, p in enumerate(pl))

    # -- flat parameter / gradient storage (training) ---------------------------------------------------------
    def flatten_parameters(self) -> torch.Tensor:
        """Re-home every parameter as a view of ONE flat fp32 tensor on its current device, in the library's
        parameter order (= the layout of the flat gradient buffer), and pre-assign ``.grad`` views of a second flat
        tensor.  Backward then writes gradients in place (no per-parameter copies), the data-parallel all-reduce is a
        single collective over ``flat_grads`` and the optimizer is one fused kernel over both buffers
        (``ldm3d.optim.FlatAdam``).  In this mode every backward OVERWRITES the gradients (no accumulation)."""
        pl = self._param_list()
        dev = pl[0].device
        L = _lib.lib()
        total = int(L.ldm_model_param_numel_total(self._h))
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        grads = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for i, p in enumerate(pl):
                off = int(L.ldm_model_param_offset(self._h, i))
                v = flat[off:off + p.numel()].view(p.shape)
                v.copy_(p.detach().to(device=dev, dtype=torch.float32))
                p.data = v
                p.grad = grads[off:off + p.numel()].view(p.shape)
        self.flat_params, self.flat_grads = flat, grads
        self._dirty = True
        return flat

    # -- precision of the inference plans ---------------------------------------------------------------------
    def set_precision(self, precision: str = "bf16"):
        """``"bf16"`` (default): bf16 storage / bf16 MFMA / fp32 accumulate, the headline path.  ``"fp32"``: the
        reference's own arithmetic (autocast is off at 3d_ldm/train_diffusion.py:177 and absent from inference.py:91-99):
        fp32 activations and weights; inference convolutions as three bf16 MFMAs per product on hi / lo splits of the fp32 operands
        (fp32-class accuracy, ~5e-5 rel-L2 from the CPU path, about 0.3 of the bf16 throughput), everything else and every
        training plan on the exact fp32 matrix instruction (gradients ~1e-5 from fp32 autograd).  Inference and training plans
        of both networks.  Environment default: ``LDM_PRECISION=fp32``."""
        code = {"bf16": 0, "fp32": 1}.get(str(precision).lower())
        if code is None:
            raise ValueError(f"precision must be 'bf16' or 'fp32', got {precision!r}")
        _lib.check(_lib.lib().ldm_model_set_precision(self._h, code))
        self.precision = "fp32" if code else "bf16"
        self._uploaded_versions.clear()                # the unrounded copies are made at upload time
        self._dev_sig = None
        self._dirty = True
        return self

    def _default_precision(self):
        import os
        self.precision = "bf16"
        if os.environ.get("LDM_PRECISION", "").lower() == "fp32":
            self.set_precision("fp32")

    # -- debug taps (stage-wise parity tests) ---------------------------------------------------------------------
    def _tap_layout(self, kind: str, B, D, H, W):
        L = _lib.lib()
        n = L.ldm_model_tap_count(self._h, kind.encode(), B, D, H, W)
        if n < 0:
            raise _lib.LdmError((L.ldm_last_error() or b"tap query failed").decode())
        total = int(L.ldm_model_tap_elems(self._h, kind.encode(), B, D, H, W))
        out = []
        name = C.create_string_buffer(128)
        dims = (C.c_int * 5)()
        off = C.c_int64()
        for i in range(n):
            _lib.check(L.ldm_model_tap_info(self._h, kind.encode(), B, D, H, W, i, name, 128, dims, C.byref(off)))
            out.append((name.value.decode(), tuple(int(v) for v in dims), int(off.value)))
        return out, total

    def _tap_buffers(self, kind, B, D, H, W, force, device):
        layout, total = self._tap_layout(kind, B, D, H, W)
        taps_out = torch.zeros(max(total, 1), dtype=torch.float32, device=device)
        taps_in = None
        if force is not None:
            taps_in = torch.zeros(max(total, 1), dtype=torch.float32, device=device)
            for name, dims, off in layout:
                if name not in force:
                    raise KeyError(f"force is missing the tap '{name}'")
                t = force[name].to(device=device, dtype=torch.float32).contiguous()
                if tuple(t.shape) != dims:
                    raise ValueError(f"force['{name}'] has shape {tuple(t.shape)}, expected {dims}")
                taps_in[off:off + t.numel()] = t.reshape(-1)
        L = _lib.lib()
        nbytes = L.ldm_model_taps_workspace_bytes(self._h, kind.encode(), B, D, H, W, 1 if force is not None else 0)
        if nbytes == 0:
            raise _lib.LdmError((L.ldm_last_error() or b"workspace query failed").decode())
        ws = self._workspace((kind + "-taps", B, D, H, W, force is not None), nbytes, device)
        return layout, taps_out, taps_in, ws

    @staticmethod
    def _tap_dict(layout, taps_out):
        out = {}
        for name, dims, off in layout:
            n = 1
            for v in dims:
                n *= v
            out[name] = taps_out[off:off + n].view(dims).clone()
        return out

    def _workspace(self, key: tuple, nbytes: int, device) -> torch.Tensor:
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes or ws.device != device:
            ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    @staticmethod
    def _need_cuda(t: torch.Tensor, what: str):
        if not t.is_cuda:
            raise _lib.LdmError(f"{what}: tensor is on {t.device}; this implementation runs on the GPU only "
                                f"(no CPU fallback - the CPU oracle lives in oracle/ and is test infrastructure)")

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _lib.lib().ldm_model_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class _UNetTrainFn(torch.autograd.Function):
    """eps_hat = UNet(x_t, t; params) with the hand-written backward plan behind ``loss.backward()``."""

    @staticmethod
    def forward(ctx, module, x, timesteps, cond, *params):
        ctx.module = module
        out, ctx.token = module._train_forward(x, timesteps, cond)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        grads = ctx.module._train_backward(ctx.token, grad_out)
        return (None, None, None, None) + tuple(grads)


class DiffusionModelUNet(_LdmModule):
    """MI355X-native DiffusionModelUNet (kwargs of ``diffusion_def``, 3d_ldm/config/config_train_16g.json:39-48)."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int,
                 num_res_blocks: Union[Sequence[int], int] = (2, 2, 2, 2),
                 channels: Sequence[int] = (32, 64, 64, 64),
                 attention_levels: Sequence[bool] = (False, False, True, True),
                 norm_num_groups: int = 32, norm_eps: float = 1e-6, resblock_updown: bool = False,
                 num_head_channels: Union[int, Sequence[int]] = 8, with_conditioning: bool = False,
                 transformer_num_layers: int = 1, cross_attention_dim: Optional[int] = None,
                 num_class_embeds: Optional[int] = None, upcast_attention: bool = False,
                 dropout_cattn: float = 0.0, include_fc: bool = True, use_combined_linear: bool = False,
                 use_flash_attention: bool = False):
        super().__init__()
        if with_conditioning or cross_attention_dim is not None or num_class_embeds is not None:
            raise NotImplementedError("cross-attention / class conditioning is not on the reference's path "
                                      "(with_conditioning=False everywhere; conditioning is channel concat)")
        if resblock_updown or not include_fc or use_combined_linear:
            raise NotImplementedError("resblock_updown / include_fc=False / use_combined_linear are not implemented")
        n = len(channels)
        if len(attention_levels) != n:
            raise ValueError("attention_levels must have one entry per level")
        cfg = _lib.UNetCfg()
        cfg.spatial_dims, cfg.in_channels, cfg.out_channels, cfg.num_levels = spatial_dims, in_channels, out_channels, n
        nrb = _per_level(num_res_blocks, n, "num_res_blocks")
        nhc = _per_level(num_head_channels, n, "num_head_channels")
        for i in range(n):
            cfg.channels[i] = int(channels[i])
            cfg.attention_levels[i] = int(bool(attention_levels[i]))
            cfg.num_head_channels[i] = nhc[i]
            cfg.num_res_blocks[i] = nrb[i]
        cfg.norm_num_groups, cfg.norm_eps = int(norm_num_groups), float(norm_eps)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.block_out_channels = list(channels)
        _lib.check(_lib.lib().ldm_unet_create(C.byref(cfg), C.byref(self._h)))
        self._build_params()
        self._default_precision()

    def _zero_init(self):
        # MONAI zero_module(): every ResBlock conv2 and the output conv start at zero
        with torch.no_grad():
            for name, p in self.named_parameters():
                if ".conv2.conv." in name or name.startswith("out.2.conv."):
                    p.zero_()

    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, context: Optional[torch.Tensor] = None,
                class_labels=None, down_block_additional_residuals=None, mid_block_additional_residual=None,
                cond: Optional[torch.Tensor] = None) -> torch.Tensor:
        """eps_hat = UNet(x_t, t).  ``cond`` (extension) is a second tensor channel-concatenated after ``x``
        inside the input packing kernel, so mode="concat" callers need no torch.cat."""
        if context is not None or class_labels is not None or down_block_additional_residuals is not None \
                or mid_block_additional_residual is not None:
            raise NotImplementedError("context / class_labels / ControlNet residuals are not on the reference's path")
        if torch.is_grad_enabled() and (x.requires_grad or (cond is not None and cond.requires_grad)):
            raise NotImplementedError("gradients w.r.t. the UNet input are not implemented (the reference's trainer "
                                      "never asks for them: the noisy latent is built under no_grad)")
        self._need_cuda(x, "DiffusionModelUNet.forward")
        if x.dim() != 5:
            raise ValueError(f"expected [B, C, D, H, W], got {tuple(x.shape)}")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list()):
            return _UNetTrainFn.apply(self, x, timesteps, cond, *self._param_list())
        B, cx, D, H, W = x.shape
        x = x.detach().to(torch.float32).contiguous()
        cc = 0
        if cond is not None:
            cond = cond.detach().to(device=x.device, dtype=torch.float32).contiguous()
            if cond.dim() != 5 or cond.shape[0] != B or tuple(cond.shape[2:]) != (D, H, W):
                raise ValueError(f"cond must be [{B}, C, {D}, {H}, {W}], got {tuple(cond.shape)}")
            cc = cond.shape[1]
        t = timesteps.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        if t.numel() != B:
            raise ValueError(f"timesteps must have {B} entries, got {t.numel()}")
        self._sync_weights()
        L = _lib.lib()
        nbytes = L.ldm_unet_workspace_bytes(self._h, B, D, H, W)
        if nbytes == 0:
            raise _lib.LdmError((L.ldm_last_error() or b"workspace query failed").decode())
        ws = self._workspace(("unet", B, D, H, W), nbytes, x.device)
        if getattr(self, "_graph", False):
            # HIP-graph replay needs fixed addresses: stage the inputs through persistent tensors and hand out a persistent
            # output (valid until the next forward of this module: the sampling loop consumes eps_hat immediately)
            key = ("g", B, cx, cc, D, H, W, str(x.device))
            st = self._gstage.get(key)
            if st is None:
                st = (torch.empty_like(x), torch.empty_like(t), None if cond is None else torch.empty_like(cond),
                      torch.empty((B, self.out_channels, D, H, W), dtype=torch.float32, device=x.device))
                self._gstage[key] = st
            gx, gt, gc, out = st
            if gx.data_ptr() != x.data_ptr():
                gx.copy_(x)
            if gt.data_ptr() != t.data_ptr():
                gt.copy_(t)
            if gc is not None and gc.data_ptr() != cond.data_ptr():
                gc.copy_(cond)
            with torch.cuda.device(x.device):
                _lib.check(L.ldm_unet_forward(self._h, gx.data_ptr(), cx, _lib.ptr(gc), cc, gt.data_ptr(), out.data_ptr(),
                                              B, D, H, W, ws.data_ptr(), ws.numel(), _lib.current_stream()))
            return out
        out = torch.empty((B, self.out_channels, D, H, W), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.ldm_unet_forward(self._h, x.data_ptr(), cx, _lib.ptr(cond), cc, t.data_ptr(), out.data_ptr(),
                                          B, D, H, W, ws.data_ptr(), ws.numel(), _lib.current_stream()))
        return out


    def forward_taps(self, x: torch.Tensor, timesteps: torch.Tensor, cond: Optional[torch.Tensor] = None,
                     force: Optional[Dict[str, torch.Tensor]] = None):
        """Debug: ``(eps_hat, {block name: output as fp32 NCDHW})``.  With ``force`` (a dict holding a tensor for EVERY tap)
        each block output is overwritten with the given tensor after it has been exported, so every block runs on the
        reference's input (teacher forcing; tests/test_gpu_taps.py)."""
        self._need_cuda(x, "DiffusionModelUNet.forward_taps")
        x, cx, cond, cc, t = self._prep(x, timesteps, cond)
        B, _, D, H, W = x.shape
        self._sync_weights()
        layout, taps_out, taps_in, ws = self._tap_buffers("unet", B, D, H, W, force, x.device)
        out = torch.empty((B, self.out_channels, D, H, W), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ldm_unet_forward_taps(self._h, x.data_ptr(), cx, _lib.ptr(cond), cc, t.data_ptr(), out.data_ptr(),
                                                        B, D, H, W, taps_out.data_ptr(), _lib.ptr(taps_in), ws.data_ptr(), ws.numel(),
                                                        _lib.current_stream()))
        return out, self._tap_dict(layout, taps_out)

    def denoise_step(self, x: torch.Tensor, tbuf: torch.Tensor, sampler, cond: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One whole denoising step IN PLACE: ``x := sampler.step(UNet(x, tbuf), x)`` with the device-resident sampler
        (``scheduler.device_sampler(seed)``; ``sampler.reset(tbuf)`` first).  ``x`` must be a persistent contiguous fp32 CUDA
        tensor and ``tbuf`` a persistent fp32 [B] tensor: with ``enable_graph_replay`` the forward plan and the scheduler step
        replay as ONE HIP graph launch per call, nothing else runs on the host or the device between steps."""
        self._need_cuda(x, "DiffusionModelUNet.denoise_step")
        if not (x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 5 and tbuf.dtype == torch.float32 and tbuf.is_cuda):
            raise _lib.LdmError("denoise_step: x must be a contiguous fp32 [B, C, D, H, W] CUDA tensor and tbuf fp32 [B] on the same device")
        B, cx, D, H, W = x.shape
        cc = 0
        if cond is not None:
            if not (cond.is_cuda and cond.dtype == torch.float32 and cond.is_contiguous()):
                raise _lib.LdmError("denoise_step: cond must be a contiguous fp32 CUDA tensor")
            cc = cond.shape[1]
        self._sync_weights()
        L = _lib.lib()
        nbytes = L.ldm_unet_workspace_bytes(self._h, B, D, H, W)
        if nbytes == 0:
            raise _lib.LdmError((L.ldm_last_error() or b"workspace query failed").decode())
        ws = self._workspace(("unet", B, D, H, W), nbytes, x.device)
        key = ("eps", B, D, H, W, str(x.device))
        eps = self._ws.get(key)
        if eps is None:
            eps = self._ws[key] = torch.empty((B, self.out_channels, D, H, W), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.ldm_unet_denoise_step(self._h, sampler._h, x.data_ptr(), cx, _lib.ptr(cond), cc, tbuf.data_ptr(), eps.data_ptr(),
                                               B, D, H, W, ws.data_ptr(), ws.numel(), _lib.current_stream()))
        return x

    def enable_graph_replay(self, on: bool = True):
        """Inference: replay the forward plan as ONE HIP graph launch per call instead of ~150 kernel launches (same
        kernels and results; the host cost per step drops from ~1.6 ms to ~0.1 ms, which matters when many ranks share a
        host).  The returned tensor is then a persistent buffer that the next forward overwrites."""
        _lib.check(_lib.lib().ldm_model_set_graph_mode(self._h, 1 if on else 0))
        self._graph = bool(on)
        self._gstage = {}
        return self

    # -- training plan ------------------------------------------------------------------------------------------
    def _prep(self, x, timesteps, cond):
        B, cx, D, H, W = x.shape
        x = x.detach().to(torch.float32).contiguous()
        cc = 0
        if cond is not None:
            cond = cond.detach().to(device=x.device, dtype=torch.float32).contiguous()
            if cond.dim() != 5 or cond.shape[0] != B or tuple(cond.shape[2:]) != (D, H, W):
                raise ValueError(f"cond must be [{B}, C, {D}, {H}, {W}], got {tuple(cond.shape)}")
            cc = cond.shape[1]
        t = timesteps.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        if t.numel() != B:
            raise ValueError(f"timesteps must have {B} entries, got {t.numel()}")
        return x, cx, cond, cc, t

    def _train_forward(self, x, timesteps, cond):
        if not all(p.is_cuda for p in self._param_list()):
            raise _lib.LdmError("training needs the parameters on the GPU: call .to('cuda') first")
        x, cx, cond, cc, t = self._prep(x, timesteps, cond)
        B, _, D, H, W = x.shape
        self._sync_weights()
        L = _lib.lib()
        nbytes = L.ldm_unet_train_workspace_bytes(self._h, B, D, H, W)
        if nbytes == 0:
            raise _lib.LdmError((L.ldm_last_error() or b"workspace query failed").decode())
        ws = self._workspace(("train", B, D, H, W), nbytes, x.device)
        out = torch.empty((B, self.out_channels, D, H, W), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.ldm_unet_train_forward(self._h, x.data_ptr(), cx, _lib.ptr(cond), cc, t.data_ptr(), out.data_ptr(),
                                                B, D, H, W, ws.data_ptr(), ws.numel(), _lib.current_stream()))
        self._train_serial = getattr(self, "_train_serial", 0) + 1
        return out, (self._train_serial, B, D, H, W)

    def _train_backward(self, token, grad_out):
        serial, B, D, H, W = token
        if serial != getattr(self, "_train_serial", 0):
            raise _lib.LdmError("backward of a stale forward: the training workspace holds one forward at a time "
                                "(call backward before the next grad-enabled forward of the same module)")
        L = _lib.lib()
        pl = self._param_list()
        g = grad_out.detach().to(torch.float32).contiguous()
        flat_mode = getattr(self, "flat_grads", None) is not None
        total = int(L.ldm_model_param_numel_total(self._h))
        flat = self.flat_grads if flat_mode else torch.empty(total, dtype=torch.float32, device=g.device)
        ws = self._ws[("train", B, D, H, W)]
        with torch.cuda.device(g.device):
            _lib.check(L.ldm_unet_train_backward(self._h, g.data_ptr(), flat.data_ptr(), B, D, H, W, ws.data_ptr(), ws.numel(),
                                                 _lib.current_stream()))
        self._train_serial += 1                        # the workspace has been consumed
        offs = self._offsets()
        if flat_mode:                                  # .grad aliases flat_grads (flatten_parameters): nothing to hand to autograd
            for i, p in enumerate(pl):
                if p.grad is None or p.grad.data_ptr() != flat.data_ptr() + 4 * offs[i]:
                    p.grad = flat[offs[i]:offs[i] + p.numel()].view(p.shape)
            return [None] * len(pl)
        return [flat[offs[i]:offs[i] + p.numel()].view(p.shape) if p.requires_grad else None for i, p in enumerate(pl)]

    def _offsets(self):
        o = getattr(self, "_offs", None)
        if o is None:
            L = _lib.lib()
            o = [int(L.ldm_model_param_offset(self._h, i)) for i in range(L.ldm_model_num_params(self._h))]
            self._offs = o
        return o


class _VaeTrainFn(torch.autograd.Function):
    """(reconstruction, z_mu, z_sigma) = AutoencoderKL(images; params) with the hand-written backward plan."""

    @staticmethod
    def forward(ctx, module, x, eps, *params):
        ctx.module = module
        recon, z_mu, z_sigma, ctx.token = module._train_forward(x, eps)
        return recon, z_mu, z_sigma

    @staticmethod
    def backward(ctx, g_recon, g_mu, g_sigma):
        grads = ctx.module._train_backward(ctx.token, g_recon, g_mu, g_sigma)
        return (None, None, None) + tuple(grads)


class AutoencoderKL(_LdmModule):
    """MI355X-native AutoencoderKL (kwargs of ``autoencoder_def``, 3d_ldm/config/config_train_16g.json:7-28)."""

    def __init__(self, spatial_dims: int, in_channels: int = 1, out_channels: int = 1,
                 num_res_blocks: Union[Sequence[int], int] = (2, 2, 2, 2), channels: Sequence[int] = (32, 64, 64, 64),
                 attention_levels: Sequence[bool] = (False, False, True, True), latent_channels: int = 3,
                 norm_num_groups: int = 32, norm_eps: float = 1e-6, with_encoder_nonlocal_attn: bool = True,
                 with_decoder_nonlocal_attn: bool = True, use_checkpoint: bool = False, use_convtranspose: bool = False,
                 include_fc: bool = True, use_combined_linear: bool = False, use_flash_attention: bool = False):
        super().__init__()
        if use_convtranspose:
            raise NotImplementedError("use_convtranspose is not on the reference's path")
        n = len(channels)
        cfg = _lib.VaeCfg()
        cfg.spatial_dims, cfg.in_channels, cfg.out_channels = spatial_dims, in_channels, out_channels
        cfg.latent_channels, cfg.num_levels = latent_channels, n
        nrb = _per_level(num_res_blocks, n, "num_res_blocks")
        for i in range(n):
            cfg.channels[i] = int(channels[i])
            cfg.num_res_blocks[i] = nrb[i]
            cfg.attention_levels[i] = int(bool(attention_levels[i]))
        cfg.norm_num_groups, cfg.norm_eps = int(norm_num_groups), float(norm_eps)
        cfg.with_encoder_nonlocal_attn = int(bool(with_encoder_nonlocal_attn))
        cfg.with_decoder_nonlocal_attn = int(bool(with_decoder_nonlocal_attn))
        self.in_channels, self.out_channels, self.latent_channels = in_channels, out_channels, latent_channels
        self.factor = 2 ** (n - 1)
        _lib.check(_lib.lib().ldm_vae_create(C.byref(cfg), C.byref(self._h)))
        self._build_params()
        self._default_precision()

    # -- encode ----------------------------------------------------------------------------------------
    def _encode(self, x: torch.Tensor, eps: Optional[torch.Tensor], want_z: bool):
        self._need_cuda(x, "AutoencoderKL.encode")
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("backward through the HIP AutoencoderKL is not implemented yet (inference only)")
        if x.dim() != 5 or x.shape[1] != self.in_channels:
            raise _lib.LdmError(f"AutoencoderKL.encode: expected an image [B, {self.in_channels}, D, H, W], got {tuple(x.shape)}")
        B, _, D, H, W = x.shape
        if eps is not None and tuple(eps.shape) != (B, self.latent_channels, D // self.factor, H // self.factor, W // self.factor):
            raise _lib.LdmError(f"AutoencoderKL.encode: eps has shape {tuple(eps.shape)}, expected "
                                f"{(B, self.latent_channels, D // self.factor, H // self.factor, W // self.factor)}")
        x = x.detach().to(torch.float32).contiguous()
        self._sync_weights()
        L = _lib.lib()
        nbytes = L.ldm_vae_encode_workspace_bytes(self._h, B, D, H, W)
        if nbytes == 0:
            raise _lib.LdmError((L.ldm_last_error() or b"workspace query failed").decode())
        ws = self._workspace(("enc", B, D, H, W), nbytes, x.device)
        f = self.factor
        shp = (B, self.latent_channels, D // f, H // f, W // f)
        z_mu = torch.empty(shp, dtype=torch.float32, device=x.device)
        z_sigma = torch.empty_like(z_mu)
        z = torch.empty_like(z_mu) if want_z else None
        if eps is not None:
            eps = eps.detach().to(device=x.device, dtype=torch.float32).contiguous()
        with torch.cuda.device(x.device):
            _lib.check(L.ldm_vae_encode(self._h, x.data_ptr(), _lib.ptr(eps), z_mu.data_ptr(), z_sigma.data_ptr(),
                                        _lib.ptr(z), B, D, H, W, ws.data_ptr(), ws.numel(), _lib.current_stream()))
        return z_mu, z_sigma, z

    def encode(self, x: torch.Tensor):
        z_mu, z_sigma, _ = self._encode(x, None, False)
        return z_mu, z_sigma

    def sampling(self, z_mu: torch.Tensor, z_sigma: torch.Tensor) -> torch.Tensor:
        return z_mu + torch.randn_like(z_sigma) * z_sigma

    def encode_stage_2_inputs(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None) -> torch.Tensor:
        """z = mu + sigma * eps, eps ~ N(0, I) drawn with torch.randn_like unless given (tests pass it explicitly)."""
        if eps is None:
            B, _, D, H, W = x.shape
            f = self.factor
            eps = torch.randn((B, self.latent_channels, D // f, H // f, W // f), dtype=torch.float32, device=x.device)
        return self._encode(x, eps, True)[2]

    # -- decode ----------------------------------------------------------------------------------------
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        self._need_cuda(z, "AutoencoderKL.decode")
        if torch.is_grad_enabled() and z.requires_grad:
            raise NotImplementedError("backward through the HIP AutoencoderKL is not implemented yet (inference only)")
        if z.dim() != 5 or z.shape[1] != self.latent_channels:     # the C entry takes a raw pointer: a wrong channel count would read past the tensor
            raise _lib.LdmError(f"AutoencoderKL.decode: expected a latent [B, {self.latent_channels}, d, h, w], got {tuple(z.shape)}")
        B, _, d, h, w = z.shape
        z = z.detach().to(torch.float32).contiguous()
        self._sync_weights()
        L = _lib.lib()
        nbytes = L.ldm_vae_decode_workspace_bytes(self._h, B, d, h, w)
        if nbytes == 0:
            raise _lib.LdmError((L.ldm_last_error() or b"workspace query failed").decode())
        ws = self._workspace(("dec", B, d, h, w), nbytes, z.device)
        f = self.factor
        out = torch.empty((B, self.out_channels, d * f, h * f, w * f), dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device):
            _lib.check(L.ldm_vae_decode(self._h, z.data_ptr(), out.data_ptr(), B, d, h, w, ws.data_ptr(), ws.numel(),
                                        _lib.current_stream()))
        return out

    def encode_taps(self, x: torch.Tensor, force: Optional[Dict[str, torch.Tensor]] = None):
        """Debug: ``(z_mu, z_sigma, {block name: output})`` (see DiffusionModelUNet.forward_taps)."""
        self._need_cuda(x, "AutoencoderKL.encode_taps")
        B, _, D, H, W = x.shape
        x = x.detach().to(torch.float32).contiguous()
        self._sync_weights()
        layout, taps_out, taps_in, ws = self._tap_buffers("enc", B, D, H, W, force, x.device)
        f = self.factor
        z_mu = torch.empty((B, self.latent_channels, D // f, H // f, W // f), dtype=torch.float32, device=x.device)
        z_sigma = torch.empty_like(z_mu)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().ldm_vae_encode_taps(self._h, x.data_ptr(), None, z_mu.data_ptr(), z_sigma.data_ptr(), None, B, D, H, W,
                                                      taps_out.data_ptr(), _lib.ptr(taps_in), ws.data_ptr(), ws.numel(),
                                                      _lib.current_stream()))
        return z_mu, z_sigma, self._tap_dict(layout, taps_out)

    def decode_taps(self, z: torch.Tensor, force: Optional[Dict[str, torch.Tensor]] = None):
        """Debug: ``(reconstruction, {block name: output})``."""
        self._need_cuda(z, "AutoencoderKL.decode_taps")
        B, _, d, h, w = z.shape
        z = z.detach().to(torch.float32).contiguous()
        self._sync_weights()
        layout, taps_out, taps_in, ws = self._tap_buffers("dec", B, d, h, w, force, z.device)
        f = self.factor
        out = torch.empty((B, self.out_channels, d * f, h * f, w * f), dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device):
            _lib.check(_lib.lib().ldm_vae_decode_taps(self._h, z.data_ptr(), out.data_ptr(), B, d, h, w, taps_out.data_ptr(),
                                                      _lib.ptr(taps_in), ws.data_ptr(), ws.numel(), _lib.current_stream()))
        return out, self._tap_dict(layout, taps_out)

    def decode_stage_2_outputs(self, z: torch.Tensor) -> torch.Tensor:
        return self.decode(z)

    def reconstruct(self, x: torch.Tensor) -> torch.Tensor:
        return self.decode(self.encode(x)[0])

    def forward(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None):
        """-> (reconstruction, z_mu, z_sigma)  (3d_ldm/train_autoencoder.py:366).  Differentiable w.r.t. the parameters
        (``loss_g.backward()`` of the stage-1 trainer); ``eps`` (extension) passes the sampling draw explicitly."""
        B, _, D, H, W = x.shape
        f = self.factor
        if eps is None:
            eps = torch.randn((B, self.latent_channels, D // f, H // f, W // f), dtype=torch.float32, device=x.device)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list()):
            if x.requires_grad:
                raise NotImplementedError("gradients w.r.t. the AutoencoderKL input are not implemented")
            self._need_cuda(x, "AutoencoderKL.forward")
            return _VaeTrainFn.apply(self, x, eps, *self._param_list())
        z_mu, z_sigma, z = self._encode(x, eps, True)
        return self.decode(z), z_mu, z_sigma

    # -- training plan ------------------------------------------------------------------------------------------
    def _train_forward(self, x, eps):
        if not all(p.is_cuda for p in self._param_list()):
            raise _lib.LdmError("training needs the parameters on the GPU: call .to('cuda') first")
        B, _, D, H, W = x.shape
        x = x.detach().to(torch.float32).contiguous()
        eps = eps.detach().to(device=x.device, dtype=torch.float32).contiguous()
        self._sync_weights()
        L = _lib.lib()
        nbytes = L.ldm_vae_train_workspace_bytes(self._h, B, D, H, W)
        if nbytes == 0:
            raise _lib.LdmError((L.ldm_last_error() or b"workspace query failed").decode())
        ws = self._workspace(("train", B, D, H, W), nbytes, x.device)
        f = self.factor
        recon = torch.empty((B, self.out_channels, D, H, W), dtype=torch.float32, device=x.device)
        z_mu = torch.empty((B, self.latent_channels, D // f, H // f, W // f), dtype=torch.float32, device=x.device)
        z_sigma = torch.empty_like(z_mu)
        with torch.cuda.device(x.device):
            _lib.check(L.ldm_vae_train_forward(self._h, x.data_ptr(), eps.data_ptr(), recon.data_ptr(), z_mu.data_ptr(), z_sigma.data_ptr(),
                                               B, D, H, W, ws.data_ptr(), ws.numel(), _lib.current_stream()))
        self._train_serial = getattr(self, "_train_serial", 0) + 1
        return recon, z_mu, z_sigma, (self._train_serial, B, D, H, W)

    def _train_backward(self, token, g_recon, g_mu, g_sigma):
        serial, B, D, H, W = token
        if serial != getattr(self, "_train_serial", 0):
            raise _lib.LdmError("backward of a stale forward: the training workspace holds one forward at a time")
        L = _lib.lib()
        pl = self._param_list()
        dev = pl[0].device

        def prep(g):
            return None if g is None else g.detach().to(torch.float32).contiguous()
        g_recon, g_mu, g_sigma = prep(g_recon), prep(g_mu), prep(g_sigma)
        if g_recon is None:
            f = self.factor
            g_recon = torch.zeros((B, self.out_channels, D, H, W), dtype=torch.float32, device=dev)
        flat_mode = getattr(self, "flat_grads", None) is not None
        total = int(L.ldm_model_param_numel_total(self._h))
        flat = self.flat_grads if flat_mode else torch.empty(total, dtype=torch.float32, device=dev)
        ws = self._ws[("train", B, D, H, W)]
        with torch.cuda.device(dev):
            _lib.check(L.ldm_vae_train_backward(self._h, g_recon.data_ptr(), _lib.ptr(g_mu), _lib.ptr(g_sigma), flat.data_ptr(),
                                                B, D, H, W, ws.data_ptr(), ws.numel(), _lib.current_stream()))
        self._train_serial += 1
        if getattr(self, "_offs", None) is None:
            self._offs = [int(L.ldm_model_param_offset(self._h, i)) for i in range(len(pl))]
        offs = self._offs
        if flat_mode:
            for i, p in enumerate(pl):
                if p.grad is None or p.grad.data_ptr() != flat.data_ptr() + 4 * offs[i]:
                    p.grad = flat[offs[i]:offs[i] + p.numel()].view(p.shape)
            return [None] * len(pl)
        return [flat[offs[i]:offs[i] + p.numel()].view(p.shape) if p.requires_grad else None for i, p in enumerate(pl)]
