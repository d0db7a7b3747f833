"""PatchDiscriminator + LSGAN adversarial loss of the stage-1 trainer, on the HIP kernels.

Reference: 3d_ldm/train_autoencoder.py:150-158 builds ``monai.networks.nets.PatchDiscriminator(spatial_dims=3, num_layers_d=3,
channels=32, in_channels=1, out_channels=1, norm="INSTANCE")`` and :225,407-424,454-494 use it through
``PatchAdversarialLoss(criterion="least_squares")``: the generator term ``mse(D(recon)[-1], 1)`` after the warm-up epochs and the
discriminator step ``0.5 * (mse(D(recon.detach())[-1], 0) + mse(D(images)[-1], 1))``.

[MONAI-ext] layer structure restated from MONAI's source (not vendored, not installable here):
    initial_conv : Conv3d(in, C, k4 s2 p1, bias) -> LeakyReLU(0.2)
    0 .. L-1     : Conv3d(C 2^l, C 2^(l+1), k4, stride 2 (1 for the last), p1, no bias) -> InstanceNorm3d -> LeakyReLU(0.2)
    final_conv   : Conv3d(C 2^L, out, k4 s1 p1, bias)
state_dict keys ``initial_conv.conv.{weight,bias}``, ``<l>.conv.weight``, ``final_conv.conv.{weight,bias}``; weights ~ N(0, 0.02).
``forward`` returns the list of every layer's output (the trainer takes ``[-1]``).

MI355X side: every 4^3 convolution is ``ldm_op_im2col`` + the bf16-MFMA 1x1 GEMM (``ldm_op_conv3d`` with ksize 1); its data gradient is
the same GEMM on the transposed weights followed by ``ldm_op_col2im`` (gather form, no atomics), its weight gradient
``ldm_op_conv3d_wgrad``; InstanceNorm + LeakyReLU is ``ldm_op_group_norm`` with groups = C and activation code 2.  Activations are
bf16 NDHWC between the layers; the module boundary is fp32 NCDHW like the rest of the library.  No CPU fallback.

``precision="fp32"`` (``set_precision``, or ``LDM_PRECISION=fp32`` / ``--precision fp32`` at construction): the reference trains the
discriminator in fp32 when AMP is off (3d_ldm/train_autoencoder.py:150-158,454-494).  The same structure then runs on fp32 NDHWC
tensors through the ``ldm_op_*_f32`` entries: im2col + exact-fp32-MFMA GEMM, col2im, fp32 weight gradient, InstanceNorm + LeakyReLU on
the fp32 GroupNorm kernels.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from . import _lib


def _rup(v, m):
    return (v + m - 1) // m * m


def _st():
    return _lib.current_stream()


class _ConvFn(torch.autograd.Function):
    """y[M][couts] (bf16 NDHWC) = conv_k(x NDHWC bf16) + bias, as im2col + GEMM.  Saves the column matrix for the weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, geom):
        L = _lib.lib()
        N, D, H, W, Cs, C, k, stride, pad = geom
        cout, cin = weight.shape[0], weight.shape[1]
        Do, Ho, Wo = [(s + 2 * pad - k) // stride + 1 for s in (D, H, W)]
        M, taps = N * Do * Ho * Wo, k ** 3
        Kp, cout_pad, couts = _rup(taps * C, 32), _rup(cout, 64), _rup(cout, 32)
        dev = x.device
        col = torch.empty((M, Kp), dtype=torch.bfloat16, device=dev)
        _lib.check(L.ldm_op_im2col(x.data_ptr(), col.data_ptr(), N, D, H, W, Cs, C, k, stride, pad, Kp, _st()))
        # MONAI layout [cout][cin][kd][kh][kw] -> GEMM rows [cout_pad][Kp] with column = tap * C + c (C == cin)
        wm = torch.zeros((cout_pad, Kp), dtype=torch.bfloat16, device=dev)
        wm[:cout, :taps * C] = weight.detach().permute(0, 2, 3, 4, 1).reshape(cout, taps * cin).to(torch.bfloat16)
        bp = torch.zeros((cout_pad,), dtype=torch.float32, device=dev)
        if bias is not None:
            bp[:cout] = bias.detach().float()
        y = torch.empty((N, Do, Ho, Wo, couts), dtype=torch.bfloat16, device=dev)
        _lib.check(L.ldm_op_conv3d(col.data_ptr(), Kp, None, 0, wm.data_ptr(), bp.data_ptr(), None, 0, None, 0, None, None, None, 0, None,
                                   y.data_ptr(), None, 1, M, 1, 1, 1, 1, 0, 0, cout, cout_pad, 0, 1, None, 0, _st()))   # splitk 1: no slab scratch
        ctx.save_for_backward(col, wm)
        ctx.geom, ctx.shape, ctx.has_bias = geom, (cout, cin, Kp, cout_pad, couts, M, Do, Ho, Wo), bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        col, wm = ctx.saved_tensors
        N, D, H, W, Cs, C, k, stride, pad = ctx.geom
        cout, cin, Kp, cout_pad, couts, M, Do, Ho, Wo = ctx.shape
        taps = k ** 3
        dev = dy.device
        dy = dy.contiguous()
        # weight gradient: dW[co][kk] = sum_m dy[m][co] col[m][kk]   (ksize-1 weight-gradient kernel over the column matrix)
        dw = torch.empty((cout, Kp), dtype=torch.float32, device=dev)
        _lib.check(L.ldm_op_conv3d_wgrad(dy.data_ptr(), couts, col.data_ptr(), Kp, dw.data_ptr(), cout, Kp, 1, M, 1, 1, 1, 1, 0, 0, 1, _st()))
        gw = dw[:, :taps * C].reshape(cout, k, k, k, cin).permute(0, 4, 1, 2, 3).contiguous()
        gb = None
        if ctx.has_bias:
            gb = dy.reshape(M, couts)[:, :cout].float().sum(0)
        gx = None
        if ctx.needs_input_grad[0]:
            # data gradient: dcol = dy W (the same GEMM on the transposed weight matrix), then the adjoint of im2col
            wt = torch.empty((_rup(Kp, 64), couts), dtype=torch.bfloat16, device=dev)
            _lib.check(L.ldm_op_weight_flip_transpose(wm.data_ptr(), wt.data_ptr(), 1, cout, cout_pad, Kp, _st()))
            dcol = torch.empty((M, Kp), dtype=torch.bfloat16, device=dev)
            _lib.check(L.ldm_op_conv3d(dy.data_ptr(), couts, None, 0, wt.data_ptr(), None, None, 0, None, 0, None, None, None, 0, None,
                                       dcol.data_ptr(), None, 1, M, 1, 1, 1, 1, 0, 0, Kp, _rup(Kp, 64), 0, 1, None, 0, _st()))
            gx = torch.empty((N, D, H, W, Cs), dtype=torch.bfloat16, device=dev)
            _lib.check(L.ldm_op_col2im(dcol.data_ptr(), gx.data_ptr(), N, D, H, W, Cs, C, k, stride, pad, Kp, _st()))
        return gx, gw, gb, None


class _InstanceNormLeakyFn(torch.autograd.Function):
    """LeakyReLU(0.2)(InstanceNorm3d(x)) on bf16 NDHWC: GroupNorm kernels with one channel per group, no affine, eps 1e-5."""

    @staticmethod
    def forward(ctx, x, C):
        L = _lib.lib()
        N, DHW = x.shape[0], x.shape[1] * x.shape[2] * x.shape[3]
        dev = x.device
        ones = torch.ones((C,), dtype=torch.float32, device=dev)
        zeros = torch.zeros((C,), dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        nb = L.ldm_op_group_norm_scratch_bytes(N, C, DHW)
        scratch = torch.empty((nb,), dtype=torch.uint8, device=dev)
        _lib.check(L.ldm_op_group_norm(x.data_ptr(), C, None, 0, ones.data_ptr(), zeros.data_ptr(), C, 1e-5, 2, y.data_ptr(), N, DHW,
                                       scratch.data_ptr(), scratch.numel(), _st()))
        ctx.save_for_backward(x, ones, zeros)
        ctx.C = C
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, ones, zeros = ctx.saved_tensors
        C = ctx.C
        N, DHW = x.shape[0], x.shape[1] * x.shape[2] * x.shape[3]
        dev = x.device
        dx = torch.empty_like(x)
        dg, db = torch.empty((C,), dtype=torch.float32, device=dev), torch.empty((C,), dtype=torch.float32, device=dev)
        nb = L.ldm_op_group_norm_bwd_scratch_bytes(N, C, DHW, C)
        scratch = torch.empty((nb,), dtype=torch.uint8, device=dev)
        _lib.check(L.ldm_op_group_norm_bwd(dy.contiguous().data_ptr(), x.data_ptr(), C, None, 0, ones.data_ptr(), zeros.data_ptr(), C, 1e-5, 2,
                                           None, None, dx.data_ptr(), None, dg.data_ptr(), db.data_ptr(), N, DHW, scratch.data_ptr(),
                                           scratch.numel(), _st()))
        return dx, None


class _LeakyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        _lib.check(_lib.lib().ldm_op_leaky_relu(x.data_ptr(), y.data_ptr(), x.numel(), 0.2, _st()))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().ldm_op_leaky_relu_bwd(x.data_ptr(), dy.contiguous().data_ptr(), dx.data_ptr(), x.numel(), 0.2, _st()))
        return dx


class _PackFn(torch.autograd.Function):
    """fp32 NCDHW -> bf16 NDHWC (channels padded to 32) and its adjoint."""

    @staticmethod
    def forward(ctx, x, Cs):
        N, C = x.shape[:2]
        DHW = x[0, 0].numel()
        out = torch.empty((N, *x.shape[2:], Cs), dtype=torch.bfloat16, device=x.device)
        _lib.check(_lib.lib().ldm_op_pack_ncdhw(x.contiguous().data_ptr(), out.data_ptr(), N, C, Cs, DHW, _st()))
        ctx.meta = (N, C, Cs, DHW, tuple(x.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, Cs, DHW, shape = ctx.meta
        out = torch.empty(shape, dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().ldm_op_unpack_ndhwc(g.contiguous().data_ptr(), out.data_ptr(), N, C, Cs, DHW, _st()))
        return out, None


class _UnpackFn(torch.autograd.Function):
    """bf16 NDHWC -> fp32 NCDHW (first C channels) and its adjoint."""

    @staticmethod
    def forward(ctx, a, C):
        N, Cs = a.shape[0], a.shape[-1]
        DHW = a.shape[1] * a.shape[2] * a.shape[3]
        out = torch.empty((N, C, *a.shape[1:4]), dtype=torch.float32, device=a.device)
        _lib.check(_lib.lib().ldm_op_unpack_ndhwc(a.data_ptr(), out.data_ptr(), N, C, Cs, DHW, _st()))
        ctx.meta = (N, C, Cs, DHW, tuple(a.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, Cs, DHW, shape = ctx.meta
        out = torch.empty(shape, dtype=torch.bfloat16, device=g.device)
        _lib.check(_lib.lib().ldm_op_pack_ncdhw(g.contiguous().float().data_ptr(), out.data_ptr(), N, C, Cs, DHW, _st()))
        return out, None


# ------------------------------------------------------------------------------------------------ fp32 forms (precision="fp32")
class _ConvFn32(torch.autograd.Function):
    """y[M][couts] (fp32 NDHWC) = conv_k(x NDHWC fp32) + bias as im2col + the exact-fp32 GEMM; saves the column matrix."""

    @staticmethod
    def forward(ctx, x, weight, bias, geom):
        L = _lib.lib()
        N, D, H, W, Cs, C, k, stride, pad = geom
        cout, cin = weight.shape[0], weight.shape[1]
        Do, Ho, Wo = [(s + 2 * pad - k) // stride + 1 for s in (D, H, W)]
        M, taps = N * Do * Ho * Wo, k ** 3
        Kp, cout_pad, couts = _rup(taps * C, 16), _rup(cout, 64), _rup(cout, 16)
        dev = x.device
        col = torch.empty((M, Kp), dtype=torch.float32, device=dev)
        _lib.check(L.ldm_op_im2col_f32(x.data_ptr(), col.data_ptr(), N, D, H, W, Cs, C, k, stride, pad, Kp, _st()))
        wm = torch.zeros((cout_pad, Kp), dtype=torch.float32, device=dev)
        wm[:cout, :taps * C] = weight.detach().float().permute(0, 2, 3, 4, 1).reshape(cout, taps * cin)
        bp = torch.zeros((cout_pad,), dtype=torch.float32, device=dev)
        if bias is not None:
            bp[:cout] = bias.detach().float()
        y = torch.empty((N, Do, Ho, Wo, couts), dtype=torch.float32, device=dev)
        _lib.check(L.ldm_op_gemm_f32(col.data_ptr(), Kp, wm.data_ptr(), bp.data_ptr(), y.data_ptr(), M, cout, cout_pad, couts, _st()))
        ctx.save_for_backward(col, wm)
        ctx.geom, ctx.shape, ctx.has_bias = geom, (cout, cin, Kp, cout_pad, couts, M, Do, Ho, Wo), bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        col, wm = ctx.saved_tensors
        N, D, H, W, Cs, C, k, stride, pad = ctx.geom
        cout, cin, Kp, cout_pad, couts, M, Do, Ho, Wo = ctx.shape
        taps = k ** 3
        dev = dy.device
        dy = dy.contiguous().float()
        ks = max(1, min(16, M // 4096))                          # row ranges of the contraction: partial matrices summed below
        dw = torch.empty((ks, cout, Kp), dtype=torch.float32, device=dev)
        _lib.check(L.ldm_op_gemm_wgrad_f32(dy.data_ptr(), couts, col.data_ptr(), Kp, dw.data_ptr(), cout, M, ks, _st()))
        dw = dw.sum(0) if ks > 1 else dw[0]
        gw = dw[:, :taps * C].reshape(cout, k, k, k, cin).permute(0, 4, 1, 2, 3).contiguous()
        gb = dy.reshape(M, couts)[:, :cout].sum(0) if ctx.has_bias else None
        gx = None
        if ctx.needs_input_grad[0]:
            # data gradient: dcol[M][Kp] = dy[M][couts] W[couts][Kp]: the same GEMM with the transposed weight matrix as its "weights"
            kp_pad = _rup(Kp, 64)
            wt = torch.zeros((kp_pad, couts), dtype=torch.float32, device=dev)
            wt[:Kp, :min(couts, cout_pad)] = wm[:couts].t()
            dcol = torch.empty((M, Kp), dtype=torch.float32, device=dev)
            _lib.check(L.ldm_op_gemm_f32(dy.data_ptr(), couts, wt.data_ptr(), None, dcol.data_ptr(), M, Kp, kp_pad, Kp, _st()))
            gx = torch.empty((N, D, H, W, Cs), dtype=torch.float32, device=dev)
            _lib.check(L.ldm_op_col2im_f32(dcol.data_ptr(), gx.data_ptr(), N, D, H, W, Cs, C, k, stride, pad, Kp, _st()))
        return gx, gw, gb, None


class _InstanceNormLeakyFn32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, C):
        L = _lib.lib()
        N, DHW = x.shape[0], x.shape[1] * x.shape[2] * x.shape[3]
        dev = x.device
        ones = torch.ones((C,), dtype=torch.float32, device=dev)
        zeros = torch.zeros((C,), dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        scratch = torch.empty((L.ldm_op_group_norm_f32_scratch_bytes(N, C, DHW, C),), dtype=torch.uint8, device=dev)
        _lib.check(L.ldm_op_group_norm_f32(x.data_ptr(), C, ones.data_ptr(), zeros.data_ptr(), C, 1e-5, 2, y.data_ptr(), N, DHW,
                                           scratch.data_ptr(), scratch.numel(), _st()))
        ctx.save_for_backward(x, ones, zeros)
        ctx.C = C
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, ones, zeros = ctx.saved_tensors
        C = ctx.C
        N, DHW = x.shape[0], x.shape[1] * x.shape[2] * x.shape[3]
        dev = x.device
        dx = torch.empty_like(x)
        dg, db = torch.empty((C,), dtype=torch.float32, device=dev), torch.empty((C,), dtype=torch.float32, device=dev)
        scratch = torch.empty((L.ldm_op_group_norm_f32_scratch_bytes(N, C, DHW, C),), dtype=torch.uint8, device=dev)
        _lib.check(L.ldm_op_group_norm_bwd_f32(dy.contiguous().float().data_ptr(), x.data_ptr(), C, ones.data_ptr(), zeros.data_ptr(), C, 1e-5, 2,
                                               dx.data_ptr(), dg.data_ptr(), db.data_ptr(), N, DHW, scratch.data_ptr(), scratch.numel(), _st()))
        return dx, None


class _LeakyFn32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        _lib.check(_lib.lib().ldm_op_leaky_relu_f32(x.data_ptr(), y.data_ptr(), x.numel(), 0.2, _st()))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().ldm_op_leaky_relu_bwd_f32(x.data_ptr(), dy.contiguous().float().data_ptr(), dx.data_ptr(), x.numel(), 0.2, _st()))
        return dx


class _PackFn32(torch.autograd.Function):
    """fp32 NCDHW -> fp32 NDHWC (channels padded to 16) and its adjoint."""

    @staticmethod
    def forward(ctx, x, Cs):
        N, C = x.shape[:2]
        DHW = x[0, 0].numel()
        out = torch.empty((N, *x.shape[2:], Cs), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().ldm_op_pack_ncdhw_f32(x.contiguous().data_ptr(), out.data_ptr(), N, C, Cs, DHW, _st()))
        ctx.meta = (N, C, Cs, DHW, tuple(x.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, Cs, DHW, shape = ctx.meta
        out = torch.empty(shape, dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().ldm_op_unpack_ndhwc_f32(g.contiguous().float().data_ptr(), out.data_ptr(), N, C, Cs, DHW, _st()))
        return out, None


class _UnpackFn32(torch.autograd.Function):
    """fp32 NDHWC -> fp32 NCDHW (first C channels) and its adjoint."""

    @staticmethod
    def forward(ctx, a, C):
        N, Cs = a.shape[0], a.shape[-1]
        DHW = a.shape[1] * a.shape[2] * a.shape[3]
        out = torch.empty((N, C, *a.shape[1:4]), dtype=torch.float32, device=a.device)
        _lib.check(_lib.lib().ldm_op_unpack_ndhwc_f32(a.data_ptr(), out.data_ptr(), N, C, Cs, DHW, _st()))
        ctx.meta = (N, C, Cs, DHW, tuple(a.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, Cs, DHW, shape = ctx.meta
        out = torch.empty(shape, dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().ldm_op_pack_ncdhw_f32(g.contiguous().float().data_ptr(), out.data_ptr(), N, C, Cs, DHW, _st()))
        return out, None


class _Conv(nn.Module):
    """``Convolution`` wrapper of MONAI: parameters live under ``.conv``."""

    def __init__(self, cin, cout, bias):
        super().__init__()
        self.conv = nn.Module()
        self.conv.weight = nn.Parameter(torch.empty(cout, cin, 4, 4, 4))
        self.conv.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.normal_(self.conv.weight, 0.0, 0.02)            # PatchDiscriminator.initialise_weights
        if bias:
            nn.init.zeros_(self.conv.bias)


class PatchDiscriminator(nn.Module):
    def __init__(self, spatial_dims: int = 3, channels: int = 32, in_channels: int = 1, out_channels: int = 1, num_layers_d: int = 3,
                 kernel_size: int = 4, norm: str = "INSTANCE", bias: bool = False, padding: int = 1, dropout: float = 0.0, **unused):
        super().__init__()
        if spatial_dims != 3 or kernel_size != 4 or padding != 1 or str(norm).upper() != "INSTANCE" or dropout:
            raise NotImplementedError("PatchDiscriminator: only the reference's configuration (3-D, k4 p1, INSTANCE norm, no dropout) "
                                      "is implemented (3d_ldm/train_autoencoder.py:151-158)")
        if channels % 32:
            raise NotImplementedError("PatchDiscriminator: channels must be a multiple of 32")
        self.in_channels, self.out_channels, self.num_layers_d = in_channels, out_channels, num_layers_d
        import os
        self.precision = "fp32" if os.environ.get("LDM_PRECISION", "bf16").lower() == "fp32" else "bf16"
        self.add_module("initial_conv", _Conv(in_channels, channels, True))
        cin, cout = channels, channels * 2
        for l_ in range(num_layers_d):
            self.add_module(str(l_), _Conv(cin, cout, bias))
            cin, cout = cout, cout * 2
        self.add_module("final_conv", _Conv(cin, out_channels, True))

    def set_precision(self, precision: str) -> "PatchDiscriminator":
        """"bf16" (default) or "fp32": the reference's arithmetic when AMP is off (3d_ldm/train_autoencoder.py:150-158,454-494)."""
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.precision = precision
        return self

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        if not x.is_cuda:
            raise _lib.LdmError("PatchDiscriminator: CUDA tensors only (no CPU fallback)")
        N, C, D, H, W = x.shape
        if C != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {C}")
        outs = []
        f32 = self.precision == "fp32"
        Pack, Unpack, Conv, Leaky, Norm = ((_PackFn32, _UnpackFn32, _ConvFn32, _LeakyFn32, _InstanceNormLeakyFn32) if f32 else
                                           (_PackFn, _UnpackFn, _ConvFn, _LeakyFn, _InstanceNormLeakyFn))
        with torch.cuda.device(x.device):
            Cs = _rup(C, 16 if f32 else 32)
            h = Pack.apply(x.float(), Cs)
            geom = (N, D, H, W, Cs, C, 4, 2, 1)
            layer = self.initial_conv
            h = Leaky.apply(Conv.apply(h, layer.conv.weight, layer.conv.bias, geom))
            c = layer.conv.weight.shape[0]
            outs.append((h, c))
            for l_ in range(self.num_layers_d):
                layer = getattr(self, str(l_))
                stride = 1 if l_ == self.num_layers_d - 1 else 2
                geom = (N, h.shape[1], h.shape[2], h.shape[3], h.shape[4], c, 4, stride, 1)
                h = Conv.apply(h, layer.conv.weight, layer.conv.bias, geom)
                c = layer.conv.weight.shape[0]
                h = Norm.apply(h, c)
                outs.append((h, c))
            layer = self.final_conv
            geom = (N, h.shape[1], h.shape[2], h.shape[3], h.shape[4], c, 4, 1, 1)
            h = Conv.apply(h, layer.conv.weight, layer.conv.bias, geom)
            outs.append((h, self.out_channels))
            return [Unpack.apply(t, cc) for t, cc in outs]


class PatchAdversarialLoss(nn.Module):
    """``criterion="least_squares"`` (LSGAN), the only one the reference uses (3d_ldm/train_autoencoder.py:225): the mean squared
    distance of the discriminator's last output to 1 (real target) or 0 (fake target); ``for_discriminator`` only matters for the
    hinge criteria MONAI also offers."""

    def __init__(self, criterion: str = "least_squares", **unused):
        super().__init__()
        if criterion != "least_squares":
            raise NotImplementedError("only criterion='least_squares' is on the reference's path")

    def forward(self, logits, target_is_real: bool, for_discriminator: bool = True) -> torch.Tensor:
        if isinstance(logits, (list, tuple)):
            logits = logits[-1]
        target = 1.0 if target_is_real else 0.0
        return torch.mean((logits.float() - target) ** 2)
