"""Test helpers: layout packing for the operator-level ABI and error metrics."""
import torch


def rup(v, m):
    return (v + m - 1) // m * m


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def to_ndhwc_bf16(x: torch.Tensor, cs: int | None = None) -> torch.Tensor:
    """fp32 NCDHW -> bf16 NDHWC with channels zero-padded to cs."""
    n, c = x.shape[:2]
    cs = cs or rup(c, 32)
    out = torch.zeros((n, *x.shape[2:], cs), dtype=torch.bfloat16)
    out[..., :c] = x.permute(0, 2, 3, 4, 1).to(torch.bfloat16)
    return out.contiguous()


def from_ndhwc(x: torch.Tensor, c: int) -> torch.Tensor:
    """bf16 NDHWC -> fp32 NCDHW, first c channels."""
    return x[..., :c].to(torch.float32).permute(0, 4, 1, 2, 3).contiguous()


def pack_conv_weight(w: torch.Tensor, cin_s: int, cout_pad: int) -> torch.Tensor:
    """[cout][cin][k][k][k] fp32 -> [k^3][cout_pad][cin_s] bf16 (the arena layout of libldm3d)."""
    cout, cin = w.shape[:2]
    taps = w[0, 0].numel()
    out = torch.zeros((taps, cout_pad, cin_s), dtype=torch.bfloat16)
    out[:, :cout, :cin] = w.reshape(cout, cin, taps).permute(2, 0, 1).to(torch.bfloat16)
    return out.contiguous()


def pad_vec(v: torch.Tensor, n: int) -> torch.Tensor:
    out = torch.zeros((n,), dtype=torch.float32)
    out[: v.numel()] = v
    return out
