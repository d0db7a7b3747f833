"""One-launch split-K finalize + GroupNorm (csrc/fin_gn.h, -m gpu).

With ``LDM_FIN_GN=1`` the plans fold a split-K conv's slabs and normalise the folded tensor in ONE launch (two phases separated
by a grid-wide barrier) wherever the GroupNorm directly follows the conv.  It measured slower than the two launches (fin_gn.h
has the numbers) and is off by default; it stays in the tree, tested, as the record of that experiment.  The kernel runs the
bodies of the two kernels it replaces in the same summation order, so the bar is bit-exact equality with the default plans --
for inference, graph replay, batch > 1, concatenated skip inputs (up blocks) and the training plan's forward + backward; and
the barrier must never time out (``ldm_model_sync_faults`` == 0).
"""
import ctypes as C
import os

import pytest
import torch

import cfgs

pytestmark = pytest.mark.gpu


def _pair(cfg, seed, cuda, train=False):
    """Two modules with the same weights: plans with and without the fused launch (the knob is read when a plan is built)."""
    from ldm3d.networks import DiffusionModelUNet
    from oracle import unet as ou
    sd = ou.init_state_dict(ou.unet_param_shapes(cfg), seed, gain=0.5 if train else 1.0)
    out = []
    for knob in ("1", "0"):
        m = DiffusionModelUNet(**cfg)
        m.load_state_dict(sd)
        m = m.to(cuda)
        m = m.train() if train else m.eval()
        m._fin_gn_knob = knob
        out.append(m)
    return out


def _with_knob(m, fn):
    old = os.environ.get("LDM_FIN_GN")
    os.environ["LDM_FIN_GN"] = m._fin_gn_knob
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop("LDM_FIN_GN", None)
        else:
            os.environ["LDM_FIN_GN"] = old


def _launches(m, dims, b):
    from ldm3d import _lib
    f = C.c_int(0)
    n = _with_knob(m, lambda: _lib.lib().ldm_model_plan_launches(m._h, b"unet", b, *dims, C.byref(f)))
    assert n > 0
    return n, f.value


def _faults(m):
    from ldm3d import _lib
    return _lib.lib().ldm_model_sync_faults(m._h)


@pytest.mark.parametrize("name,dims,b", [("UNET_FULL", (24, 24, 24), 1), ("UNET_FULL", (16, 16, 16), 2), ("UNET_TINY", (8, 8, 8), 2),
                                         ("UNET_TINY_ALT", (6, 10, 8), 1), ("UNET_TINY_COND", (8, 8, 8), 1), ("UNET_TINY_ODD", (8, 8, 8), 2)])
def test_fused_finalize_groupnorm_is_bit_identical(cuda, name, dims, b):
    cfg = getattr(cfgs, name)
    fused, plain = _pair(cfg, 21, cuda)
    g = torch.Generator().manual_seed(22)
    cin = cfg["in_channels"]
    x = torch.randn((b, cin, *dims), generator=g).to(cuda)
    t = torch.tensor([13.0, 640.0][:b]).to(cuda)
    with torch.no_grad():
        a = _with_knob(fused, lambda: fused(x=x, timesteps=t)).clone()
        r = _with_knob(plain, lambda: plain(x=x, timesteps=t)).clone()
    n_f, k_f = _launches(fused, dims, b)
    n_p, k_p = _launches(plain, dims, b)
    print(f"{name} {dims} b{b}: {n_p} launches -> {n_f} ({k_f} fused finalize+GroupNorm)")
    assert k_p == 0 and n_p - n_f == k_f
    if name == "UNET_FULL" and dims == (24, 24, 24):
        assert k_f >= 20 and n_f <= 130, (n_f, k_f)      # the headline plan: 152 -> 125 launches
    assert torch.isfinite(a).all()
    assert torch.equal(a, r), float((a - r).abs().max())
    assert _faults(fused) == 0


def test_fused_launch_under_graph_replay(cuda):
    cfg = cfgs.UNET_FULL
    fused, plain = _pair(cfg, 23, cuda)
    g = torch.Generator().manual_seed(24)
    x = torch.randn((1, 4, 24, 24, 24), generator=g).to(cuda)
    t = torch.tensor([500.0]).to(cuda)
    with torch.no_grad():
        ref = _with_knob(plain, lambda: plain(x=x, timesteps=t)).clone()
        fused.enable_graph_replay(True)
        outs = [_with_knob(fused, lambda: fused(x=x, timesteps=t)).clone() for _ in range(6)]   # the barrier state resets itself between replays
    for o in outs:
        assert torch.equal(o, ref)
    assert _faults(fused) == 0


def test_fused_launch_in_training_plans(cuda):
    import torch.nn.functional as F
    cfg = cfgs.UNET_TINY
    fused, plain = _pair(cfg, 25, cuda, train=True)
    g = torch.Generator().manual_seed(26)
    x = torch.randn((2, 4, 8, 8, 8), generator=g).to(cuda)
    tgt = torch.randn((2, 4, 8, 8, 8), generator=g).to(cuda)
    t = torch.tensor([7.0, 800.0]).to(cuda)
    grads = []
    for m in (fused, plain):
        def step():
            loss = F.mse_loss(m(x=x, timesteps=t), tgt)
            loss.backward()
            return float(loss)
        loss = _with_knob(m, step)
        grads.append((loss, torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()))
    assert grads[0][0] == grads[1][0]
    assert torch.equal(grads[0][1], grads[1][1])
    assert _faults(fused) == 0
