"""Negative controls for the bf16 parity gates (-m gpu): a defect injected into ONE layer must turn the gates red.

The bf16 gates (per block, teacher-forced: 5e-3 / 1e-2 for attention blocks, tests/test_gpu_taps.py; model level: the bf16 noise
floor, tests/test_gpu_models.py::floor_gate) are only worth their green if a real 5 - 10 % defect in one layer cannot pass them.
Each case below runs the headline network (benchmark UNet, 1x4x24^3, the golden's seeds) with the fp32 CPU oracle UNTOUCHED and a
defect on the GPU side only, injected through what a user can reach -- the ``state_dict`` or a constructor argument -- never through
the launch planner:

  * ``tap``   one of the 27 taps of one 24^3 convolution (down_blocks.0.resnets.1.conv1) zeroed: 3.7 % of that layer's weights;
  * ``eps``   GroupNorm epsilon off by 10 x (1e-5 instead of 1e-6) on a network whose first activations are small (conv_in scaled so
              that the variance norm1 sees is of the order of epsilon -- with unit-variance activations NO bf16 gate can see an epsilon
              error of 9e-6, so the control is run where epsilon matters);
  * ``vswap`` the V projections of two heads of one attention block (down_blocks.1.attentions.0) exchanged.

For every case: the clean module passes both gates, the defective one must FAIL the per-block gate at the injected block and FAIL the
model-level gate (free-running, against the fp32 oracle).  Reference arithmetic: MONAI DiffusionModelUNet as reached from
3d_ldm/inference.py:94-99; oracle/unet.py restates it.
"""
import os

import pytest
import torch

import cfgs
from util import rel_l2

pytestmark = pytest.mark.gpu


def _module(cfg, sd, cuda):
    from ldm3d.networks import DiffusionModelUNet
    m = DiffusionModelUNet(**cfg)
    m.load_state_dict(sd)
    return m.to(cuda).eval()


def _block_errors(m, x, t, ref, cuda):
    with torch.no_grad():
        _, got = m.forward_taps(x.to(cuda), t.to(cuda), force=ref)
    return {k: rel_l2(got[k].cpu(), ref[k]) for k in ref}


def _gates(errs, free_err, floor):
    """(names of blocks over the per-block gate, model-level gate holds?) with the gates of test_gpu_taps / test_gpu_models."""
    from test_gpu_models import FLOOR_FACTOR_FP32
    from test_gpu_taps import BLOCK_TOL_BF16
    over = [k for k, e in errs.items() if e > (2 * BLOCK_TOL_BF16 if ".attention" in k else BLOCK_TOL_BF16)]
    return over, free_err <= FLOOR_FACTOR_FP32 * floor + 1e-3


@pytest.fixture(scope="module")
def case():
    """Golden's seeds: weights seed 0, input seed 0, t = 500; fp32 oracle with block taps (one ~10 s CPU forward)."""
    from oracle import unet as ou
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "unet_full_24.pt"), weights_only=True)
    cfg = cfgs.UNET_FULL
    sd = ou.init_state_dict(ou.unet_param_shapes(cfg), gold["weight_seed"])
    x = torch.randn((1, 4, 24, 24, 24), generator=torch.Generator().manual_seed(gold["input_seed"]))
    t = torch.tensor([gold["t"]])
    taps = {}
    ref_out = ou.unet_forward(sd, cfg, x, t, taps=taps)
    assert rel_l2(ref_out, gold["eps_fp32_oracle"].float()) <= 1e-5          # the oracle still produces its golden
    ref = {k: v for k, v in taps.items() if "." in k or k == "conv_in"}
    floor = rel_l2(gold["eps_bf16_oracle"].float(), gold["eps_fp32_oracle"].float())
    return cfg, sd, x, t, ref, ref_out, floor


def _run(cfg, sd_gpu, x, t, ref, ref_out, floor, cuda, what, cfg_gpu=None):
    m = _module(cfg_gpu or cfg, sd_gpu, cuda)
    errs = _block_errors(m, x, t, ref, cuda)
    with torch.no_grad():
        free = rel_l2(m(x=x.to(cuda), timesteps=t.to(cuda)).cpu(), ref_out)
    over, model_ok = _gates(errs, free, floor)
    worst = max(errs, key=errs.get)
    print(f"{what}: worst block {worst} {errs[worst]:.2e}, blocks over their gate {len(over)}, free-running vs fp32 oracle {free:.2e} "
          f"= {free / floor:.2f} x floor ({floor:.2e}) -> model-level gate {'holds' if model_ok else 'FAILS'}")
    del m
    torch.cuda.empty_cache()
    return errs, over, model_ok


def test_clean_module_passes_and_one_zeroed_tap_fails_both_gates(cuda, case):
    cfg, sd, x, t, ref, ref_out, floor = case
    _, over, ok = _run(cfg, sd, x, t, ref, ref_out, floor, cuda, "clean")
    assert not over and ok                                               # the control's baseline: green
    bad = {k: v.clone() for k, v in sd.items()}
    bad["down_blocks.0.resnets.1.conv1.conv.weight"][:, :, 1, 1, 2] = 0.0     # one tap of one 24^3 conv
    errs, over, ok = _run(cfg, bad, x, t, ref, ref_out, floor, cuda, "one tap of down_blocks.0.resnets.1.conv1 zeroed")
    assert "down_blocks.0.resnets.1" in over, errs["down_blocks.0.resnets.1"]
    assert over == ["down_blocks.0.resnets.1"], over                     # teacher forcing localises the defect to its block
    assert not ok


def test_swapped_value_heads_fail_both_gates(cuda, case):
    cfg, sd, x, t, ref, ref_out, floor = case
    bad = {k: v.clone() for k, v in sd.items()}
    for suffix in ("weight", "bias"):
        v = bad[f"down_blocks.1.attentions.0.attn.to_v.{suffix}"]
        a, b = v[0:64].clone(), v[64:128].clone()
        v[0:64], v[64:128] = b, a                                        # head 0's V <-> head 1's V
    errs, over, ok = _run(cfg, bad, x, t, ref, ref_out, floor, cuda, "V of heads 0 / 1 of down_blocks.1.attentions.0 swapped")
    assert over == ["down_blocks.1.attentions.0"], over
    assert not ok


def test_group_norm_epsilon_off_by_ten_fails_both_gates_where_epsilon_matters(cuda):
    """conv_in scaled so that resnets.0.norm1 sees a variance of the order of epsilon: rstd = (var + eps)^-1/2 then moves by tens of
    per cent between eps = 1e-6 and 1e-5.  Own oracle run (the scaled weights are not the golden's)."""
    from oracle import unet as ou
    cfg = cfgs.UNET_FULL
    sd = ou.init_state_dict(ou.unet_param_shapes(cfg), 0)
    sd["conv_in.conv.weight"] *= 2e-3
    sd["conv_in.conv.bias"] *= 2e-3
    x = torch.randn((1, 4, 16, 16, 16), generator=torch.Generator().manual_seed(0))
    t = torch.tensor([500.0])
    taps = {}
    ref_out = ou.unet_forward(sd, cfg, x, t, taps=taps)
    ref = {k: v for k, v in taps.items() if "." in k or k == "conv_in"}
    var = float(ref["conv_in"].var())
    assert 1e-7 < var < 1e-4, var
    floor = rel_l2(ou.unet_forward(sd, cfg, x, t, emulate_bf16=True), ref_out)
    _, over, ok = _run(cfg, sd, x, t, ref, ref_out, floor, cuda, f"small activations (var {var:.1e}), eps as configured")
    assert not over and ok
    errs, over, ok = _run(cfg, sd, x, t, ref, ref_out, floor, cuda, "GroupNorm eps 1e-5 instead of 1e-6", cfg_gpu=dict(cfg, norm_eps=1e-5))
    assert "down_blocks.0.resnets.0" in over, errs["down_blocks.0.resnets.0"]
    assert not ok
