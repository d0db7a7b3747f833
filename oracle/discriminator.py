"""CPU oracle of the stage-1 GAN tail: PatchDiscriminator + LSGAN loss.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

[MONAI-ext] restated from monai.networks.nets.PatchDiscriminator / monai.losses.PatchAdversarialLoss as the reference builds and
uses them (3d_ldm/train_autoencoder.py:150-158: num_layers_d=3, channels=32, in=out=1, norm INSTANCE; :235 criterion
"least_squares"; :410-413 generator term, :459-468 discriminator term).  PARITY UNPINNED (MONAI not importable here)."""
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


def param_shapes(in_channels=1, channels=32, out_channels=1, num_layers_d=3) -> Dict[str, tuple]:
    out = {"initial_conv.conv.weight": (channels, in_channels, 4, 4, 4), "initial_conv.conv.bias": (channels,)}
    cin, cout = channels, channels * 2
    for l_ in range(num_layers_d):
        out[f"{l_}.conv.weight"] = (cout, cin, 4, 4, 4)
        cin, cout = cout, cout * 2
    out["final_conv.conv.weight"] = (out_channels, cin, 4, 4, 4)
    out["final_conv.conv.bias"] = (out_channels,)
    return out


def _leaky(u: torch.Tensor, branch: Optional[torch.Tensor]) -> torch.Tensor:
    """LeakyReLU(0.2).  ``branch`` (bool, True = slope 1) fixes which side of the kink every element is on: the derivative jumps at 0,
    so two correct fp32 implementations whose pre-activations differ in the last bits can land on opposite sides of it for an element
    within rounding distance of 0 (|u| ~ 1e-7).  A parity test passes the branches the implementation under test took; the value
    changes by at most 0.8 |u| there, the gradient is then compared on the same branch."""
    return F.leaky_relu(u, 0.2) if branch is None else torch.where(branch, u, 0.2 * u)


def forward(sd, x: torch.Tensor, num_layers_d=3, branches: Optional[List[torch.Tensor]] = None) -> List[torch.Tensor]:
    """Every layer's output, as PatchDiscriminator.forward returns them.  ``branches``: see ``_leaky`` (one mask per activation)."""
    outs = []
    br = list(branches) if branches is not None else [None] * (num_layers_d + 1)
    h = _leaky(F.conv3d(x, sd["initial_conv.conv.weight"], sd["initial_conv.conv.bias"], stride=2, padding=1), br[0])
    outs.append(h)
    for l_ in range(num_layers_d):
        stride = 1 if l_ == num_layers_d - 1 else 2
        h = F.conv3d(h, sd[f"{l_}.conv.weight"], None, stride=stride, padding=1)
        h = _leaky(F.instance_norm(h, eps=1e-5), br[l_ + 1])
        outs.append(h)
    outs.append(F.conv3d(h, sd["final_conv.conv.weight"], sd["final_conv.conv.bias"], stride=1, padding=1))
    return outs


def lsgan(logits: torch.Tensor, target_is_real: bool) -> torch.Tensor:
    return torch.mean((logits - (1.0 if target_is_real else 0.0)) ** 2)
