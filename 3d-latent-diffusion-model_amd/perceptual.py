"""Optional perceptual term of the stage-1 loss, from USER-SUPPLIED weights (``train_autoencoder.py --perceptual-weights file.pt``).

The reference builds ``PerceptualLoss(spatial_dims=3, network_type="squeeze", is_fake_3d=True, fake_3d_ratio=0.2)``
(3d_ldm/train_autoencoder.py:236) and adds ``perceptual_weight * loss_perceptual(reconstruction.float(), images.float())`` to the
generator loss (:397-406).  MONAI's class downloads an LPIPS network (SqueezeNet 1.1 features + learned 1x1 "lin" layers); neither MONAI
nor those weights exist offline, so without a weights file the trainer drops the term and records that it did (trainer.py).

[MONAI-ext / lpips-ext] restated from the published definitions, NOT pinned against either package (none is installable here):
  * 2.5-D evaluation: for each spatial axis the volume is cut into 2-D slices along that axis ([B * n, C, h, w]); a random
    ``fake_3d_ratio`` of the slices (``torch.randperm(n)[: int(n * ratio)]``, the same indices for input and target) is scored and the
    mean over the slices taken; the loss is the SUM of the three axes' means.
  * LPIPS(net="squeeze", version 0.1): inputs (1 channel is repeated to 3) go through the scaling layer ((x - shift) / scale with
    shift = (-.030, -.088, -.188), scale = (.458, .448, .450)), then SqueezeNet 1.1 ``features`` cut into 7 slices ([0:2], [2:5], [5:8],
    [8:10], [10:11], [11:12], [12:13]: 64, 128, 256, 384, 384, 512, 512 channels); per slice: channel-unit-normalise both activations,
    squared difference, the slice's 1x1 "lin" layer (no bias), spatial mean; summed over the slices.
  * Weights file = ``lpips.LPIPS(net="squeeze").state_dict()`` saved with ``torch.save`` (keys ``net.slice<k>.<i>[.squeeze|.expand1x1|
    .expand3x3].{weight,bias}``, ``lin<k>.model.1.weight``; ``scaling_layer.*`` optional).

The 2-D convolutions run on the tensor library (``torch.nn.functional.conv2d``): this is an auxiliary loss outside the hot path
(SURVEY.md section 8f-1 marks it optional); its gradient reaches the AutoencoderKL through ``reconstruction`` and the HIP backward plan.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

# SqueezeNet 1.1 ``features``: index -> ("conv", cin, cout, k, stride) | ("pool",) | ("fire", cin, squeeze, e1, e3); ReLU follows a conv
_FEATURES = {0: ("conv", 3, 64, 3, 2), 2: ("pool",), 3: ("fire", 64, 16, 64, 64), 4: ("fire", 128, 16, 64, 64), 5: ("pool",),
             6: ("fire", 128, 32, 128, 128), 7: ("fire", 256, 32, 128, 128), 8: ("pool",), 9: ("fire", 256, 48, 192, 192),
             10: ("fire", 384, 48, 192, 192), 11: ("fire", 384, 64, 256, 256), 12: ("fire", 512, 64, 256, 256)}
_SLICES = [(0, 2), (2, 5), (5, 8), (8, 10), (10, 11), (11, 12), (12, 13)]
_CHANNELS = [64, 128, 256, 384, 384, 512, 512]


def expected_keys() -> Dict[str, tuple]:
    """name -> shape of every tensor the weights file must hold."""
    out = {}
    for k, (lo, hi) in enumerate(_SLICES, start=1):
        for i in range(lo, hi):
            spec = _FEATURES.get(i)
            if spec is None or spec[0] == "pool":
                continue
            p = f"net.slice{k}.{i}"
            if spec[0] == "conv":
                out[f"{p}.weight"], out[f"{p}.bias"] = (spec[2], spec[1], spec[3], spec[3]), (spec[2],)
            else:
                _, cin, sq, e1, e3 = spec
                out[f"{p}.squeeze.weight"], out[f"{p}.squeeze.bias"] = (sq, cin, 1, 1), (sq,)
                out[f"{p}.expand1x1.weight"], out[f"{p}.expand1x1.bias"] = (e1, sq, 1, 1), (e1,)
                out[f"{p}.expand3x3.weight"], out[f"{p}.expand3x3.bias"] = (e3, sq, 3, 3), (e3,)
    for k, c in enumerate(_CHANNELS):
        out[f"lin{k}.model.1.weight"] = (1, c, 1, 1)
    return out


class LpipsSqueeze(torch.nn.Module):
    def __init__(self, state_dict: Dict[str, torch.Tensor]):
        super().__init__()
        need = expected_keys()
        missing = [k for k in need if k not in state_dict]
        if missing:
            raise KeyError(f"perceptual weights: {len(missing)} tensors missing, first: {missing[:3]} (expected the state_dict of "
                           "lpips.LPIPS(net='squeeze'))")
        for k, shape in need.items():
            t = state_dict[k]
            if tuple(t.shape) != shape:
                raise ValueError(f"perceptual weights: {k} has shape {tuple(t.shape)}, expected {shape}")
            self.register_buffer(k.replace(".", "_"), t.detach().float().clone(), persistent=False)
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1), persistent=False)
        self.register_buffer("scale", torch.tensor([.458, .448, .450]).view(1, 3, 1, 1), persistent=False)

    def _w(self, name):
        return getattr(self, name.replace(".", "_"))

    def _features(self, x):
        feats = []
        for k, (lo, hi) in enumerate(_SLICES, start=1):
            for i in range(lo, hi):
                spec = _FEATURES.get(i)
                if spec is None:
                    continue                                   # index 1: the ReLU behind the first conv (applied with it)
                p = f"net.slice{k}.{i}"
                if spec[0] == "conv":
                    x = F.relu(F.conv2d(x, self._w(p + ".weight"), self._w(p + ".bias"), stride=spec[4]))
                elif spec[0] == "pool":
                    x = F.max_pool2d(x, kernel_size=3, stride=2, ceil_mode=True)
                else:
                    s = F.relu(F.conv2d(x, self._w(p + ".squeeze.weight"), self._w(p + ".squeeze.bias")))
                    x = torch.cat([F.relu(F.conv2d(s, self._w(p + ".expand1x1.weight"), self._w(p + ".expand1x1.bias"))),
                                   F.relu(F.conv2d(s, self._w(p + ".expand3x3.weight"), self._w(p + ".expand3x3.bias"), padding=1))], 1)
            feats.append(x)
        return feats

    def forward(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """LPIPS distance per image: [B, 1, 1, 1]."""
        if a.shape[1] == 1:
            a, b = a.repeat(1, 3, 1, 1), b.repeat(1, 3, 1, 1)
        elif a.shape[1] != 3:
            # neither grey nor RGB (the 2-channel images of config_train_16g.json): the network has 3 input channels, so the channels
            # are scored one by one as grey images and averaged (what MONAI's ``channel_wise=True`` does; the reference's default
            # ``channel_wise=False`` would fail on such an input)
            return sum(self.forward(a[:, c:c + 1], b[:, c:c + 1]) for c in range(a.shape[1])) / a.shape[1]
        fa, fb = self._features((a - self.shift) / self.scale), self._features((b - self.shift) / self.scale)
        total = 0.0
        for k, (x, y) in enumerate(zip(fa, fb)):
            xn = x / (x.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            yn = y / (y.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
            total = total + F.conv2d((xn - yn) ** 2, self._w(f"lin{k}.model.1.weight")).mean((2, 3), keepdim=True)
        return total


class PerceptualLoss(torch.nn.Module):
    """``PerceptualLoss(spatial_dims=3, network_type="squeeze", is_fake_3d=True, fake_3d_ratio=0.2)`` (train_autoencoder.py:236)."""

    def __init__(self, weights: Dict[str, torch.Tensor], fake_3d_ratio: float = 0.2):
        super().__init__()
        self.net = LpipsSqueeze(weights)
        self.fake_3d_ratio = fake_3d_ratio

    @classmethod
    def from_file(cls, path: str, **kw) -> "PerceptualLoss":
        return cls(torch.load(path, map_location="cpu", weights_only=True), **kw)

    def _axis_loss(self, x: torch.Tensor, y: torch.Tensor, axis: int) -> torch.Tensor:
        keep = [a for a in (2, 3, 4) if a != axis]
        perm = (0, axis, 1, *keep)

        def slices(v):
            v = v.float().permute(*perm).contiguous()
            return v.view(-1, v.shape[2], v.shape[3], v.shape[4])
        xs, ys = slices(x), slices(y)
        idx = torch.randperm(xs.shape[0])[: int(xs.shape[0] * self.fake_3d_ratio)].to(xs.device)
        return self.net(xs.index_select(0, idx), ys.index_select(0, idx)).mean()

    def forward(self, inp: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if inp.shape != target.shape or inp.dim() != 5:
            raise ValueError(f"PerceptualLoss: two [B, C, D, H, W] tensors of one shape expected, got {tuple(inp.shape)} / {tuple(target.shape)}")
        return self._axis_loss(inp, target, 2) + self._axis_loss(inp, target, 3) + self._axis_loss(inp, target, 4)
