"""Oracle: AutoencoderKL encode / decode (test infrastructure only, see oracle/__init__.py).

Restates monai.networks.nets.AutoencoderKL (MONAI >= 1.4, [MONAI-ext])
for the definitions the reference instantiates (3d_ldm/config/config_train_16g.json:7-28,
config_train_32g.json:7-29).  Reference call sites: 3d_ldm/train_diffusion.py:104,180,195,249,258,310,324
(``encode_stage_2_inputs``), 3d_ldm/train_autoencoder.py:366,579 (``forward``) and, through the inferer,
``decode_stage_2_outputs`` (3d_ldm/inference.py:94-99).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

from .unet import SD, attention_block, conv, group_norm, rbf


def norm_cfg(cfg: dict) -> dict:
    c = dict(cfg)
    ch = list(c["channels"])
    n = len(ch)
    c["channels"] = ch
    c.setdefault("spatial_dims", 3)
    c.setdefault("norm_num_groups", 32)
    c.setdefault("norm_eps", 1e-6)
    c.setdefault("latent_channels", 3)
    nrb = c.get("num_res_blocks", 2)
    c["num_res_blocks"] = [nrb] * n if isinstance(nrb, int) else list(nrb)
    c["attention_levels"] = [bool(a) for a in c.get("attention_levels", [False] * n)]
    c.setdefault("with_encoder_nonlocal_attn", True)
    c.setdefault("with_decoder_nonlocal_attn", True)
    assert c["spatial_dims"] == 3
    return c


def encoder_layout(c: dict) -> List[Tuple[str, tuple]]:
    """Flat ``encoder.blocks.<k>`` list: (kind, args).  Mirrors [MONAI-ext] Encoder.__init__."""
    ch = c["channels"]
    blocks: List[Tuple[str, tuple]] = [("conv", (c["in_channels"], ch[0]))]
    oc = ch[0]
    for i in range(len(ch)):
        ic, oc = oc, ch[i]
        for _ in range(c["num_res_blocks"][i]):
            blocks.append(("res", (ic, oc)))
            ic = oc
            if c["attention_levels"][i]:
                blocks.append(("attn", (ic,)))
        if i != len(ch) - 1:
            blocks.append(("down", (ic,)))
    if c["with_encoder_nonlocal_attn"]:
        blocks += [("res", (ch[-1], ch[-1])), ("attn", (ch[-1],)), ("res", (ch[-1], ch[-1]))]
    blocks.append(("gn", (ch[-1],)))
    blocks.append(("conv", (ch[-1], c["latent_channels"])))
    return blocks


def decoder_layout(c: dict) -> List[Tuple[str, tuple]]:
    """Flat ``decoder.blocks.<k>`` list.  Mirrors [MONAI-ext] Decoder.__init__ (nearest-upsample + post conv)."""
    rev = list(reversed(c["channels"]))
    rev_attn = list(reversed(c["attention_levels"]))
    rev_nrb = list(reversed(c["num_res_blocks"]))
    blocks: List[Tuple[str, tuple]] = [("conv", (c["latent_channels"], rev[0]))]
    if c["with_decoder_nonlocal_attn"]:
        blocks += [("res", (rev[0], rev[0])), ("attn", (rev[0],)), ("res", (rev[0], rev[0]))]
    oc = rev[0]
    for i in range(len(rev)):
        ic, oc = oc, rev[i]
        for _ in range(rev_nrb[i]):
            blocks.append(("res", (ic, oc)))
            ic = oc
            if rev_attn[i]:
                blocks.append(("attn", (ic,)))
        if i != len(rev) - 1:
            blocks.append(("up", (ic,)))
    blocks.append(("gn", (oc,)))
    blocks.append(("conv", (oc, c["out_channels"])))
    return blocks


def _res_block(sd: SD, p: str, x, c, bf):
    """AEKLResBlock: GN-SiLU-conv3-GN-SiLU-conv3 + (1x1 nin_shortcut | identity).  No time embedding."""
    g, eps = c["norm_num_groups"], c["norm_eps"]
    h = rbf(F.silu(group_norm(sd, p + ".norm1", x, g, eps)), bf)
    h = rbf(conv(sd, p + ".conv1", h, bf), bf)
    h = rbf(F.silu(group_norm(sd, p + ".norm2", h, g, eps)), bf)
    h = conv(sd, p + ".conv2", h, bf)
    if (p + ".nin_shortcut.conv.weight") in sd:
        skip = conv(sd, p + ".nin_shortcut", x, bf, padding=0)
    else:
        skip = x
    return rbf(skip + h, bf)


def _run(sd: SD, prefix: str, layout, x, c, bf, taps=None):
    """``taps`` (optional dict) receives every block output under "<prefix>.blocks.<k>" (ldm_model_tap_info's names)."""
    h = x
    for k, (kind, _args) in enumerate(layout):
        p = f"{prefix}.blocks.{k}"
        if taps is not None and k > 0:
            taps[f"{prefix}.blocks.{k - 1}"] = h.detach().clone()
        if kind == "conv":
            last = k == len(layout) - 1
            h = conv(sd, p, h, bf)
            h = h if last else rbf(h, bf)         # final conv output stays fp32
        elif kind == "res":
            h = _res_block(sd, p, h, c, bf)
        elif kind == "attn":
            h = attention_block(sd, p, h, 0, c, bf)   # AE attention is single-head ([MONAI-ext])
        elif kind == "down":
            # AEKLDownsample: F.pad (0,1) per spatial dim, then 3^3 conv stride 2 pad 0
            h = F.pad(h, (0, 1, 0, 1, 0, 1), mode="constant", value=0.0)
            h = rbf(conv(sd, p + ".conv", h, bf, stride=2, padding=0), bf)
        elif kind == "up":
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = rbf(conv(sd, p + ".postconv", h, bf), bf)
        elif kind == "gn":
            # final GroupNorm feeds the last conv directly: NO SiLU here ([MONAI-ext])
            h = rbf(group_norm(sd, p, h, c["norm_num_groups"], c["norm_eps"]), bf)
    return h


def encode(sd: SD, cfg: dict, x: torch.Tensor, emulate_bf16: bool = False, taps=None):
    """-> (z_mu, z_sigma).  log-variance clamped to [-30, 20] before exp(./2)."""
    c = norm_cfg(cfg)
    bf = emulate_bf16
    h = _run(sd, "encoder", encoder_layout(c), rbf(x, bf), c, bf, taps)
    if taps is not None:
        taps[f"encoder.blocks.{len(encoder_layout(c)) - 1}"] = h.detach().clone()
    h = rbf(h, bf)                                  # encoder output is stored bf16 before the 1x1 heads
    z_mu = conv(sd, "quant_conv_mu", h, bf, padding=0)
    z_log_var = conv(sd, "quant_conv_log_sigma", h, bf, padding=0)
    z_log_var = torch.clamp(z_log_var, -30.0, 20.0)
    z_sigma = torch.exp(z_log_var / 2)
    return z_mu, z_sigma


def sampling(z_mu, z_sigma, eps):
    """z = mu + sigma * eps with eps supplied explicitly (goldens must not depend on an RNG stream)."""
    return z_mu + eps * z_sigma


def decode(sd: SD, cfg: dict, z: torch.Tensor, emulate_bf16: bool = False, taps=None) -> torch.Tensor:
    c = norm_cfg(cfg)
    bf = emulate_bf16
    h = rbf(conv(sd, "post_quant_conv", rbf(z, bf), bf, padding=0), bf)
    if taps is not None:
        taps["post_quant_conv"] = h.detach().clone()
    return _run(sd, "decoder", decoder_layout(c), h, c, bf, taps)


def encode_stage_2_inputs(sd, cfg, x, eps, emulate_bf16=False):
    z_mu, z_sigma = encode(sd, cfg, x, emulate_bf16)
    return sampling(z_mu, z_sigma, eps)


def decode_stage_2_outputs(sd, cfg, z, emulate_bf16=False):
    return decode(sd, cfg, z, emulate_bf16)


def forward(sd, cfg, x, eps, emulate_bf16=False):
    """AutoencoderKL.forward -> (reconstruction, z_mu, z_sigma) (3d_ldm/train_autoencoder.py:366)."""
    z_mu, z_sigma = encode(sd, cfg, x, emulate_bf16)
    z = sampling(z_mu, z_sigma, eps)
    return decode(sd, cfg, z, emulate_bf16), z_mu, z_sigma


def kl_loss(z_mu: torch.Tensor, z_sigma: torch.Tensor) -> torch.Tensor:
    """Clamped KL of the reference glue, restated from 3d_ldm/utils.py:249-262."""
    eps = 1e-8
    s = torch.clamp(z_sigma, min=eps)
    kl = 0.5 * torch.sum(z_mu.pow(2) + s.pow(2) - torch.log(s.pow(2) + eps) - 1,
                         dim=list(range(1, z_sigma.dim())))
    return torch.clamp(kl / kl.shape[0], 0.0, 1000.0)


def ae_param_shapes(cfg: dict) -> Dict[str, Sequence[int]]:
    c = norm_cfg(cfg)
    out: Dict[str, Sequence[int]] = {}

    def conv_p(name, cin, cout, k):
        out[name + ".conv.weight"] = (cout, cin, k, k, k)
        out[name + ".conv.bias"] = (cout,)

    def gn_p(name, cch):
        out[name + ".weight"] = (cch,)
        out[name + ".bias"] = (cch,)

    def fill(prefix, layout):
        for k, (kind, a) in enumerate(layout):
            p = f"{prefix}.blocks.{k}"
            if kind == "conv":
                conv_p(p, a[0], a[1], 3)
            elif kind == "res":
                gn_p(p + ".norm1", a[0]); conv_p(p + ".conv1", a[0], a[1], 3)
                gn_p(p + ".norm2", a[1]); conv_p(p + ".conv2", a[1], a[1], 3)
                if a[0] != a[1]:
                    conv_p(p + ".nin_shortcut", a[0], a[1], 1)
            elif kind == "attn":
                gn_p(p + ".norm", a[0])
                for n in ("to_q", "to_k", "to_v", "out_proj"):
                    out[f"{p}.attn.{n}.weight"] = (a[0], a[0])
                    out[f"{p}.attn.{n}.bias"] = (a[0],)
            elif kind == "down":
                conv_p(p + ".conv", a[0], a[0], 3)
            elif kind == "up":
                conv_p(p + ".postconv", a[0], a[0], 3)
            elif kind == "gn":
                gn_p(p, a[0])

    fill("encoder", encoder_layout(c))
    fill("decoder", decoder_layout(c))
    lc = c["latent_channels"]
    conv_p("quant_conv_mu", lc, lc, 1)
    conv_p("quant_conv_log_sigma", lc, lc, 1)
    conv_p("post_quant_conv", lc, lc, 1)
    return out
