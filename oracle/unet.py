"""Oracle: DiffusionModelUNet forward (test infrastructure only, see oracle/__init__.py).

Restates monai.networks.nets.DiffusionModelUNet (MONAI >= 1.4, [MONAI-ext] in
SURVEY.md section 8a rows a2, a2.1-a2.4) for the configuration family the reference
instantiates (3d_ldm/config/config_train_16g.json:39-48, config_train_32g.json:40-49):
spatial_dims=3, with_conditioning=False, resblock_updown=False, include_fc=True.
Call sites in the reference: 3d_ldm/train_diffusion.py:197-205,260-268,326-333 and
3d_ldm/inference.py:94-99 (through LatentDiffusionInferer).

Everything is plain torch.nn.functional on a MONAI-shaped ``state_dict``
(name -> fp32 tensor).  ``emulate_bf16`` inserts the HIP path's rounding points.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------- helpers
def rbf(x: torch.Tensor, on: bool) -> torch.Tensor:
    """Round to bf16 (nearest-even) and come back to fp32 when emulation is on."""
    return x.to(torch.bfloat16).to(torch.float32) if on else x


def norm_cfg(cfg: dict) -> dict:
    """Fill MONAI defaults and broadcast per-level scalars."""
    c = dict(cfg)
    ch = list(c["channels"])
    n = len(ch)
    c["channels"] = ch
    c.setdefault("spatial_dims", 3)
    c.setdefault("norm_num_groups", 32)
    c.setdefault("norm_eps", 1e-6)
    nrb = c.get("num_res_blocks", 2)
    c["num_res_blocks"] = [nrb] * n if isinstance(nrb, int) else list(nrb)
    nhc = c.get("num_head_channels", 8)
    c["num_head_channels"] = [nhc] * n if isinstance(nhc, int) else list(nhc)
    c["attention_levels"] = [bool(a) for a in c.get("attention_levels", [False] * n)]
    assert c["spatial_dims"] == 3
    return c


def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """[MONAI-ext] get_timestep_embedding: cat(cos, sin), cos FIRST."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32)
    freqs = torch.exp(exponent / half)
    args = timesteps[:, None].float() * freqs[None, :]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1))
    return emb


def conv(sd: SD, name: str, x: torch.Tensor, bf: bool, stride: int = 1, padding: int = 1) -> torch.Tensor:
    """MONAI Convolution(conv_only=True): keys ``<name>.conv.{weight,bias}``.  fp32 result, NOT rounded."""
    w = rbf(sd[name + ".conv.weight"], bf)
    return F.conv3d(x, w, sd[name + ".conv.bias"], stride=stride, padding=padding)


def group_norm(sd: SD, name: str, x: torch.Tensor, groups: int, eps: float) -> torch.Tensor:
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps)


def linear(sd: SD, name: str, x: torch.Tensor, bf: bool) -> torch.Tensor:
    return F.linear(x, rbf(sd[name + ".weight"], bf), sd[name + ".bias"])


# ----------------------------------------------------------------------------- blocks
def resnet_block(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor, c: dict, bf: bool) -> torch.Tensor:
    """DiffusionUNetResnetBlock: additive time bias only, 1x1 skip when C changes.

    Rounding points (bf16 emulation): GN+SiLU output, conv1+bias+temb output,
    GN+SiLU output, (conv2 + bias + skip(x)) output.  The 1x1 skip is accumulated in
    fp32 together with conv2 (the HIP kernel fuses it as extra K-steps).
    """
    g, eps = c["norm_num_groups"], c["norm_eps"]
    h = rbf(F.silu(group_norm(sd, p + ".norm1", x, g, eps)), bf)
    h = conv(sd, p + ".conv1", h, bf)
    temb = linear(sd, p + ".time_emb_proj", F.silu(emb), bf)
    h = rbf(h + temb[:, :, None, None, None], bf)
    h = rbf(F.silu(group_norm(sd, p + ".norm2", h, g, eps)), bf)
    h = conv(sd, p + ".conv2", h, bf)
    if (p + ".skip_connection.conv.weight") in sd:
        skip = conv(sd, p + ".skip_connection", x, bf, padding=0)
    else:
        skip = x
    return rbf(skip + h, bf)


def _flash_emulated_attention(q, k, v, scale: float, kv_tile: int = 64):
    """Online-softmax attention with the HIP kernel's rounding points: P is rounded to bf16
    against the RUNNING row max of kv tiles of ``kv_tile`` keys; O and l accumulate in fp32."""
    B, H, N, D = q.shape
    m = torch.full((B, H, N, 1), -float("inf"))
    l = torch.zeros((B, H, N, 1))
    o = torch.zeros((B, H, N, D))
    for j0 in range(0, N, kv_tile):
        s = torch.matmul(q, k[:, :, j0:j0 + kv_tile].transpose(-1, -2)) * scale
        m_new = torch.maximum(m, s.amax(dim=-1, keepdim=True))
        alpha = torch.exp(m - m_new)
        p = torch.exp(s - m_new)
        l = l * alpha + p.sum(dim=-1, keepdim=True)
        o = o * alpha + torch.matmul(rbf(p, True), v[:, :, j0:j0 + kv_tile])
        m = m_new
    return o / l


def attention_block(sd: SD, p: str, x: torch.Tensor, head_ch: int, c: dict, bf: bool) -> torch.Tensor:
    """SpatialAttentionBlock -> SABlock: GN (no SiLU) -> q,k,v Linear(bias) ->
    softmax(q k^T d^-1/2) v -> out_proj -> + residual.  Tokens are D*H*W in NCDHW order."""
    B, C = x.shape[:2]
    sp = x.shape[2:]
    heads = C // head_ch if head_ch else 1
    d = C // heads
    h = rbf(group_norm(sd, p + ".norm", x, c["norm_num_groups"], c["norm_eps"]), bf)
    t = h.reshape(B, C, -1).transpose(1, 2)  # [B, N, C]
    q = rbf(linear(sd, p + ".attn.to_q", t, bf), bf)
    k = rbf(linear(sd, p + ".attn.to_k", t, bf), bf)
    v = rbf(linear(sd, p + ".attn.to_v", t, bf), bf)

    def split(z):
        return z.reshape(B, -1, heads, d).permute(0, 2, 1, 3)  # [B, h, N, d]

    q, k, v = split(q), split(k), split(v)
    scale = d ** -0.5
    if bf:
        o = _flash_emulated_attention(q, k, v, scale)
    else:
        a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * scale, dim=-1)
        o = torch.matmul(a, v)
    o = rbf(o, bf).permute(0, 2, 1, 3).reshape(B, -1, C)
    o = linear(sd, p + ".attn.out_proj", o, bf)
    o = o.transpose(1, 2).reshape(B, C, *sp)
    return rbf(o + x, bf)


def upsample_nearest_conv(sd: SD, name: str, x: torch.Tensor, bf: bool) -> torch.Tensor:
    """[MONAI-ext] nearest x2 interpolate then 3^3 conv pad 1."""
    x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    return rbf(conv(sd, name, x, bf), bf)


# ----------------------------------------------------------------------------- model
def unet_forward(sd: SD, cfg: dict, x: torch.Tensor, timesteps: torch.Tensor,
                 emulate_bf16: bool = False, taps: dict | None = None, force: dict | None = None) -> torch.Tensor:
    """eps_hat = UNet(x_t, t).  x: [B, C_in, D, H, W] fp32; timesteps: [B].

    ``taps`` (optional dict) receives named intermediate tensors for per-op parity tests: the coarse ones
    ("emb", "conv_in", "down{i}", "mid", "up{i}") and every block output under the block's state_dict prefix
    ("down_blocks.0.resnets.1", "middle_block.attention", "up_blocks.1.upsampler.postconv", ...: the names
    ldm_model_tap_info reports).  ``force`` (optional dict with the same block names) replaces each block output
    after it has been recorded: teacher forcing, the mirror of ldm_unet_forward_taps(taps_in=...).
    """
    c = norm_cfg(cfg)
    bf = emulate_bf16
    ch = c["channels"]
    nlev = len(ch)

    def tap(name, t):
        if taps is not None:
            taps[name] = t.detach().clone()

    def blk(name, t):
        tap(name, t)
        return force[name].to(t.dtype) if force is not None else t

    t_emb = timestep_embedding(timesteps, ch[0])
    emb = linear(sd, "time_embed.0", t_emb, bf)
    emb = linear(sd, "time_embed.2", F.silu(emb), bf)
    tap("emb", emb)

    h = blk("conv_in", rbf(conv(sd, "conv_in", rbf(x, bf), bf), bf))
    skips: List[torch.Tensor] = [h]
    for i in range(nlev):
        for j in range(c["num_res_blocks"][i]):
            h = blk(f"down_blocks.{i}.resnets.{j}", resnet_block(sd, f"down_blocks.{i}.resnets.{j}", h, emb, c, bf))
            if c["attention_levels"][i]:
                h = blk(f"down_blocks.{i}.attentions.{j}",
                        attention_block(sd, f"down_blocks.{i}.attentions.{j}", h, c["num_head_channels"][i], c, bf))
            skips.append(h)
        tap(f"down{i}", h)
        if i != nlev - 1:
            h = blk(f"down_blocks.{i}.downsampler.op",
                    rbf(conv(sd, f"down_blocks.{i}.downsampler.op", h, bf, stride=2, padding=1), bf))
            skips.append(h)

    h = blk("middle_block.resnet_1", resnet_block(sd, "middle_block.resnet_1", h, emb, c, bf))
    h = blk("middle_block.attention", attention_block(sd, "middle_block.attention", h, c["num_head_channels"][-1], c, bf))
    h = blk("middle_block.resnet_2", resnet_block(sd, "middle_block.resnet_2", h, emb, c, bf))
    tap("mid", h)

    for i in range(nlev):
        lvl = nlev - 1 - i
        for j in range(c["num_res_blocks"][lvl] + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = blk(f"up_blocks.{i}.resnets.{j}", resnet_block(sd, f"up_blocks.{i}.resnets.{j}", h, emb, c, bf))
            if c["attention_levels"][lvl]:
                h = blk(f"up_blocks.{i}.attentions.{j}",
                        attention_block(sd, f"up_blocks.{i}.attentions.{j}", h, c["num_head_channels"][lvl], c, bf))
        if i != nlev - 1:
            h = blk(f"up_blocks.{i}.upsampler.postconv", upsample_nearest_conv(sd, f"up_blocks.{i}.upsampler.postconv", h, bf))
        tap(f"up{i}", h)
    assert not skips

    h = rbf(F.silu(group_norm(sd, "out.0", h, c["norm_num_groups"], c["norm_eps"])), bf)
    return conv(sd, "out.2", h, bf)  # final output stays fp32 (no rounding)


# ----------------------------------------------------------------------------- parameters
def unet_param_shapes(cfg: dict) -> Dict[str, Sequence[int]]:
    """MONAI-shaped state_dict layout: name -> shape (SURVEY section 5 'Checkpoint', 8c 'Residual risk')."""
    c = norm_cfg(cfg)
    ch = c["channels"]
    nlev = len(ch)
    temb = ch[0] * 4
    out: Dict[str, Sequence[int]] = {}

    def conv_p(name, cin, cout, k):
        out[name + ".conv.weight"] = (cout, cin, k, k, k)
        out[name + ".conv.bias"] = (cout,)

    def lin_p(name, cin, cout):
        out[name + ".weight"] = (cout, cin)
        out[name + ".bias"] = (cout,)

    def gn_p(name, cch):
        out[name + ".weight"] = (cch,)
        out[name + ".bias"] = (cch,)

    def res_p(p, cin, cout):
        gn_p(p + ".norm1", cin)
        conv_p(p + ".conv1", cin, cout, 3)
        lin_p(p + ".time_emb_proj", temb, cout)
        gn_p(p + ".norm2", cout)
        conv_p(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv_p(p + ".skip_connection", cin, cout, 1)

    def attn_p(p, cch):
        gn_p(p + ".norm", cch)
        for n in ("to_q", "to_k", "to_v", "out_proj"):
            lin_p(p + ".attn." + n, cch, cch)

    conv_p("conv_in", c["in_channels"], ch[0], 3)
    lin_p("time_embed.0", ch[0], temb)
    lin_p("time_embed.2", temb, temb)
    oc = ch[0]
    for i in range(nlev):
        ic, oc = oc, ch[i]
        for j in range(c["num_res_blocks"][i]):
            res_p(f"down_blocks.{i}.resnets.{j}", ic if j == 0 else oc, oc)
            if c["attention_levels"][i]:
                attn_p(f"down_blocks.{i}.attentions.{j}", oc)
        if i != nlev - 1:
            conv_p(f"down_blocks.{i}.downsampler.op", oc, oc, 3)
    res_p("middle_block.resnet_1", ch[-1], ch[-1])
    attn_p("middle_block.attention", ch[-1])
    res_p("middle_block.resnet_2", ch[-1], ch[-1])
    rev = list(reversed(ch))
    oc = rev[0]
    for i in range(nlev):
        prev, oc = oc, rev[i]
        ic = rev[min(i + 1, nlev - 1)]
        lvl = nlev - 1 - i
        nres = c["num_res_blocks"][lvl] + 1
        for j in range(nres):
            skip_c = ic if j == nres - 1 else oc
            rin = prev if j == 0 else oc
            res_p(f"up_blocks.{i}.resnets.{j}", rin + skip_c, oc)
            if c["attention_levels"][lvl]:
                attn_p(f"up_blocks.{i}.attentions.{j}", oc)
        if i != nlev - 1:
            conv_p(f"up_blocks.{i}.upsampler.postconv", oc, oc, 3)
    gn_p("out.0", ch[0])
    conv_p("out.2", ch[0], c["out_channels"], 3)
    return out


def init_state_dict(shapes: Dict[str, Sequence[int]], seed: int, gain: float = 1.0) -> SD:
    """Deterministic test weights: W ~ N(0, gain^2/fan_in), b ~ N(0, 0.05^2), GN gamma = 1 + 0.1 N, beta = 0.1 N.

    Deliberately NOT MONAI's init (conv2/out zero-init would make eps_hat == 0 and parity vacuous,
    SURVEY.md section 8d config 1).  CPU generator => identical on every machine with the same torch.
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: SD = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        if name.endswith(".weight") and len(shape) == 1:      # GroupNorm gamma
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            is_gn = (name[:-5] + ".weight") in shapes and len(tuple(shapes[name[:-5] + ".weight"])) == 1
            sd[name] = (0.1 if is_gn else 0.05) * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            gg = gain * (2.0 if (".to_q." in name or ".to_k." in name) else 1.0)
            sd[name] = (gg / math.sqrt(fan_in)) * torch.randn(shape, generator=g)
    return sd
