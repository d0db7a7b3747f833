"""Pins the CPU oracle with closed-form known answers: the reference ships no golden vectors
(3d_ldm/test_losses.py:11-86 asserts nothing) and MONAI is not importable, so these are what stands behind it."""
import math

import pytest
import torch
import torch.nn.functional as F

import cfgs
from oracle import autoencoder as oa
from oracle import unet as ou
from oracle.schedulers import OracleDDIM, OracleDDPM
from util import rel_l2


def test_scheduler_tables():
    s = OracleDDPM(**cfgs.SCHED)
    assert abs(float(s.betas[0]) - 0.0015) < 1e-9 and abs(float(s.betas[-1]) - 0.0195) < 1e-8
    assert torch.all(s.alphas_cumprod[1:] < s.alphas_cumprod[:-1])          # abar strictly decreasing
    assert s.timesteps.tolist() == list(range(999, -1, -1))
    # scaled_linear_beta: sqrt(beta) is linear
    sb = s.betas.sqrt()
    assert torch.allclose(sb[1:] - sb[:-1], (sb[-1] - sb[0]) / 999 * torch.ones(999), atol=1e-7)


def test_ddpm_step_known_answers():
    s = OracleDDPM(**cfgs.SCHED)
    g = torch.Generator().manual_seed(0)
    x0 = torch.rand((1, 2, 3, 3, 3), generator=g) * 1.6 - 0.8               # inside the clip range
    eps = torch.randn(x0.shape, generator=g)
    for t in (0, 1, 500, 999):
        xt = s.add_noise(x0, eps, torch.tensor([t]))
        prev, x0_hat = s.step(eps, t, xt, torch.zeros_like(x0))
        assert rel_l2(x0_hat, x0) < 2e-5 * (1 / float(s.alphas_cumprod[t]) ** 0.5)   # exact eps recovers x0
    # t == 0: no noise is added whatever z is; x_{-1} = x0_hat (abar_{-1} := 1 -> c0 = 1, c1 = 0)
    xt = s.add_noise(x0, eps, torch.tensor([0]))
    p1, x0h = s.step(eps, 0, xt, torch.full_like(x0, 1e6))
    assert torch.equal(p1, s.step(eps, 0, xt, None)[0]) and rel_l2(p1, x0h) < 1e-4   # 1 - abar_0 cancels in fp32
    # posterior variance at t: sigma^2 = (1-abar_{t-1})/(1-abar_t) beta_t
    t = 400
    z = torch.ones_like(x0)
    d = s.step(eps, t, xt, z)[0] - s.step(eps, t, xt, torch.zeros_like(z))[0]
    var = (1 - s.alphas_cumprod[t - 1]) / (1 - s.alphas_cumprod[t]) * s.betas[t]
    assert torch.allclose(d, var.sqrt() * z, rtol=1e-5, atol=1e-7)


def test_ddim_known_answers():
    s = OracleDDIM(**cfgs.SCHED)
    s.set_timesteps(10)
    assert s.timesteps.tolist() == [900, 800, 700, 600, 500, 400, 300, 200, 100, 0]
    g = torch.Generator().manual_seed(1)
    x0 = torch.rand((1, 2, 3, 3, 3), generator=g) - 0.5
    eps = torch.randn(x0.shape, generator=g)
    # eta = 0 with the exact eps walks the same (x0, eps) pair down the schedule: x_prev = sqrt(abar_prev) x0 + sqrt(1-abar_prev) eps
    xt = s.add_noise(x0, eps, torch.tensor([900]))
    prev, x0_hat = s.step(eps, 900, xt)
    assert rel_l2(prev, s.add_noise(x0, eps, torch.tensor([800]))) < 1e-4
    # last step lands exactly on x0 (set_alpha_to_one)
    x_last = s.add_noise(x0, eps, torch.tensor([0]))
    assert rel_l2(s.step(eps, 0, x_last)[0], x0) < 1e-5
    # clip: x0_hat is clamped to [-1, 1] but eps is NOT recomputed
    big = 5 * torch.ones_like(x0)
    prev, x0_hat = s.step(torch.zeros_like(x0), 500, big)
    assert float(x0_hat.max()) == 1.0
    assert torch.allclose(prev, s.alphas_cumprod[400].sqrt() * torch.ones_like(x0))


def test_timestep_embedding_cos_first():
    e = ou.timestep_embedding(torch.tensor([0.0, 3.0]), 8)
    assert torch.equal(e[0], torch.tensor([1, 1, 1, 1, 0, 0, 0, 0.0]))       # t = 0: cos = 1 first, sin = 0 after
    f = torch.exp(-math.log(10000) * torch.arange(4) / 4)
    assert torch.allclose(e[1], torch.cat([torch.cos(3 * f), torch.sin(3 * f)]), atol=1e-6)


def test_zero_init_blocks_are_identity_and_fresh_unet_is_zero():
    cfg = cfgs.UNET_TINY
    sd = ou.init_state_dict(ou.unet_param_shapes(cfg), 0)
    for k in sd:
        if ".conv2.conv." in k or k.startswith("out.2.conv."):
            sd[k] = torch.zeros_like(sd[k])                                   # MONAI zero_module()
    x = torch.randn((1, 4, 8, 8, 8))
    assert float(ou.unet_forward(sd, cfg, x, torch.tensor([7.0])).abs().max()) == 0.0
    c = ou.norm_cfg(cfg)
    h = torch.randn((1, 64, 4, 4, 4))
    emb = torch.randn((1, 256))
    out = ou.resnet_block(sd, "down_blocks.0.resnets.0", h, emb, c, False)   # Cin == Cout: identity skip
    assert torch.equal(out, h)


def test_group_norm_constant_and_attention_constant_v():
    c = ou.norm_cfg(cfgs.UNET_TINY)
    sd = {"n.weight": torch.rand(64) + 0.5, "n.bias": torch.randn(64)}
    y = ou.group_norm(sd, "n", torch.full((1, 64, 2, 2, 2), 3.7), 32, 1e-6)
    assert torch.allclose(y, sd["n.bias"][None, :, None, None, None].expand_as(y), atol=2e-3)   # GN(const) = beta
    # attention with V rows all equal returns that row, whatever Q, K are
    sd = ou.init_state_dict({k: v for k, v in ou.unet_param_shapes(cfgs.UNET_TINY).items()
                             if k.startswith("middle_block.attention")}, 3)
    sd["middle_block.attention.attn.to_v.weight"] = torch.zeros_like(sd["middle_block.attention.attn.to_v.weight"])
    x = torch.randn((1, 128, 2, 2, 2))
    out = ou.attention_block(sd, "middle_block.attention", x, 64, c, False)
    const = F.linear(sd["middle_block.attention.attn.to_v.bias"][None], sd["middle_block.attention.attn.out_proj.weight"],
                     sd["middle_block.attention.attn.out_proj.bias"])
    assert torch.allclose(out - x, const[:, :, None, None, None].expand_as(x), atol=1e-5)


def test_flash_emulation_equals_softmax_attention():
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn((1, 2, 150, 64), generator=g) for _ in range(3))
    ref = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v
    got = ou._flash_emulated_attention(q, k, v, 0.125, kv_tile=64)
    assert rel_l2(got, ref) < 5e-3                                            # only P's bf16 rounding separates them


def test_nearest_upsample_then_subsample_is_identity():
    x = torch.randn((1, 3, 4, 5, 6))
    assert torch.equal(F.interpolate(x, scale_factor=2.0, mode="nearest")[:, :, ::2, ::2, ::2], x)


def test_vae_logvar_clamp_and_kl():
    cfg = cfgs.VAE_TINY
    sd = ou.init_state_dict(oa.ae_param_shapes(cfg), 1)
    sd["quant_conv_log_sigma.conv.weight"].zero_()
    sd["quant_conv_log_sigma.conv.bias"] = torch.tensor([100.0] * 4 + [-100.0] * 4)
    mu, sigma = oa.encode(sd, cfg, torch.rand((1, 2, 8, 8, 8)))
    assert torch.allclose(sigma[:, :4], torch.full_like(sigma[:, :4], math.exp(10.0)), rtol=1e-6)
    assert torch.allclose(sigma[:, 4:], torch.full_like(sigma[:, 4:], math.exp(-15.0)), rtol=1e-6)
    # KL of N(0, 1) is 0; of N(mu, 1) is 0.5 |mu|^2 / batch (3d_ldm/utils.py:249-262, incl. its /batch and clamp)
    z = torch.zeros((2, 4, 2, 2, 2))
    assert torch.allclose(oa.kl_loss(z, torch.ones_like(z)), torch.zeros(2), atol=1e-6)
    m = torch.full_like(z, 0.5)
    assert torch.allclose(oa.kl_loss(m, torch.ones_like(z)), torch.full((2,), 0.5 * 0.25 * 32 / 2), rtol=1e-5)
    assert float(oa.kl_loss(100 * torch.ones_like(z), torch.ones_like(z)).max()) == 1000.0


def test_unet_param_count_matches_survey():
    n = sum(math.prod(s) for s in ou.unet_param_shapes(cfgs.UNET_FULL).values())
    assert n == 191_175_172                                                   # "191.18 M params" 
    n = sum(math.prod(s) for s in oa.ae_param_shapes(cfgs.VAE_FULL).values())
    assert abs(n - 20.94e6) < 0.02e6


def test_bf16_network_noise_floor():
    """The fact the end-to-end parity gate rests on: under bf16 storage a 1e-6 input perturbation moves the CPU
    oracle's OWN output by about as much as bf16 differs from fp32, while the fp32 path moves ~1e-5."""
    cfg = cfgs.UNET_TINY
    sd = ou.init_state_dict(ou.unet_param_shapes(cfg), 1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn((1, 4, 8, 8, 8), generator=g)
    t = torch.tensor([37.0])
    b0, b1 = ou.unet_forward(sd, cfg, x, t, True), ou.unet_forward(sd, cfg, x * (1 + 1e-6), t, True)
    f0, f1 = ou.unet_forward(sd, cfg, x, t, False), ou.unet_forward(sd, cfg, x * (1 + 1e-6), t, False)
    floor = rel_l2(b0, f0)
    assert rel_l2(f1, f0) < 1e-4
    assert 0.3 * floor < rel_l2(b1, b0) < 3 * floor and floor > 5e-3


def test_golden_vector_is_reproducible_from_its_seeds():
    """The committed golden only stores outputs; check its inputs regenerate and a cheap slice of the recipe holds."""
    import os
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "unet_full_24.pt"), weights_only=True)
    assert gold["eps_bf16_oracle"].shape == (1, 4, 24, 24, 24)
    floor = rel_l2(gold["eps_bf16_oracle"], gold["eps_fp32_oracle"])
    assert 5e-3 < floor < 0.1
    tabs = torch.load(os.path.join(os.path.dirname(__file__), "golden", "sched_tables.pt"), weights_only=True)
    s = OracleDDPM(**cfgs.SCHED)
    assert torch.equal(tabs["betas"], s.betas) and torch.equal(tabs["alphas_cumprod"], s.alphas_cumprod)


def test_training_golden_is_reproducible_from_its_seeds():
    """tests/golden/train_step_tiny.pt (loss, gradient norms) regenerates from the seeds in make_golden.train_case()."""
    import os
    import sys
    import torch.nn.functional as F
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden as mg
    from oracle import unet as ou
    gold = torch.load(os.path.join(here, "golden", "train_step_tiny.pt"), weights_only=True)
    cfg, sd, x, t, target = mg.train_case()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss = F.mse_loss(ou.unet_forward(leaves, cfg, x, t), target)
    loss.backward()
    assert abs(float(loss) - gold["loss_fp32"]) <= 1e-5 * gold["loss_fp32"]
    total = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in leaves.values()))
    assert abs(float(total) - gold["total_grad_norm_fp32"]) <= 1e-4 * gold["total_grad_norm_fp32"]
    name = "middle_block.resnet_1.conv1.conv.weight"
    assert abs(float(leaves[name].grad.norm()) - gold["grad_norm_fp32"][name]) <= 1e-4 * gold["grad_norm_fp32"][name]


def test_baseline_config1_cpu_plumbing_full_unet_ddim10():
    """BASELINE configs[0] (CPU plumbing, no GPU): the benchmark UNet definition (191 M parameters, seeded non-zero weights) and the
    DDIM schedule run end to end in the oracle on a 1x4x16^3 latent: 10 steps at t = 900 ... 0, finite, reproducible from the seed,
    final sample = the clamped x0 estimate."""
    from oracle import unet as ou
    from oracle.schedulers import OracleDDIM
    cfg = cfgs.UNET_FULL
    sd = ou.init_state_dict(ou.unet_param_shapes(cfg), 0)

    def chain():
        x = torch.randn((1, 4, 16, 16, 16), generator=torch.Generator().manual_seed(0))
        sch = OracleDDIM(**cfgs.SCHED)
        sch.set_timesteps(10)
        assert sch.timesteps.tolist() == list(range(900, -1, -100))
        for t in sch.timesteps.tolist():
            eps = ou.unet_forward(sd, cfg, x, torch.tensor([float(t)]), emulate_bf16=False)
            assert eps.shape == x.shape and torch.isfinite(eps).all()
            x, x0 = sch.step(eps, t, x)
        return x, x0
    torch.manual_seed(123)
    a, a0 = chain()
    assert torch.isfinite(a).all() and float(a.abs().max()) <= 1.0 + 1e-6 and torch.equal(a, a0)


def test_vae_golden_regenerates_from_its_seeds():
    """tests/golden/vae_full_96.pt (BASELINE configs[1]: 1x1x96^3 through the full AutoencoderKL): the fp32 half regenerates bit for bit
    from the stored seed and the recipe in make_golden.vae_case (same torch build), and the two oracles differ by a sane bf16 floor."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "vae_full_96.pt"), weights_only=True)
    assert gold["mu_fp32"].shape == (1, 4, 24, 24, 24) and gold["rec_sub_fp32"].shape == (1, 1, 24, 24, 24)
    assert 1e-3 < rel_l2(gold["mu_bf16"], gold["mu_fp32"]) < 5e-2 and 1e-3 < rel_l2(gold["rec_sub_bf16"], gold["rec_sub_fp32"]) < 1e-1
    sd = ou.init_state_dict(oa.ae_param_shapes(cfgs.VAE_FULL), gold["weight_seed"])
    mu, _ = oa.encode(sd, cfgs.VAE_FULL, make_golden.vae_case(), emulate_bf16=False)
    assert rel_l2(mu, gold["mu_fp32"]) <= (0.0 if gold["torch_version"] == torch.__version__ else 1e-5)
