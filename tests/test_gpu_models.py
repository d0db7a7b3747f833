"""Model-level parity (-m gpu): the nn.Module shells (-> C ABI -> HIP plan) against the CPU oracle on identical
seeded weights and inputs.

How the gate is stated (DESIGN.md "Parity"):
  * every kernel is gated on its own at rel-L2 <= 3e-4 with identical rounding points (tests/test_gpu_ops.py);
  * a whole bf16 network is CHAOTIC under rounding: perturbing the input of the CPU oracle by 1e-6 moves its own
    bf16-emulated output by ~4e-2 (fp32: 5e-6) because every flipped rounding is re-amplified by ~110 later
    roundings (tests/test_oracle_known_answers.py::test_bf16_network_noise_floor measures this).  Two correct bf16
    implementations that sum in different orders are therefore two independent draws of the same rounding noise,
    and no end-to-end 1e-3 bound can hold between them.  The end-to-end gate is the NOISE FLOOR itself:
        floor   = rel-L2(oracle bf16-emulated, oracle fp32)          (what bf16 costs the CPU reference)
        rel-L2(GPU, oracle fp32)          <= FLOOR_FACTOR_FP32 * floor + 1e-3      (GPU is as close to fp32 as the CPU bf16 path)
        rel-L2(GPU, oracle bf16-emulated) <= FLOOR_FACTOR_BF16 * floor + 1e-3      (two draws of the same noise: ~sqrt(2) * floor)
    The factors are the measured ratios plus margin.  Round 5 printed the ratio of all 57 calls (gpurun_out/r05b_floor_ratios.txt):
    GPU vs fp32 oracle 0.34 - 1.37 x floor (median 1.00; the two above 1.25 are small random-weight cases -- the concat-conditioned tiny
    UNet 1.37, one step of the 16^3 DDIM chain 1.29 -- i.e. draws of the noise, not defects: the 24^3 headline network sits at 0.98),
    GPU vs bf16 oracle 0.42 - 1.20.  tests/test_gpu_negative_controls.py shows what a defect in ONE layer does to the same ratio: a
    zeroed conv tap 7.1 x, a GroupNorm epsilon off by 10 x 12.8 x, two swapped attention heads 11.9 x floor -- all fail this gate and
    the per-block one.
"""
import pytest
import torch

import cfgs
from util import rel_l2

pytestmark = pytest.mark.gpu


FLOOR_FACTOR_FP32 = 1.5         # GPU vs fp32 oracle, in units of the CPU oracle's own bf16 floor (measured <= 1.37; round 4: 2.0)
FLOOR_FACTOR_BF16 = 1.5         # GPU vs bf16-emulating oracle: two draws of the same rounding noise (measured <= 1.20; round 4: 2.5)


def floor_gate(got, ref_bf, ref_32, what):
    floor = rel_l2(ref_bf, ref_32)
    e_bf, e_32 = rel_l2(got, ref_bf), rel_l2(got, ref_32)
    print(f"FLOOR_GATE {what}: bf16 floor {floor:.2e} | GPU vs fp32-oracle {e_32:.2e} = {e_32 / floor:.2f} x floor | "
          f"GPU vs bf16-oracle {e_bf:.2e} = {e_bf / floor:.2f} x floor")
    assert torch.isfinite(got).all(), what
    assert e_32 <= FLOOR_FACTOR_FP32 * floor + 1e-3, (what, e_32, floor, e_32 / floor)
    assert e_bf <= FLOOR_FACTOR_BF16 * floor + 1e-3, (what, e_bf, floor, e_bf / floor)
    return floor, e_32, e_bf


def _unet_pair(cfg, seed, cuda):
    from ldm3d.networks import DiffusionModelUNet
    from oracle import unet as ou
    sd = ou.init_state_dict(ou.unet_param_shapes(cfg), seed)
    m = DiffusionModelUNet(**cfg)
    m.load_state_dict(sd)
    return m.to(cuda).eval(), sd


@pytest.mark.parametrize("name,dims,b", [("UNET_TINY", (8, 8, 8), 1), ("UNET_TINY", (8, 12, 4), 2), ("UNET_TINY_ALT", (6, 10, 8), 2),
                                         ("UNET_TINY_HEAD32", (8, 8, 8), 2), ("UNET_TINY_ODD", (8, 8, 8), 2)])
def test_unet_tiny_matches_oracle(cuda, name, dims, b):
    from oracle import unet as ou
    cfg = getattr(cfgs, name)
    m, sd = _unet_pair(cfg, 1, cuda)
    g = torch.Generator().manual_seed(2)
    x = torch.randn((b, cfg["in_channels"], *dims), generator=g)
    t = torch.tensor([37.0, 911.0][:b])
    with torch.no_grad():
        got = m(x=x.to(cuda), timesteps=t.to(cuda), context=None).cpu()
    floor_gate(got, ou.unet_forward(sd, cfg, x, t, emulate_bf16=True), ou.unet_forward(sd, cfg, x, t, emulate_bf16=False),
               f"{name} {dims}")


def test_unet_first_layers_tight(cuda):
    """Before the rounding noise has had layers to compound, model output and oracle agree tightly: a 1-level UNet
    with a single ResBlock (conv_in -> Res -> mid(Res, Attn, Res) -> 2 Res -> out) stays within 1.5e-2 of the
    bf16-emulating oracle and the floor gate holds."""
    from oracle import unet as ou
    cfg = dict(spatial_dims=3, in_channels=4, out_channels=4, channels=[64], attention_levels=[False],
               num_head_channels=64, num_res_blocks=1, norm_num_groups=32)
    m, sd = _unet_pair(cfg, 41, cuda)
    g = torch.Generator().manual_seed(42)
    x = torch.randn((1, 4, 6, 6, 6), generator=g)
    t = torch.tensor([250.0])
    with torch.no_grad():
        got = m(x=x.to(cuda), timesteps=t.to(cuda)).cpu()
    ref_bf = ou.unet_forward(sd, cfg, x, t, emulate_bf16=True)
    floor_gate(got, ref_bf, ou.unet_forward(sd, cfg, x, t, emulate_bf16=False), "1-level UNet")
    assert rel_l2(got, ref_bf) <= 1.5e-2


def test_unet_concat_conditioning_paths_agree(cuda):
    """mode="concat": passing cond separately (packed by the kernel) == torch.cat on the caller side == oracle."""
    from oracle import unet as ou
    cfg = cfgs.UNET_TINY_COND
    m, sd = _unet_pair(cfg, 3, cuda)
    g = torch.Generator().manual_seed(4)
    x = torch.randn((2, 4, 8, 8, 8), generator=g)
    c = torch.randn((2, 4, 8, 8, 8), generator=g)
    t = torch.tensor([5.0, 640.0])
    with torch.no_grad():
        a = m(x=x.to(cuda), timesteps=t.to(cuda), cond=c.to(cuda)).cpu()
        bb = m(x=torch.cat([x, c], 1).to(cuda), timesteps=t.to(cuda)).cpu()
    assert torch.equal(a, bb)
    xc = torch.cat([x, c], 1)
    floor_gate(a, ou.unet_forward(sd, cfg, xc, t, emulate_bf16=True), ou.unet_forward(sd, cfg, xc, t, emulate_bf16=False),
               "concat-conditioned UNet")


def test_unet_fresh_module_outputs_zero(cuda):
    """MONAI zero-inits conv2 / out: a freshly constructed UNet predicts exactly 0 (known answer)."""
    from ldm3d.networks import DiffusionModelUNet
    m = DiffusionModelUNet(**cfgs.UNET_TINY).to(cuda).eval()
    x = torch.randn((1, 4, 8, 8, 8), device=cuda)
    with torch.no_grad():
        out = m(x=x, timesteps=torch.tensor([10.0], device=cuda))
    assert float(out.abs().max()) == 0.0


def test_unet_is_deterministic_and_weights_resync(cuda):
    cfg = cfgs.UNET_TINY
    m, sd = _unet_pair(cfg, 5, cuda)
    x = torch.randn((1, 4, 8, 8, 8), device=cuda)
    t = torch.tensor([100.0], device=cuda)
    with torch.no_grad():
        a = m(x=x, timesteps=t)
        b = m(x=x, timesteps=t)
        assert torch.equal(a, b)                         # split-K slabs, no atomics: bitwise reproducible
        sd2 = {k: v * 1.5 for k, v in sd.items()}
        m.load_state_dict(sd2)
        c = m(x=x, timesteps=t)
    assert not torch.equal(a, c)                         # new weights were re-packed into the arena


def test_unet_full_size_16cube_matches_oracle(cuda):
    """BASELINE config 1 shapes: benchmark UNet on 1x4x16^3."""
    from oracle import unet as ou
    cfg = cfgs.UNET_FULL
    m, sd = _unet_pair(cfg, 0, cuda)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, 4, 16, 16, 16), generator=g)
    t = torch.tensor([500.0])
    with torch.no_grad():
        got = m(x=x.to(cuda), timesteps=t.to(cuda)).cpu()
    floor_gate(got, ou.unet_forward(sd, cfg, x, t, emulate_bf16=True), ou.unet_forward(sd, cfg, x, t, emulate_bf16=False),
               "UNET_FULL 16^3")


def test_unet_full_size_24cube_golden(cuda):
    """Headline shape 1x4x24^3 against the committed golden vector (tests/golden/make_golden.py)."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "unet_full_24.pt")
    gold = torch.load(path, weights_only=True)
    cfg = cfgs.UNET_FULL
    m, _ = _unet_pair(cfg, gold["weight_seed"], cuda)
    g = torch.Generator().manual_seed(gold["input_seed"])
    x = torch.randn((1, 4, 24, 24, 24), generator=g)
    with torch.no_grad():
        got = m(x=x.to(cuda), timesteps=torch.tensor([gold["t"]], device=cuda)).cpu()
    floor_gate(got, gold["eps_bf16_oracle"].float(), gold["eps_fp32_oracle"].float(), "UNET_FULL 24^3 (golden)")


# ---------------------------------------------------------------------------------------------- AutoencoderKL
def _vae_pair(cfg, seed, cuda):
    from ldm3d.networks import AutoencoderKL
    from oracle import autoencoder as oa
    from oracle.unet import init_state_dict
    sd = init_state_dict(oa.ae_param_shapes(cfg), seed)
    m = AutoencoderKL(**cfg)
    m.load_state_dict(sd)
    return m.to(cuda).eval(), sd


# VAE_TINY at 40 x 48 x 44 / VAE_FULL at 64^3: enough 64-row tiles for the first layer's fused im2col GEMM (gemm_light_kernel<.., IM2>, K = 64
# with two input channels, K = 32 with one; ragged last tile at 84 480 rows)
@pytest.mark.parametrize("name,dims,b", [("VAE_TINY", (16, 16, 16), 1), ("VAE_TINY", (8, 16, 12), 2), ("VAE_FULL", (32, 32, 32), 1),
                                         ("VAE_TINY_ATTN", (16, 16, 16), 2), ("VAE_FULL_ATTN", (32, 32, 32), 1),
                                         ("VAE_TINY", (40, 48, 44), 1), ("VAE_TINY", (32, 32, 32), 2), ("VAE_FULL", (64, 64, 64), 1)])
def test_vae_encode_decode_match_oracle(cuda, name, dims, b):
    from oracle import autoencoder as oa
    cfg = getattr(cfgs, name)
    m, sd = _vae_pair(cfg, 7, cuda)
    g = torch.Generator().manual_seed(8)
    x = torch.rand((b, cfg["in_channels"], *dims), generator=g)
    f = 2 ** (len(cfg["channels"]) - 1)
    eps = torch.randn((b, cfg["latent_channels"], *[d // f for d in dims]), generator=g)
    with torch.no_grad():
        mu, sigma = m.encode(x.to(cuda))
        z = m.encode_stage_2_inputs(x.to(cuda), eps.to(cuda))
    b_mu, b_sigma = oa.encode(sd, cfg, x, emulate_bf16=True)
    f_mu, f_sigma = oa.encode(sd, cfg, x, emulate_bf16=False)
    floor_gate(mu.cpu(), b_mu, f_mu, f"{name} encode mu")
    floor_gate(sigma.cpu(), b_sigma, f_sigma, f"{name} encode sigma")
    floor_gate(z.cpu(), oa.sampling(b_mu, b_sigma, eps), oa.sampling(f_mu, f_sigma, eps), f"{name} encode z")
    assert rel_l2(z, mu + sigma * eps.to(cuda)) <= 1e-6        # the fused sampling head is exact
    # decode the ORACLE's latent on both sides so the decoder is tested in isolation
    r_z = oa.sampling(f_mu, f_sigma, eps)
    with torch.no_grad():
        rec = m.decode_stage_2_outputs(r_z.to(cuda)).cpu()
    floor_gate(rec, oa.decode(sd, cfg, r_z, emulate_bf16=True), oa.decode(sd, cfg, r_z, emulate_bf16=False), f"{name} decode")


def test_vae_forward_tuple_and_logvar_clamp(cuda):
    """forward -> (recon, z_mu, z_sigma); log-variance clamp edges [-30, 20] (known answer)."""
    import math
    cfg = cfgs.VAE_TINY
    m, sd = _vae_pair(cfg, 9, cuda)
    sd = dict(sd)
    sd["quant_conv_log_sigma.conv.weight"] = torch.zeros_like(sd["quant_conv_log_sigma.conv.weight"])
    half = cfg["latent_channels"] // 2
    bias = torch.full((cfg["latent_channels"],), 100.0)
    bias[half:] = -100.0
    sd["quant_conv_log_sigma.conv.bias"] = bias
    m.load_state_dict(sd)
    x = torch.rand((1, cfg["in_channels"], 8, 8, 8), device=cuda)
    with torch.no_grad():
        rec, mu, sigma = m(x)
    assert rec.shape == x.shape and mu.shape == (1, cfg["latent_channels"], 2, 2, 2)
    assert torch.allclose(sigma[:, :half], torch.full_like(sigma[:, :half], math.exp(10.0)), rtol=1e-5)
    assert torch.allclose(sigma[:, half:], torch.full_like(sigma[:, half:], math.exp(-15.0)), rtol=1e-5)


# ---------------------------------------------------------------------------------------------- schedulers / inferer
def test_scheduler_steps_match_oracle(cuda):
    from ldm3d.schedulers import DDIMScheduler, DDPMScheduler
    from oracle.schedulers import OracleDDIM, OracleDDPM
    g = torch.Generator().manual_seed(11)
    x = torch.randn((2, 4, 6, 6, 6), generator=g)
    e = torch.randn((2, 4, 6, 6, 6), generator=g)
    z = torch.randn((2, 4, 6, 6, 6), generator=g)
    d, od = DDPMScheduler(**cfgs.SCHED), OracleDDPM(**cfgs.SCHED)
    for t in (999, 500, 1, 0):
        p, x0 = d.step(e.to(cuda), t, x.to(cuda), noise=z.to(cuda))
        rp, rx0 = od.step(e, t, x, z)
        assert rel_l2(p, rp) <= 1e-6 and rel_l2(x0, rx0) <= 1e-6, t
    i, oi = DDIMScheduler(**cfgs.SCHED), OracleDDIM(**cfgs.SCHED)
    i.set_timesteps(10); oi.set_timesteps(10)
    assert i.timesteps.tolist() == oi.timesteps.tolist() == [900, 800, 700, 600, 500, 400, 300, 200, 100, 0]
    for t in (900, 400, 0):
        p, x0 = i.step(e.to(cuda), t, x.to(cuda))
        rp, rx0 = oi.step(e, t, x)
        assert rel_l2(p, rp) <= 1e-6 and rel_l2(x0, rx0) <= 1e-6, t
    ts = torch.tensor([3, 977])
    n = d.add_noise(x.to(cuda), e.to(cuda), ts.to(cuda))
    assert rel_l2(n, od.add_noise(x, e, ts)) <= 1e-6


def test_ddim_sampling_teacher_forced(cuda):
    """BASELINE config 1 (scaled down): 10 DDIM steps.  A free-running reverse trajectory is chaotic (x0 = (x - s eps)/
    sqrt(abar) amplifies eps by up to ~100x before the clamp), so each GPU step is TEACHER-FORCED with the oracle's
    x_t and compared with the oracle's x_{t-1}; the decode is compared on the oracle's final latent.  The free-running
    LatentDiffusionInferer.sample is then checked for shape / finiteness / determinism."""
    from ldm3d.inferer import LatentDiffusionInferer
    from ldm3d.schedulers import DDIMScheduler
    from oracle import autoencoder as oa
    from oracle import unet as ou
    from oracle.schedulers import OracleDDIM
    ucfg, vcfg = dict(cfgs.UNET_TINY, in_channels=8, out_channels=8), cfgs.VAE_TINY
    unet, usd = _unet_pair(ucfg, 21, cuda)
    vae, vsd = _vae_pair(vcfg, 22, cuda)
    g = torch.Generator().manual_seed(23)
    x = torch.randn((1, 8, 8, 8, 8), generator=g)
    sch, osch = DDIMScheduler(**cfgs.SCHED), OracleDDIM(**cfgs.SCHED)
    sch.set_timesteps(10); osch.set_timesteps(10)
    for t in osch.timesteps.tolist():
        ts = torch.tensor([float(t)])
        e_bf = ou.unet_forward(usd, ucfg, x, ts, emulate_bf16=True)
        e_32 = ou.unet_forward(usd, ucfg, x, ts, emulate_bf16=False)
        with torch.no_grad():
            e_gpu = unet(x=x.to(cuda), timesteps=ts.to(cuda))
            p_gpu, _ = sch.step(e_gpu, t, x.to(cuda))
        floor_gate(e_gpu.cpu(), e_bf, e_32, f"DDIM t={t} eps")
        # same eps in -> same x_{t-1} out (element-wise kernel is exact to 1e-6)
        p_ref, _ = osch.step(e_gpu.cpu(), t, x)
        assert rel_l2(p_gpu, p_ref) <= 1e-6
        x, _ = osch.step(e_32, t, x)
    with torch.no_grad():
        dec = vae.decode_stage_2_outputs((x / 0.9).to(cuda)).cpu()
    floor_gate(dec, oa.decode(vsd, vcfg, x / 0.9, emulate_bf16=True), oa.decode(vsd, vcfg, x / 0.9, emulate_bf16=False), "decode")
    inf = LatentDiffusionInferer(sch, scale_factor=0.9)
    noise = torch.randn((1, 8, 8, 8, 8), generator=g).to(cuda)
    out1 = inf.sample(noise, vae, unet, sch)
    out2, inter = inf.sample(noise, vae, unet, sch, save_intermediates=True, intermediate_steps=300)
    assert out1.shape == (1, 2, 32, 32, 32) and torch.isfinite(out1).all() and torch.equal(out1, out2)
    assert len(inter) == 4                                   # t = 900, 600, 300, 0


def test_inferer_call_concat_mode(cuda):
    """Training-time forward of train_diffusion.py:197-205: encode -> scale -> add_noise -> concat cond -> UNet."""
    from ldm3d.inferer import LatentDiffusionInferer
    from ldm3d.schedulers import DDPMScheduler
    from oracle import inferer as oi
    from oracle.schedulers import OracleDDPM
    ucfg, vcfg = dict(cfgs.UNET_TINY, in_channels=16, out_channels=8), cfgs.VAE_TINY
    unet, usd = _unet_pair(ucfg, 31, cuda)
    vae, vsd = _vae_pair(vcfg, 32, cuda)
    g = torch.Generator().manual_seed(33)
    img = torch.rand((2, 2, 16, 16, 16), generator=g)
    veps = torch.randn((2, 8, 4, 4, 4), generator=g)
    noise = torch.randn((2, 8, 4, 4, 4), generator=g)
    cond = torch.randn((2, 8, 4, 4, 4), generator=g)
    ts = torch.tensor([12, 850])
    inf = LatentDiffusionInferer(DDPMScheduler(**cfgs.SCHED), scale_factor=1.3)
    got = inf(inputs=img.to(cuda), autoencoder_model=vae, diffusion_model=unet, noise=noise.to(cuda),
              timesteps=ts.to(cuda), condition=cond.to(cuda), mode="concat", vae_eps=veps.to(cuda)).cpu()
    args = (usd, ucfg, vsd, vcfg, OracleDDPM(**cfgs.SCHED), 1.3, img, noise, ts.float(), veps, cond, "concat")
    floor_gate(got, oi.inferer_call(*args, emulate_bf16=True), oi.inferer_call(*args, emulate_bf16=False), "inferer __call__ concat")


def test_unet_graph_replay_is_bit_identical_to_eager(cuda):
    """ldm_model_set_graph_mode: the forward plan replayed as a HIP graph (second and later calls on the same buffers)
    gives exactly the eager result, also after the weights and the inputs change."""
    m, sd = _unet_pair(cfgs.UNET_TINY, 5, cuda)
    g = torch.Generator().manual_seed(6)
    xs = [torch.randn((2, 4, 8, 8, 8), generator=g).to(cuda) for _ in range(4)]
    t = torch.tensor([10.0, 700.0], device=cuda)
    with torch.no_grad():
        eager = [m(x=x, timesteps=t).clone() for x in xs]
        m.enable_graph_replay(True)
        replay = [m(x=x, timesteps=t).clone() for x in xs]         # call 1 eager, call 2 captures, calls 3-4 replay
        for a, b in zip(eager, replay):
            assert torch.equal(a, b)
        for p in m.parameters():
            p.mul_(1.01)
        r2 = m(x=xs[0], timesteps=t).clone()
        m.enable_graph_replay(False)
        assert torch.equal(r2, m(x=xs[0], timesteps=t))


def test_upsample_phase_convs_match_fused_upsample_form(cuda, monkeypatch):
    """Inference plans run (nearest x2 upsample -> 3^3 conv) as eight 2^3 phase convolutions with pre-summed weights
    (DESIGN.md section 3.1c); LDM_CONV_PHASE=0 keeps the 27-tap fused-upsample form.  Both forms sit inside the noise-floor
    gate of the oracle, for the UNet (one upsampler) and for the VAE decoder (two)."""
    from oracle import autoencoder as oa
    from oracle import unet as ou
    g = torch.Generator().manual_seed(2)                       # the input of test_unet_tiny_matches_oracle's (8, 12, 4) case
    x = torch.randn((2, 4, 8, 12, 4), generator=g)
    t = torch.tensor([37.0, 911.0])
    z = torch.randn((1, cfgs.VAE_TINY["latent_channels"], 4, 4, 4), generator=g)
    outs = {}
    for phase in ("1", "0"):
        monkeypatch.setenv("LDM_CONV_PHASE", phase)
        m, sd = _unet_pair(cfgs.UNET_TINY, 1, cuda)          # plans are built per module instance: the knob is read then
        v, vsd = _vae_pair(cfgs.VAE_TINY, 7, cuda)
        with torch.no_grad():
            outs[phase] = (m(x=x.to(cuda), timesteps=t.to(cuda)).cpu(), v.decode_stage_2_outputs(z.to(cuda)).cpu())
    ref_bf, ref32 = ou.unet_forward(sd, cfgs.UNET_TINY, x, t, emulate_bf16=True), ou.unet_forward(sd, cfgs.UNET_TINY, x, t, emulate_bf16=False)
    dec_bf, dec32 = oa.decode(vsd, cfgs.VAE_TINY, z, emulate_bf16=True), oa.decode(vsd, cfgs.VAE_TINY, z, emulate_bf16=False)
    for phase in ("1", "0"):
        floor_gate(outs[phase][0], ref_bf, ref32, f"UNet, LDM_CONV_PHASE={phase}")
        floor_gate(outs[phase][1], dec_bf, dec32, f"VAE decode, LDM_CONV_PHASE={phase}")
    assert not torch.equal(outs["1"][0], outs["0"][0])        # the two forms really are different launch plans


def test_sample_concurrent_equals_sequential_sampling(cuda):
    """LatentDiffusionInferer.sample_concurrent: two chains advanced round-robin on their own streams (own module instance
    each, graph replay on) give bit for bit what ``sample`` gives for each noise tensor on its own."""
    from ldm3d.inferer import LatentDiffusionInferer
    from ldm3d.schedulers import DDIMScheduler
    m1, sd = _unet_pair(cfgs.UNET_TINY, 3, cuda)
    m2, _ = _unet_pair(cfgs.UNET_TINY, 3, cuda)
    m_ref, _ = _unet_pair(cfgs.UNET_TINY, 3, cuda)
    v, _ = _vae_pair(cfgs.VAE_TINY, 7, cuda)
    for m in (m1, m2):
        m.enable_graph_replay(True)
    sch = DDIMScheduler(**cfgs.SCHED)
    sch.set_timesteps(5)
    inf = LatentDiffusionInferer(sch, scale_factor=0.8)
    g = torch.Generator().manual_seed(31)
    zs = [torch.randn((1, 4, 4, 4, 4), generator=g).to(cuda), torch.randn((2, 4, 4, 4, 4), generator=g).to(cuda)]
    with pytest.raises(Exception, match="expected a latent"):       # 4-channel UNet latents do not fit the 8-channel tiny VAE
        v.decode_stage_2_outputs(zs[0])
    v, _ = _vae_pair(dict(cfgs.VAE_TINY, latent_channels=4), 7, cuda)
    with torch.no_grad():
        seq = [inf.sample(input_noise=z, autoencoder_model=v, diffusion_model=m_ref, scheduler=sch) for z in zs]
        con = inf.sample_concurrent(zs, v, [m1, m2], scheduler=sch)
    torch.cuda.synchronize()
    for a, b in zip(seq, con):
        assert a.shape == b.shape and torch.equal(a, b)


def test_first_conv_im2col_form_matches_3x3x3_form(cuda, monkeypatch):
    """Inference plans run the networks' first conv (Cin <= 9) as im2col + light GEMM with derived weights (pack_im2col_kernel,
    im2col_weights_kernel); LDM_CONV_IM2COL=0 keeps the 3^3 conv over the channel-padded input.  Same products, different fp32
    summation order: a 1-level UNet (where rounding noise has no depth to grow) agrees to 1e-2 between the forms, with and without
    a concatenated condition, and the VAE encoder (Cin = 2) stays inside the oracle's noise-floor gate."""
    from oracle import autoencoder as oa
    cfg = dict(spatial_dims=3, in_channels=8, out_channels=4, channels=[64], attention_levels=[False], num_head_channels=64,
               num_res_blocks=1, norm_num_groups=32)
    g = torch.Generator().manual_seed(51)
    x, cond = torch.randn((2, 4, 6, 5, 7), generator=g), torch.randn((2, 4, 6, 5, 7), generator=g)
    xfull = torch.randn((1, 8, 6, 6, 6), generator=g)
    t = torch.tensor([100.0, 800.0])
    img = torch.rand((1, cfgs.VAE_TINY["in_channels"], 16, 8, 12), generator=g)
    outs = {}
    monkeypatch.setenv("LDM_GEMM_LIGHT", "1")                   # the im2col form needs the light GEMM
    for mode in ("1", "0"):
        monkeypatch.setenv("LDM_CONV_IM2COL", mode)
        m, sd = _unet_pair(cfg, 5, cuda)
        v, vsd = _vae_pair(cfgs.VAE_TINY, 7, cuda)
        with torch.no_grad():
            outs[mode] = (m(x=x.to(cuda), timesteps=t.to(cuda), cond=cond.to(cuda)).cpu(), m(x=xfull.to(cuda), timesteps=t[:1].to(cuda)).cpu(),
                          v.encode(img.to(cuda))[0].cpu())
    assert rel_l2(outs["1"][0], outs["0"][0]) <= 1e-2 and rel_l2(outs["1"][1], outs["0"][1]) <= 1e-2
    assert not torch.equal(outs["1"][0], outs["0"][0])
    mu_bf, _ = oa.encode(vsd, cfgs.VAE_TINY, img, emulate_bf16=True)
    mu_32, _ = oa.encode(vsd, cfgs.VAE_TINY, img, emulate_bf16=False)
    for mode in ("1", "0"):
        floor_gate(outs[mode][2], mu_bf, mu_32, f"VAE encode mu, LDM_CONV_IM2COL={mode}")


def test_baseline_config1_full_unet_16cube_ddim10_teacher_forced(cuda):
    """BASELINE configs[0] at full size: the benchmark UNet (191 M parameters, SURVEY.md section 8d config 1: seeded N(0, .) weights,
    conv2 / out NOT zero) on a 1x4x16^3 latent, 10 DDIM steps (900, 800, ..., 0).  Every GPU step is teacher-forced with the
    fp32 oracle's x_t; eps is gated against the oracle's bf16 noise floor (measured at the first and the last step), the
    scheduler step against the oracle's to 1e-6."""
    from ldm3d.schedulers import DDIMScheduler
    from oracle import unet as ou
    from oracle.schedulers import OracleDDIM
    cfg = cfgs.UNET_FULL
    unet, sd = _unet_pair(cfg, 0, cuda)
    x = torch.randn((1, 4, 16, 16, 16), generator=torch.Generator().manual_seed(0))
    sch, osch = DDIMScheduler(**cfgs.SCHED), OracleDDIM(**cfgs.SCHED)
    sch.set_timesteps(10); osch.set_timesteps(10)
    assert osch.timesteps.tolist() == list(range(900, -1, -100))
    floor = None
    for t in osch.timesteps.tolist():
        ts = torch.tensor([float(t)])
        e_32 = ou.unet_forward(sd, cfg, x, ts, emulate_bf16=False)
        with torch.no_grad():
            e_gpu = unet(x=x.to(cuda), timesteps=ts.to(cuda))
            p_gpu, _ = sch.step(e_gpu, t, x.to(cuda))
        if t in (900, 0):
            f, _, _ = floor_gate(e_gpu.cpu(), ou.unet_forward(sd, cfg, x, ts, emulate_bf16=True), e_32, f"config 1, t={t}")
            floor = f if floor is None else max(floor, f)
        else:
            err = rel_l2(e_gpu.cpu(), e_32)
            print(f"config 1, t={t}: GPU vs fp32-oracle {err:.2e}")
            assert torch.isfinite(e_gpu).all() and err <= 2.5 * floor + 1e-3, (t, err, floor)
        p_ref, _ = osch.step(e_gpu.cpu(), t, x)
        assert rel_l2(p_gpu, p_ref) <= 1e-6
        x, _ = osch.step(e_32, t, x)
    assert torch.isfinite(x).all() and float(x.abs().max()) <= 1.0 + 1e-6          # clip_sample keeps x0 in [-1, 1]; the last step returns x0


def test_vae_full_size_96cube_golden(cuda):
    """BASELINE configs[1] at full size (AutoencoderKL 64/128/256, 1x1x96^3) against the committed oracle golden
    (tests/golden/make_golden.py vae): encoder mean, and the decoder on the fp32 oracle's latent (stride-4 sub-lattice + moments)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "vae_full_96.pt"), weights_only=True)
    m, _ = _vae_pair(cfgs.VAE_FULL, gold["weight_seed"], cuda)
    with torch.no_grad():
        mu, _ = m.encode(make_golden.vae_case().to(cuda))
        rec = m.decode(gold["mu_for_decode"].to(cuda)).cpu()
    floor_gate(mu.cpu(), gold["mu_bf16"], gold["mu_fp32"], "VAE_FULL 96^3 encode mu (golden)")
    floor_gate(rec[..., ::4, ::4, ::4], gold["rec_sub_bf16"], gold["rec_sub_fp32"], "VAE_FULL 96^3 decode, stride-4 sub-lattice (golden)")
    tol = 3.0 * abs(gold["rec_mean_bf16"] - gold["rec_mean_fp32"]) + 1e-3 * abs(gold["rec_mean_fp32"]) + 1e-4
    assert abs(float(rec.double().mean()) - gold["rec_mean_fp32"]) <= tol
    assert abs(float((rec.double() ** 2).mean()) - gold["rec_msq_fp32"]) <= 3.0 * abs(gold["rec_msq_bf16"] - gold["rec_msq_fp32"]) + 1e-2 * gold["rec_msq_fp32"]


def test_fused_finalize_group_norm_plan_matches_the_default_plan(cuda):
    """The inference plans run every split-K finalize -> GroupNorm pair as ONE launch in which a workgroup owns a whole (sample, group)
    (csrc/fin_gn.h; LDM_FIN_GN=0 keeps the three-launch form).  Same rounding points as the three-launch plan: the benchmark UNet's
    output at 16^3 agrees to a few bf16 ulps of a few elements amplified by the network, the plan has >= 15 launches fewer, graph
    replays are bit-stable.  Runs in child processes: the knob is read once per process."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import json, sys, torch
sys.path.insert(0, "tests")
import cfgs
from ldm3d import _lib
from ldm3d.networks import DiffusionModelUNet
from oracle import unet as ou
dev = torch.device("cuda:0")
m = DiffusionModelUNet(**cfgs.UNET_FULL)
m.load_state_dict(ou.init_state_dict(ou.unet_param_shapes(cfgs.UNET_FULL), 0))
m = m.to(dev).eval()
g = torch.Generator().manual_seed(1)
x = torch.randn((1, 4, 16, 16, 16), generator=g).to(dev)
t = torch.tensor([500.0], device=dev)
with torch.no_grad():
    a = m(x=x, timesteps=t).clone()
    m.enable_graph_replay(True)
    outs = [m(x=x, timesteps=t).clone() for _ in range(20)]
L = _lib.lib()
print(json.dumps({"sum": float(a.double().sum()), "abs": float(a.double().abs().sum()), "replays_equal": all(bool(torch.equal(o, a)) for o in outs),
                  "launches": L.ldm_model_plan_launches(m._h, b"unet", 1, 16, 16, 16),
                  "out": a.flatten()[::97].cpu().tolist()}))
'''
    recs = {}
    for v in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, LDM_FIN_GN=v), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        recs[v] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    a, b = torch.tensor(recs["0"]["out"]), torch.tensor(recs["1"]["out"])
    rel = float((a - b).norm() / a.norm())
    print(f"fused finalize + GroupNorm plan vs default: rel-L2 {rel:.2e}, launches {recs['0']['launches']} -> {recs['1']['launches']}")
    assert recs["0"]["replays_equal"] and recs["1"]["replays_equal"]
    assert recs["1"]["launches"] <= recs["0"]["launches"] - 15
    assert rel <= 5e-2                       # two correct bf16 evaluations of this network (statistics folded in another order): its noise floor


def test_block_conv_plans_match_the_halo_plans_forward_and_gradients(cuda):
    """conv3_block_kernel / conv3_block128_kernel inside the launch plans (inference AND training: forward convs, data-gradient convs, GroupNorm
    partials per block): an AutoencoderKL with 64 channels at full resolution and 128 below on a ragged 12 x 20 x 24 volume, batch 2, with the
    kernels forced on (LDM_CONV_BLOCK_MIN=1, LDM_CONV_BLOCK128_MIN=1, and LDM_CONV_THIN_MIN=1 for the last layer's conv3_thin_kernel, which the
    training forward uses too; the plans use them from 512 / 192 / 512 blocks up) against the same run with all three off.  Both are bf16 evaluations with
    the same rounding points: they agree to the network's bf16 floor, and both sit equally close to the fp32 CPU oracle.  Child processes: the knobs are read
    once per process."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import json, sys, torch
import torch.nn.functional as F
sys.path.insert(0, "tests")
from ldm3d import _lib
from ldm3d.networks import AutoencoderKL
from oracle import autoencoder as oa
from oracle.unet import init_state_dict
cfg = dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=4, channels=[64, 128, 128], num_res_blocks=[2, 2, 1],
           norm_num_groups=16, norm_eps=1e-6, attention_levels=[False, False, False], with_encoder_nonlocal_attn=False,
           with_decoder_nonlocal_attn=False)
sd = init_state_dict(oa.ae_param_shapes(cfg), 3, gain=0.7)
g = torch.Generator().manual_seed(4)
dims = (12, 20, 24)
x = torch.rand((2, 1, *dims), generator=g)
eps = torch.randn((2, 4, 3, 5, 6), generator=g)
dev = torch.device("cuda:0")
m = AutoencoderKL(**cfg); m.load_state_dict(sd); m = m.to(dev)
m.eval()
with torch.no_grad():
    mu, sig = m.encode(x.to(dev))
    rec = m.decode(mu)
L = _lib.lib()
buf = (__import__("ctypes").c_int * 2048)()
n = L.ldm_model_plan_conv_cfgs(m._h, b"enc", 2, *dims, buf, 512)
blocks = sum(1 for i in range(n) if (buf[4 * i + 2] >> 8) in (3, 4))
m.train()
recon, mu2, sigma2 = m(x.to(dev), eps=eps.to(dev))
loss = F.l1_loss(recon, x.to(dev)) + 1e-3 * oa.kl_loss(mu2, sigma2).mean()
loss.backward()
torch.cuda.synchronize()
names = sorted(k for k, _ in m.named_parameters())
got = dict(m.named_parameters())
grads = torch.cat([got[k].grad.reshape(-1) for k in names]).double().cpu()
# the fp32 CPU oracle of the same forward / loss / gradients
leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
o_rec, o_mu, o_sigma = oa.forward(leaves, cfg, x, eps, emulate_bf16=False)
o_loss = F.l1_loss(o_rec, x) + 1e-3 * oa.kl_loss(o_mu, o_sigma).mean()
o_loss.backward()
o_grads = torch.cat([leaves[k].grad.reshape(-1) for k in names]).double()
with torch.no_grad():
    o_dec = oa.decode(sd, cfg, oa.encode(sd, cfg, x)[0])
rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())
print(json.dumps({"blocks": blocks, "rec": rec.flatten()[::7].double().cpu().tolist(), "mu": mu.flatten().double().cpu().tolist(),
                  "loss": float(loss), "grads": grads[::11].tolist(), "gnorm": float(grads.norm()),
                  "e_rec": rel(rec, o_dec), "e_mu": rel(mu, o_mu.detach()), "e_grads": rel(grads, o_grads), "o_loss": float(o_loss)}))
'''
    recs = {}
    for tag, env in (("halo", {"LDM_CONV_BLOCK": "0", "LDM_CONV_BLOCK128": "0", "LDM_CONV_THIN": "0"}), ("block", {"LDM_CONV_BLOCK_MIN": "1", "LDM_CONV_BLOCK128_MIN": "1", "LDM_CONV_THIN_MIN": "1"})):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        recs[tag] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert recs["halo"]["blocks"] == 0 and recs["block"]["blocks"] >= 6, (recs["halo"]["blocks"], recs["block"]["blocks"])
    errs = {}
    for k in ("rec", "mu", "grads"):
        a, b = torch.tensor(recs["halo"][k], dtype=torch.float64), torch.tensor(recs["block"][k], dtype=torch.float64)
        errs[k] = float((a - b).norm() / a.norm())
    print(f"block-conv plans vs halo plans: decode {errs['rec']:.2e}, latent mean {errs['mu']:.2e}, parameter gradients {errs['grads']:.2e}; "
          f"{recs['block']['blocks']} block launches in the encoder plan")
    h, b = recs["halo"], recs["block"]
    print(f"  vs the fp32 CPU oracle, halo / block plans: decode(encode) {h['e_rec']:.2e} / {b['e_rec']:.2e}, latent mean {h['e_mu']:.2e} / "
          f"{b['e_mu']:.2e}, parameter gradients {h['e_grads']:.2e} / {b['e_grads']:.2e}; loss {h['loss']:.5f} / {b['loss']:.5f} / oracle {h['o_loss']:.5f}")
    # two bf16 evaluations of one network differ by its rounding floor (same rounding points, other summation order inside a conv); what
    # must hold is that the block plans sit as close to the fp32 oracle as the halo plans do
    for k in ("e_rec", "e_mu", "e_grads"):
        assert b[k] <= 1.3 * h[k] + 2e-3, (k, h[k], b[k])
    assert errs["rec"] <= 2.0 * h["e_rec"] + 2e-3 and errs["mu"] <= 2.0 * h["e_mu"] + 2e-3 and errs["grads"] <= 2.0 * h["e_grads"] + 2e-3, errs
    assert abs(h["loss"] - b["loss"]) <= 5e-3 * abs(h["loss"])
