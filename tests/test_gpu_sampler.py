"""Device-resident sampler (-m gpu): in-kernel Philox noise, device step counter, one graph per denoising step."""
import numpy as np
import pytest
import torch

import cfgs
from util import rel_l2

pytestmark = pytest.mark.gpu


def test_philox_noise_is_standard_normal_and_reproducible(cuda):
    from scipy import stats
    from ldm3d.schedulers import DDPMScheduler
    sch = DDPMScheduler(**cfgs.SCHED)
    a, b = sch.device_sampler(seed=1234), sch.device_sampler(seed=1234)
    z = a.noise(7, (1, 4, 24, 24, 24), cuda)
    assert torch.equal(z, b.noise(7, (1, 4, 24, 24, 24), cuda))                      # a seed and a step pin the draw
    assert not torch.equal(z, a.noise(8, (1, 4, 24, 24, 24), cuda))                  # ... every step gets its own
    assert not torch.equal(z, sch.device_sampler(seed=1235).noise(7, (1, 4, 24, 24, 24), cuda))
    big = a.noise(3, (4_000_003,), cuda).double().cpu().numpy()                       # odd length: the quad tail
    assert abs(big.mean()) < 3e-3 and abs(big.var() - 1.0) < 5e-3
    assert abs(stats.skew(big)) < 1e-2 and abs(stats.kurtosis(big)) < 2e-2
    ks = stats.kstest(big[:200_000], "norm")
    assert ks.pvalue > 1e-3, ks
    # no correlation between neighbouring elements or between consecutive steps of one element
    assert abs(np.corrcoef(big[:-1], big[1:])[0, 1]) < 3e-3
    nxt = a.noise(4, (4_000_003,), cuda).double().cpu().numpy()
    assert abs(np.corrcoef(big, nxt)[0, 1]) < 3e-3
    assert np.isfinite(big).all() and np.abs(big).max() < 7.0


@pytest.mark.parametrize("kind,nsteps", [("ddpm", 1000), ("ddim", 50)])
def test_fused_step_equals_the_host_driven_step(cuda, kind, nsteps):
    """Every step of a chain: the fused kernel (device coefficients / counter / noise) == DDPMScheduler.step / DDIMScheduler.step
    driven from the host with the same z, bit for bit; the timestep buffer walks scheduler.timesteps."""
    from ldm3d.schedulers import DDIMScheduler, DDPMScheduler
    sch = DDPMScheduler(**cfgs.SCHED) if kind == "ddpm" else DDIMScheduler(**cfgs.SCHED)
    if kind == "ddim":
        sch.set_timesteps(nsteps)
    smp = sch.device_sampler(seed=99)
    g = torch.Generator(device=cuda).manual_seed(5)
    x = torch.randn((2, 4, 8, 8, 8), device=cuda, generator=g)
    ref = x.clone()
    tbuf = torch.empty((2,), device=cuda)
    smp.reset(tbuf)
    ts = sch.timesteps.tolist()
    for k in list(range(6)) + [len(ts) - 2, len(ts) - 1]:
        if k >= 6:                                        # jump: replay the counter up to k (cheap steps on a dummy tensor)
            smp.reset(tbuf)
            dummy = torch.zeros_like(x)
            for _ in range(k):
                smp.step(dummy, dummy.clone(), tbuf)
        t = ts[k]
        assert tbuf.tolist() == [float(t)] * 2
        eps = torch.randn(x.shape, device=cuda, generator=g)
        z = smp.noise(k, x.shape, cuda)
        if kind == "ddpm":
            want, want_x0 = sch.step(eps, t, ref, noise=z)
        else:
            want, want_x0 = sch.step(eps, t, ref)
        x0 = torch.empty_like(x)
        xin = ref.clone()
        smp.step(eps, xin, tbuf, x0_out=x0)
        # same formula, same fp32 coefficients, same z; the two kernels may contract multiply-adds differently: <= 1 ulp apart
        assert rel_l2(xin, want) <= 2e-7 and rel_l2(x0, want_x0) <= 2e-7, (kind, k, rel_l2(xin, want))
        ref = want
    smp.step(eps, xin, tbuf)                               # beyond the last step: x unchanged, t stays at the last timestep
    assert torch.equal(xin, ref) and tbuf.tolist() == [float(ts[-1])] * 2


def test_denoise_step_graph_equals_forward_plus_step(cuda):
    """UNet forward + fused scheduler step as one call (and as one HIP graph) == the two calls made separately."""
    from ldm3d.networks import DiffusionModelUNet
    from ldm3d.schedulers import DDPMScheduler
    from oracle import unet as ou
    cfg = cfgs.UNET_TINY_COND
    m = DiffusionModelUNet(**cfg)
    m.load_state_dict(ou.init_state_dict(ou.unet_param_shapes(cfg), 1))
    m = m.to(cuda).eval()
    sch = DDPMScheduler(**cfgs.SCHED)
    g = torch.Generator(device=cuda).manual_seed(2)
    x0 = torch.randn((1, 4, 8, 8, 8), device=cuda, generator=g)
    cond = torch.randn((1, 4, 8, 8, 8), device=cuda, generator=g)
    tbuf = torch.empty((1,), device=cuda)
    with torch.no_grad():
        # reference: separate calls
        a = sch.device_sampler(seed=7)
        xa = x0.clone()
        a.reset(tbuf)
        for _ in range(5):
            eps = m(x=xa, timesteps=tbuf, cond=cond)
            a.step(eps, xa, tbuf)
        results = []
        for graph in (False, True):
            m.enable_graph_replay(graph)
            b = sch.device_sampler(seed=7)
            xb = x0.clone()
            b.reset(tbuf)
            for _ in range(5):
                m.denoise_step(xb, tbuf, b, cond=cond)
            results.append(xb.clone())
        m.enable_graph_replay(False)
    assert torch.equal(xa, results[0]) and torch.equal(xa, results[1])
    assert tbuf.tolist() == [994.0]


def test_inferer_sample_with_fused_seed(cuda):
    from ldm3d.inferer import LatentDiffusionInferer
    from ldm3d.networks import DiffusionModelUNet
    from ldm3d.schedulers import DDIMScheduler
    from oracle import unet as ou
    cfg = cfgs.UNET_TINY
    m = DiffusionModelUNet(**cfg)
    m.load_state_dict(ou.init_state_dict(ou.unet_param_shapes(cfg), 1))
    m = m.to(cuda).eval()
    sch = DDIMScheduler(**cfgs.SCHED)
    sch.set_timesteps(10)
    inf = LatentDiffusionInferer(sch)
    noise = torch.randn((1, 4, 8, 8, 8), device=cuda)
    fused = inf.sample(noise, None, m, fused_seed=3)
    plain = inf.sample(noise, None, m)                       # DDIM (eta = 0) draws no noise: the two loops must agree exactly
    assert rel_l2(fused, plain) <= 1e-5
    assert torch.isfinite(fused).all()


def test_graph_replay_never_reuses_a_graph_recorded_for_a_destroyed_sampler(cuda):
    """``inferer.sample(fused_seed=...)`` builds a new DeviceSampler per call and frees the old one; malloc readily hands the new
    ldm_sampler (and torch the new x / tbuf) the old addresses.  The captured sampler kernel bakes seed, step table and state pointers
    in by value, so the replay cache must be keyed on the sampler's identity, not its address: a second chain with another seed and
    another schedule (DDPM: the seed matters) must equal its own eager run, not replay the first chain's graph."""
    import gc
    from ldm3d.inferer import LatentDiffusionInferer
    from ldm3d.networks import DiffusionModelUNet
    from ldm3d.schedulers import DDPMScheduler
    from oracle import unet as ou
    cfg = cfgs.UNET_TINY
    m = DiffusionModelUNet(**cfg)
    m.load_state_dict(ou.init_state_dict(ou.unet_param_shapes(cfg), 1))
    m = m.to(cuda).eval()
    noise = torch.randn((1, 4, 8, 8, 8), device=cuda)

    def chain(seed, n_train, graph):
        sch = DDPMScheduler(**dict(cfgs.SCHED, num_train_timesteps=n_train))
        m.enable_graph_replay(graph)
        out = LatentDiffusionInferer(sch).sample(noise, None, m, fused_seed=seed, verbose=False).clone()
        gc.collect()                                        # the DeviceSampler of this call is destroyed here
        return out
    eager_a, eager_b = chain(11, 12, False), chain(12, 9, False)
    assert not torch.equal(eager_a, eager_b)
    ga = chain(11, 12, True)
    gb = chain(12, 9, True)                                 # same model, same staging tensors, recycled sampler address
    ga2 = chain(11, 12, True)
    m.enable_graph_replay(False)
    assert torch.equal(ga, eager_a) and torch.equal(gb, eager_b) and torch.equal(ga2, eager_a)


def test_time_embedding_table_follows_weight_uploads_schedules_and_precision(cuda):
    """``denoise_step`` takes the 17 stacked time_emb_proj outputs of the sampler's current step from a table built once per
    parameter upload (csrc temb_row_kernel), ``forward`` computes them from ``timesteps`` (sinusoid + 3 GEMVs).
    Same kernels at another batch size, so the two must agree bit for bit -- after a second ``load_state_dict`` (stale table), with
    another schedule on the same module (DDIM, 7 steps), past the end of the schedule, and in the fp32 precision mode."""
    from ldm3d.networks import DiffusionModelUNet
    from ldm3d.schedulers import DDIMScheduler, DDPMScheduler
    from oracle import unet as ou
    cfg = cfgs.UNET_TINY
    m = DiffusionModelUNet(**cfg)
    m.load_state_dict(ou.init_state_dict(ou.unet_param_shapes(cfg), 1))
    m = m.to(cuda).eval()
    x0 = torch.randn((2, 4, 8, 8, 8), device=cuda, generator=torch.Generator(device=cuda).manual_seed(5))
    tbuf = torch.empty((2,), device=cuda)

    def both(sch, steps, graph):
        with torch.no_grad():
            a = sch.device_sampler(seed=3)
            xa = x0.clone()
            a.reset(tbuf)
            m.enable_graph_replay(False)
            for _ in range(steps):
                a.step(m(x=xa, timesteps=tbuf), xa, tbuf)
            b = sch.device_sampler(seed=3)
            xb = x0.clone()
            b.reset(tbuf)
            m.enable_graph_replay(graph)
            for _ in range(steps):
                m.denoise_step(xb, tbuf, b)
            m.enable_graph_replay(False)
        return xa, xb
    ddpm = DDPMScheduler(**cfgs.SCHED)
    ddim = DDIMScheduler(**cfgs.SCHED)
    ddim.set_timesteps(7)
    xa, xb = both(ddpm, 4, True)
    assert torch.equal(xa, xb)
    first = xb.clone()
    m.load_state_dict({k: v * 1.25 for k, v in ou.init_state_dict(ou.unet_param_shapes(cfg), 1).items()})
    xa, xb = both(ddpm, 4, True)
    assert torch.equal(xa, xb) and not torch.equal(xb, first)        # the table was rebuilt from the new weights
    xa, xb = both(ddim, 9, False)                                    # another schedule; two calls beyond its 7 steps
    assert torch.equal(xa, xb)
    xa, xb = both(ddpm, 3, True)                                     # and back
    assert torch.equal(xa, xb)
    m.set_precision("fp32")
    xa, xb = both(ddpm, 3, True)
    assert torch.equal(xa, xb)
    m.set_precision("bf16")
    xa, xb = both(ddpm, 3, False)
    assert torch.equal(xa, xb)
