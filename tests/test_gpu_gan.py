"""Stage-1 GAN tail (-m gpu): PatchDiscriminator + LSGAN on the HIP kernels against the CPU oracle (oracle/discriminator.py)."""
import pytest
import torch
import torch.nn.functional as F

import cfgs
from util import rel_l2

pytestmark = pytest.mark.gpu


def _pair(cuda, in_channels=1, seed=0):
    from ldm3d.discriminator import PatchDiscriminator
    from oracle import discriminator as od
    g = torch.Generator().manual_seed(seed)
    sd = {k: (0.02 * torch.randn(v, generator=g) if k.endswith("weight") else 0.05 * torch.randn(v, generator=g))
          for k, v in od.param_shapes(in_channels).items()}
    d = PatchDiscriminator(spatial_dims=3, num_layers_d=3, channels=32, in_channels=in_channels, out_channels=1, norm="INSTANCE")
    assert {k: tuple(v.shape) for k, v in d.state_dict().items()} == {k: tuple(v) for k, v in od.param_shapes(in_channels).items()}
    d.load_state_dict(sd)
    return d.to(cuda), sd


@pytest.mark.parametrize("dims,b,cin", [((32, 32, 32), 2, 1), ((48, 32, 40), 1, 2)])
def test_patch_discriminator_forward_and_gradients(cuda, dims, b, cin):
    """All five layer outputs, the LSGAN generator-side gradient w.r.t. the INPUT (what flows back into the autoencoder) and every
    parameter gradient of the discriminator step, against torch autograd through the fp32 oracle."""
    from oracle import discriminator as od
    d, sd = _pair(cuda, cin)
    g = torch.Generator().manual_seed(3)
    fake, real = torch.rand((b, cin, *dims), generator=g), torch.rand((b, cin, *dims), generator=g)
    # ---- oracle
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xf = fake.clone().requires_grad_(True)
    o_outs = od.forward(leaves, xf)
    od.lsgan(o_outs[-1], True).backward()                       # generator term: d/d fake
    gx_ref = xf.grad.clone()
    for v in leaves.values():
        v.grad = None
    loss_d_ref = 0.5 * (od.lsgan(od.forward(leaves, fake)[-1], False) + od.lsgan(od.forward(leaves, real)[-1], True))
    loss_d_ref.backward()
    # ---- HIP
    xd = fake.to(cuda).requires_grad_(True)
    outs = d(xd)
    assert len(outs) == 5 and [tuple(o.shape) for o in outs] == [tuple(o.shape) for o in o_outs]
    errs = [rel_l2(a.detach().cpu(), r.detach()) for a, r in zip(outs, o_outs)]
    print("PatchDiscriminator layer outputs vs fp32 oracle:", " ".join(f"{e:.2e}" for e in errs))
    assert max(errs) <= 2.5e-2                                     # bf16 storage between five conv / norm layers
    F.mse_loss(outs[-1], torch.ones_like(outs[-1])).backward()
    e_gx = rel_l2(xd.grad.cpu(), gx_ref)
    d.zero_grad(set_to_none=True)
    loss_d = 0.5 * (torch.mean(d(fake.to(cuda))[-1] ** 2) + torch.mean((d(real.to(cuda))[-1] - 1.0) ** 2))
    loss_d.backward()
    torch.cuda.synchronize()
    assert abs(float(loss_d) - float(loss_d_ref)) <= 2e-2 * abs(float(loss_d_ref))
    errs = {k: rel_l2(p.grad.cpu(), leaves[k].grad) for k, p in d.named_parameters()}
    print("parameter gradients vs fp32 autograd:", " ".join(f"{k.replace('.conv', '')} {e:.2e}" for k, e in errs.items()),
          f"| input gradient {e_gx:.2e}")
    # LeakyReLU has a kink: an element whose pre-activation changes sign between the bf16 and the fp32 forward (|u| below the ~1e-2
    # bf16 noise: ~1 % of the elements) gets slope 0.2 instead of 1, so every LeakyReLU the gradient crosses adds a few 1e-2 that no
    # implementation with bf16 storage can avoid (final_conv sits behind no kink: ~1e-2).  The layers are pinned tightly one by one
    # in test_discriminator_layers_match_torch_on_identical_inputs; here: bounded, and the direction is right.
    assert errs["final_conv.conv.weight"] <= 3e-2 and errs["final_conv.conv.bias"] <= 3e-2
    assert max(errs.values()) <= 0.2 and e_gx <= 0.2
    a = torch.cat([p.grad.reshape(-1).double().cpu() for _, p in d.named_parameters()])
    r = torch.cat([leaves[k].grad.reshape(-1).double() for k, _ in d.named_parameters()])
    assert float(a @ r / (a.norm() * r.norm())) >= 0.99


def test_vae_gan_step_trains_both_networks(cuda):
    """AutoencoderTrainer after the warm-up epochs: the adversarial term reaches the autoencoder's gradients (they differ from the
    warm-up step's), the discriminator step lowers its own loss on a fixed batch, every loss is finite."""
    from ldm3d.networks import AutoencoderKL
    from ldm3d.trainer import AutoencoderTrainer
    from oracle import autoencoder as oa
    from oracle.unet import init_state_dict
    cfg = cfgs.VAE_TINY_ATTN
    ae = AutoencoderKL(**cfg)
    ae.load_state_dict(init_state_dict(oa.ae_param_shapes(cfg), 3, gain=0.7))
    ae = ae.to(cuda)
    torch.manual_seed(0)                                     # the discriminator's N(0, 0.02) initialisation
    with pytest.warns(UserWarning, match="perceptual"):
        tr = AutoencoderTrainer(ae, lr=1e-4, kl_weight=1e-6, perceptual_weight=1e-3, warm_up_epochs=1, adv_weight=0.5)
    g = torch.Generator().manual_seed(1)
    x = torch.rand((2, 1, 32, 32, 32), generator=g).to(cuda)
    eps = torch.randn((2, 8, 8, 8, 8), generator=g).to(cuda)
    l0, skipped = tr.train_step(x, epoch=0, eps=eps)
    assert not skipped and "adv_g" not in l0 and "adv_d" not in l0          # warm-up: reconstruction + KL only
    g_warm = ae.flat_grads.clone()
    d_before = tr.optimizer_d.flat_params.clone()
    l, skipped = tr.train_step(x, epoch=2, eps=eps)
    assert not skipped and all(bool(torch.isfinite(v)) for v in l.values())
    assert "adv_g" in l and "adv_d" in l and not torch.equal(g_warm, ae.flat_grads)   # the adversarial term reached the autoencoder
    assert not torch.equal(d_before, tr.optimizer_d.flat_params)                      # ... and the discriminator took a step
    # the discriminator step alone on a FIXED (fake, real) pair must make progress on its own objective
    with torch.no_grad():
        ae.eval()
        fake = ae(x, eps=eps)[0].detach()
        ae.train()
    hist = []
    for _ in range(12):
        tr.optimizer_d.zero_grad()
        loss_d = 0.5 * (tr.adv_loss(tr.discriminator(fake), False) + tr.adv_loss(tr.discriminator(x), True))
        loss_d.backward()
        tr.optimizer_d.step()
        hist.append(float(loss_d))
    print("discriminator loss on a fixed batch:", " ".join(f"{v:.4f}" for v in hist))
    assert all(v == v for v in hist) and min(hist[-3:]) < 0.7 * hist[0]

def test_non_finite_adversarial_term_is_dropped_and_a_skipped_generator_step_skips_the_discriminator(cuda):
    """3d_ldm/train_autoencoder.py:417-422: a NaN / inf adversarial term is dropped and the generator step continues on
    reconstruction + KL; :362-365,426-437 `continue` before BOTH optimizers when the batch itself is bad.  Both decisions are taken on
    the device.  (i) A discriminator that returns NaN logits (one poisoned weight): the generator step is NOT skipped, the
    autoencoder's gradients are finite and equal those of the warm-up step (recon + KL only) bit for bit -- the branch's NaN gradient
    is selected away, not multiplied by zero.  (ii) A NaN image: the generator step is skipped and the discriminator's parameters
    and moments do not move either."""
    from ldm3d.networks import AutoencoderKL
    from ldm3d.trainer import AutoencoderTrainer
    from oracle import autoencoder as oa
    from oracle.unet import init_state_dict
    cfg = cfgs.VAE_TINY
    g = torch.Generator().manual_seed(1)
    x = torch.rand((1, 2, 32, 32, 32), generator=g).to(cuda)               # 32^3: the PatchDiscriminator's five 4^3 convs need >= 32 voxels per axis
    eps = torch.randn((1, cfg["latent_channels"], 8, 8, 8), generator=g).to(cuda)

    def make():
        ae = AutoencoderKL(**cfg)
        ae.load_state_dict(init_state_dict(oa.ae_param_shapes(cfg), 3, gain=0.7))
        torch.manual_seed(0)
        return AutoencoderTrainer(ae.to(cuda), lr=1e-4, kl_weight=1e-6, warm_up_epochs=1, adv_weight=0.5), ae
    tr0, ae0 = make()
    tr0.train_step(x, epoch=0, eps=eps)                       # warm-up step: reconstruction + KL only
    g_ref = ae0.flat_grads.clone()
    tr, ae = make()
    with torch.no_grad():
        next(iter(tr.discriminator.parameters())).view(-1)[0] = float("nan")
    out, skipped = tr.train_step(x, epoch=2, eps=eps)
    assert not bool(skipped) and not bool(torch.isfinite(out["adv_g"]))
    assert bool(torch.isfinite(ae.flat_grads).all()) and torch.equal(ae.flat_grads, g_ref)
    assert bool(torch.isfinite(ae.flat_params).all()) and bool(torch.isfinite(out["loss_g"]))
    # (ii) a bad batch: neither network moves
    tr2, ae2 = make()
    p_g, p_d = ae2.flat_params.clone(), tr2.optimizer_d.flat_params.clone()
    bad = x.clone()
    bad[0, 0, 3, 3, 3] = float("nan")
    out, skipped = tr2.train_step(bad, epoch=2, eps=eps)
    assert bool(skipped)
    assert torch.equal(ae2.flat_params, p_g) and torch.equal(tr2.optimizer_d.flat_params, p_d)
    assert float(tr2.optimizer_d.exp_avg.abs().max()) == 0.0 and float(tr2.optimizer_d.sq_norm[1]) == 1.0
    out, skipped = tr2.train_step(x, epoch=2, eps=eps)        # and the next good batch trains both
    assert not bool(skipped) and not torch.equal(tr2.optimizer_d.flat_params, p_d) and not torch.equal(ae2.flat_params, p_g)


@pytest.mark.parametrize("cin,cout,dims,stride,n", [(1, 32, (16, 16, 16), 2, 2), (32, 64, (12, 10, 8), 2, 1), (64, 32, (7, 7, 7), 1, 2), (128, 1, (6, 6, 6), 1, 1)])
def test_discriminator_layers_match_torch_on_identical_inputs(cuda, cin, cout, dims, stride, n):
    """The building blocks one by one against torch autograd on the SAME bf16-rounded inputs (no compounding, no kink flips):
    the 4^3 conv as im2col + GEMM (output, dX through col2im, dW, db) and InstanceNorm + LeakyReLU (output, dX)."""
    from ldm3d.discriminator import _ConvFn, _InstanceNormLeakyFn, _PackFn, _UnpackFn, _rup
    g = torch.Generator().manual_seed(cin + cout)
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.randn((n, cin, *dims), generator=g)).requires_grad_(True)
    w = bf(0.05 * torch.randn((cout, cin, 4, 4, 4), generator=g)).requires_grad_(True)
    b = (0.1 * torch.randn((cout,), generator=g)).requires_grad_(True)
    y = F.conv3d(x, w, b, stride=stride, padding=1)
    dy = bf(torch.randn(y.shape, generator=g))
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), dy)
    xd, wd, bd = x.detach().to(cuda).requires_grad_(True), w.detach().to(cuda).requires_grad_(True), b.detach().to(cuda).requires_grad_(True)
    Cs = _rup(cin, 32)
    h = _PackFn.apply(xd, Cs)
    yk = _UnpackFn.apply(_ConvFn.apply(h, wd, bd, (n, *dims, Cs, cin, 4, stride, 1)), cout)
    yk.backward(dy.to(cuda))
    torch.cuda.synchronize()
    e = dict(y=rel_l2(yk.detach().cpu(), y.detach()), dx=rel_l2(xd.grad.cpu(), gx), dw=rel_l2(wd.grad.cpu(), gw), db=rel_l2(bd.grad.cpu(), gb))
    print(f"conv4 {cin}->{cout} {dims} s{stride}: " + " ".join(f"{k} {v:.2e}" for k, v in e.items()))
    assert e["y"] <= 4e-3 and e["dx"] <= 6e-3 and e["dw"] <= 2e-3 and e["db"] <= 2e-3          # bf16 output / dcol / dx roundings only
    if cout % 32 == 0:
        c = cout
        u = bf(torch.randn((n, c, 6, 5, 7), generator=g)).requires_grad_(True)
        v = F.leaky_relu(F.instance_norm(u, eps=1e-5), 0.2)
        dv = bf(torch.randn(v.shape, generator=g))
        (gu,) = torch.autograd.grad(v, u, dv)
        ud = u.detach().to(cuda).requires_grad_(True)
        vk = _UnpackFn.apply(_InstanceNormLeakyFn.apply(_PackFn.apply(ud, c), c), c)
        vk.backward(dv.to(cuda))
        e_v, e_u = rel_l2(vk.detach().cpu(), v.detach()), rel_l2(ud.grad.cpu(), gu)
        print(f"instance norm + leaky relu C={c}: y {e_v:.2e} dx {e_u:.2e}")
        assert e_v <= 4e-3 and e_u <= 6e-3


def test_perceptual_term_from_user_weights_reaches_the_autoencoder(cuda, tmp_path):
    """AutoencoderTrainer(perceptual_weight > 0, perceptual_weights=file): the term of 3d_ldm/train_autoencoder.py:236,386,406 from a
    user-supplied LPIPS-squeeze state_dict (synthetic here): it is reported, finite, and its gradient reaches the AutoencoderKL through
    ``reconstruction`` and the HIP backward plan (the gradients differ from the step without it); without a file the term is dropped
    and flagged."""
    from ldm3d.networks import AutoencoderKL
    from ldm3d.perceptual import expected_keys
    from ldm3d.trainer import AutoencoderTrainer
    from oracle import autoencoder as oa
    from oracle.unet import init_state_dict
    g = torch.Generator().manual_seed(0)
    sd = {k: (torch.rand(s, generator=g) if k.startswith("lin") else 0.1 * torch.randn(s, generator=g)) for k, s in expected_keys().items()}
    path = str(tmp_path / "lpips_squeeze.pt")
    torch.save(sd, path)
    cfg = cfgs.VAE_TINY                                      # 2 image channels: scored channel by channel (perceptual.py)
    x = torch.rand((1, 2, 32, 32, 32), generator=g).to(cuda)
    eps = torch.randn((1, 8, 8, 8, 8), generator=g).to(cuda)

    def one(weights_file):
        ae = AutoencoderKL(**cfg)
        ae.load_state_dict(init_state_dict(oa.ae_param_shapes(cfg), 3, gain=0.7))
        ae = ae.to(cuda)
        torch.manual_seed(0)
        if weights_file is None:
            with pytest.warns(UserWarning, match="perceptual"):
                tr = AutoencoderTrainer(ae, lr=1e-4, kl_weight=1e-6, perceptual_weight=0.5)
        else:
            tr = AutoencoderTrainer(ae, lr=1e-4, kl_weight=1e-6, perceptual_weight=0.5, perceptual_weights=weights_file)
        torch.manual_seed(4)                                 # the slice draw of the fake 3-D evaluation
        out, skipped = tr.train_step(x, epoch=0, eps=eps)
        return tr, out, bool(skipped), ae.flat_grads.clone()
    tr0, out0, s0, g0 = one(None)
    tr1, out1, s1, g1 = one(path)
    assert tr0.perceptual_dropped and "perceptual" not in out0 and not s0
    assert not tr1.perceptual_dropped and not s1 and bool(torch.isfinite(out1["perceptual"])) and float(out1["perceptual"]) > 0
    assert torch.isfinite(g1).all() and not torch.equal(g0, g1)
    assert abs(float(out1["loss_g"]) - (float(out1["recons"]) + 1e-6 * float(out1["kl"]) + 0.5 * float(out1["perceptual"]))) <= 1e-5


@pytest.mark.parametrize("dims,b,cin", [((32, 32, 32), 2, 1), ((48, 32, 40), 1, 2)])
def test_patch_discriminator_fp32_mode_meets_the_reference_arithmetic(cuda, dims, b, cin):
    """``PatchDiscriminator.set_precision("fp32")`` (``--precision fp32``): the reference trains the discriminator in fp32 when AMP is off
    (3d_ldm/train_autoencoder.py:150-158,454-494).  Same checks as the bf16 test, against the fp32 oracle, at the 1e-3 parity bar of
    BASELINE.json with no floor allowance: all five layer outputs, the generator-side gradient w.r.t. the input, the discriminator loss
    and every parameter gradient of the discriminator step."""
    from oracle import discriminator as od
    d, sd = _pair(cuda, cin)
    d.set_precision("fp32")
    g = torch.Generator().manual_seed(3)
    fake, real = torch.rand((b, cin, *dims), generator=g), torch.rand((b, cin, *dims), generator=g)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xd = fake.to(cuda).requires_grad_(True)
    outs = d(xd)
    # LeakyReLU's derivative jumps at 0: the oracle takes, element by element, the side of the kink the HIP path took (an element
    # whose normalised value is within rounding distance of 0 may legitimately fall on either side; one such element in a 64-voxel
    # instance moves the whole instance's gradient by ~1e-2, seen at seed 0).  Values are compared WITHOUT that help first.
    free = od.forward(sd, fake)
    errs = [rel_l2(a.detach().cpu(), r) for a, r in zip(outs, free)]
    br_f = [o.detach().cpu() > 0 for o in outs[:4]]
    def other_side(ref_outs, br):                       # (count, largest |oracle activation| among them)
        vals = [r[(r > 0) != m].abs() for r, m in zip(ref_outs[:4], br)]
        return sum(v.numel() for v in vals), max([float(v.max()) for v in vals if v.numel()] or [0.0])
    flips, flip_mag = other_side(free, br_f)
    xf = fake.clone().requires_grad_(True)
    o_outs = od.forward(leaves, xf, branches=br_f)
    od.lsgan(o_outs[-1], True).backward()
    gx_ref = xf.grad.clone()
    for v in leaves.values():
        v.grad = None
    F.mse_loss(outs[-1], torch.ones_like(outs[-1])).backward()
    e_gx = rel_l2(xd.grad.cpu(), gx_ref)
    d.zero_grad(set_to_none=True)
    of, orl = d(fake.to(cuda)), d(real.to(cuda))
    loss_d = 0.5 * (torch.mean(of[-1] ** 2) + torch.mean((orl[-1] - 1.0) ** 2))
    loss_d.backward()
    torch.cuda.synchronize()
    br_r = [o.detach().cpu() > 0 for o in orl[:4]]
    more, mag2 = other_side(od.forward(sd, real), br_r)
    flips, flip_mag = flips + more, max(flip_mag, mag2)
    loss_d_ref = 0.5 * (od.lsgan(od.forward(leaves, fake, branches=br_f)[-1], False) + od.lsgan(od.forward(leaves, real, branches=br_r)[-1], True))
    loss_d_ref.backward()
    gerrs = {k: rel_l2(p.grad.cpu(), leaves[k].grad) for k, p in d.named_parameters()}
    print(f"activations on the other side of the LeakyReLU kink than the free-running oracle: {flips} (largest |value| {flip_mag:.1e})")
    assert flips <= 8 and flip_mag <= 1e-5, "only elements within rounding distance of 0 may change sides"
    print("fp32 PatchDiscriminator vs fp32 oracle: layer outputs", " ".join(f"{e:.1e}" for e in errs), "| input gradient", f"{e_gx:.1e}",
          "| parameter gradients", " ".join(f"{e:.1e}" for e in gerrs.values()))
    assert max(errs) <= 1e-3 and e_gx <= 1e-3 and max(gerrs.values()) <= 1e-3
    assert abs(float(loss_d) - float(loss_d_ref)) <= 1e-5 * abs(float(loss_d_ref))
