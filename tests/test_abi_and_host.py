"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports every symbol include/ldm3d.h
declares; the host mirror (config resolver, scheduler tables, module shells) behaves like the reference's objects."""
import ctypes
import json
import os
import re
import types

import pytest
import torch

import cfgs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "ldm3d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ldm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built_lib):
    from ldm3d import _lib
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(built_lib, n), f"{n} declared in include/ldm3d.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes signature table and header disagree"
    assert built_lib.ldm_version() == 1


def test_error_reporting_never_throws(built_lib):
    from ldm3d import _lib
    h = ctypes.c_void_p()
    cfg = _lib.UNetCfg()
    cfg.spatial_dims = 2
    assert built_lib.ldm_unet_create(ctypes.byref(cfg), ctypes.byref(h)) == -2            # LDM_ERR_UNSUPPORTED
    assert b"spatial_dims" in built_lib.ldm_last_error()
    assert built_lib.ldm_unet_create(None, ctypes.byref(h)) == -1                        # LDM_ERR_BAD_ARG
    with pytest.raises(_lib.LdmError):
        _lib.check(-1)


def test_param_layout_matches_independent_oracle_layout(built_lib):
    """The library enumerates MONAI-shaped state_dict names; the oracle derives them independently."""
    from ldm3d.networks import AutoencoderKL, DiffusionModelUNet
    from oracle import autoencoder as oa
    from oracle import unet as ou
    for cfg in (cfgs.UNET_FULL, cfgs.UNET_TINY, cfgs.UNET_TINY_ALT, cfgs.UNET_TINY_COND):
        m = DiffusionModelUNet(**cfg)
        sd, ref = m.state_dict(), ou.unet_param_shapes(cfg)
        assert set(sd) == set(ref)
        assert all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref)
    for cfg in (cfgs.VAE_FULL, cfgs.VAE_TINY):
        m = AutoencoderKL(**cfg)
        sd, ref = m.state_dict(), oa.ae_param_shapes(cfg)
        assert set(sd) == set(ref)
        assert all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref)


def test_module_surface_the_reference_relies_on(built_lib):
    """state_dict round trip, .parameters(), train()/eval(), DDP-style .module unwrap, MONAI zero init
    (3d_ldm/train_diffusion.py:129-149,291-303)."""
    from ldm3d.networks import DiffusionModelUNet
    m = DiffusionModelUNet(**cfgs.UNET_TINY)
    sd = m.state_dict()
    assert float(sd["out.2.conv.weight"].abs().max()) == 0.0 and float(sd["down_blocks.0.resnets.0.conv2.conv.weight"].abs().max()) == 0.0
    assert float(sd["down_blocks.0.resnets.0.conv1.conv.weight"].abs().max()) > 0.0
    m2 = DiffusionModelUNet(**cfgs.UNET_TINY)
    m2.load_state_dict({k: v + 1 for k, v in sd.items()})
    assert torch.equal(m2.state_dict()["conv_in.conv.bias"], sd["conv_in.conv.bias"] + 1)
    wrapper = types.SimpleNamespace(module=m)
    assert wrapper.module.state_dict().keys() == sd.keys()
    assert m.train().training and not m.eval().training
    assert sum(p.numel() for p in m.parameters()) == sum(v.numel() for v in sd.values())
    with pytest.raises(Exception):                                                       # no CPU fallback
        m(x=torch.zeros((1, 4, 8, 8, 8)), timesteps=torch.zeros(1))


def test_unsupported_configs_fail_loudly(built_lib):
    from ldm3d import _lib
    from ldm3d.networks import AutoencoderKL, DiffusionModelUNet
    with pytest.raises(NotImplementedError):
        DiffusionModelUNet(**dict(cfgs.UNET_TINY, with_conditioning=True, cross_attention_dim=64))
    with pytest.raises(_lib.LdmError):
        DiffusionModelUNet(**dict(cfgs.UNET_TINY, num_head_channels=[0, 48, 48]))       # head_dim must be 32 | 64 | 128 | 256
    with pytest.raises(_lib.LdmError):
        DiffusionModelUNet(**dict(cfgs.UNET_TINY, num_head_channels=[0, 128, 256]))     # ... and divide the level's channels (64, 128)
    DiffusionModelUNet(**dict(cfgs.UNET_TINY, num_head_channels=[0, 32, 32]))           # config_train_stable.json:45-46
    # AutoencoderKL attention blocks (level flags and the non-local ones) construct with the oracle's MONAI-shaped names
    from oracle import autoencoder as oa
    acfg = dict(cfgs.VAE_TINY, attention_levels=[False, False, True], with_encoder_nonlocal_attn=True, with_decoder_nonlocal_attn=True)
    sd = AutoencoderKL(**acfg).state_dict()
    shapes = oa.ae_param_shapes(acfg)
    assert set(sd) == set(shapes) and all(tuple(sd[k].shape) == tuple(v) for k, v in shapes.items())
    assert "encoder.blocks.7.attn.to_q.weight" in sd and "decoder.blocks.2.attn.out_proj.bias" in sd


REFERENCE_STYLE_CONFIG = {
    # same schema as 3d_ldm/config/config_train_16g.json (values reduced): "_target_", "@ref", "$@ref"
    "spatial_dims": 3, "image_channels": 2, "latent_channels": 8,
    "autoencoder_def": {"_target_": "networks.AutoencoderKL", "spatial_dims": "@spatial_dims",
                        "in_channels": "$@image_channels", "out_channels": "@image_channels",
                        "latent_channels": "@latent_channels", "channels": [32, 64, 64], "num_res_blocks": 2,
                        "norm_num_groups": 32, "norm_eps": 1e-06, "attention_levels": [False, False, False],
                        "with_encoder_nonlocal_attn": False, "with_decoder_nonlocal_attn": False},
    "diffusion_def": {"_target_": "monai.networks.nets.DiffusionModelUNet", "spatial_dims": "@spatial_dims",
                      "in_channels": "$@latent_channels * 2", "out_channels": "@latent_channels",
                      "channels": [64, 64, 128], "attention_levels": [False, True, True],
                      "num_head_channels": [0, 64, 64], "num_res_blocks": 2},
    "NoiseScheduler": {"num_train_timesteps": 1000, "beta_start": 0.0015, "beta_end": 0.0195},
}


def test_define_instance_resolves_reference_schema(built_lib, tmp_path):
    from ldm3d.config import define_instance, load_config_namespace
    from ldm3d.networks import AutoencoderKL, DiffusionModelUNet
    (tmp_path / "c.json").write_text(json.dumps(REFERENCE_STYLE_CONFIG))
    (tmp_path / "e.json").write_text(json.dumps({"model_dir": "./m", "output_dir": "./o"}))
    args = load_config_namespace(str(tmp_path / "e.json"), str(tmp_path / "c.json"), gpus=1)
    ae = define_instance(args, "autoencoder_def")
    un = define_instance(args, "diffusion_def")
    assert isinstance(ae, AutoencoderKL) and isinstance(un, DiffusionModelUNet)
    assert ae.in_channels == 2 and ae.latent_channels == 8 and un.in_channels == 16 and un.out_channels == 8
    assert args.model_dir == "./m" and args.NoiseScheduler["beta_end"] == 0.0195
    with pytest.raises(ImportError):
        define_instance({"x": {"_target_": "no.such.Class"}}, "x")
    with pytest.raises(KeyError):
        define_instance({"x": {"_target_": "networks.AutoencoderKL", "spatial_dims": "@missing"}}, "x")


def test_host_scheduler_tables_equal_oracle(built_lib):
    from ldm3d.schedulers import DDIMScheduler, DDPMScheduler
    from oracle.schedulers import OracleDDPM
    d, o = DDPMScheduler(**cfgs.SCHED), OracleDDPM(**cfgs.SCHED)
    assert torch.equal(d.betas, o.betas) and torch.equal(d.alphas_cumprod, o.alphas_cumprod)
    assert d.num_train_timesteps == 1000 and d.timesteps.tolist() == o.timesteps.tolist()
    for t in (0, 1, 500, 999):                                                           # per-step scalars as MONAI computes them
        a_t = o.alphas_cumprod[t]
        a_p = o.alphas_cumprod[t - 1] if t > 0 else o.one
        assert d._c0[t] == pytest.approx(float(a_p ** 0.5 * o.betas[t] / (1 - a_t)), rel=1e-6)
        assert d._c1[t] == pytest.approx(float(o.alphas[t] ** 0.5 * (1 - a_p) / (1 - a_t)), rel=1e-6, abs=1e-12)
    i = DDIMScheduler(**cfgs.SCHED)
    i.set_timesteps(50)
    assert i.timesteps.tolist() == list(range(980, -1, -20))
    with pytest.raises(ValueError):
        i.set_timesteps(1001)
    with pytest.raises(Exception):                                                       # CUDA tensors only
        d.step(torch.zeros(2), 5, torch.zeros(2))


REF_CFG_DIR = "/root/reference/3d_ldm/config"


@pytest.mark.skipif(not os.path.isdir(REF_CFG_DIR), reason="the reference checkout is only present in the authoring container")
def test_every_reference_config_constructs(built_lib):
    """define_instance on autoencoder_def / diffusion_def of ALL shipped reference configs, read in place (never copied):
    attention levels + non-local attention in the AutoencoderKL (config_train_32g.json:21-25, config_train_multigpu.json:21-27,
    config_optimized.json:25-31), num_head_channels 32 (config_train_stable.json:45-46).  Parameter names / shapes must equal the
    oracle's MONAI-shaped layout for the same kwargs."""
    import glob
    from ldm3d.config import ConfigResolver, define_instance
    from oracle import autoencoder as oa, unet as ou
    files = sorted(glob.glob(os.path.join(REF_CFG_DIR, "config_*.json")))
    assert len(files) >= 5
    seen = 0
    for f in files:
        cfg = json.load(open(f))
        for key, shapes_of in (("autoencoder_def", oa.ae_param_shapes), ("diffusion_def", ou.unet_param_shapes)):
            if key not in cfg:
                continue                                  # config_optimized.json (written by check_system.py) has no diffusion_def
            m = define_instance(cfg, key)
            kwargs = {k: v for k, v in ConfigResolver(cfg).get_parsed_content(key, instantiate=False).items() if k != "_target_"}
            want = shapes_of(kwargs)
            got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
            assert got == {k: tuple(v) for k, v in want.items()}, (f, key)
            seen += 1
    assert seen >= 9


def test_state_dict_emits_monai_core_names_and_loads_legacy_checkpoints(built_lib):
    """state_dict() emits the MONAI >= 1.4 ("core") names the reference's ``monai.networks.nets.*`` targets save
    (3d_ldm/train_diffusion.py:92-95,129-136: ``upsampler.postconv``, ``attn.to_q / to_k / to_v / out_proj``), and load_state_dict
    also accepts the legacy MONAI-GenerativeModels names (``upsampler.conv``, ``to_q`` ... ``proj_attn`` directly on the attention
    block, ``decoder.blocks.N.conv``): the renames MONAI's own load_old_state_dict performs."""
    from ldm3d.networks import AutoencoderKL, DiffusionModelUNet
    from oracle import autoencoder as oa, unet as ou

    def legacy(k):
        k = k.replace("upsampler.postconv.", "upsampler.conv.").replace(".postconv.", ".conv.")
        for n in ("to_q", "to_k", "to_v"):
            k = k.replace(f".attn.{n}.", f".{n}.")
        return k.replace(".attn.out_proj.", ".proj_attn.")
    for cls, cfg, shapes in ((DiffusionModelUNet, cfgs.UNET_TINY, ou.unet_param_shapes), (AutoencoderKL, cfgs.VAE_TINY_ATTN, oa.ae_param_shapes)):
        m = cls(**cfg)
        sd = ou.init_state_dict(shapes(cfg), 1)
        assert set(m.state_dict()) == set(sd)
        assert any("postconv" in k for k in sd) and any(".attn.to_q." in k for k in sd) and not any("proj_attn" in k for k in sd)
        old = {legacy(k): v for k, v in sd.items()}
        assert len(set(old) - set(sd)) > 8
        m.load_state_dict(old)                              # strict: every key must land
        assert all(torch.equal(m.state_dict()[k], sd[k]) for k in sd)
        m.load_state_dict({k: v + 1 for k, v in sd.items()})     # current names pass through untouched
        assert torch.equal(m.state_dict()[next(iter(sd))], sd[next(iter(sd))] + 1)
        with pytest.raises(RuntimeError):
            m.load_state_dict({k.replace("norm1", "norm_one"): v for k, v in sd.items()})


def test_headline_plan_launch_count(built_lib):
    """Plans are host-side objects: the launch count of the headline step (BASELINE configs[2]) is checked without a GPU."""
    from ldm3d import _lib
    from ldm3d.networks import DiffusionModelUNet
    m = DiffusionModelUNet(**cfgs.UNET_FULL)
    n = _lib.lib().ldm_model_plan_launches(m._h, b"unet", 1, 24, 24, 24)
    assert 100 <= n <= 155, n


def _bf16_conv_launches(model, kind, batch, dims):
    """(wgm, wgn, bk, halo, splitk) of every launch of the bf16 conv kernels in a cached plan (`ldm_model_plan_conv_cfgs`)."""
    import ctypes as C
    from ldm3d import _lib
    buf = (C.c_int * (4 * 512))()
    n = _lib.lib().ldm_model_plan_conv_cfgs(model._h, kind, batch, *dims, buf, 512)
    assert 0 <= n <= 512, n
    return [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2] & 255, buf[4 * i + 2] >> 8, buf[4 * i + 3]) for i in range(n)]


def test_planner_rules_added_in_round_3(built_lib, monkeypatch):
    """Host-side planner decisions, checked without a GPU.
    * fp32 precision mode, inference: every 3^3 ResBlock convolution and the last conv run as the 3 x bf16 product on conv3_halo_kernel, the
      Upsample convolutions in the 8-tap phase form on the general kernel (the only launches of the bf16 conv kernels in such a plan),
      DESIGN.md section 3.6; the bf16 plan of the same network uses both kernels too.
    * AutoencoderKL decoder at 96^3: the phase-upsample and fused-skip convolutions with >= 512 tiles and <= 64 K steps take 32-channel K
      steps (two four-wave workgroups per CU); LDM_IGEMM_2WG=0 plans them as before."""
    from ldm3d.networks import AutoencoderKL, DiffusionModelUNet
    unet = DiffusionModelUNet(**cfgs.UNET_FULL)
    bf16 = _bf16_conv_launches(unet, b"unet", 1, (24, 24, 24))
    assert any(c[3] for c in bf16) and any(not c[3] for c in bf16)
    unet.set_precision("fp32")
    fp32 = _bf16_conv_launches(unet, b"unet", 1, (24, 24, 24))
    assert len(fp32) >= 30 and sum(1 for c in fp32 if not c[3]) == 2, fp32     # all on the halo kernel but the two phase-form Upsample convs
    assert all(c[4] == 1 for c in fp32[:4]), "the 24^3 level runs unsplit (and finishes in the conv's own epilogue)"
    vae32 = AutoencoderKL(**cfgs.VAE_FULL).set_precision("fp32")
    dec32 = _bf16_conv_launches(vae32, b"dec", 1, (24, 24, 24))
    # the two Upsample convs in the 8-tap phase form on the general kernel (two views of the split tensor as its concatenated sources),
    # every other 3^3 conv of the decoder, its last one included, on the halo kernel
    assert sum(1 for c in dec32 if not c[3]) == 2 and sum(1 for c in dec32 if c[3]) >= 12, dec32
    two = [c for c in _bf16_conv_launches(AutoencoderKL(**cfgs.VAE_FULL), b"dec", 1, (24, 24, 24)) if c[:4] == (2, 2, 32, 0)]
    monkeypatch.setenv("LDM_IGEMM_2WG", "0")
    one = [c for c in _bf16_conv_launches(AutoencoderKL(**cfgs.VAE_FULL), b"dec", 1, (24, 24, 24)) if c[:4] == (2, 2, 32, 0)]
    assert len(one) == 1 and len(two) == 3, (one, two)      # conv_in (Cin = 32) always; + the two phase-upsample convs
    # round 4: the fused-skip convs (ResBlock conv2 + 1x1 nin_shortcut) run on the halo kernel's second K loop, unsplit; LDM_HALO_SKIP=0
    # sends them back to the general kernel (where the 48^3 one takes the two-workgroups-per-CU form again)
    monkeypatch.delenv("LDM_IGEMM_2WG")
    halo_now = sum(1 for c in _bf16_conv_launches(AutoencoderKL(**cfgs.VAE_FULL), b"dec", 1, (24, 24, 24)) if c[3])
    bf16_unet = _bf16_conv_launches(DiffusionModelUNet(**cfgs.UNET_FULL), b"unet", 1, (24, 24, 24))
    assert sum(1 for c in bf16_unet if c[3] and c[4] == 1) >= 10        # 7 plain + 3 fused-skip convs at 24^3 on the halo kernel, unsplit
    # round 5: the split-K convs with a fused skip (12^3 / 6^3) share the skip's steps among their splits on the halo kernel as well: only
    # the two stride-2 Downsample convs and the two phase-form Upsample convs are left on the general kernel
    assert sum(1 for c in bf16_unet if not c[3]) == 4, [c for c in bf16_unet if not c[3]]
    # (LDM_HALO_SKIP_SPLIT=0, read once per process, plans the seven of them on the general kernel as round 4 did)
    assert halo_now >= 12
    # the 128 -> 64 and the two plain 64 -> 64 convs of the 96^3 level on conv3_block_kernel (halo code 3); the fused-skip one stays on the
    # 254 x 64 halo tile; the encoder's 96^3 level has four 64 -> 64 convs
    dec = _bf16_conv_launches(AutoencoderKL(**cfgs.VAE_FULL), b"dec", 1, (24, 24, 24))
    assert sum(1 for c in dec if c[3] == 3) == 3 and sum(1 for c in dec if c[3] == 2) == 1, dec
    enc = _bf16_conv_launches(AutoencoderKL(**cfgs.VAE_FULL), b"enc", 1, (96, 96, 96))
    assert sum(1 for c in enc if c[3] == 3) >= 3, enc
    # the 48^3 level's plain convs with 128 output channels and Cin <= 128 on conv3_block128_kernel (halo code 4): 64 -> 128, 128 -> 128 x 2 in the
    # encoder (the fused-skip conv stays on the halo kernel), 128 -> 128 x 2 in the decoder (256 -> 128 stays: the halo tile gains with K)
    assert sum(1 for c in enc if c[3] == 4) == 3 and sum(1 for c in dec if c[3] == 4) == 2, (enc, dec)


def test_shapes_the_networks_cannot_take_fail_loudly(built_lib):
    """The reference crops / pads every volume to a multiple of 2^(levels-1) (utils.py:87-95 `size_divisible`, DivisiblePadd) because
    MONAI's skip concatenation fails on odd sizes; here the planner refuses them with a message instead of a size mismatch deep inside."""
    import ctypes as C
    from ldm3d import _lib
    from ldm3d.networks import AutoencoderKL, DiffusionModelUNet
    L = _lib.lib()
    unet = DiffusionModelUNet(**cfgs.UNET_TINY)
    for dims in ((7, 8, 8), (8, 8, 6), (2, 2, 2), (1, 4, 4)):
        assert L.ldm_model_plan_launches(unet._h, b"unet", 1, *dims) == -2, dims     # LDM_ERR_UNSUPPORTED
        assert b"odd spatial size" in L.ldm_last_error()
    assert L.ldm_model_plan_launches(unet._h, b"unet", 1, 4, 4, 4) > 0               # the smallest legal latent
    assert L.ldm_model_plan_launches(unet._h, b"unet", 0, 8, 8, 8) == -1             # empty batch: LDM_ERR_BAD_ARG
    vae = AutoencoderKL(**cfgs.VAE_TINY)
    for dims in ((7, 8, 8), (2, 2, 2), (16, 16, 18)):
        assert L.ldm_model_plan_launches(vae._h, b"enc", 1, *dims) == -2, dims
    assert L.ldm_model_plan_launches(vae._h, b"enc", 1, 4, 4, 4) > 0


def test_launcher_script_uses_torchrun_and_keeps_peer_to_peer_on():
    """train_LDM.sh = the reference's 3d_ldm/train_LDM.sh:71-76 / train_stable.sh:54-66 for this build: one
    torch.distributed.run line per stage over a 127.0.0.1 rendezvous, and -- unlike the reference (train_LDM.sh:41-42) -- it never sets
    NCCL_P2P_DISABLE / NCCL_IB_DISABLE: xGMI peer-to-peer is the fabric."""
    import subprocess
    path = os.path.join(ROOT, "train_LDM.sh")
    assert os.access(path, os.X_OK)
    assert subprocess.run(["bash", "-n", path]).returncode == 0
    text = open(path).read()
    code = [ln for ln in text.splitlines() if not ln.lstrip().startswith("#")]
    assert any("torch.distributed.run" in ln and "--nproc-per-node" in ln for ln in code)
    assert any("--master-addr 127.0.0.1" in ln for ln in code)
    assert not any("NCCL_P2P_DISABLE=" in ln or "NCCL_IB_DISABLE=" in ln for ln in code)
    assert any("HSA_ENABLE_IPC_MODE_LEGACY" in ln for ln in code)
    for stage in ("train_autoencoder.py", "train_diffusion.py"):
        assert any(stage in ln for ln in code) and os.path.exists(os.path.join(ROOT, stage))


def test_launcher_refuses_shared_stage_flags_before_training_anything():
    """`-s both` with trailing `-- flags` would hand the autoencoder stage's options to the diffusion stage's strict parser AFTER stage 1
    has trained: the launcher refuses up front and names -A / -D; per-stage flags reach only their stage."""
    import subprocess
    path = os.path.join(ROOT, "train_LDM.sh")
    r = subprocess.run(["bash", path, "-n", "1", "-s", "both", "--", "--no-images"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "-A" in r.stderr and "-D" in r.stderr, (r.returncode, r.stderr[-300:])
    text = open(path).read()
    assert "train_autoencoder.py $AE_FLAGS" in text and "train_diffusion.py $DM_FLAGS" in text
    assert ">/dev/null" not in text.replace("2>/dev/null", "")          # a failing build must show its log
