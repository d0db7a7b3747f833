"""Paired-volume data path (3d_ldm/utils.py:66-240) - CPU only."""
import argparse
import os

import numpy as np
import pytest
import torch

from ldm3d import data


def test_npz_pair_keys_and_shape_checks(tmp_path):
    vol = np.arange(2 * 4 * 5 * 6, dtype=np.float32).reshape(2, 4, 5, 6)
    for key in ("arr0", "arr_0", "whatever"):
        p = tmp_path / f"{key}.npz"
        np.savez(p, **{key: vol})
        low, high = data.load_pair(str(p))
        assert np.array_equal(low, vol[0]) and np.array_equal(high, vol[1])
    bad = tmp_path / "bad.npz"
    np.savez(bad, arr0=vol[0])
    with pytest.raises(RuntimeError):
        data.load_pair(str(bad))


def test_center_crop_and_percentile_scaling_known_answers():
    assert data.crop_start((10, 11, 12), (4, 4, 4), None) == [3, 3, 4]
    assert data.crop_start((3, 11, 12), (4, 4, 4), None) == [0, 3, 4]          # roi clipped to the volume
    v = np.arange(1001, dtype=np.float32).reshape(1, 7, 11, 13)
    s = data.scale_percentiles(v)                                               # a_min = 0, a_max = p99.5 = 995
    assert s.min() == 0.0 and abs(float(s.reshape(-1)[995]) - 1.0) < 1e-6
    assert float(s.max()) > 1.0                                                 # no clipping (MONAI default clip=False)
    assert np.all(data.scale_percentiles(np.full((2, 2, 2), 3.0, np.float32)) == 0.0)


def test_loader_split_crop_and_pairing(tmp_path):
    files = data.write_synthetic_pairs(str(tmp_path / "d"), 5, (12, 10, 8), seed=1)
    assert len(files) == 5
    args = argparse.Namespace(npz_dir=str(tmp_path / "d"), seed=0, val_fraction=0.2)
    tr, va = data.split_files(args)
    assert len(tr) == 4 and len(va) == 1 and not set(tr) & set(va)
    tl, vl = data.prepare_dataloader(args, 2, (8, 8, 8))
    b = next(iter(tl))
    assert b["image"].shape == (2, 1, 8, 8, 8) and b["label"].shape == (2, 1, 8, 8, 8) and b["image"].dtype == torch.float32
    assert float(b["label"].min()) == 0.0
    # random crops (train_autoencoder.py:133-145 asks for randcrop=True): a FRESH crop on every access, as MONAI's RandSpatialCropd
    # (3d_ldm/utils.py:87); the stream is reproducible from the seed, pairs stay aligned, epochs differ
    big = data.write_synthetic_pairs(str(tmp_path / "big"), 2, (24, 20, 16), seed=2)
    ds, ds2 = data.PairVolumes(big, (8, 8, 8), randcrop=True, seed=3), data.PairVolumes(big, (8, 8, 8), randcrop=True, seed=3)
    e0 = [ds[0]["label"] for _ in range(4)]
    assert any(not torch.equal(e0[0], e) for e in e0[1:])                       # later accesses (= later epochs) see other crops
    assert all(torch.equal(a, ds2[0]["label"]) for a in e0)                     # ... the same ones for the same seed
    ds.set_epoch(5); ds2.set_epoch(5)
    assert torch.equal(ds[1]["image"], ds2[1]["image"])
    ds3 = data.PairVolumes(big, (8, 8, 8), randcrop=True, seed=4)
    assert any(not torch.equal(ds3[0]["label"], e) for e in e0)
    # validation under random-crop training: centre crop of 1.5 x patch rounded up to a multiple of 16 (3d_ldm/utils.py:88), clipped
    _, vl2 = data.prepare_dataloader(argparse.Namespace(npz_dir=str(tmp_path / "big"), seed=0, val_fraction=0.5), 1, (8, 8, 8), randcrop=True)
    assert next(iter(vl2))["image"].shape == (1, 1, 16, 16, 16)
    with pytest.raises(ValueError):
        data.split_files(argparse.Namespace(npz_dir=str(tmp_path / "none")))


def test_trainer_kl_loss_matches_oracle_and_closed_form():
    from ldm3d.trainer import kl_loss
    from oracle import autoencoder as oa
    g = torch.Generator().manual_seed(0)
    mu, sigma = torch.randn((2, 3, 4, 4, 4), generator=g), torch.rand((2, 3, 4, 4, 4), generator=g) + 0.1
    assert torch.allclose(kl_loss(mu, sigma), oa.kl_loss(mu, sigma))
    assert float(kl_loss(torch.zeros(1, 1, 2, 2, 2), torch.ones(1, 1, 2, 2, 2)).abs().max()) < 1e-6      # KL(N(0,1) || N(0,1)) = 0


def test_perceptual_loss_from_a_weights_file(tmp_path):
    """--perceptual-weights (3d_ldm/train_autoencoder.py:236,386,406: PerceptualLoss(squeeze, fake 3-D 0.2)): host-side structure of the
    restated LPIPS-squeeze network from a synthetic weights file: 57 tensors with the shapes of lpips.LPIPS(net='squeeze'), zero for equal
    volumes, symmetric, positive, differentiable, 20 % of the slices of each axis, and loud errors for a wrong file.  The arithmetic
    itself is unpinned (neither MONAI nor lpips can be installed here): ldm3d/perceptual.py states so."""
    import pytest
    import torch
    from ldm3d.perceptual import PerceptualLoss, expected_keys
    g = torch.Generator().manual_seed(0)
    keys = expected_keys()
    assert len(keys) == 57 and keys["net.slice1.0.weight"] == (64, 3, 3, 3) and keys["lin6.model.1.weight"] == (1, 512, 1, 1)
    sd = {k: (torch.rand(s, generator=g) if k.startswith("lin") else 0.1 * torch.randn(s, generator=g)) for k, s in keys.items()}
    path = str(tmp_path / "lpips_squeeze.pt")
    torch.save(sd, path)
    pl = PerceptualLoss.from_file(path)
    x = torch.rand((1, 1, 40, 40, 40), generator=g).requires_grad_(True)
    y = torch.rand((1, 1, 40, 40, 40), generator=g)
    torch.manual_seed(1)
    loss = pl(x, y)
    loss.backward()
    torch.manual_seed(1)
    assert float(pl(y, x.detach())) == pytest.approx(float(loss.detach()), rel=1e-6)
    assert float(loss.detach()) > 0 and float(pl(y, y)) == 0.0 and float(x.grad.abs().sum()) > 0
    picked = []
    pl.net.forward = lambda a, b, orig=pl.net.forward: (picked.append(a.shape[0]), orig(a, b))[1]
    pl(x.detach(), y)
    assert picked == [8, 8, 8]                                # int(40 * 0.2) slices per axis
    bad = dict(sd)
    bad.pop("lin3.model.1.weight")
    with pytest.raises(KeyError):
        PerceptualLoss(bad)
    bad = dict(sd, **{"net.slice2.3.squeeze.weight": torch.zeros(16, 64, 3, 3)})
    with pytest.raises(ValueError):
        PerceptualLoss(bad)
