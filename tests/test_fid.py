"""FID arithmetic: the oracle against the reference-generated golden values (CPU), the device
path against both (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import fid as OF

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "fid_kat.npz"))
CASES = ["a", "b", "c", "d"]


def _case(tag):
    d, n, s1, s2, shift, scale = GOLD[f"{tag}_params"]
    a1 = OF.synth_activations(int(s1), int(n), int(d))
    a2 = OF.synth_activations(int(s2), int(n), int(d), float(shift), float(scale))
    return a1, a2, float(GOLD[f"{tag}_fid"])


@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_reference_golden(tag):
    a1, a2, want = _case(tag)
    mu1, s1 = OF.activation_statistics(a1)
    mu2, s2 = OF.activation_statistics(a2)
    assert abs(mu1.sum() - float(GOLD[f"{tag}_mu1_sum"])) <= 1e-9 * abs(mu1.sum())
    assert abs(np.trace(s1) - float(GOLD[f"{tag}_sig1_trace"])) <= 1e-9 * np.trace(s1)
    assert abs(np.linalg.norm(s2) - float(GOLD[f"{tag}_sig2_fro"])) <= 1e-9 * np.linalg.norm(s2)
    got = OF.frechet_distance(mu1, s1, mu2, s2)
    assert abs(got - want) <= 1e-9 * max(abs(want), 1.0), (got, want)


def test_oracle_statistics_elementwise():
    mu, sigma = OF.activation_statistics(GOLD["small_act"])
    assert np.allclose(mu, GOLD["small_mu"], rtol=1e-13, atol=0) and np.allclose(sigma, GOLD["small_sigma"], rtol=1e-12, atol=1e-15)


def test_get_fid_without_inception_fails_loudly(tmp_path):
    from disentangle_mlp_amd import fid
    (tmp_path / "imgs").mkdir()
    np.savez(tmp_path / "s.npz", mu=np.zeros(4), sigma=np.eye(4))
    with pytest.raises(RuntimeError, match="Inception"):
        fid.get_fid(str(tmp_path / "imgs"), str(tmp_path / "s.npz"), device="cpu")
    with pytest.raises(RuntimeError, match="Invalid path"):
        fid.get_fid(str(tmp_path / "nope"), str(tmp_path / "s.npz"), device="cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_device_fid_matches_reference_golden(tag):
    """fp64 on the GPU, eigenvalue formulation: within 1e-7 relative (abs 1e-7 near zero) of the
    imported reference's scipy.linalg.sqrtm result."""
    from disentangle_mlp_amd import fid
    a1, a2, want = _case(tag)
    mu1, s1 = fid.calculate_activation_statistics(a1, batch_size=173)      # ragged streaming batches
    mu2, s2 = fid.calculate_activation_statistics(torch.from_numpy(a2).cuda())
    omu1, os1 = OF.activation_statistics(a1)
    assert np.allclose(mu1.cpu().numpy(), omu1, rtol=1e-12, atol=1e-14)
    assert np.allclose(s1.cpu().numpy(), os1, rtol=1e-10, atol=1e-13)
    got = fid.calculate_frechet_distance(mu1, s1, mu2, s2)
    assert abs(got - want) <= 1e-7 * max(abs(want), 1.0), (got, want)
    # numpy inputs, as the reference's callers pass them
    got2 = fid.calculate_frechet_distance(omu1, os1, *OF.activation_statistics(a2))
    assert abs(got2 - want) <= 1e-7 * max(abs(want), 1.0)


@pytest.mark.gpu
def test_device_fid_npz_round_trip_and_folder(tmp_path):
    from PIL import Image
    from disentangle_mlp_amd import fid
    a1, a2, want = _case("a")
    for name, a in (("gen.npz", a1), ("data.npz", a2)):
        mu, sigma = fid.calculate_activation_statistics(a)
        fid.save_statistics(str(tmp_path / name), mu, sigma)
    with np.load(tmp_path / "gen.npz") as f:
        assert sorted(f.files) == ["mu", "sigma"] and f["sigma"].shape == (64, 64) and f["mu"].dtype == np.float64
    got = fid.get_fid(str(tmp_path / "gen.npz"), str(tmp_path / "data.npz"))
    assert abs(got - want) <= 1e-7 * want
    # image folder + pluggable feature extractor; 7 images, batch 50 -> one batch of 7 (fid.py:84-86)
    d = tmp_path / "imgs"
    d.mkdir()
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, size=(7, 8, 8, 3), dtype=np.uint8)
    for i, im in enumerate(imgs):
        Image.fromarray(im).save(d / f"{i}.png")
    proj = rng.standard_normal((8 * 8 * 3, 4))
    fe = lambda x: x.reshape(x.shape[0], -1) @ proj
    val = fid.get_fid(str(d), str(d), feature_extractor=fe)
    assert abs(val) < 1e-6
    # singular covariances (7 samples, 4 dims is fine; 3 samples is rank 2): still finite
    mu, sigma = fid.calculate_activation_statistics(fe(imgs[:3].astype(np.float32)))
    assert np.isfinite(fid.calculate_frechet_distance(mu, sigma, mu * 1.01, sigma * 0.9))
