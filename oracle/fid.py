"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the FID arithmetic of
/root/reference/scoring/fid.py: calculate_frechet_distance (:109-160) and the statistics of
calculate_activation_statistics (:163-183).  Pinned by tests/golden/fid_kat.npz, which
tests/golden/make_golden_fid.py generated with the imported reference functions.
The Inception pool_3 network itself (fid.py:34-105) needs weights that are not in the repo:
activations are inputs here.
"""
import numpy as np
from scipy import linalg


def activation_statistics(act):
    """fid.py:181-183."""
    act = np.asarray(act, dtype=np.float64)
    return np.mean(act, axis=0), np.cov(act, rowvar=False)


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """d^2 = |mu1-mu2|^2 + Tr(C1 + C2 - 2 sqrt(C1 C2)); fid.py:132-160."""
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


def synth_activations(seed, n, d, shift=0.0, scale=1.0):
    """The synthetic activations of tests/golden/make_golden_fid.py (same generator)."""
    rng = np.random.default_rng(seed)
    mix = rng.standard_normal((d, d)) / np.sqrt(d)
    z = rng.standard_normal((n, d)) @ mix * scale + 0.3 + shift
    return np.maximum(z, 0.0)
