"""FID on the device: the arithmetic of /root/reference/scoring/fid.py.

What the reference does on the CPU with NumPy / SciPy -- mean and covariance of the Inception
pool_3 activations (fid.py:181-183) and the Frechet distance with a Schur-based
``scipy.linalg.sqrtm`` of a 2048 x 2048 product (:132-160) -- runs here in fp64 on the GPU:

* statistics are accumulated batch by batch (``ActivationStatistics.update``) as shifted sums
  ``sum(x - c)`` and ``(x - c)^T (x - c)`` (rocBLAS dgemm), so 10k x 2048 activations never have
  to exist at once and the generator's samples can stream straight from the decoder;
* ``Tr sqrt(C1 C2)`` is the sum of the square roots of the eigenvalues of the symmetric
  ``C1^{1/2} C2 C1^{1/2}`` (two ``eigh``): always real and finite, where the reference's sqrtm of
  the non-symmetric product needs its "singular product" and "imaginary component" branches.
  For non-singular inputs the two agree to ~1e-9 relative (tests/golden/fid_kat.npz, generated
  with the imported reference).  ``singular_offset=True`` applies the reference's eps*I offset
  (fid.py:146-150) explicitly -- the reference applies it only when sqrtm went non-finite.

File contract kept: ``.npz`` statistics with keys ``mu`` and ``sigma`` (fid.py:287-290),
``get_fid(path_data, path_pretrained, inception="", lowprofile=False)`` (fid.py:320-323).
The Inception pool_3 network is disentangle_mlp_amd/inception.py (the FID Inception-v3 of scoring/inception.py
on the device).  Its pretrained weights (fid.py:268-283 downloads a TensorFlow GraphDef; scoring/inception.py:13
the PyTorch port of the same weights) are NOT part of this repo and cannot be fetched here: pass the
``pt_inception-2015-12-05`` state_dict file as ``inception=`` (the reference's ``inception_path`` argument), or any
``feature_extractor`` callable (images [n,h,w,3] with values 0..255 -> [n,2048]); without either ``get_fid`` accepts
``.npz`` statistics on both sides and raises otherwise.  Absolute FID of images: unpinned (no weights, no reference
statistics offline).
"""
import os
import pathlib

import numpy as np
import torch


def _dev64(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float64)
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=device)


class ActivationStatistics:
    """Streaming mean / covariance (np.mean(axis=0), np.cov(rowvar=False)) in fp64 on the device."""

    def __init__(self, dim=2048, device="cuda"):
        self.device = torch.device(device)
        self.dim, self.n = int(dim), 0
        self.shift = None
        self.s1 = torch.zeros(dim, dtype=torch.float64, device=self.device)
        self.s2 = torch.zeros(dim, dim, dtype=torch.float64, device=self.device)

    def update(self, act):
        act = _dev64(act, self.device).reshape(-1, self.dim)
        if self.shift is None:                      # centre on the first batch: no cancellation later
            self.shift = act.mean(dim=0)
        x = act - self.shift
        self.s1 += x.sum(dim=0)
        self.s2.addmm_(x.t(), x)
        self.n += act.size(0)
        return self

    def finalize(self):
        """(mu [d], sigma [d,d]) as fp64 device tensors."""
        if self.n < 2:
            raise ValueError("covariance needs at least 2 samples")
        m = self.s1 / self.n
        sigma = (self.s2 - self.n * torch.outer(m, m)) / (self.n - 1)
        return m + self.shift, sigma


def calculate_activation_statistics(act, device="cuda", batch_size=4096):
    """fid.py:163-183 for precomputed activations [n, d] (numpy or tensor)."""
    d = act.shape[1]
    st = ActivationStatistics(d, device)
    for s in range(0, act.shape[0], batch_size):
        st.update(act[s:s + batch_size])
    return st.finalize()


def _sym_sqrt(c):
    w, v = torch.linalg.eigh((c + c.t()) * 0.5)
    return (v * w.clamp_min(0).sqrt()) @ v.t()


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6, device="cuda", singular_offset=False):
    """fid.py:109-160.  Accepts numpy arrays or tensors; returns a Python float."""
    device = torch.device(device)
    mu1, mu2 = _dev64(mu1, device).reshape(-1), _dev64(mu2, device).reshape(-1)
    sigma1, sigma2 = _dev64(sigma1, device), _dev64(sigma2, device)
    sigma1 = sigma1.reshape(mu1.numel(), -1)
    sigma2 = sigma2.reshape(mu2.numel(), -1)
    if mu1.shape != mu2.shape:
        raise AssertionError("Training and test mean vectors have different lengths")
    if sigma1.shape != sigma2.shape:
        raise AssertionError("Training and test covariances have different dimensions")
    diff = mu1 - mu2
    a, b = sigma1, sigma2
    if singular_offset:
        off = torch.eye(a.size(0), dtype=torch.float64, device=device) * eps
        a, b = a + off, b + off
    ra = _sym_sqrt(a)
    m = ra @ b @ ra
    lam = torch.linalg.eigvalsh((m + m.t()) * 0.5)
    tr_covmean = lam.clamp_min(0).sqrt().sum()
    val = diff.dot(diff) + torch.trace(sigma1) + torch.trace(sigma2) - 2 * tr_covmean
    return float(val)


def save_statistics(path, mu, sigma):
    """The .npz layout fid.py:287-290 reads."""
    to_np = lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    np.savez(path, mu=to_np(mu), sigma=to_np(sigma))


def load_statistics(path):
    with np.load(path) as f:
        return f["mu"][:], f["sigma"][:]


def _load_images(files):
    from PIL import Image
    return np.stack([np.asarray(Image.open(str(fn)).convert("RGB"), dtype=np.float32) for fn in files])


def _handle_path(path, feature_extractor, device, batch_size=50):
    """fid.py:286-300: .npz statistics or a folder of *.jpg / *.png images."""
    if str(path).endswith(".npz"):
        return load_statistics(path)
    if feature_extractor is None:
        raise RuntimeError(
            "FID of an image folder needs the Inception pool_3 network; its weights "
            "(classify_image_graph_def.pb, fid.py:268-283) are not distributed with this package and cannot be "
            "downloaded here.  Pass feature_extractor=callable(images[n,h,w,3] float 0..255) -> [n,2048], or "
            "precomputed .npz statistics.")
    p = pathlib.Path(path)
    files = list(p.glob("*.jpg")) + list(p.glob("*.png"))
    st = None
    n_batches = len(files) // min(batch_size, max(len(files), 1))   # fid.py:88: the remainder is never propagated
    bs = min(batch_size, len(files))
    for i in range(n_batches):
        act = feature_extractor(_load_images(files[i * bs:(i + 1) * bs]))
        act = torch.as_tensor(np.asarray(act) if not isinstance(act, torch.Tensor) else act)
        act = act.reshape(act.shape[0], -1)
        st = st or ActivationStatistics(act.shape[1], device)
        st.update(act)
    if st is None:
        raise RuntimeError(f"no *.jpg / *.png images under {path}")
    return st.finalize()


def _find_inception_weights(inception_path):
    """The reference's ``inception_path`` is a directory or file of the Inception model (fid.py:268-283); here: the
    ``pt_inception-2015-12-05*.pth`` state_dict, given directly or looked up in a directory."""
    if not inception_path:
        return None
    p = pathlib.Path(inception_path)
    if p.is_file():
        return str(p)
    if p.is_dir():
        hits = sorted(p.glob("pt_inception-2015-12-05*.pth"))
        if hits:
            return str(hits[0])
    raise RuntimeError(f"no Inception state_dict (pt_inception-2015-12-05*.pth) at {inception_path!r}")


def calculate_fid_given_paths(paths, inception_path="", low_profile=False, feature_extractor=None, device="cuda"):
    """fid.py:303-317."""
    for p in paths:
        if not os.path.exists(p):
            raise RuntimeError("Invalid path: %s" % p)
    if feature_extractor is None and not all(str(p).endswith(".npz") for p in paths):
        weights = _find_inception_weights(inception_path)
        if weights is not None:
            from .inception import InceptionFeatureExtractor
            feature_extractor = InceptionFeatureExtractor(weights, device=device)
    m1, s1 = _handle_path(paths[0], feature_extractor, device)
    m2, s2 = _handle_path(paths[1], feature_extractor, device)
    return calculate_frechet_distance(m1, s1, m2, s2, device=device)


def get_fid(path_data, path_pretrained, inception="", lowprofile=False, feature_extractor=None, device="cuda"):
    """fid.py:320-323."""
    return calculate_fid_given_paths([path_data, path_pretrained], inception, lowprofile, feature_extractor, device)
