#!/usr/bin/env python3
"""Golden vectors for the FID arithmetic, generated with the IMPORTED reference
(/root/reference/scoring/fid.py: calculate_frechet_distance :109-160; the statistics recipe of
calculate_activation_statistics :163-183 = np.mean / np.cov).  fid.py imports tensorflow and
imageio at module level (:23-24, used only by the Inception path); empty stubs let the NumPy /
SciPy part import.  Run in the authoring container only:

    python tests/golden/make_golden_fid.py   ->  tests/golden/fid_kat.npz

Activations are synthetic (ReLU-like, correlated) -- the Inception weights are not in the
repo and there is no network, so absolute FID values of images stay unpinned.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference_fid():
    for name in ("tensorflow", "imageio"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["imageio"].imread = None
    sys.path.insert(0, os.path.join(REF, "scoring"))
    import fid as ref_fid  # noqa
    return ref_fid


def synth_activations(seed, n, d, shift=0.0, scale=1.0):
    rng = np.random.default_rng(seed)
    mix = rng.standard_normal((d, d)) / np.sqrt(d)
    z = rng.standard_normal((n, d)) @ mix * scale + 0.3 + shift
    return np.maximum(z, 0.0)        # pool_3 activations are non-negative


def main():
    ref = import_reference_fid()
    out = {}
    cases = [("a", 64, 500, 1, 2, 0.0, 1.0), ("b", 64, 500, 3, 3, 0.0, 1.0), ("c", 256, 1500, 4, 5, 0.15, 1.3),
             ("d", 512, 2000, 6, 7, -0.05, 0.8)]
    for tag, d, n, s1, s2, shift, scale in cases:
        a1 = synth_activations(s1, n, d)
        a2 = synth_activations(s2, n, d, shift, scale)
        mu1, sig1 = np.mean(a1, axis=0), np.cov(a1, rowvar=False)
        mu2, sig2 = np.mean(a2, axis=0), np.cov(a2, rowvar=False)
        val = ref.calculate_frechet_distance(mu1, sig1, mu2, sig2)
        out[f"{tag}_params"] = np.asarray([d, n, s1, s2, shift, scale], dtype=np.float64)
        out[f"{tag}_fid"] = np.float64(val)
        out[f"{tag}_mu1_sum"], out[f"{tag}_sig1_trace"] = np.float64(mu1.sum()), np.float64(np.trace(sig1))
        out[f"{tag}_sig2_fro"] = np.float64(np.linalg.norm(sig2))
        print(tag, d, n, float(val))
    # full small tensors for one case so the statistics path has an element-wise pin
    a = synth_activations(11, 40, 8)
    out["small_act"], out["small_mu"], out["small_sigma"] = a, np.mean(a, axis=0), np.cov(a, rowvar=False)
    np.savez_compressed(os.path.join(HERE, "fid_kat.npz"), **out)


if __name__ == "__main__":
    main()
