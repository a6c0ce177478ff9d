import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, so a plain
    `pytest tests/` on a CPU box stays green; `-m gpu` on the GPU box runs them."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kats():
    return dict(np.load(os.path.join(GOLDEN, "kernel_kats.npz")))


# Biases added right before a train-mode BatchNorm have an analytically zero
# gradient; what the reference accumulates there is rounding noise that Adam
# amplifies to lr-sized steps.  Parity checks
# exclude them (they are functionally inert: BN cancels them, eval() is never used).
BN_SHADOWED = {
    "eg": ["features.0.bias", "features.3.bias", "features.6.bias", "x_to_mu.0.bias",
           "x_to_logvar.0.bias", "preprocess.0.bias", "deconv1.bias", "deconv2.bias",
           "deconv3.bias", "x_to_mu.3.bias"],
    "d": ["convs.0.bias", "convs.3.bias", "convs.6.bias", "convs.9.bias"],
    "g": ["preprocess.0.bias", "deconv1.bias", "deconv2.bias", "deconv3.bias"],
}


# ---- stated tolerances for one full beta-VAE-GAN iteration in fp32 -------------------------
# Phase 1 (discriminator) is tight.  Everything after the first Adam step (a sign-like update:
# lr*g/(|g|+eps)) is chaotic in ANY fp32 evaluation order -- the reference's own CPU path moves
# kld by 0.5 % (55235 / 55388 / 55501 at B=16) when only torch's thread count changes 1/3/8 and
# by 0.85 % between two hosts at B=4 -- so those quantities get the loose bounds below.
LOSS_TOL = dict(D_x=2e-5, errD_real=2e-5, errD_fake=2e-5, errG_fake=1e-3, errG_recon=1e-3, sim=1e-3,
                mse_dec=1e-4, mse_enc=2e-3, kld=3e-2, errG=1e-3, mse=1e-4, loss=1e-4)
GRADNORM_TOL = {"D": 5e-3, "EG2": 1e-2, "EG3": 0.5}


def gap(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-30)


def check_state(state, gold32, gold64, skip, lr, rel=1e-4, abs_=1e-4):
    """Checksums (sum, abs-sum) vs the fp32 golden values.  Per tensor the tolerance is the
    larger of `rel` and 5x the reference's own fp32-vs-fp64 gap, plus a sign-flip budget:
    Adam's first update is lr*sign(g), so an element whose gradient is within rounding noise
    of zero moves by 2*lr; up to max(4, 0.2 % of the elements) such flips are tolerated."""
    bad = []
    for k, v in state.items():
        if k in skip:
            continue
        s, a = float(v.double().sum()), float(v.double().abs().sum())
        # + lr*sqrt(n): tensors that took a second (no longer sign-like, chaotic) Adam step
        flips = 0.0 if ("running" in k or "num_batches" in k) else \
            2 * lr * max(4, 2e-3 * v.numel()) + lr * v.numel() ** 0.5
        # BatchNorm running statistics absorb the (chaotic) later forwards with momentum 0.1
        base = 2e-2 if "running" in k else rel
        tol = max(base, 5 * gap(gold32[k][1], gold64[k][1]))
        tol_s = max(base, 5 * abs(gold32[k][0] - gold64[k][0]) / max(gold32[k][1], 1e-30))
        if abs(a - gold32[k][1]) > abs_ + flips + tol * a or abs(s - gold32[k][0]) > abs_ + flips + tol_s * a:
            bad.append((k, s, a, gold32[k], gold64[k]))
    assert not bad, bad[:5]
