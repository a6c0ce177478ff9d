"""CPU oracle for the CelebA 64x64 beta-VAE-GAN hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32/fp64)
restatement of the algorithm the reference implements in
``/root/reference/models/model.py`` and ``experiments/new_betavaegan.py``.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / timed CPU
baseline -- never as a compute path of ``disentangle_mlp_amd``.

Pinning: the restatement is validated against the *imported* reference
(``tests/golden/make_golden.py`` run in the authoring container, where
``/root/reference`` exists) and against the committed golden vectors under
``tests/golden/`` (which travel to the GPU box; the reference does not).
The reference itself has no tests or golden vectors.
"""
from .modules import (OracleOpt, weights_init, VAE, Encoder_celeba,
                      Generator_celeba, Discriminator_celeba)
from .steps import (kld_loss, sim_loss, recon_loss, bce_loss,
                    betavaegan_step, vae_step, gan_step, build_nets,
                    synthetic_batch)

__all__ = [
    "OracleOpt", "weights_init", "VAE", "Encoder_celeba", "Generator_celeba",
    "Discriminator_celeba", "kld_loss", "sim_loss", "recon_loss", "bce_loss",
    "betavaegan_step", "vae_step", "gan_step", "build_nets", "synthetic_batch",
]
