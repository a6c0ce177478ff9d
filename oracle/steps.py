"""Oracle (CPU) restatement of the reference's training iterations.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

  * betavaegan_step : experiments/new_betavaegan.py:87-193 (one loop body)
  * vae_step        : experiments/new_vae.py:39-48, 53-59
  * gan_step        : experiments/new_gan.py:66-141 (one loop body)
  * losses          : new_betavaegan.py:53 (BCE), :64-65 (KLD), :67-69 (SIM),
                      :71-75 (reconstruction_loss)

Every source of randomness of the reference loop (the soft labels drawn with
``np.random.choice`` at :89-90, ``noise`` at :111, the two ``randn_like`` eps
draws inside ``VAE.forward``) is an explicit argument here, so the oracle and
the HIP path can be driven with identical numbers.  The step executes the
reference's schedule literally (separate ``backward`` calls, un-detached
``sim_real``, no-op ``requires_grad`` flags omitted because they are no-ops,
SURVEY.md section 3.1 item 1).
"""
from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import nn, optim

from .modules import OracleOpt, VAE, Discriminator_celeba, Generator_celeba, weights_init


def kld_loss(mu, logvar, beta):
    return beta * (-0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp()))


def sim_loss(sim_recon, sim_real):
    return 0.5 * F.mse_loss(sim_recon, sim_real, reduction="sum")


def recon_loss(recon_x, x):
    return F.mse_loss(recon_x, x, reduction="sum")


def bce_loss(p, label_value):
    """nn.BCELoss() (mean) against a constant label, as new_betavaegan.py:97,101."""
    label = torch.full((p.size(0),), label_value, dtype=p.dtype, device=p.device)
    return F.binary_cross_entropy(p, label)


def build_nets(seed=999, dtype=torch.float32, opt: Optional[OracleOpt] = None):
    """The reference's construction recipe (new_betavaegan.py:36,41-50):
    seed -> VAE -> Discriminator -> apply(weights_init) x2 -> two Adams, lr 1e-3."""
    opt = opt or OracleOpt()
    torch.manual_seed(seed)
    net_eg = VAE(opt)
    net_d = Discriminator_celeba(opt)
    net_eg.apply(weights_init)
    net_d.apply(weights_init)
    if dtype != torch.float32:
        net_eg = net_eg.to(dtype)
        net_d = net_d.to(dtype)
    opt_eg = optim.Adam(net_eg.parameters(), lr=1e-3)
    opt_d = optim.Adam(net_d.parameters(), lr=1e-3)
    return net_eg, net_d, opt_eg, opt_d


def synthetic_batch(batch, seed=1234, dtype=torch.float32, n_hidden=128):
    """SURVEY.md section 8(d) synthetic inputs: data U(-1,1) (B,3,64,64);
    eps/noise N(0,1) (B,n_hidden), all from one CPU generator in a fixed order."""
    g = torch.Generator().manual_seed(seed)
    data = torch.rand(batch, 3, 64, 64, generator=g) * 2 - 1
    eps2 = torch.randn(batch, n_hidden, generator=g)
    noise = torch.randn(batch, n_hidden, generator=g)
    eps3 = torch.randn(batch, n_hidden, generator=g)
    return {k: v.to(dtype) for k, v in
            dict(data=data, eps2=eps2, noise=noise, eps3=eps3).items()}


def betavaegan_step(net_eg: VAE, net_d: Discriminator_celeba, opt_eg, opt_d,
                    data, noise, eps2, eps3, beta=25.0,
                    real_label=0.9, fake_label=0.1,
                    bce_divisor: Optional[int] = None,
                    grad_hook=None) -> Dict[str, float]:
    """One iteration of new_betavaegan.py:87-193.

    ``bce_divisor`` (default: the local batch) lets an N-replica emulation divide
    the BCE sums by the *global* batch so that summed replica gradients equal
    the DataParallel gradient.  ``grad_hook(phase, net)``
    is called right before each optimizer step (used by tests to snapshot or
    all-reduce gradients).
    """
    net_d.train()
    net_eg.train()
    bs = data.size(0)
    scale = 1.0 if bce_divisor is None else bs / float(bce_divisor)
    out: Dict[str, float] = {}

    # ---- phase 1: discriminator (:95-123)
    net_d.zero_grad()
    p_real, sim_real = net_d(data)
    err_d_real = bce_loss(p_real, real_label) * scale
    err_d_real.backward()
    out["D_x"] = p_real.mean().item()
    fake = net_eg.decode(noise)
    p_fake, _ = net_d(fake.detach())
    err_d_fake = bce_loss(p_fake, fake_label) * scale
    err_d_fake.backward()
    if grad_hook:
        grad_hook("D", net_d)
    opt_d.step()
    out["errD_real"] = err_d_real.item()
    out["errD_fake"] = err_d_fake.item()

    # ---- phase 2: "decoder" -- in fact all of EG moves (:127-164)
    net_eg.zero_grad()
    p_real2, sim_real = net_d(data)
    recon, mu, logvar = net_eg(data, eps2)
    p_fake2, _ = net_d(fake)
    p_rec, sim_rec = net_d(recon)
    err_g_fake = bce_loss(p_fake2, real_label) * scale
    err_g_rec = bce_loss(p_rec, real_label) * scale
    err_g_fake.backward(retain_graph=True)
    err_g_rec.backward(retain_graph=True)
    sim = sim_loss(sim_rec, sim_real)
    sim.backward(retain_graph=True)
    mse2 = recon_loss(recon, data)
    mse2.backward()
    if grad_hook:
        grad_hook("EG2", net_eg)
    opt_eg.step()
    out.update(errG_fake=err_g_fake.item(), errG_recon=err_g_rec.item(),
               sim=sim.item(), mse_dec=mse2.item())

    # ---- phase 3: "encoder" -- again all of EG moves (:167-193)
    net_eg.zero_grad()
    recon, mu, logvar = net_eg(data, eps3)
    kld = kld_loss(mu, logvar, beta)
    kld.backward(retain_graph=True)
    mse3 = recon_loss(recon, data)
    mse3.backward()
    if grad_hook:
        grad_hook("EG3", net_eg)
    opt_eg.step()
    out.update(kld=kld.item(), mse_enc=mse3.item())
    return out


def vae_step(model: VAE, optimizer, data, eps, beta=1.0) -> Dict[str, float]:
    """new_vae.py:53-59 with loss_function of :39-48 (MSE_sum + KLD, beta=1)."""
    model.train()
    optimizer.zero_grad()
    recon, mu, logvar = model(data, eps)
    mse = recon_loss(recon, data)
    kld = kld_loss(mu, logvar, beta)
    loss = mse + kld
    loss.backward()
    optimizer.step()
    return dict(loss=loss.item(), mse=mse.item(), kld=kld.item())


def gan_step(net_g: Generator_celeba, net_d: Discriminator_celeba, opt_g, opt_d,
             data, noise, real_label=0.9, fake_label=0.1,
             bce_divisor: Optional[int] = None, grad_hook=None) -> Dict[str, float]:
    """new_gan.py:66-141: D(real)+D(fake.detach()) -> step D; D(fake) -> step G.
    ``bce_divisor`` / ``grad_hook``: as in `betavaegan_step` (N-replica emulation of the
    reference's nn.DataParallel run, new_gan.py:51-53)."""
    net_g.train()
    net_d.train()
    scale = 1.0 if bce_divisor is None else data.size(0) / float(bce_divisor)
    net_d.zero_grad()
    p_real, _ = net_d(data)
    err_real = bce_loss(p_real, real_label) * scale
    err_real.backward()
    fake = net_g(noise)
    p_fake, _ = net_d(fake.detach())
    err_fake = bce_loss(p_fake, fake_label) * scale
    err_fake.backward()
    if grad_hook:
        grad_hook("D", net_d)
    opt_d.step()
    net_g.zero_grad()
    p_fake2, _ = net_d(fake)
    err_g = bce_loss(p_fake2, real_label) * scale
    err_g.backward()
    if grad_hook:
        grad_hook("G", net_g)
    opt_g.step()
    return dict(errD_real=err_real.item(), errD_fake=err_fake.item(), errG=err_g.item(),
                D_x=p_real.mean().item())
