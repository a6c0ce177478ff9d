#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the IMPORTED reference.

Run in the authoring container only (``/root/reference`` must exist):

    python tests/golden/make_golden.py

The reference's Python never travels to the GPU box; only the small data files
this script writes do.  What is pinned:

  kat0.json        SURVEY.md section 8(c) KAT-0: weight/input/output checksums and
                   loss scalars of the imported reference models (fp32 + fp64).
  step_b{4,16}.json one full beta-VAE-GAN iteration (new_betavaegan.py:87-193)
                   driven on the imported reference modules: losses, per-parameter
                   gradient norms per phase, post-step parameter / BN-buffer
                   checksums, fp32 and fp64.
  vae_step_b16.json one new_vae.py iteration (BASELINE config 1).
  gan_step_b4.json  one new_gan.py iteration.
  kernel_kats.npz  tiny full-tensor known answers for each hot-path op (conv 5x5
                   s1/s2, conv-transpose with output_padding, train-mode BN +
                   activation, reparam+KL, Dis_l, pixel MSE, BCE incl. the -100
                   clamp) with their backward results, computed with the same
                   torch ops the reference dispatches to.

The reference script itself cannot be imported (argparse + dataset at import
time), so the iteration is driven by ``oracle.steps`` on top of the *reference*
modules via a thin adapter that injects eps (VAE.forward draws it internally).
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

REF = "/root/reference"


def import_reference_model():
    """model.py:3 imports torchvision only for the out-of-scope ResNet classes;
    an empty stub lets the in-scope classes import."""
    for name in ("torchvision", "torchvision.models"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.path.insert(0, os.path.join(REF, "models"))
    import model as ref_model  # noqa
    return ref_model


class RefVAEAdapter:
    """Makes the reference VAE callable as net(data, eps) (eps injected)."""

    def __init__(self, m):
        self.m = m

    def __call__(self, x, eps):
        mu, lv = self.m.encode(x)
        return self.m.decode(mu + eps * torch.exp(0.5 * lv)), mu, lv

    def __getattr__(self, k):
        return getattr(self.m, k)


def f64(t):
    return t.detach().double()


def checksum(t):
    t = f64(t)
    return [float(t.sum()), float(t.abs().sum())]


def build_ref(ref, dtype):
    from oracle.modules import OracleOpt
    opt = OracleOpt()
    torch.manual_seed(999)
    eg = ref.VAE(opt)
    d = ref.Discriminator_celeba(opt)
    eg.apply(ref.weights_init)
    d.apply(ref.weights_init)
    if dtype != torch.float32:
        eg, d = eg.to(dtype), d.to(dtype)
    return eg, d


def kat0(ref):
    from oracle import steps
    out = {}
    for dtype, tag in ((torch.float32, "fp32"), (torch.float64, "fp64")):
        eg, d = build_ref(ref, dtype)
        g = torch.Generator().manual_seed(1234)
        x = (torch.rand(4, 3, 64, 64, generator=g) * 2 - 1).to(dtype)
        eps = torch.randn(4, 128, generator=g).to(dtype)
        noise = torch.randn(4, 128, generator=g).to(dtype)
        r = {}
        for k in ("features.0.weight", "x_to_mu.0.weight", "deconv1.weight"):
            r["w_sum/" + k] = float(f64(eg.state_dict()[k]).sum())
        for k in ("convs.0.weight", "lth_features.0.weight"):
            r["w_sum/D." + k] = float(f64(d.state_dict()[k]).sum())
        r["x_sum"], r["eps_sum"], r["noise_sum"] = (float(f64(v).sum()) for v in (x, eps, noise))
        with torch.no_grad():
            mu, lv = eg.encode(x)
            recon = eg.decode(mu + eps * torch.exp(0.5 * lv))
            fake = eg.decode(noise)
            p_real, f_real = d(x)
            p_rec, f_rec = d(recon)
            p_fake, _ = d(fake)
        r["mu"], r["logvar"] = checksum(mu), checksum(lv)
        r["recon"], r["fake"] = checksum(recon), checksum(fake)
        r["p_real"], r["p_rec"], r["p_fake"] = (f64(v).tolist() for v in (p_real, p_rec, p_fake))
        r["kl_beta1"] = float(steps.kld_loss(mu, lv, 1.0))
        r["mse"] = float(steps.recon_loss(recon, x))
        r["dis_l"] = float(steps.sim_loss(f_rec, f_real))
        r["bce_real_0.9"] = float(steps.bce_loss(p_real, 0.9))
        r["bce_rec_0.9"] = float(steps.bce_loss(p_rec, 0.9))
        r["bce_fake_0.1"] = float(steps.bce_loss(p_fake, 0.1))
        r["recon_slice"] = f64(recon[0, :, ::16, ::16]).flatten().tolist()
        r["mu_slice"] = f64(mu[0, :8]).tolist()
        r["feat_slice"] = f64(f_rec[0, :8]).tolist()
        r["bn_nbt"] = int(eg.state_dict()["features.1.num_batches_tracked"])
        out[tag] = r
    # self-check against the values SURVEY.md recorded
    a = out["fp32"]
    assert abs(a["w_sum/features.0.weight"] - 1.374517) < 1e-4, a["w_sum/features.0.weight"]
    assert abs(a["recon"][0] - (-6764.510742)) < 0.05, a["recon"]
    assert abs(a["kl_beta1"] - 66.10232) < 1e-3
    return out


def step_fixture(ref, batch):
    from oracle import steps
    from torch import optim
    out = {}
    for dtype, tag in ((torch.float32, "fp32"), (torch.float64, "fp64")):
        eg, d = build_ref(ref, dtype)
        opt_eg = optim.Adam(eg.parameters(), lr=1e-3)
        opt_d = optim.Adam(d.parameters(), lr=1e-3)
        b = steps.synthetic_batch(batch, dtype=dtype)
        grads = {}

        def hook(phase, net):
            grads[phase] = {k: float(f64(p.grad).norm()) for k, p in net.named_parameters()
                            if p.grad is not None}
        losses = steps.betavaegan_step(RefVAEAdapter(eg), d, opt_eg, opt_d, b["data"], b["noise"],
                                       b["eps2"], b["eps3"], beta=25.0, grad_hook=hook)
        r = dict(losses=losses, grad_norms=grads)
        r["eg_state"] = {k: checksum(v) for k, v in eg.state_dict().items()}
        r["d_state"] = {k: checksum(v) for k, v in d.state_dict().items()}
        # outputs after the step (train-mode BN, batch statistics)
        with torch.no_grad():
            mu, lv = eg.encode(b["data"])
            recon = eg.decode(mu)
            p, feat = d(recon)
        r["post"] = dict(mu=checksum(mu), recon=checksum(recon), p=f64(p).tolist(),
                         recon_slice=f64(recon[0, :, ::16, ::16]).flatten().tolist())
        out[tag] = r
    return out


def vae_step_fixture(ref, batch=16):
    from oracle import steps
    from oracle.modules import OracleOpt
    from torch import optim
    out = {}
    for dtype, tag in ((torch.float32, "fp32"), (torch.float64, "fp64")):
        torch.manual_seed(999)
        m = ref.VAE(OracleOpt())
        m.apply(ref.weights_init)
        m = m.to(dtype)
        o = optim.Adam(m.parameters(), lr=3e-3)   # envsetter.py:43 default --lr
        b = steps.synthetic_batch(batch, dtype=dtype)
        losses = steps.vae_step(RefVAEAdapter(m), o, b["data"], b["eps2"], beta=1.0)
        out[tag] = dict(losses=losses, state={k: checksum(v) for k, v in m.state_dict().items()})
    return out


def gan_step_fixture(ref, batch=4):
    from oracle import steps
    from oracle.modules import OracleOpt
    from torch import optim
    out = {}
    for dtype, tag in ((torch.float32, "fp32"), (torch.float64, "fp64")):
        # new_gan.py builds the nets BEFORE seeding (:47-57 vs :155-156); the
        # fixture seeds first so it is reproducible.
        torch.manual_seed(999)
        g = ref.Generator_celeba(OracleOpt())
        d = ref.Discriminator_celeba(OracleOpt())
        g.apply(ref.weights_init)
        d.apply(ref.weights_init)
        g, d = g.to(dtype), d.to(dtype)
        og = optim.Adam(g.parameters(), lr=3e-3)
        od = optim.Adam(d.parameters(), lr=3e-3)
        b = steps.synthetic_batch(batch, dtype=dtype)
        losses = steps.gan_step(g, d, og, od, b["data"], b["noise"])
        out[tag] = dict(losses=losses, g_state={k: checksum(v) for k, v in g.state_dict().items()},
                        d_state={k: checksum(v) for k, v in d.state_dict().items()})
    return out


def kernel_kats():
    """Tiny full-tensor known answers, fp64 math on fp32-representable inputs."""
    g = torch.Generator().manual_seed(4242)
    K = {}

    def rnd(*s):
        return torch.randn(*s, generator=g).float().double()

    # conv 5x5 p2, strides 2 and 1 (model.py:450, :389)
    for tag, s, (B, ci, co, H) in (("conv_s2", 2, (2, 3, 4, 8)), ("conv_s1", 1, (2, 3, 5, 6)),
                                   ("conv_s2b", 2, (3, 6, 7, 12))):
        x, w, b = rnd(B, ci, H, H).requires_grad_(), rnd(co, ci, 5, 5).requires_grad_(), rnd(co).requires_grad_()
        y = F.conv2d(x, w, b, stride=s, padding=2)
        gy = rnd(*y.shape)
        y.backward(gy)
        K.update({f"{tag}/x": x, f"{tag}/w": w, f"{tag}/b": b, f"{tag}/y": y, f"{tag}/gy": gy,
                  f"{tag}/gx": x.grad, f"{tag}/gw": w.grad, f"{tag}/gb": b.grad})
    # conv-transpose 5x5 p2 s2 output_padding 1 (model.py:495 + :558) and s1 (model.py:507)
    for tag, s, op, (B, ci, co, H) in (("convT_s2", 2, 1, (2, 4, 3, 4)), ("convT_s1", 1, 0, (2, 5, 3, 6)),
                                       ("convT_s2b", 2, 1, (3, 7, 6, 6))):
        x, w, b = rnd(B, ci, H, H).requires_grad_(), rnd(ci, co, 5, 5).requires_grad_(), rnd(co).requires_grad_()
        y = F.conv_transpose2d(x, w, b, stride=s, padding=2, output_padding=op)
        gy = rnd(*y.shape)
        y.backward(gy)
        K.update({f"{tag}/x": x, f"{tag}/w": w, f"{tag}/b": b, f"{tag}/y": y, f"{tag}/gy": gy,
                  f"{tag}/gx": x.grad, f"{tag}/gw": w.grad, f"{tag}/gb": b.grad})
    # train-mode BN2d + ReLU / LeakyReLU(0.2) (model.py:451-452, :390-391), BN1d + ReLU (:462-463)
    for tag, shape, slope in (("bn2d_relu", (4, 5, 6, 6), 0.0), ("bn2d_lrelu", (4, 5, 6, 6), 0.2),
                              ("bn1d_relu", (6, 10), 0.0)):
        x = rnd(*shape).requires_grad_()
        C = shape[1]
        wt, bs = (1 + 0.1 * rnd(C)).requires_grad_(), (0.1 * rnd(C)).requires_grad_()
        rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
        z = F.batch_norm(x, rm, rv, wt, bs, True, 0.1, 1e-5)
        y = F.leaky_relu(z, slope) if slope else F.relu(z)
        gy = rnd(*shape)
        y.backward(gy)
        K.update({f"{tag}/x": x, f"{tag}/w": wt, f"{tag}/b": bs, f"{tag}/y": y, f"{tag}/gy": gy,
                  f"{tag}/gx": x.grad, f"{tag}/gw": wt.grad, f"{tag}/gb": bs.grad,
                  f"{tag}/rm": rm, f"{tag}/rv": rv})
    # reparam + KL (model.py:532-535, new_betavaegan.py:64-65)
    mu, lv, eps = rnd(5, 8).requires_grad_(), (0.5 * rnd(5, 8)).requires_grad_(), rnd(5, 8)
    z = mu + eps * torch.exp(0.5 * lv)
    kl = 25.0 * (-0.5 * torch.sum(1 + lv - mu.pow(2) - lv.exp()))
    gz = rnd(5, 8)
    (kl + (z * gz).sum()).backward()
    K.update({"rkl/mu": mu, "rkl/lv": lv, "rkl/eps": eps, "rkl/z": z, "rkl/kl": kl, "rkl/gz": gz,
              "rkl/gmu": mu.grad, "rkl/glv": lv.grad})
    # Dis_l (new_betavaegan.py:67-69) and pixel MSE (:71-75)
    a, b = rnd(4, 16).requires_grad_(), rnd(4, 16)
    l = 0.5 * F.mse_loss(a, b, reduction="sum")
    l.backward()
    K.update({"disl/a": a, "disl/b": b, "disl/l": l, "disl/ga": a.grad})
    a, b = rnd(2, 3, 4, 4).requires_grad_(), rnd(2, 3, 4, 4)
    l = F.mse_loss(a, b, reduction="sum")
    l.backward()
    K.update({"mse/a": a, "mse/b": b, "mse/l": l, "mse/ga": a.grad})
    # BCE mean with the log clamp at -100 (new_betavaegan.py:53): p at 0, 1, tiny, interior
    p = torch.tensor([0.0, 1.0, 1e-45, 0.3, 0.7, 1 - 1e-7, 0.5, 0.999], dtype=torch.float32)
    for y in (0.9, 0.1):
        pp = p.clone().requires_grad_()
        l = F.binary_cross_entropy(pp, torch.full_like(pp, y))
        l.backward()
        K.update({f"bce{y}/p": pp, f"bce{y}/l": l, f"bce{y}/gp": pp.grad})
    # tanh / sigmoid epilogues
    t = rnd(3, 7).requires_grad_()
    y = torch.tanh(t)
    gy = rnd(3, 7)
    y.backward(gy)
    K.update({"tanh/x": t, "tanh/y": y, "tanh/gy": gy, "tanh/gx": t.grad})
    return {k: v.detach().numpy() for k, v in K.items()}


def main():
    torch.set_num_threads(8)
    ref = import_reference_model()
    os.makedirs(HERE, exist_ok=True)

    def dump(name, obj):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(obj, f, indent=1, sort_keys=True)
        print("wrote", name)

    dump("kat0.json", kat0(ref))
    dump("step_b4.json", step_fixture(ref, 4))
    dump("step_b16.json", step_fixture(ref, 16))
    dump("vae_step_b16.json", vae_step_fixture(ref, 16))
    dump("gan_step_b4.json", gan_step_fixture(ref, 4))
    np.savez_compressed(os.path.join(HERE, "kernel_kats.npz"), **kernel_kats())
    print("wrote kernel_kats.npz")


if __name__ == "__main__":
    main()
