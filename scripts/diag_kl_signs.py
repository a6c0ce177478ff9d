"""Element-level comparison of the EG2 gradients of the big Linear weights (build vs oracle fp32 vs oracle fp64), all at the
oracle's post-step D weights: exact zeros, sign disagreements, and the |g| range in which they occur (Adam's first step is
lr * g / (|g| + 1e-8): sign-like above ~1e-7, proportional below)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import steps as osteps
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer

B = 16
torch.set_num_threads(16)
g = torch.Generator().manual_seed(7)
base = torch.randn(4 * B, 3, 8, 8, generator=g)
data = torch.tanh(torch.nn.functional.interpolate(base, size=64, mode="bilinear"))[:B]
no, e2, e3 = (torch.randn(B, 128, generator=g) for _ in range(3))


def oracle(dtype, d_grads=None):
    got = {}
    eg, d, oeg, od = osteps.build_nets(dtype=dtype)

    def hook(ph, net):
        if ph == "D" and d_grads is not None:
            for k, p in net.named_parameters():
                p.grad = d_grads[k].to(dtype)
        got[ph] = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    out = osteps.betavaegan_step(eg, d, oeg, od, data.to(dtype), no.to(dtype), e2.to(dtype), e3.to(dtype), beta=25.0, grad_hook=hook)
    return out, got


o32, g32 = oracle(torch.float32)
o64, g64 = oracle(torch.float64, d_grads=g32["D"])          # fp64 arithmetic at (nearly) the fp32 oracle's D weights
print(f"oracle fp32 kld {o32['kld']:.1f}; oracle fp64 with the fp32 oracle's D gradients: kld {o64['kld']:.1f}", flush=True)
tr = BetaVAEGANTrainer(beta=25.0)
gb = {}


def hook(ph, net):
    if ph == "D":
        for k, p in net.named_parameters():
            p.grad = g32["D"][k].cuda()
    gb[ph] = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}
out = tr.step(data.cuda(), no.cuda(), e2.cuda(), e3.cuda(), grad_hook=hook)
print(f"build with the oracle's D gradients: kld {float(out['kld']):.1f}", flush=True)


def adam1(gr):
    gr = gr.double()
    return gr / (gr.abs() + 1e-8)


for k in ("x_to_mu.0.weight", "x_to_logvar.0.weight", "x_to_mu.3.weight", "features.6.weight", "features.0.weight", "x_to_mu.1.weight"):
    a, b, c = gb["EG2"][k].double(), g32["EG2"][k].double(), g64["EG2"][k]
    n = a.numel()
    print(f"--- {k}: {n} elements; |g| median {float(c.abs().median()):.2e}, max {float(c.abs().max()):.2e}", flush=True)
    for name, t in (("build", a), ("oracle32", b), ("oracle64", c)):
        print(f"   {name:9s} exact zeros {int((t == 0).sum()):9d}  |g|<1e-9 {int((t.abs() < 1e-9).sum()):9d}  |g|<1e-8 {int((t.abs() < 1e-8).sum()):9d}  "
              f"|g|<1e-7 {int((t.abs() < 1e-7).sum()):9d}  |g|<1e-6 {int((t.abs() < 1e-6).sum()):9d}", flush=True)
    for name, t in (("build", a), ("oracle32", b)):
        d1 = adam1(t) - adam1(c)
        flips = (torch.sign(t) != torch.sign(c))
        print(f"   {name:9s} vs oracle64: sign differs on {int(flips.sum()):8d}; |first Adam update difference|/lr: mean {float(d1.abs().mean()):.3e}, "
              f"sum of squares {float((d1 ** 2).sum()):.1f}; rel L2 of g {float((t - c).norm() / c.norm()):.2e}; "
              f"mean update (build - fp64) {float(d1.mean()):+.3e}", flush=True)
        if flips.any():
            m = c.abs()[flips]
            print(f"             |g64| where the sign differs: median {float(m.median()):.2e}, 90% {float(m.quantile(0.9)) if m.numel() < 1e7 else -1:.2e}, max {float(m.max()):.2e}", flush=True)
