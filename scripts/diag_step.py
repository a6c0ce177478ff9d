"""Diagnostic: HIP step vs live oracle, per-phase gradient / weight comparison."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import steps as osteps
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
LR0 = len(sys.argv) > 2 and sys.argv[2] == "lr0"
torch.set_num_threads(16)
eg, d, oeg, od = osteps.build_nets(dtype=torch.float64 if LR0 else torch.float32)
if LR0:
    for o in (oeg, od):
        o.param_groups[0]["lr"] = 0.0
b = osteps.synthetic_batch(batch, dtype=torch.float64 if LR0 else torch.float32)
ref_g, ref_w = {}, {}
def rh(ph, net):
    ref_g[ph] = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
ref_l = osteps.betavaegan_step(eg, d, oeg, od, b["data"], b["noise"], b["eps2"], b["eps3"], beta=25.0, grad_hook=rh)
tr = BetaVAEGANTrainer(beta=25.0, lr=0.0 if LR0 else 1e-3)
got_g = {}
def gh(ph, net):
    got_g[ph] = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}
out = tr.step(*(b[k].float().cuda() for k in ("data", "noise", "eps2", "eps3")), grad_hook=gh)
for k in ref_l:
    if k in out:
        print(f"{k:12s} ref {ref_l[k]:.6f} got {float(out[k]):.6f} rel {abs(ref_l[k]-float(out[k]))/abs(ref_l[k]):.2e}")
for ph in ("D", "EG2", "EG3"):
    errs = []
    for k, r in ref_g[ph].items():
        g = got_g[ph][k]
        e = float((g.double() - r.double()).norm() / max(float(r.double().norm()), 1e-30))
        errs.append((e, k, float(r.norm())))
    errs.sort(reverse=True)
    print(ph, "all:", [(f"{e:.1e}", k) for e, k, n in errs if "bias" not in k])
for name, a, r in (("EG", tr.netEG, eg), ("D", tr.netD, d)):
    errs = []
    for (k, v), (_, w) in zip(a.state_dict().items(), r.state_dict().items()):
        if "num_batches" in k: continue
        diff = (v.cpu().double() - w.double()).abs()
        errs.append((float(diff.mean()), float(diff.max()), k))
    errs.sort(reverse=True)
    print(name, "post-step mean/max abs diff worst:", [(f"{m:.2e}", f"{x:.2e}", k) for m, x, k in errs[:8]])
