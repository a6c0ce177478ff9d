"""Why is the build's beta*KL after the first two Adam steps systematically below the reference's (B = 16: 64 138 / 65 544
against 65 956 .. 66 040 for the reference's own three evaluations)?  Bisect: run the build's iteration but replace, right
before an optimizer step, the gradients of chosen parameter groups by the oracle's (fp32, 16 threads), and read the KL of
phase 3.  If replacing a group moves the build's KL onto the reference's, that group's gradient is where the two differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import steps as osteps
from disentangle_mlp_amd import ops
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer

B = int(os.environ.get("B", "16"))
torch.set_num_threads(16)
g = torch.Generator().manual_seed(7)
base = torch.randn(4 * B, 3, 8, 8, generator=g)
data = torch.tanh(torch.nn.functional.interpolate(base, size=64, mode="bilinear"))[:B]
no, e2, e3 = (torch.randn(B, 128, generator=g) for _ in range(3))

ref_g = {}
eg, d, oeg, od = osteps.build_nets()
ref = osteps.betavaegan_step(eg, d, oeg, od, data, no, e2, e3, beta=25.0,
                             grad_hook=lambda ph, net: ref_g.__setitem__(ph, {k: p.grad.detach().clone() for k, p in net.named_parameters()}))
print(f"oracle fp32: kld {ref['kld']:.1f} mse_enc {ref['mse_enc']:.1f} errG_recon {ref['errG_recon']:.5f}", flush=True)

SHADOWED = ["features.0.bias", "features.3.bias", "features.6.bias", "x_to_mu.0.bias", "x_to_logvar.0.bias",
            "preprocess.0.bias", "deconv1.bias", "deconv2.bias", "deconv3.bias", "x_to_mu.3.bias"]


def run(name, pick_eg=None, pick_d=None, zero=None, arith=None):
    """pick_eg(k) / pick_d(k) -> True: take the oracle's gradient for parameter k at the EG2 / D step."""
    if arith:
        ops.CONV_ARITH = arith
    tr = BetaVAEGANTrainer(beta=25.0)
    stats = {}

    def hook(ph, net):
        pick = {"D": pick_d, "EG2": pick_eg}.get(ph)
        for k, p in net.named_parameters():
            if ph == "EG2" and p.grad is not None:
                r = ref_g["EG2"][k]
                e = float((p.grad.cpu().double() - r.double()).norm() / max(float(r.double().norm()), 1e-30))
                stats[k] = (e, float(r.abs().max()), float(p.grad.abs().max()))
            if pick is not None and pick(k):
                p.grad = ref_g[ph][k].cuda()
            if zero is not None and ph == "EG2" and k in zero:
                p.grad = torch.zeros_like(p)
    out = tr.step(data.cuda(), no.cuda(), e2.cuda(), e3.cuda(), grad_hook=hook)
    print(f"{name:58s} kld {float(out['kld']):10.1f} ({float(out['kld']) / ref['kld'] - 1:+.4f})  mse_enc {float(out['mse_enc']):9.1f}", flush=True)
    return stats


# the reference's own fp32-vs-fp64 gradient error on this batch (the yardstick)
ref64_g = {}
eg64, d64, oeg64, od64 = osteps.build_nets(dtype=torch.float64)
ref64 = osteps.betavaegan_step(eg64, d64, oeg64, od64, data.double(), no.double(), e2.double(), e3.double(), beta=25.0,
                               grad_hook=lambda ph, net: ref64_g.__setitem__(ph, {k: p.grad.detach().clone() for k, p in net.named_parameters()}))
print(f"oracle fp64: kld {ref64['kld']:.1f}", flush=True)
e3264 = sorted(((float((ref_g['EG2'][k].double() - r).norm() / max(float(r.norm()), 1e-30)), k) for k, r in ref64_g['EG2'].items()
                if k not in SHADOWED), reverse=True)
print("  oracle fp32 vs fp64, EG2 gradients (own D steps: chaos included):", [(f"{e:.1e}", k) for e, k in e3264[:8]], flush=True)
st = run("D gradients from the oracle (bf16x6)", None, lambda k: True, arith="bf16x6")
allg = sorted(((v[0], k) for k, v in st.items() if k not in SHADOWED), reverse=True)
print("  EG2 gradient rel-L2 vs oracle fp32 at the oracle's D weights:", [(f"{e:.1e}", k) for e, k in allg], flush=True)
if os.environ.get("ONLY_D"):
    sys.exit(0)
for arith in ("bf16x6", "fp32"):
    print("==== build arithmetic", arith, flush=True)
    st = run("build as is", arith=arith)
    worst = sorted(((v[0], k, v[1], v[2]) for k, v in st.items()), reverse=True)[:12]
    print("  EG2 gradient rel-L2 vs oracle fp32, worst:", [(f"{e:.1e}", k, f"{a:.1e}", f"{b:.1e}") for e, k, a, b in worst], flush=True)
    run("all EG2 + D gradients from the oracle", lambda k: True, lambda k: True)
    run("all EG2 gradients from the oracle", lambda k: True)
    run("D gradients from the oracle", None, lambda k: True)
    run("EG2: shadowed-bias gradients from the oracle", lambda k: k in SHADOWED)
    run("EG2: all but the shadowed biases from the oracle", lambda k: k not in SHADOWED)
    run("EG2: encoder conv/BN (features.*) from the oracle", lambda k: k.startswith("features"))
    run("EG2: x_to_mu.* + x_to_logvar.* from the oracle", lambda k: k.startswith("x_to_"))
    run("EG2: the two 16384x2048 weights from the oracle", lambda k: k in ("x_to_mu.0.weight", "x_to_logvar.0.weight"))
    run("EG2: decoder from the oracle", lambda k: k.startswith(("preprocess", "deconv", "act")))
# the other direction: the oracle stepping with the shadowed-bias gradients zeroed (what the build defines them as)
eg, d, oeg, od = osteps.build_nets()


def zero_shadowed(ph, net):
    if ph == "EG2":
        for k, p in net.named_parameters():
            if k in SHADOWED:
                p.grad.zero_()
r2 = osteps.betavaegan_step(eg, d, oeg, od, data, no, e2, e3, beta=25.0, grad_hook=zero_shadowed)
print(f"oracle fp32, shadowed-bias gradients zeroed at EG2: kld {r2['kld']:.1f} ({r2['kld'] / ref['kld'] - 1:+.4f})", flush=True)
