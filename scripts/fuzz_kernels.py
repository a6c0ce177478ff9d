"""Random-shape sweep (not part of the test suite): N cases per op against the fp64 oracle."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops as H
from oracle import ops as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
def rel(a, r):
    r = r.double(); return float((a.detach().cpu().double() - r).norm() / max(float(r.norm()), 1e-30))
worst = 0.0
for case in range(N):
    s = rng.choice([1, 2])
    B = rng.choice([1, 2, 3, 4, 7, 16, 33])
    Cin = rng.choice([1, 2, 3, 4, 6, 16, 31, 64, 70])
    Cout = rng.choice([1, 2, 3, 4, 5, 31, 32, 64, 96, 128, 130, 200])
    Hs = rng.choice([2, 4, 6, 8, 12, 16, 24, 32, 40])
    Ws = rng.choice([2, 4, 8, 16, 20, 32, 36, 64, 68, 80])
    g = torch.Generator().manual_seed(case)
    x = torch.randn(B, Cin, Hs, Ws, generator=g); w = 0.2 * torch.randn(Cout, Cin, 5, 5, generator=g)
    b = torch.randn(Cout, generator=g)
    y_ref = O.conv5x5(x, w, b, s)
    gy = torch.randn(*y_ref.shape, generator=g)
    gx_ref, gw_ref = O.conv5x5_grads(x, w, gy, s)
    wt = 0.2 * torch.randn(Cin, Cout, 5, 5, generator=g)
    errs = dict(fwd=rel(H.conv5x5_fwd(x.cuda(), w.cuda(), b.cuda(), s), y_ref),
                dgrad=rel(H.convT5x5_fwd(gy.cuda(), w.cuda(), None, s), gx_ref),
                wgrad=rel(H.conv5x5_wgrad(x.cuda(), gy.cuda(), s), gw_ref),
                convT=rel(H.convT5x5_fwd(x.cuda(), wt.cuda(), b.cuda(), s), O.convT5x5(x, wt, b, s)))
    m = max(errs.values()); worst = max(worst, m)
    if m > 3e-6:
        print("FAIL", case, (B, Cin, Cout, Hs, Ws, s), errs, flush=True)
# BatchNorm shapes
for case in range(N // 3):
    B = rng.choice([2, 3, 8, 17, 64]); C = rng.choice([1, 3, 32, 100, 256]); hw = rng.choice([(1, 1), (3, 5), (8, 8), (16, 16), (7, 12)])
    act = rng.choice(["none", "relu", "lrelu"])
    g = torch.Generator().manual_seed(1000 + case)
    shape = (B, C) if hw == (1, 1) and rng.random() < 0.5 else (B, C, hw[0], hw[1])
    x = 2 * torch.randn(*shape, generator=g) + 0.3
    gamma, beta, gy = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g), torch.randn(*shape, generator=g)
    ref = O.bn_act(x, gamma, beta, act, gy=gy)
    code = {"none": 0, "relu": 1, "lrelu": 2}[act]
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    y, mean, invstd = H.bn_act_fwd(x.cuda(), gamma.cuda(), beta.cuda(), rm, rv, 1e-5, 0.1, code)
    gx, gw, gb = H.bn_act_bwd(gy.cuda(), x.cuda(), gamma.cuda(), beta.cuda(), mean, invstd, code)
    n_el = x.numel() // C
    errs = dict(y=rel(y, ref["y"]), rm=rel(rm, ref["rm"]), rv=rel(rv, ref["rv"]))
    if n_el >= 16:   # tiny per-channel counts make the input gradient ill-conditioned
        errs.update(gx=rel(gx, ref["gx"]), gw=rel(gw, ref["gw"]), gb=rel(gb, ref["gb"]))
    m = max(errs.values())
    if m > 5e-5:
        print("FAIL BN", case, shape, act, errs, flush=True)
print("done; worst conv rel-L2", worst)
