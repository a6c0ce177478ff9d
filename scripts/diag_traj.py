"""kld / mse over the first iterations: HIP engine in each arithmetic (and fused / unfused BatchNorm) vs the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import steps as osteps
from disentangle_mlp_amd import ops, model as M
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer
N, B = 4, 16
torch.set_num_threads(16)
def inputs():
    g = torch.Generator().manual_seed(7)
    base = torch.randn(4 * B, 3, 8, 8, generator=g)
    data = torch.tanh(torch.nn.functional.interpolate(base, size=64, mode="bilinear"))
    rnd = [[torch.randn(B, 128, generator=g) for _ in range(3)] for _ in range(N)]
    return data, rnd
data, rnd = inputs()
rows = {}
for th in (16, 1):
    torch.set_num_threads(th)
    eg, d, oeg, od = osteps.build_nets()
    rows[f"oracle fp32 {th} thr"] = [osteps.betavaegan_step(eg, d, oeg, od, data[(it % 4) * B:(it % 4 + 1) * B], *rnd[it], beta=25.0) for it in range(N)]
torch.set_num_threads(16)
eg, d, oeg, od = osteps.build_nets(dtype=torch.float64)
rows["oracle fp64"] = [osteps.betavaegan_step(eg, d, oeg, od, data[(it % 4) * B:(it % 4 + 1) * B].double(), *[t.double() for t in rnd[it]], beta=25.0) for it in range(N)]
for arith in ("fp32", "bf16x6"):
    for fused in (False, True):
        ops.CONV_ARITH, M.FUSE_CONV_BN = arith, fused
        tr = BetaVAEGANTrainer(beta=25.0)
        outs = []
        for it in range(N):
            o = tr.step(data[(it % 4) * B:(it % 4 + 1) * B].cuda(), *[t.cuda() for t in rnd[it]])
            outs.append({k: float(v) for k, v in o.items()})
        rows[f"hip {arith} fused={fused}"] = outs
for name, outs in rows.items():
    print(f"{name:28s} kld " + " ".join(f"{o['kld']:11.1f}" for o in outs) + " | mse_enc " + " ".join(f"{o['mse_enc']:9.1f}" for o in outs), flush=True)
