"""Qualitative trajectory check: the HIP engine and the CPU oracle on the same synthetic data for
N iterations (identical noise / eps every iteration).  Past the first Adam step the two are
chaotic twins, so compare trends (D(x), reconstruction error), not digits."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import steps as osteps
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer
N, B = int(sys.argv[1]) if len(sys.argv) > 1 else 24, 16
torch.set_num_threads(16)
g = torch.Generator().manual_seed(7)
base = torch.randn(4 * B, 3, 8, 8, generator=g)
data = torch.tanh(torch.nn.functional.interpolate(base, size=64, mode="bilinear"))
rnd = [[torch.randn(B, 128, generator=g) for _ in range(3)] for _ in range(N)]
tr = BetaVAEGANTrainer(beta=25.0)
eg, d, oeg, od = osteps.build_nets()
for it in range(N):
    x = data[(it % 4) * B:(it % 4 + 1) * B]
    no, e2, e3 = rnd[it]
    out = tr.step(x.cuda(), no.cuda(), e2.cuda(), e3.cuda())
    ref = osteps.betavaegan_step(eg, d, oeg, od, x, no, e2, e3, beta=25.0)
    if it % 3 == 0 or it == N - 1:
        print(f"it {it:3d}  D(x) hip {float(out['D_x_sum'])/B:.4f} ref {ref['D_x']:.4f} | errD_fake hip {float(out['errD_fake']):8.4f} "
              f"ref {ref['errD_fake']:8.4f} | mse_enc hip {float(out['mse_enc']):10.1f} ref {ref['mse_enc']:10.1f} | "
              f"kld hip {float(out['kld']):10.1f} ref {ref['kld']:10.1f}", flush=True)
