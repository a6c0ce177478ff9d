"""Same-process A/B of the Conv <-> BatchNorm fusion (model.FUSE_CONV_BN) on the full iteration, interleaved rounds."""
import sys, os, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import model as M
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer
B = 128
tr = BetaVAEGANTrainer(beta=25.0)
g = torch.Generator().manual_seed(1)
data = (torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).cuda()
nz = [torch.randn(B, 128, generator=g).cuda() for _ in range(3)]
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.step(data, *nz)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
res = {True: [], False: []}
for r in range(5):
    for fused in (False, True):
        M.FUSE_CONV_BN = fused
        run(2)
        res[fused].append(run(10))
for k, v in res.items():
    print(f"FUSE_CONV_BN={k}: median {statistics.median(v):.3f} ms  min {min(v):.3f}  all {[round(x, 2) for x in v]}")
