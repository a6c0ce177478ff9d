import sys, os, cProfile, pstats, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer
tr = BetaVAEGANTrainer(beta=25.0)
x = (torch.rand(128, 3, 64, 64) * 2 - 1).cuda()
n = [torch.randn(128, 128).cuda() for _ in range(3)]
for _ in range(3): tr.step(x, *n)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5): tr.step(x, *n)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
