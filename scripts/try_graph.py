"""Feasibility: capture one whole training iteration in a HIP graph (torch.cuda.CUDAGraph) and replay it."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer, VAETrainer
which, B = sys.argv[1], int(sys.argv[2])
tr = BetaVAEGANTrainer(beta=25.0, capturable=True) if which == "vaegan" else VAETrainer(beta=1.0, capturable=True)
x = (torch.rand(B, 3, 64, 64) * 2 - 1).cuda()
n = [torch.randn(B, 128).cuda() for _ in range(3)]
args = (x, *n) if which == "vaegan" else (x, n[0])
def eager(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): tr.step(*args)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
for _ in range(5): tr.step(*args)
print(f"eager  : {eager(30)*1e3:.3f} ms/step")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): tr.step(*args)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = tr.step(*args)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): g.replay()
torch.cuda.synchronize()
print(f"graphed: {(time.perf_counter()-t0)/30*1e3:.3f} ms/step", {k: round(float(v), 3) for k, v in out.items()})
