"""Host cost of enqueuing one iteration: at batch 2 the GPU work is far shorter than the enqueue, so
wall time per step = host time per step (launches, autograd, Python)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer
tr = BetaVAEGANTrainer(beta=25.0)
for B in (2, 128):
    x = (torch.rand(B, 3, 64, 64) * 2 - 1).cuda()
    for _ in range(5): tr.step(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): tr.step(x)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B}: host loop {(t1-t0)/30*1e3:.2f} ms/step, wall {(t2-t0)/30*1e3:.2f} ms/step")
