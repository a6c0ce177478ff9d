"""HipAdam (vg_adam_step) vs torch's fused Adam on the reference's parameter set (109.5 M fp32 parameters)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd.optim import HipAdam
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer
tr = BetaVAEGANTrainer(beta=25.0)
params = [p for p in list(tr.netEG.parameters()) + list(tr.netD.parameters())]
n = sum(p.numel() for p in params)
for p in params:
    p.grad = torch.randn_like(p)
def timeit(opt, reps=10):
    for _ in range(3): opt.step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): opt.step()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for name, opt in (("HipAdam", HipAdam(params, lr=1e-3)), ("torch fused Adam", torch.optim.Adam(params, lr=1e-3, fused=True))):
    ms = timeit(opt)
    print(f"{name:18s}: {ms*1e3:7.1f} us for {n/1e6:.1f} M parameters = {n*28/ms/1e9:6.2f} TB/s of ~8 (28 B per parameter)")
