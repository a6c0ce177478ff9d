"""Timings of the secondary drivers: new_vae.py step (BASELINE config 1 shape, B=16 and B=128) and
new_gan.py step at 64x64 (the reference's Generator is hard-wired to 64x64)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd.trainer import VAETrainer, GANTrainer
def run(tr, args, n=30, w=5):
    for _ in range(w): tr.step(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.step(*args)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for B in (16, 128):
    x = (torch.rand(B, 3, 64, 64) * 2 - 1).cuda(); e = torch.randn(B, 128).cuda()
    dt = run(VAETrainer(beta=1.0), (x, e))
    print(f"new_vae step  B={B:4d}: {dt*1e3:7.2f} ms  {B/dt:8.1f} images/s  ({3.653*B/dt/1e3:.1f} TFLOP/s algorithmic)")
    dt = run(GANTrainer(), (x, e))
    print(f"new_gan step  B={B:4d}: {dt*1e3:7.2f} ms  {B/dt:8.1f} images/s  ({9.993*B/dt/1e3:.1f} TFLOP/s algorithmic)")
