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

# BASELINE configs 4-5 (sizes the reference never defined: image side = 8 * n_z[1])
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer, ModelOpt
def flops_img(side):          # conv MACs scale with the pixel count; FC input with (side/8)^2
    s = (side / 64.0) ** 2
    fe = 109_772_800 * s + 2 * (16384 * s * 2048 + 2048 * 128)
    fg = 429_260_800 * s + 128 * 16384 * s
    fd = 429_260_800 * s + 16384 * s * 2048 + 2048
    return fe, fg, fd
for name, side, B, kind in (("config 4: new_gan 128x128", 128, 256, "gan"), ("config 5: beta=75 VAE-GAN 256x256 (512/8 per GPU)", 256, 64, "vaegan"),
                            ("beta=25 VAE-GAN 128x128", 128, 128, "vaegan")):
    opt = ModelOpt(n_z=[256, side // 8, side // 8])
    x = (torch.rand(B, 3, side, side) * 2 - 1).cuda(); e = torch.randn(B, 128).cuda()
    fe, fg, fd = flops_img(side)
    if kind == "gan":
        tr = GANTrainer(opt=opt); args = (x, e); gf = 2 * (8 * fd + 3 * fg) / 1e9
    else:
        tr = BetaVAEGANTrainer(beta=75.0 if side == 256 else 25.0, opt=opt); args = (x,); gf = 2 * (11 * fd + 9 * fg + 6 * fe) / 1e9
    dt = run(tr, args, n=10, w=3)
    print(f"{name}: B={B} {dt*1e3:8.2f} ms  {B/dt:8.1f} images/s  ({gf*B/dt/1e3:.1f} TFLOP/s algorithmic)", flush=True)
    del tr; torch.cuda.empty_cache()
