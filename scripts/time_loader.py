"""Throughput of the HBM-resident input pipeline (N2) and of the grid writer (N3) on one GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import data as D, ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 202599        # CelebA
u8 = torch.randint(0, 256, (N, 64, 64, 3), dtype=torch.uint8)
t0 = time.perf_counter()
ds = D.DeviceImageDataset(u8, device="cuda")
torch.cuda.synchronize()
print(f"upload of {u8.numel()/1e9:.2f} GB cache: {time.perf_counter()-t0:.2f} s")
for bs in (128, 1024):
    ld = D.DeviceLoader(ds, bs, shuffle=True)
    for _ in ld:      # warm-up epoch
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nb = 0
    for data, lab in ld:
        nb += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    byts = N * (64 * 64 * 3 * 5)          # 1 byte in + 4 bytes out per element
    print(f"batch {bs:5d}: epoch of {N} images in {dt*1e3:7.1f} ms = {N/dt/1e6:6.2f} M images/s, "
          f"{byts/dt/1e9:7.1f} GB/s algorithmic, {dt/nb*1e6:6.1f} us/batch (host-enqueue bound)")
# kernel-only rate: one launch over 16384 images
idx = torch.randint(0, N, (16384,), device="cuda")
for _ in range(3):
    ops.u8_gather_normalize(ds.images, idx)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    ops.u8_gather_normalize(ds.images, idx)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print(f"gather+normalize kernel, 16384 images: {ms*1e3:.1f} us = {16384*64*64*3*5/ms/1e6:.0f} GB/s of ~8000 peak")
x = torch.tanh(torch.randn(64, 3, 64, 64, device="cuda"))
for _ in range(3):
    ops.image_grid_u8(x, normalize=True)
a.record()
for _ in range(50):
    ops.image_grid_u8(x, normalize=True)
b.record(); torch.cuda.synchronize()
print(f"64-image grid (min-max + tile + quantise): {a.elapsed_time(b)/50*1e3:.1f} us on device")
