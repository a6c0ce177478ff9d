"""Per-shape timing of every convolution launch of one beta-VAE-GAN iteration (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tr = BetaVAEGANTrainer(beta=25.0)
x = (torch.rand(B, 3, 64, 64) * 2 - 1).cuda()
for _ in range(3):
    tr.step(x)
torch.cuda.synchronize()
ops.start_timing()
N = 5
for _ in range(N):
    tr.step(x)
t = ops.stop_timing()


def gflop(key):
    op, b, cin, h, w, cout, s = key
    if op == "conv_fwd":
        return 2.0 * b * ((h - 1) // s + 1) * ((w - 1) // s + 1) * cin * cout * 25 / 1e9
    if op == "convT_fwd":
        return 2.0 * b * h * w * cin * cout * 25 / 1e9
    return 2.0 * b * ((h - 1) // s + 1) * ((w - 1) // s + 1) * cin * cout * 25 / 1e9   # wgrad: x shape


rows = []
for key, ms in t.items():
    gf = gflop(key)
    avg = sum(ms) / len(ms)
    rows.append((sum(ms) / N, len(ms) / N, avg, gf / avg, key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"conv launches: {tot:.2f} ms / iteration")
for tot_ms, n, avg, tf, key in rows:
    lost = tot_ms * (1 - tf / 130.0)
    print(f"{tot_ms:7.3f} ms  x{n:4.1f}  avg {avg*1e3:7.1f} us  {tf:6.1f} TF  (vs 130 TF: {lost:5.2f} ms)  {key}")
