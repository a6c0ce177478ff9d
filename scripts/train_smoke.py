"""Longer run: 120 iterations on a fixed synthetic set of 4 batches; losses must stay finite and the
reconstruction error must fall.  Also reports host enqueue time per step vs wall time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer
if len(sys.argv) > 1:
    ops.CONV_ARITH = sys.argv[1]          # "fp32" (default) or "bf16x3"
print("convolution arithmetic:", ops.CONV_ARITH)
tr = BetaVAEGANTrainer(beta=25.0)
g = torch.Generator().manual_seed(7)
B = 128
# smooth synthetic "images": low-frequency patterns so that there is something to learn
base = torch.randn(4 * B, 3, 8, 8, generator=g)
data = torch.tanh(torch.nn.functional.interpolate(base, size=64, mode="bilinear")).cuda()
t_host = 0.0
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(120):
    x = data[(it % 4) * B:(it % 4 + 1) * B]
    h0 = time.perf_counter()
    out = tr.step(x)
    t_host += time.perf_counter() - h0
    if it % 20 == 0 or it == 119:
        print(it, {k: round(float(v), 3) for k, v in out.items()}, flush=True)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"wall {wall/120*1e3:.1f} ms/step (incl. syncs for printing), host enqueue {t_host/120*1e3:.1f} ms/step, "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
