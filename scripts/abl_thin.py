"""Ablation timings of the thin forward convolution (conv_thin_fwd.hip) on convs.0 (3 -> 32, 64 x 64, B = 128): one process
per one-off library built by experiments/abl_build.sh tfwd <bits> (1 no stores, 2 no MFMAs, 4 input rows built once,
8 no statistics).  Usage: abl_thin.py <bits> ..."""
import sys, os, statistics, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2:
    for b in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, b], check=False)
    sys.exit(0)
bits = int(sys.argv[1])
sys.path.insert(0, ROOT)
import torch
from disentangle_mlp_amd import _lib
if bits:
    _lib.LIB_PATH = os.path.join(ROOT, "experiments", "abl", f"libabl_tfwd_{bits}.so")
from disentangle_mlp_amd import ops
B = 128
x3 = torch.randn(B, 3, 64, 64, device="cuda"); w32 = torch.randn(32, 3, 5, 5, device="cuda") * 0.05; b32 = torch.randn(32, device="cuda")
def run(n):
    for _ in range(n): ops.conv5x5_fwd(x3, w32, b32, 1, want_stats=True)
run(50); torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(20); b.record(); torch.cuda.synchronize()      # 20 back-to-back launches: launch overhead amortised
    ts.append(a.elapsed_time(b) / 20)
names = {1: "no stores", 2: "no MFMA", 4: "rows built once", 8: "no statistics"}
print(f"abl {bits:3d} [{', '.join(v for k, v in names.items() if bits & k) or 'full'}]: {statistics.median(ts)*1e3:7.1f} us per call", flush=True)
