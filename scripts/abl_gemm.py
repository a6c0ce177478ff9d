"""Ablation timings of the Linear GEMM (gemm_split.hip) at the 16384 <-> 2048 shapes: one process per one-off library built
by experiments/abl_build.sh gemm <bits> (1 B rows cache-resident, 2 A rows cache-resident, 4 no plane split, 8 no MFMAs).
Usage: abl_gemm.py <bits> ..."""
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
    _lib.LIB_PATH = os.path.join(ROOT, "experiments", "abl", f"libabl_gemm_{bits}.so")
from disentangle_mlp_amd import ops
def timeit(fn, n=15):
    for _ in range(40): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts) * 1e3
out = []
for (M, K, N) in ((128, 16384, 2048), (256, 16384, 2048), (128, 2048, 16384)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02
    gy = torch.randn(M, N, device="cuda") * 1e-3
    with ops.packed_filter_scope():
        out.append(f"M={M} K={K} N={N}: fwd {timeit(lambda: ops.linear_fwd(x, w, None)):6.1f} dgrad {timeit(lambda: ops.linear_dgrad(gy, w)):6.1f} "
                   f"wgrad {timeit(lambda: ops.linear_wgrad(gy, x)):6.1f} us")
names = {1: "B cached", 2: "A cached", 4: "no split", 8: "no MFMA", 16: "split on the last groups"}
print(f"abl {bits:3d} [{', '.join(v for k, v in names.items() if bits & k) or 'full'}]: " + " | ".join(out), flush=True)
