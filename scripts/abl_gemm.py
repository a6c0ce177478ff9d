"""Ablation timings of the split-bf16 GEMM (gemm_split.hip) on the 16384 -> 2048 forward at B = 128: one process per
one-off library built by experiments/abl_build.sh gemm <bits> (1 B from cache-resident rows, 2 A likewise, 4 no plane
split).  Kernel + slab sum.  Usage: abl_gemm.py <bits> ..."""
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
def timeit(fn, n=20):
    for _ in range(60): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)
M, N, K = 128, 2048, 16384
x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5, torch.randn(N, device="cuda")
gy = torch.randn(M, N, device="cuda")
t = [timeit(f) for f in (lambda: ops.linear_fwd(x, w, b), lambda: ops.linear_dgrad(gy, w), lambda: ops.linear_wgrad(gy, x))]
names = {1: "B cached", 2: "A cached", 4: "no split"}
print(f"abl {bits:3d} [{', '.join(v for k, v in names.items() if bits & k) or 'full'}]: fwd {t[0]*1e3:6.1f}  dgrad {t[1]*1e3:6.1f}  wgrad {t[2]*1e3:6.1f} us", flush=True)
