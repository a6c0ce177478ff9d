"""Linear GEMMs of the 16384 <-> 2048 layers at B = 128: split-bf16 kernel vs the vendor fp32 GEMM (torch)."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops
def timeit(fn, n=20):
    for _ in range(60): fn()          # the clock takes tens of launches to settle after idle
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)
for (M, N, K) in ((128, 2048, 16384), (128, 16384, 128)):
    x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5, torch.randn(N, device="cuda")
    gy = torch.randn(M, N, device="cuda")
    gf = 2.0 * M * N * K / 1e9
    for name, f_split, f_lib in (("fwd", lambda: ops.linear_fwd(x, w, b), lambda: torch.nn.functional.linear(x, w, b)),
                                 ("dgrad", lambda: ops.linear_dgrad(gy, w), lambda: gy @ w),
                                 ("wgrad", lambda: ops.linear_wgrad(gy, x), lambda: gy.t() @ x)):
        a, c = timeit(f_split), timeit(f_lib)
        print(f"{name:6s} M={M} N={N} K={K}: split {a*1e3:7.1f} us ({gf/a:6.1f} TF)   vendor fp32 {c*1e3:7.1f} us ({gf/c:6.1f} TF)", flush=True)
