"""The Linear layers' GEMMs at the benchmark's shapes: this package's fp16x3 GEMM (csrc/gemm_split.hip) against the vendor
fp32 GEMM with the shipped algorithm table, forward / data gradient / weight gradient, each with its error against fp64.
Backs DESIGN.md section 4.5 (round 4)."""
import statistics, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops, tuned_gemms
tuned_gemms.enable()


def timeit(fn, n=20):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts) * 1e3


def err(got, ref):
    return float((got.double() - ref).norm() / ref.norm())


for (M, K, N) in ((128, 16384, 2048), (256, 16384, 2048), (384, 16384, 2048), (128, 128, 16384), (256, 128, 16384)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02
    b = torch.randn(N, device="cuda"); gy = torch.randn(M, N, device="cuda") * 1e-3
    with ops.packed_filter_scope():          # the weight's bound is measured once, as inside a training iteration
        rows = []
        for name, own, ven, ref in (
                ("fwd", lambda: ops.linear_fwd(x, w, b), lambda: torch.nn.functional.linear(x, w, b),
                 lambda: x.double() @ w.double().t() + b.double()),
                ("dgrad", lambda: ops.linear_dgrad(gy, w), lambda: gy @ w, lambda: gy.double() @ w.double()),
                ("wgrad", lambda: ops.linear_wgrad(gy, x), lambda: gy.t() @ x, lambda: gy.double().t() @ x.double())):
            r = ref()
            rows.append(f"{name}: own {timeit(own):6.1f} us ({err(own(), r):.1e}) vendor {timeit(ven):6.1f} us ({err(ven(), r):.1e})")
    print(f"M={M:4d} K={K:6d} N={N:6d} | " + " | ".join(rows), flush=True)
