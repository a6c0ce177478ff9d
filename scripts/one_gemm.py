"""One Linear GEMM of the benchmark's big layer, a few launches, for rocprofv3 (kernel trace / PMC passes):
one_gemm.py <fwd|dgrad|wgrad> [M]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 128
K, N = 16384, 2048
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02
gy = torch.randn(M, N, device="cuda") * 1e-3
fn = {"fwd": lambda: ops.linear_fwd(x, w, None), "dgrad": lambda: ops.linear_dgrad(gy, w), "wgrad": lambda: ops.linear_wgrad(gy, x)}[kind]
with ops.packed_filter_scope():
    for _ in range(30):
        fn()
torch.cuda.synchronize()
