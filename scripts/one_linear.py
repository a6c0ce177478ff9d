"""Run the split-bf16 Linear GEMMs of one layer a few times (for rocprofv3): one_linear.py M N K"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops
M, N, K = (int(a) for a in sys.argv[1:4])
x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5, torch.randn(N, device="cuda")
gy = torch.randn(M, N, device="cuda")
for _ in range(6):
    ops.linear_fwd(x, w, b); ops.linear_dgrad(gy, w); ops.linear_wgrad(gy, x)
    torch.nn.functional.linear(x, w, b); gy @ w; gy.t() @ x
torch.cuda.synchronize()
