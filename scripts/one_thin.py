"""Run the 3-channel edge layers of the benchmark (B = 128, 64 x 64) a few times each (for rocprofv3 --kernel-trace):
deconv4 forward / convs.0 data gradient (convT 32 -> 3), convs.0 forward (3 -> 32), convs.0 weight gradient,
features.0 forward (3 -> 64, stride 2) and its weight gradient."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops
B = 128
x3 = torch.randn(B, 3, 64, 64, device="cuda"); x32 = torch.randn(B, 32, 64, 64, device="cuda"); g64 = torch.randn(B, 64, 32, 32, device="cuda")
wT = torch.randn(32, 3, 5, 5, device="cuda") * 0.05; b3 = torch.randn(3, device="cuda")
w32 = torch.randn(32, 3, 5, 5, device="cuda") * 0.05; b32 = torch.randn(32, device="cuda")
w64 = torch.randn(64, 3, 5, 5, device="cuda") * 0.05; b64 = torch.randn(64, device="cuda")
for _ in range(8):
    ops.convT5x5_fwd(x32, wT, b3, 1)
    ops.conv5x5_fwd(x3, w32, b32, 1)
    ops.conv5x5_wgrad(x3, x32, 1)
    ops.conv5x5_fwd(x3, w64, b64, 2)
    ops.conv5x5_wgrad(x3, g64, 2)
torch.cuda.synchronize()
