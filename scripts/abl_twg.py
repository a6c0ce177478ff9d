"""Ablation runs of the thin weight gradient (conv_thin_wgrad.hip) on convs.0 (3 -> 32, 64 x 64, B = 128) for rocprofv3
--kernel-trace (python-side timing is launch-bound for kernels this short): abl_twg.py <bits>  (libraries from
experiments/abl_build.sh twg <bits>: 1 gy of channel 0 for every lane, 2 plane copies once, 4 no MFMAs)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bits = int(sys.argv[1])
sys.path.insert(0, ROOT)
import torch
from disentangle_mlp_amd import _lib
if bits:
    _lib.LIB_PATH = os.path.join(ROOT, "experiments", "abl", f"libabl_twg_{bits}.so")
from disentangle_mlp_amd import ops
B = 128
x3 = torch.randn(B, 3, 64, 64, device="cuda"); g32 = torch.randn(B, 32, 64, 64, device="cuda")
for _ in range(30): ops.conv5x5_wgrad(x3, g32, 1)
torch.cuda.synchronize()
