"""Run one weight-gradient shape a few times (for rocprofv3): one_wgrad.py cin cout h stride"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops
cin, cout, h, s = (int(a) for a in sys.argv[1:5])
B = 128
x = torch.randn(B, cin, h, h, device="cuda"); gy = torch.randn(B, cout, h // s, h // s, device="cuda")
for _ in range(6):
    ops.conv5x5_wgrad(x, gy, s)
torch.cuda.synchronize()
