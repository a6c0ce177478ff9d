import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops
cin, cout, h, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
x = torch.randn(128, cin, h, h, device="cuda"); gy = torch.randn(128, cout, h // s, h // s, device="cuda")
for _ in range(4): ops.conv5x5_wgrad(x, gy, s)
torch.cuda.synchronize()
