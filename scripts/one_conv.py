"""Run one conv shape/variant a few times (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops, _lib
lib = _lib.use_tuning().__enter__()      # the vg_debug_* knobs live in the tuning build only
mode, var, cin, cout, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
B = 128
x = torch.randn(B, cin, h, h, device="cuda")
if mode == "fwd":
    w = torch.randn(cout, cin, 5, 5, device="cuda") * 0.02
    lib.vg_debug_set_conv_tile(0, var)
    f = lambda: ops.conv5x5_fwd(x, w, None, 2)
else:
    w = torch.randn(cin, cout, 5, 5, device="cuda") * 0.02
    lib.vg_debug_set_conv_tile(1, var)
    f = lambda: ops.convT5x5_fwd(x, w, None, 2)
for _ in range(5):
    f()
torch.cuda.synchronize()
