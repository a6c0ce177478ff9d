import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops
x = torch.randn(128, 32, 64, 64, device="cuda"); w = torch.randn(32, 3, 5, 5, device="cuda") * 0.05
b = torch.randn(3, device="cuda")
for _ in range(3): ops.convT5x5_fwd(x, w, b, 1)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): ops.convT5x5_fwd(x, w, b, 1)
e.record(); torch.cuda.synchronize()
ms = a.elapsed_time(e) / 20
print(f"thin convT 32->3 @64x64 B128: {ms*1e3:.1f} us  {2*128*4096*32*3*25/ms/1e9:.1f} TFLOP/s")
