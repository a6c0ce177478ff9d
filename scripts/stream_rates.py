"""Streaming rates of this GPU on a 67 MB fp32 tensor (the size of the 32-channel 64 x 64 activations at B = 128) and a
1 GB one: fill (write), sum (read), copy (read + write) -- what "one pass over the tensor" costs.  Run under
rocprofv3 --kernel-trace --stats for kernel times."""
import torch
for n in (128 * 32 * 64 * 64, 256 * 1024 * 1024):
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    for _ in range(10):
        y.fill_(1.0); x.sum(); y.copy_(x); torch.relu(x, out=y) if False else None
    torch.cuda.synchronize()
