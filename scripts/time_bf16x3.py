"""fp32 (default) vs opt-in bf16x3 forward convolution on the heavy layer shapes, B = 128."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops
from oracle import ops as O

B = 128
L = [("dis.c3", 32, 128, 64, 2), ("dis.c6", 128, 256, 32, 2), ("dis.c9", 256, 256, 16, 2), ("enc.f3", 64, 128, 32, 2),
     ("enc.f6", 128, 256, 16, 2), ("dec.d3dg", 32, 128, 64, 2), ("dec.d2dg", 128, 256, 32, 2)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


scope = ops.packed_filter_scope()
scope.__enter__()
for name, cin, cout, h, s in L:
    x = torch.randn(B, cin, h, h, device="cuda")
    w = torch.randn(cout, cin, 5, 5, device="cuda") * 0.02
    gf = 2.0 * B * (h // s) ** 2 * cin * cout * 25 / 1e9
    ref = O.conv5x5(x[:2].cpu(), w.cpu(), None, s)
    res = []
    for mode in ("fp32", "bf16x3"):
        ops.CONV_FWD_ARITH = mode
        y = ops.conv5x5_fwd(x, w, None, s)
        err = float((y[:2].cpu().double() - ref).norm() / ref.norm())
        ms = timeit(lambda: ops.conv5x5_fwd(x, w, None, s))
        res.append(f"{mode}: {ms*1e3:6.0f} us {gf/ms:6.1f} TF-eq  err {err:.1e}")
    ops.CONV_FWD_ARITH = "fp32"
    print(f"{name:9s} {gf:5.1f} GF  " + "   ".join(res), flush=True)
