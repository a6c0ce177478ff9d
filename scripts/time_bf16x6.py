import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops
from oracle import ops as O
B=128
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
scope = ops.packed_filter_scope(); scope.__enter__()
for name, cin, cout, h, s in [("dis.c3", 32, 128, 64, 2), ("dis.c6", 128, 256, 32, 2), ("dis.c9", 256, 256, 16, 2), ("enc.f3", 64, 128, 32, 2)]:
    x = torch.randn(B, cin, h, h, device="cuda"); w = torch.randn(cout, cin, 5, 5, device="cuda") * 0.02
    gf = 2.0 * B * (h // s) ** 2 * cin * cout * 25 / 1e9
    ref = O.conv5x5(x[:2].cpu(), w.cpu(), None, s)
    res=[]
    for mode in ("fp32", "bf16x3", "bf16x6"):
        ops.CONV_ARITH = mode
        y = ops.conv5x5_fwd(x, w, None, s)
        err = float((y[:2].cpu().double() - ref).norm() / ref.norm())
        ms = timeit(lambda: ops.conv5x5_fwd(x, w, None, s))
        res.append(f"{mode}: {ms*1e3:5.0f}us {gf/ms:5.1f}TF e{err:.1e}")
    print(f"{name:8s} " + "   ".join(res), flush=True)
for name, cin, cout, h in [("dec.d1", 256, 256, 8), ("dec.d2", 256, 128, 16), ("dec.d3", 128, 32, 32)]:
    x = torch.randn(B, cin, h, h, device="cuda"); w = torch.randn(cin, cout, 5, 5, device="cuda") * 0.02
    gf = 2.0 * B * h * h * cin * cout * 25 / 1e9
    ref = O.convT5x5(x[:2].cpu(), w.cpu(), None, 2)
    res=[]
    for mode in ("fp32", "bf16x3", "bf16x6"):
        ops.CONV_ARITH = mode
        y = ops.convT5x5_fwd(x, w, None, 2)
        err = float((y[:2].cpu().double() - ref).norm() / ref.norm())
        ms = timeit(lambda: ops.convT5x5_fwd(x, w, None, 2))
        res.append(f"{mode}: {ms*1e3:5.0f}us {gf/ms:5.1f}TF e{err:.1e}")
    print(f"TR {name:8s} " + "   ".join(res), flush=True)
for name, cin, cout, h, s in [("dis.c3", 32, 128, 64, 2), ("dis.c6", 128, 256, 32, 2), ("dis.c9", 256, 256, 16, 2), ("enc.f3", 64, 128, 32, 2)]:
    x = torch.randn(B, cin, h, h, device="cuda"); gy = torch.randn(B, cout, h // s, h // s, device="cuda")
    gf = 2.0 * B * (h // s) ** 2 * cin * cout * 25 / 1e9
    res = []
    for mode in ("fp32", "bf16x3", "bf16x6"):
        ops.CONV_ARITH = mode
        ms = timeit(lambda: ops.conv5x5_wgrad(x, gy, s))
        res.append(f"{mode}: {ms*1e3:5.0f}us {gf/ms:5.1f}TF")
    print(f"WGRAD {name:8s} " + "   ".join(res), flush=True)
