"""fp32 (default) vs opt-in bf16x3 forward convolution on the heavy layer shapes, B = 128."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops
from oracle import ops as O
from disentangle_mlp_amd import _lib
lib = _lib.use_tuning().__enter__()      # the vg_debug_* knobs live in the tuning build only

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
    for mode, var in (("fp32", -1), ("bf16x3", 0), ("bf16x3", 1), ("bf16x3", 2), ("bf16x3", 3), ("bf16x3", -1)):
        ops.CONV_ARITH = mode
        lib.vg_debug_set_conv_bf16split_tile(var)
        y = ops.conv5x5_fwd(x, w, None, s)
        err = float((y[:2].cpu().double() - ref).norm() / ref.norm())
        ms = timeit(lambda: ops.conv5x5_fwd(x, w, None, s))
        res.append(f"{mode if mode == 'fp32' else 'x3/v%d' % var}: {ms*1e3:5.0f}us {gf/ms:5.1f}TF e{err:.0e}")
    ops.CONV_ARITH = "fp32"
    print(f"{name:9s} {gf:5.1f} GF  " + "   ".join(res), flush=True)

TR = [("dec.d1", 256, 256, 8), ("dec.d2", 256, 128, 16), ("dec.d3", 128, 32, 32), ("enc.f6dg", 256, 128, 8),
      ("enc.f3dg", 128, 64, 16)]
for name, cin, cout, h in TR:
    x = torch.randn(B, cin, h, h, device="cuda")
    w = torch.randn(cin, cout, 5, 5, device="cuda") * 0.02
    gf = 2.0 * B * h * h * cin * cout * 25 / 1e9
    ref = O.convT5x5(x[:2].cpu(), w.cpu(), None, 2)
    res = []
    for mode, var in (("fp32", -1), ("bf16x3", 1), ("bf16x3", 2), ("bf16x3", 3), ("bf16x3", 4), ("bf16x3", -1)):
        ops.CONV_ARITH = mode
        lib.vg_debug_set_conv_bf16split_tile(var)
        y = ops.convT5x5_fwd(x, w, None, 2)
        err = float((y[:2].cpu().double() - ref).norm() / ref.norm())
        ms = timeit(lambda: ops.convT5x5_fwd(x, w, None, 2))
        res.append(f"{mode if mode == 'fp32' else 'x3/v%d' % var}: {ms*1e3:5.0f}us {gf/ms:5.1f}TF e{err:.0e}")
    ops.CONV_ARITH = "fp32"
    lib.vg_debug_set_conv_bf16split_tile(-1)
    print(f"TR {name:9s} {gf:5.1f} GF  " + "   ".join(res), flush=True)

WG = [("dis.c3", 32, 128, 64, 2), ("dis.c6", 128, 256, 32, 2), ("dis.c9", 256, 256, 16, 2), ("enc.f3", 64, 128, 32, 2),
      ("enc.f6", 128, 256, 16, 2), ("dis.c0", 3, 32, 64, 1), ("enc.f0", 3, 64, 64, 2)]
for name, cin, cout, h, s in WG:
    x = torch.randn(B, cin, h, h, device="cuda")
    gy = torch.randn(B, cout, h // s, h // s, device="cuda")
    gf = 2.0 * B * (h // s) ** 2 * cin * cout * 25 / 1e9
    res = []
    ref = None
    for mode in ("fp32", "bf16x3"):
        ops.CONV_ARITH = mode
        out = ops.conv5x5_wgrad(x, gy, s)
        if ref is None:
            ref = out.double()
        err = float((out.double() - ref).norm() / ref.norm())
        ms = timeit(lambda: ops.conv5x5_wgrad(x, gy, s))
        res.append(f"{mode}: {ms*1e3:5.0f}us {gf/ms:5.1f}TF (vs fp32 kernel {err:.0e})")
    ops.CONV_ARITH = "fp32"
    print(f"WGRAD {name:7s} {gf:5.1f} GF  " + "   ".join(res), flush=True)
