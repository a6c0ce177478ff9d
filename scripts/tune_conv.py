"""Time every tile variant of the implicit-GEMM kernels on the reference's heavy layer shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops, _lib

lib = _lib.use_tuning().__enter__()      # the vg_debug_* knobs live in the tuning build only
B = 128
FWD = [("dis.c3", 32, 128, 64), ("dis.c6", 128, 256, 32), ("dis.c9", 256, 256, 16), ("enc.f3", 64, 128, 32),
       ("enc.f6", 128, 256, 16)]
TR = [("dec.d1", 256, 256, 8), ("dec.d2", 256, 128, 16), ("dis.c6dg", 256, 128, 16), ("enc.f6dg", 256, 128, 8),
      ("enc.f3dg", 128, 64, 16)]
THIN = [("dec.d3", 128, 32, 32), ("dis.c3dg", 128, 32, 32)]


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3, 4, 5]
# argv[2]: "p" packed filters (cached: the pack is outside the timed launches), "u" plain layout
ops.USE_PACKED_FILTERS = not (len(sys.argv) > 2 and sys.argv[2] == "u")
print("packed filters:", ops.USE_PACKED_FILTERS, flush=True)
scope = ops.packed_filter_scope()
scope.__enter__()
for name, cin, cout, h in FWD:
    x = torch.randn(B, cin, h, h, device="cuda")
    w = torch.randn(cout, cin, 5, 5, device="cuda") * 0.02
    gf = 2.0 * B * (h // 2) ** 2 * cin * cout * 25 / 1e9
    res = []
    for v in variants:
        lib.vg_debug_set_conv_tile(0, v)
        ms = timeit(lambda: ops.conv5x5_fwd(x, w, None, 2))
        res.append(f"v{v}:{ms*1e3:6.0f}us {gf/ms:5.1f}TF")
    lib.vg_debug_set_conv_tile(0, -1)
    print(f"FWD {name:9s} {gf:5.1f}GF  " + "  ".join(res), flush=True)
for name, cin, cout, h in TR:
    x = torch.randn(B, cin, h, h, device="cuda")
    w = torch.randn(cin, cout, 5, 5, device="cuda") * 0.02
    gf = 2.0 * B * h * h * cin * cout * 25 / 1e9
    res = []
    for v in variants:
        lib.vg_debug_set_conv_tile(1, v)
        ms = timeit(lambda: ops.convT5x5_fwd(x, w, None, 2))
        res.append(f"v{v}:{ms*1e3:6.0f}us {gf/ms:5.1f}TF")
    lib.vg_debug_set_conv_tile(1, -1)
    print(f"TR  {name:9s} {gf:5.1f}GF  " + "  ".join(res), flush=True)
for name, cin, cout, h in THIN:
    x = torch.randn(B, cin, h, h, device="cuda")
    w = torch.randn(cin, cout, 5, 5, device="cuda") * 0.02
    gf = 2.0 * B * h * h * cin * cout * 25 / 1e9
    res = []
    for v in (6, 7, -1):
        lib.vg_debug_set_conv_tile(1, v)
        ms = timeit(lambda: ops.convT5x5_fwd(x, w, None, 2))
        res.append(f"v{v}:{ms*1e3:6.0f}us {gf/ms:5.1f}TF")
    lib.vg_debug_set_conv_tile(1, -1)
    print(f"TR  {name:9s} {gf:5.1f}GF  " + "  ".join(res), flush=True)
