import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops, _lib
lib = _lib.use_tuning().__enter__()      # the vg_debug_* knobs live in the tuning build only
B = 128
L = [("dis.c3", 32, 128, 64, 2), ("dis.c6", 128, 256, 32, 2), ("dis.c9", 256, 256, 16, 2), ("enc.f3", 64, 128, 32, 2),
     ("enc.f6", 128, 256, 16, 2), ("dis.c0", 3, 32, 64, 1), ("enc.f0", 3, 64, 64, 2)]
TARGETS = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else (-1, 1024, 2048)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for name, cin, cout, h, s in L:
    x = torch.randn(B, cin, h, h, device="cuda")
    gy = torch.randn(B, cout, h // s, h // s, device="cuda")
    gf = 2.0 * B * (h // s) ** 2 * cin * cout * 25 / 1e9
    res = []
    ref = None
    for ks in (5, 4):           # 5: default; 4: same with scalar gy loads (vg_debug_set_wgrad(4, 0))
        for tgt in TARGETS:
            lib.vg_debug_set_wgrad(4, 1 if ks == 5 else 0); lib.vg_debug_set_wgrad(1, tgt)
            out = ops.conv5x5_wgrad(x, gy, s)
            if ref is None:
                ref = out.clone()
            err = float((out - ref).abs().max() / ref.abs().max())
            ms = timeit(lambda: ops.conv5x5_wgrad(x, gy, s))
            res.append(f"{'vec4' if ks == 5 else 'scal'}/b{tgt}:{ms*1e3:5.0f}us {gf/ms:5.1f}TF e{err:.0e}")
    lib.vg_debug_set_wgrad(4, 1); lib.vg_debug_set_wgrad(1, -1)
    print(f"{name:7s} {gf:5.1f}GF " + " ".join(res), flush=True)
