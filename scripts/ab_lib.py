"""Same-box A/B of two builds of the library on the ring-kernel layer shapes (B = 128, bf16x6): alternating processes, one per
library, each timing every shape (median of 20 after 60 warm-up launches) and checking the outputs against the product
build bit for bit.  Usage: ab_lib.py <libA.so|product> <libB.so> [rounds]"""
import sys, os, statistics, subprocess, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("fwd", 128, 256, 32), ("fwd", 256, 256, 16), ("fwd", 32, 128, 64), ("fwd", 64, 128, 32), ("fwd", 128, 256, 16),
          ("tr", 256, 128, 16), ("tr", 256, 256, 8), ("tr", 128, 64, 16), ("tr", 128, 32, 32)]
if len(sys.argv) > 2 and sys.argv[1] != "--one":
    a, b = sys.argv[1], sys.argv[2]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    for r in range(rounds):
        for lib in (a, b):
            subprocess.run([sys.executable, __file__, "--one", lib], check=False)
    sys.exit(0)
path = sys.argv[2]
sys.path.insert(0, ROOT)
import torch
from disentangle_mlp_amd import _lib
if path != "product":
    _lib.LIB_PATH = os.path.join(ROOT, path)
from disentangle_mlp_amd import ops
B = 128


def timeit(fn, n=20):
    for _ in range(60):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


out, tot = [], 0.0
with ops.packed_filter_scope():
    for kind, ci, co, h in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(5)
        x = torch.randn(B, ci, h, h, device="cuda", generator=g)
        w = 0.05 * torch.randn(*((co, ci, 5, 5) if kind == "fwd" else (ci, co, 5, 5)), device="cuda", generator=g)
        sc, sh = torch.rand(ci, device="cuda", generator=g) + 0.5, torch.randn(ci, device="cuda", generator=g)
        conv = ops.conv5x5_fwd if kind == "fwd" else ops.convT5x5_fwd
        fn = lambda: conv(x, w, None, 2, in_affine=(sc, sh, 2), want_stats=True)
        y, st = fn()
        dig = hashlib.md5(y.cpu().numpy().tobytes() + (st.cpu().numpy().tobytes() if st is not None else b"")).hexdigest()[:8]
        t = timeit(fn) * 1e3
        tot += t
        out.append(f"{kind}{ci}>{co}@{h} {t:6.1f} {dig}")
print(f"{os.path.basename(path):24s} sum {tot:7.1f} us | " + " | ".join(out), flush=True)
