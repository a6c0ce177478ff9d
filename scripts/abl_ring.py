"""Ablation timings of the ring kernel on the dominant shape (tuning build): which part of a K step costs what."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops, _lib
lib = _lib.use_tuning().__enter__()
ops.CONV_ARITH = "bf16x6"
B = 128
shapes = [("fwd", 128, 256, 32), ("tr", 256, 128, 16)] if len(sys.argv) < 2 else [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))]
NAMES = {0: "full", 1: "no DMA", 2: "no barrier", 4: "no staging", 8: "no vmcnt wait", 16: "no B reads", 32: "no A reads", 48: "no reads",
         64: "no MFMA", 1 + 4: "no DMA, no staging", 1 + 4 + 48: "MFMA + barrier only", 1 + 2 + 4 + 48: "MFMA only", 64 + 4: "DMA + reads + barrier",
         64 + 4 + 48: "DMA + barrier", 64 + 4 + 1: "reads + barrier"}
def timeit(fn, n=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)
with ops.packed_filter_scope():
    for kind, ci, co, h in shapes:
        x = torch.randn(B, ci, h, h, device="cuda")
        w = 0.05 * torch.randn(*((co, ci, 5, 5) if kind == "fwd" else (ci, co, 5, 5)), device="cuda")
        fn = (lambda: ops.conv5x5_fwd(x, w, None, 2)) if kind == "fwd" else (lambda: ops.convT5x5_fwd(x, w, None, 2))
        print(kind, ci, co, h)
        for bits, name in NAMES.items():
            lib.vg_debug_set_conv_ring_tile(1000 + bits)
            print(f"  {name:28s} {timeit(fn)*1e3:7.1f} us", flush=True)
        lib.vg_debug_set_conv_ring_tile(1000)
