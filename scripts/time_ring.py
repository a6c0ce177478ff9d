"""Launch times of the split-bf16 forward / transposed convolutions at the benchmark's layer shapes (B = 128),
per tile variant of the stride-2 ring kernel (conv_ring.hip); interleaved rounds in one process, median."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops, _lib
lib = _lib.use_tuning().__enter__()      # the vg_debug_* knobs live in the tuning build only
ops.CONV_ARITH = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
B = 128
FWD = [(32, 128, 64), (64, 128, 32), (128, 256, 32), (128, 256, 16), (256, 256, 16)]      # Cin, Cout, H (stride 2)
TR = [(256, 256, 8), (256, 128, 16), (256, 128, 8), (128, 64, 16), (128, 32, 32)]
peak = 2500.0 / (6 if ops.CONV_ARITH == "bf16x6" else 3)
def timeit(fn, n=20):
    for _ in range(60): fn()          # the clock takes tens of launches to settle after idle
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)
with ops.packed_filter_scope():
    for kind, layers in (("fwd", FWD), ("tr", TR)):
        for (ci, co, h) in layers:
            x = torch.randn(B, ci, h, h, device="cuda")
            w = 0.05 * torch.randn(*((co, ci, 5, 5) if kind == "fwd" else (ci, co, 5, 5)), device="cuda")
            fn = (lambda: ops.conv5x5_fwd(x, w, None, 2)) if kind == "fwd" else (lambda: ops.convT5x5_fwd(x, w, None, 2))
            oh = h // 2 if kind == "fwd" else h
            gflop = 2.0 * B * oh * oh * ci * co * 25 / 1e9
            res = []
            for v in (-1, 0, 1, 2, 3):
                if v == 3 and kind == "fwd": continue
                lib.vg_debug_set_conv_ring_tile(v)
                try:
                    ms = timeit(fn)
                    res.append(f"v{v}: {ms*1e3:6.1f} us {gflop/ms:6.1f} TF ({gflop/ms/peak*100:4.1f}%)")
                except Exception as e:
                    res.append(f"v{v}: {type(e).__name__}")
            lib.vg_debug_set_conv_ring_tile(-1)
            print(f"{kind} {ci:3d}->{co:3d} @{h:2d}: " + " | ".join(res), flush=True)
