"""Launch times of the weight gradient at the benchmark's layer shapes (B = 128) in the active arithmetic."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops, _lib
lib = _lib.use_tuning().__enter__()
ops.CONV_ARITH = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
B = 128
# x (B, Cin, H, H), gy (B, Cout, H/s, H/s): D / encoder convolutions and the decoder's transposed ones (roles swapped)
LAYERS = [(32, 128, 64, 2), (64, 128, 32, 2), (128, 256, 32, 2), (128, 256, 16, 2), (256, 256, 16, 2), (3, 32, 64, 1), (3, 64, 64, 2)]
def timeit(fn, n=15):
    for _ in range(60): fn()          # the clock takes tens of launches to settle after idle
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)
for (ci, co, h, s) in LAYERS:
    x = torch.randn(B, ci, h, h, device="cuda"); gy = torch.randn(B, co, h // s, h // s, device="cuda")
    gflop = 2.0 * B * (h // s) ** 2 * ci * co * 25 / 1e9
    res = []
    for th in (0, 1, 2):                                   # pixel rows of a chunk: planned / 1 / 2
        lib.vg_debug_set_wgrad(5, th)
        for split in (True, False):
            if not split and th > 0: continue
            ops.WGRAD_SPLIT = split
            ms = timeit(lambda: ops.conv5x5_wgrad(x, gy, s))
            res.append(f"{'split th%d' % th if split else 'fp32'}: {ms*1e3:6.1f} us {gflop/ms:6.1f} TF")
    ops.WGRAD_SPLIT = True; lib.vg_debug_set_wgrad(5, 0)
    print(f"wgrad {ci:3d}->{co:3d} @{h:2d} s{s}: " + " | ".join(res), flush=True)
