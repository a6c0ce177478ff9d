"""Ablation timings of the split-bf16 weight gradient (wgrad_bf16split.hip): one process per one-off library built by
experiments/abl_build.sh wx <bits> (1 gy from a cache-resident pixel, 2 x patch loaded once, 4 patch split + LDS store
once, 8 no MFMAs, 16 patch fragments read once, 32 every product as two 16x16x32 MFMAs -- the shape experiment).  Times the whole op (re-layout + kernel + slab sum).
Usage: abl_wx.py <bits> ..."""
import sys, os, statistics, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2:
    for b in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, b], check=False)
    sys.exit(0)
bits = int(sys.argv[1])
sys.path.insert(0, ROOT)
import torch
from disentangle_mlp_amd import _lib
if bits:
    _lib.LIB_PATH = os.path.join(ROOT, "experiments", "abl", f"libabl_wx_{bits}.so")
from disentangle_mlp_amd import ops
ops.CONV_ARITH = os.environ.get("VG_CONV_ARITH", "fp16x3")
B = 128
def timeit(fn, n=15):
    for _ in range(60): fn()          # the clock takes tens of launches to settle after idle
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)
out = []
for (ci, co, h, s) in ((128, 256, 32, 2), (32, 128, 64, 2), (256, 256, 16, 2), (64, 128, 32, 2), (128, 256, 16, 2)):
    x = torch.randn(B, ci, h, h, device="cuda"); gy = torch.randn(B, co, h // s, h // s, device="cuda")
    out.append(f"{ci}->{co}@{h} {timeit(lambda: ops.conv5x5_wgrad(x, gy, s))*1e3:7.1f} us")
names = {900: "previous kernel (libabl_wx_900.so, built by hand from an older revision)"} if bits == 900 else {1: "gy cached", 2: "x loaded once", 4: "no split/store", 8: "no MFMA", 16: "no B reads", 32: "16x16x32 MFMA shape"}
print(f"abl {bits:3d} [{', '.join(v for k, v in names.items() if bits & k) or 'full'}]: " + " | ".join(out), flush=True)
