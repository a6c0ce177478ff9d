"""Ablation timings of the ring kernel (conv_ring.hip) on the dominant shapes: one process per one-off library built by
experiments/abl_build.sh with -DVG_RING_ABL=<bits> (1 no filter DMA, 2 no barrier, 4 no patch staging, 8 no counted
vmcnt wait, 16 no pixel-fragment reads, 32 no filter-fragment reads, 64 no MFMAs).  Usage: abl_ring.py <bits> ..."""
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
    _lib.LIB_PATH = os.path.join(ROOT, "experiments", "abl", f"libabl_ring_{bits}.so")
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
with ops.packed_filter_scope():
    for kind, ci, co, h in (("fwd", 128, 256, 32), ("tr", 256, 128, 16)):
        x = torch.randn(B, ci, h, h, device="cuda")
        w = 0.05 * torch.randn(*((co, ci, 5, 5) if kind == "fwd" else (ci, co, 5, 5)), device="cuda")
        fn = (lambda: ops.conv5x5_fwd(x, w, None, 2)) if kind == "fwd" else (lambda: ops.convT5x5_fwd(x, w, None, 2))
        out.append(f"{kind} {timeit(fn)*1e3:7.1f} us")
        if bits in (0, 512):     # full kernels (512: a build with extra defines): a structural check against the vendor convolution
            import torch.nn.functional as F
            ref = F.conv2d(x, w, None, 2, 2) if kind == "fwd" else F.conv_transpose2d(x, w, None, 2, 2, 1)
            err = float((fn() - ref).abs().max() / ref.abs().max())
            out[-1] += f" (vs vendor fp32 conv {err:.1e})"
            assert err < 1e-4, err
        # the same launch reading its input through a BatchNorm + ReLU (the forward launches of a fused chain)
        sc, sh = 0.5 + torch.rand(ci, device="cuda"), torch.randn(ci, device="cuda")
        fa = (lambda: ops.conv5x5_fwd(x, w, None, 2, in_affine=(sc, sh, 1))) if kind == "fwd" else (lambda: ops.convT5x5_fwd(x, w, None, 2, in_affine=(sc, sh, 1)))
        out.append(f"{kind}+BN {timeit(fa)*1e3:7.1f} us")
names = {1: "no DMA", 2: "no barrier", 4: "no staging", 8: "no vmcnt wait", 16: "no B reads", 32: "no A reads", 64: "no MFMA"}
print(f"abl {bits:3d} [{', '.join(v for k, v in names.items() if bits & k) or 'full'}]: " + " | ".join(out), flush=True)
