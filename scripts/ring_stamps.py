"""Where a ring-kernel wavefront's cycles go (diagnostic build of conv_ring.hip with -DVG_RING_STAMP: three s_memtime stamps
per K step, summed per wavefront, returned through the statistics buffer).  Usage: ring_stamps.py <libabl_ring_XXXX.so> [arith]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from disentangle_mlp_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "experiments", "abl", sys.argv[1])
from disentangle_mlp_amd import ops
ops.CONV_ARITH = sys.argv[2] if len(sys.argv) > 2 else "fp16x3"
B = 128
with ops.packed_filter_scope():
    for kind, ci, co, h in (("fwd", 128, 256, 32), ("tr", 256, 128, 16)):
        x = torch.randn(B, ci, h, h, device="cuda")
        w = 0.05 * torch.randn(*((co, ci, 5, 5) if kind == "fwd" else (ci, co, 5, 5)), device="cuda")
        fn = (lambda: ops.conv5x5_fwd(x, w, None, 2, want_stats=True)) if kind == "fwd" else (lambda: ops.convT5x5_fwd(x, w, None, 2, want_stats=True))
        for _ in range(30):
            y, st = fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y, st = fn(); b.record(); torch.cuda.synchronize()
        if st is None:
            print(kind, ci, co, h, "no stats buffer (K-split layer)"); continue
        t = st.view(torch.int64)[:64 * 8 * 4].view(64, 8, 4).cpu().double()
        steps = t[..., 3]
        per = t[..., :3] / steps.unsqueeze(-1)
        print(f"{kind} {ci}->{co} @{h}: {a.elapsed_time(b)*1e3:.1f} us, steps/wave {steps.mean():.0f}; cycles per step: "
              f"barrier {per[...,0].mean():.0f} (min {per[...,0].min():.0f} max {per[...,0].max():.0f}), "
              f"MFMA phase {per[...,1].mean():.0f} (min {per[...,1].min():.0f} max {per[...,1].max():.0f}), "
              f"tail waits {per[...,2].mean():.0f} (max {per[...,2].max():.0f}); by wave (barrier/mfma/tail): "
              + " ".join(f"w{i}:{per[:, i, 0].mean():.0f}/{per[:, i, 1].mean():.0f}/{per[:, i, 2].mean():.0f}" for i in range(8)), flush=True)
        # the last step of workgroups 0..2: issue times of its MFMAs relative to the earliest barrier exit in the workgroup
        tl = st.view(torch.int64)[64 * 8 * 4:64 * 8 * 4 + 64 * 8 * 16].view(64, 8, 16).cpu()
        for wg in range(2):
            t0 = int(tl[wg, :, 1].min())
            for w in range(8):
                r = tl[wg, w]
                print(f"   wg{wg} w{w}: top {int(r[0]) - t0:5d} bar {int(r[1]) - t0:5d} | mfma " + " ".join(f"{int(v) - t0:5d}" for v in r[4:16] if int(v)) + f" | end {int(r[2]) - t0:5d} waited {int(r[3]) - t0:5d}")
