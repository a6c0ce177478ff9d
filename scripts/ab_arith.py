"""Same-process A/B of the convolution arithmetic on the full iteration: exact fp32 (default) vs the
opt-in bf16x3 mode (forward / transposed kernels and their use as data gradients; weight gradient
stays fp32).  Also reports how far one iteration's losses move."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = (torch.rand(B, 3, 64, 64) * 2 - 1).cuda()
n = [torch.randn(B, 128).cuda() for _ in range(3)]
first = {}
MODES = ("fp32", "bf16x6", "bf16x3")
for mode in MODES:
    ops.CONV_ARITH = mode
    tr = BetaVAEGANTrainer(beta=25.0)
    out = tr.step(x, *n)
    first[mode] = {k: float(v) for k, v in out.items()}
ops.CONV_ARITH = "fp32"
for k in first["fp32"]:
    a = first["fp32"][k]
    print(f"{k:12s} fp32 {a:14.6f}  " + "  ".join(f"{m} rel {abs(a-first[m][k])/max(abs(a),1e-30):.2e}" for m in MODES[1:]))
tr = BetaVAEGANTrainer(beta=25.0)
for _ in range(5):
    tr.step(x)
res = {m: [] for m in MODES}
for rnd in range(4):
    for mode in MODES:
        ops.CONV_ARITH = mode
        tr.step(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.step(x)
        torch.cuda.synchronize()
        res[mode].append((time.perf_counter() - t0) / 10 * 1e3)
ops.CONV_ARITH = "fp32"
for mode, ts in res.items():
    print(f"{mode:7s}: " + " ".join(f"{t:.2f}" for t in ts) + f"  -> mean {sum(ts)/len(ts):.2f} ms/step = {B/(sum(ts)/len(ts))*1e3:.0f} images/s")
