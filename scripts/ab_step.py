"""Same-process A/B of a vg_debug_set_wgrad switch on the full iteration (boxes differ by a few %,
so alternatives are compared interleaved on one box): python scripts/ab_step.py WHAT A B"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import _lib
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer

what, va, vb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lib = _lib.use_tuning().__enter__()      # the vg_debug_* knobs live in the tuning build only
tr = BetaVAEGANTrainer(beta=25.0)
x = (torch.rand(128, 3, 64, 64) * 2 - 1).cuda()
for _ in range(5):
    tr.step(x)
res = {va: [], vb: []}
for rnd in range(4):
    for v in (va, vb):
        lib.vg_debug_set_wgrad(what, v)
        tr.step(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.step(x)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 10 * 1e3)
for v, ts in res.items():
    print(f"wgrad switch {what} = {v}: " + " ".join(f"{t:.2f}" for t in ts) + f"  -> min {min(ts):.2f} ms, mean {sum(ts)/len(ts):.2f} ms")
