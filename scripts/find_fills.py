"""Which Python lines launch the small ATen kernels of an iteration (fills, adds, copies, sums)?  torch.profiler with stacks
over one step of the benchmark's trainer."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer
tr = BetaVAEGANTrainer(device="cuda", seed=1, beta=25.0)
x = torch.rand(128, 3, 64, 64, device="cuda") * 2 - 1
for _ in range(3): tr.step(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    tr.step(x)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::ones_like", "aten::add_", "aten::add", "aten::copy_", "aten::sum", "aten::mul", "aten::clone"):
        st = [f for f in (ev.stack or []) if "disentangle_mlp_amd" in f or "bench" in f]
        key = (ev.name, st[0].split("/")[-1] if st else "<autograd engine / torch internals>", str(ev.input_shapes)[:60])
        cnt[key] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:45]:
    print(v, k)
