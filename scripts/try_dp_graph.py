"""Experiment: can the data-parallel iteration -- RCCL all-reduces launched from autograd hooks on the collective stream --
be captured in the HIP graph too?  A process group of ONE rank over the real backend ("nccl" = RCCL) on this GPU, collectives
forced on (FlatGrads.exchange_when_alone), graph replay forced on for the data-parallel trainer; compared with the eager
data-parallel trainer bit for bit, and timed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
import torch, torch.distributed as dist
from disentangle_mlp_amd import trainer as T

dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
T.FlatGrads.exchange_when_alone = True
B = int(os.environ.get("B", "128"))
g = torch.Generator().manual_seed(0)
x = (torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).cuda()
lat = [torch.randn(B, 128, generator=g).cuda() for _ in range(3)]


def run(graph, n=12):
    tr = T.BetaVAEGANTrainer(beta=25.0, data_parallel=True, capturable=True)
    tr.graph = graph                       # (the constructor switches it off under data parallelism)
    tr._graphs, tr._shape_steps = {}, {}
    outs = [{k: v.clone() for k, v in tr.step(x, *lat).items()} for _ in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.step(x, *lat)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    return outs, {k: v.clone() for k, v in tr.netEG.state_dict().items()}, ms, bool(tr._graphs)


e_out, e_sd, e_ms, _ = run(False)
print(f"eager DP (1-rank RCCL): {e_ms:.2f} ms/step", flush=True)
g_out, g_sd, g_ms, captured = run(True)
print(f"graph DP (1-rank RCCL): {g_ms:.2f} ms/step, captured={captured}", flush=True)
same = all(torch.equal(a[k], b[k]) for a, b in zip(e_out, g_out) for k in a) and all(torch.equal(v, g_sd[k]) for k, v in e_sd.items())
print("bit-identical:", same, flush=True)
dist.destroy_process_group()
