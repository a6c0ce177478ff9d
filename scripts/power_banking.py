"""Does low-power time inside the iteration buy the MFMA kernels a higher clock?  The captured iteration (HIP graph replay,
B = 128) is followed by an artificial low-power kernel of duration d -- a one-thread spin (torch.cuda._sleep) or a plain
HBM copy -- and the iteration time is measured with it.  If the chip were not power / thermally limited the time would grow
by exactly d; the shortfall is what the idle (or memory-bound) interval gave back to the convolutions."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer

B = 128
tr = BetaVAEGANTrainer(beta=25.0)
g = torch.Generator().manual_seed(1234)
data = (torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).cuda()
lat = [torch.randn(B, 128, generator=g).cuda() for _ in range(3)]
for _ in range(8):
    tr.step(data, *lat)
assert tr._graphs
src = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")      # 768 MiB: read + write = 1.6 GB per copy
dst = torch.empty_like(src)


def measure(extra, n=40):
    for _ in range(5):
        tr.step(data, *lat); extra()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(data, *lat); extra()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def timed(extra, n=20):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        extra()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


base = measure(lambda: None)
print(f"iteration alone: {base:.3f} ms", flush=True)
for cycles in (250_000, 500_000, 1_000_000, 2_000_000, 4_000_000):
    f = lambda: torch.cuda._sleep(cycles)
    d = timed(f)
    t = measure(f)
    print(f"+ spin {d:6.3f} ms: iteration {t:.3f} ms -> the rest took {t - d:.3f} ms ({t - d - base:+.3f} vs alone)", flush=True)
for k in (1, 2, 4):
    def f(k=k):
        for _ in range(k):
            dst.copy_(src)
    d = timed(f)
    t = measure(f)
    print(f"+ copy {d:6.3f} ms: iteration {t:.3f} ms -> the rest took {t - d:.3f} ms ({t - d - base:+.3f} vs alone)", flush=True)
base2 = measure(lambda: None)
print(f"iteration alone again: {base2:.3f} ms", flush=True)
