"""One-iteration-ahead tracking: along the oracle's own trajectory (fp32, 16 threads, lr 1e-3, B = 16), before every
iteration the oracle's complete state (weights, BatchNorm buffers, both Adam states) is copied into (a) the oracle run with
1 thread, (b) the oracle in fp64, (c) the HIP trainer; all four then run the SAME iteration from the SAME state.  How far do
beta*KL and the reconstruction errors of that one iteration spread?"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import steps as osteps
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer

N, B = 5, 16
g = torch.Generator().manual_seed(7)
base = torch.randn(4 * B, 3, 8, 8, generator=g)
data = torch.tanh(torch.nn.functional.interpolate(base, size=64, mode="bilinear"))
rnd = [[torch.randn(B, 128, generator=g) for _ in range(3)] for _ in range(N)]


def ck(eg, d, oeg, od):
    return copy.deepcopy({"epoch": 0, "encoder_decoder_model": eg.state_dict(), "discriminator_model": d.state_dict(),
                          "encoder_decoder_optimizer": oeg.state_dict(), "discriminator_optimizer": od.state_dict()})


def oracle_from(state, dtype, threads, x, lat):
    torch.set_num_threads(threads)
    eg, d, oeg, od = osteps.build_nets(dtype=dtype)
    eg.load_state_dict(state["encoder_decoder_model"]); d.load_state_dict(state["discriminator_model"])
    oeg.load_state_dict(copy.deepcopy(state["encoder_decoder_optimizer"])); od.load_state_dict(copy.deepcopy(state["discriminator_optimizer"]))
    out = osteps.betavaegan_step(eg, d, oeg, od, x.to(dtype), *[t.to(dtype) for t in lat], beta=25.0)
    torch.set_num_threads(16)
    return out


torch.set_num_threads(16)
eg, d, oeg, od = osteps.build_nets()
tr = BetaVAEGANTrainer(beta=25.0, graph=False)
for it in range(N):
    x = data[(it % 4) * B:(it % 4 + 1) * B]
    state = ck(eg, d, oeg, od)
    a = oracle_from(state, torch.float32, 1, x, rnd[it])
    b = oracle_from(state, torch.float64, 16, x, rnd[it])
    tr.load(copy.deepcopy(state))
    c = {k: float(v) for k, v in tr.step(x.cuda(), *[t.cuda() for t in rnd[it]]).items()}
    ref = osteps.betavaegan_step(eg, d, oeg, od, x, *rnd[it], beta=25.0)        # advances the trajectory
    for k in ("kld", "mse_enc", "mse_dec", "errG_recon", "sim"):
        print(f"it {it} {k:10s} oracle16 {ref[k]:14.2f} | 1thr {a[k] / ref[k] - 1:+.4f}  fp64 {b[k] / ref[k] - 1:+.4f}  HIP {c[k] / ref[k] - 1:+.4f}", flush=True)
    print(f"it {it} D_x        oracle16 {ref['D_x']:.5f} | 1thr {a['D_x'] - ref['D_x']:+.5f} fp64 {b['D_x'] - ref['D_x']:+.5f} HIP {c['D_x_sum'] / B - ref['D_x']:+.5f}", flush=True)
