"""GPU: the data-parallel iteration with TWO ranks (one process per rank, both on cuda:0, gloo
transport: the 1-GPU box cannot host two RCCL ranks).  Exercises the real trainer code path
-- hooks, direct + bucketed all-reduce (SUM), BCE over the global batch, replica-local
BatchNorm -- and checks rank 0's exchanged gradients of all three phases against the oracle's
2-replica emulation (N replicas x B_local, per-replica BN, summed gradients)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import BN_SHADOWED

pytestmark = pytest.mark.gpu

WORLD, BATCH = 2, 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=WORLD)
        torch.cuda.set_device(0)
        from oracle import steps as osteps
        from disentangle_mlp_amd.trainer import BetaVAEGANTrainer
        b = osteps.synthetic_batch(BATCH)
        lo, hi = rank * BATCH // WORLD, (rank + 1) * BATCH // WORLD
        tr = BetaVAEGANTrainer(beta=25.0, lr=0.0)          # lr 0: every phase at the initial weights
        assert tr.dp and tr.world == WORLD
        got = {}
        out = tr.step(*(b[k][lo:hi].cuda() for k in ("data", "noise", "eps2", "eps3")),
                      grad_hook=lambda ph, net: got.__setitem__(
                          ph, {k: p.grad.detach().cpu().double() for k, p in net.named_parameters()}))
        torch.cuda.synchronize()
        worst, worst_k = 0.0, ""
        if rank == 0:
            torch.set_num_threads(8)
            ref = {}
            for r in range(WORLD):
                eg, d, oeg, od = osteps.build_nets(dtype=torch.float64)
                for o in (oeg, od):
                    o.param_groups[0]["lr"] = 0.0
                bb = osteps.synthetic_batch(BATCH, dtype=torch.float64)
                l, h = r * BATCH // WORLD, (r + 1) * BATCH // WORLD

                def hook(ph, net):
                    for k, p in net.named_parameters():
                        ref.setdefault(ph, {})
                        ref[ph][k] = ref[ph].get(k, 0) + p.grad.detach().clone()
                osteps.betavaegan_step(eg, d, oeg, od, bb["data"][l:h], bb["noise"][l:h], bb["eps2"][l:h],
                                       bb["eps3"][l:h], beta=25.0, bce_divisor=BATCH, grad_hook=hook)
            for ph, key in (("D", "d"), ("EG2", "eg"), ("EG3", "eg")):
                for k, r_ in ref[ph].items():
                    if k in BN_SHADOWED[key] or float(r_.norm()) == 0.0:
                        continue
                    e = float((got[ph][k] - r_).norm() / r_.norm())
                    if e > worst:
                        worst, worst_k = e, f"{ph}/{k}"
        # replicas stay in lock-step: same (reduced) gradients on both ranks
        g0 = got["EG3"]["deconv2.weight"].float().cuda()
        g1 = g0.clone()
        dist.broadcast(g1, src=0)
        lock = float((g0 - g1).abs().max())
        dist.barrier()
        if rank == 0:
            q.put(("ok", (worst, worst_k), lock))
        elif lock != 0.0:
            q.put(("lockstep", lock, 0.0))
        dist.destroy_process_group()
    except Exception as exc:      # surface the failure in the parent
        q.put(("error", repr(exc), 0.0))
        raise


def test_two_rank_step_matches_two_replica_oracle():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    status, worst, lock = q.get(timeout=600)
    for p in procs:
        p.join(120)
    assert status == "ok", (status, worst)
    assert all(p.exitcode == 0 for p in procs)
    assert lock == 0.0
    # 1e-2 relative L2 per tensor at a local batch of 16: one ReLU unit on the other side of zero
    # moves a gradient by ~1/sqrt(#units of its layer) (a BatchNorm1d layer has only 16 x 2048
    # units here); a wrong divisor, a missing rank or a mean-instead-of-sum reduction is O(1)
    assert worst[0] <= 1e-2, worst


def test_rccl_path_in_a_group_of_one():
    """The exchange code over the REAL backend the multi-GPU runs use ("nccl" = RCCL): a process group of one
    rank on this GPU, collectives forced on (FlatGrads.exchange_when_alone).  Every hook-launched in-place
    all-reduce, the bucketed flat buffer, the async handles and their stream waits run through RCCL; a sum over
    one rank changes nothing, so three iterations must reproduce the plain trainer bit for bit."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent("""
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, os.getcwd())
        from disentangle_mlp_amd import trainer as T
        from oracle import steps as osteps
        b = {k: v.cuda() for k, v in osteps.synthetic_batch(8).items()}
        ref = T.BetaVAEGANTrainer(beta=25.0)
        want = [{k: float(v) for k, v in ref.step(b["data"], b["noise"], b["eps2"], b["eps3"]).items()} for _ in range(3)]
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        T.FlatGrads.exchange_when_alone = True
        tr = T.BetaVAEGANTrainer(beta=25.0, data_parallel=True)
        got = [{k: float(v) for k, v in tr.step(b["data"], b["noise"], b["eps2"], b["eps3"]).items()} for _ in range(3)]
        torch.cuda.synchronize()
        assert got == want, (got, want)
        for (k, v), (_, r) in zip(tr.netEG.state_dict().items(), ref.netEG.state_dict().items()):
            assert torch.equal(v, r), k
        dist.destroy_process_group()
        print("rccl-one-rank ok")
    """)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "rccl-one-rank ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
