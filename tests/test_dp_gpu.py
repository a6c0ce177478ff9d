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
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "rccl-one-rank ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_captured_dp_iteration_equals_eager_dp():
    """The data-parallel iteration as ONE HIP graph (round 4): gradient exchange on our own RCCL communicator
    (rccl.py: ncclAllReduce on a side stream forked from / joined to the compute stream -- graph nodes and edges inside
    the capture; c10d's collectives abort there, profiles/r03_logs/r3_dpgraph.log).  One rank over RCCL on this GPU,
    exchange forced on: six iterations of a captured data-parallel trainer against an eager data-parallel one -- every
    loss and every weight bit for bit -- with label flips between replays; the exchange's byte / collective counters
    advance per replay as in eager iterations."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent("""
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, os.getcwd())
        from disentangle_mlp_amd import trainer as T, rccl
        from oracle import steps as osteps
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        T.FlatGrads.exchange_when_alone = True
        b = {k: v.cuda() for k, v in osteps.synthetic_batch(8).items()}
        eager = T.BetaVAEGANTrainer(beta=25.0, data_parallel=True, graph=False)
        cap = T.BetaVAEGANTrainer(beta=25.0, data_parallel=True, graph=True)
        assert eager.flat_d.capturable and cap.flat_eg.capturable and cap.graph and not eager.graph
        labels = [(0.9, 0.1), (0.9, 0.1), (0.9, 0.1), (0.1, 0.1), (0.9, 0.9), (0.9, 0.1)]
        for it, (rl, fl) in enumerate(labels):
            want = {k: float(v) for k, v in eager.step(b["data"], b["noise"], b["eps2"], b["eps3"], real_label=rl, fake_label=fl).items()}
            got = {k: float(v) for k, v in cap.step(b["data"], b["noise"], b["eps2"], b["eps3"], real_label=rl, fake_label=fl).items()}
            assert got == want, (it, got, want)
        assert len(cap._graphs) == 1, "the data-parallel iteration was not captured"
        torch.cuda.synchronize()
        for n in ("netEG", "netD"):
            for (k, v), (_, r) in zip(getattr(cap, n).state_dict().items(), getattr(eager, n).state_dict().items()):
                assert torch.equal(v, r), (n, k)
        for f, g in ((cap.flat_d, eager.flat_d), (cap.flat_eg, eager.flat_eg)):
            assert (f.bytes_reduced, f.collectives) == (g.bytes_reduced, g.collectives) and f.collectives > 0
        rccl.shutdown()
        dist.destroy_process_group()
        print("captured-dp ok")
    """)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "captured-dp ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def _gan_worker(rank, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=WORLD)
        torch.cuda.set_device(0)
        import oracle
        from oracle import steps as osteps
        from disentangle_mlp_amd.trainer import GANTrainer
        batch = 16
        b = osteps.synthetic_batch(batch)
        lo, hi = rank * batch // WORLD, (rank + 1) * batch // WORLD
        tr = GANTrainer(lr=0.0)
        assert tr.dp and tr.flat_g is not None and tr.flat_d is not None
        got = {}
        tr.step(b["data"][lo:hi].cuda(), b["noise"][lo:hi].cuda(),
                grad_hook=lambda ph, net: got.__setitem__(
                    ph, {k: p.grad.detach().cpu().double() for k, p in net.named_parameters()}))
        torch.cuda.synchronize()
        worst = (0.0, "")
        if rank == 0:
            torch.set_num_threads(8)
            ref = {}
            for r in range(WORLD):
                torch.manual_seed(999)
                ng, nd = oracle.Generator_celeba(oracle.OracleOpt()), oracle.Discriminator_celeba(oracle.OracleOpt())
                ng.apply(oracle.weights_init), nd.apply(oracle.weights_init)
                ng, nd = ng.double(), nd.double()
                og, od = torch.optim.Adam(ng.parameters(), lr=0.0), torch.optim.Adam(nd.parameters(), lr=0.0)
                bb = osteps.synthetic_batch(batch, dtype=torch.float64)
                l, h = r * batch // WORLD, (r + 1) * batch // WORLD

                def hook(ph, net):
                    for k, p in net.named_parameters():
                        ref.setdefault(ph, {})
                        ref[ph][k] = ref[ph].get(k, 0) + p.grad.detach().clone()
                osteps.gan_step(ng, nd, og, od, bb["data"][l:h], bb["noise"][l:h], bce_divisor=batch, grad_hook=hook)
            for ph, key in (("D", "d"), ("G", "g")):
                for k, r_ in ref[ph].items():
                    if k in BN_SHADOWED[key] or float(r_.norm()) == 0.0:
                        continue
                    worst = max(worst, (float((got[ph][k] - r_).norm() / r_.norm()), f"{ph}/{k}"))
        dist.barrier()
        if rank == 0:
            q.put(("ok", worst))
        dist.destroy_process_group()
    except Exception as exc:
        q.put(("error", repr(exc)))
        raise


def test_two_rank_gan_step_matches_two_replica_oracle():
    """new_gan.py:51-53 wraps both networks in nn.DataParallel: GANTrainer exchanges the gradients of D and of G
    (one all-reduce SUM per optimizer step, BCE over the global batch); rank 0's exchanged gradients vs the
    oracle's 2-replica emulation."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gan_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    status, worst = q.get(timeout=600)
    for p in procs:
        p.join(120)
    assert status == "ok", (status, worst)
    assert all(p.exitcode == 0 for p in procs)
    assert worst[0] <= 1e-2, worst


def test_bench_launches_two_ranks_on_this_gpu():
    """`python bench.py --gpus 2` starts its two ranks itself (no torchrun around it) and reports the whole-job
    rate: here both ranks share the one GPU of the box with gloo as the transport (a one-GPU box cannot host two
    RCCL ranks), per-GPU batch 16, 2 timed steps."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(VG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "16", "--no-cpu-baseline", "--no-opt-in"], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2" and res["config"]["global_batch"] == 32
    assert res["steps"] == 2 and res["losses_finite"] and res["value"] > 0
    assert abs(res["value"] - 32 * 2 / (res["ms_per_step"] * 2e-3)) <= 0.01 * res["value"]     # whole-job images/s
    # the line explains its own exchange: ranks the transport connected, payload per iteration (D once + EG twice:
    # (36 122 945 + 2 x 73 385 795) fp32 parameters), time the compute stream stood still for it
    dp = res["data_parallel"]
    assert dp["ranks_counted_by_all_reduce"] == 2 and dp["backend"] == "gloo"
    assert dp["bytes_all_reduced_per_step"] == 4 * (36122945 + 2 * 73385795)
    assert dp["exposed_comm_ms_per_step"] >= 0 and dp["collectives_per_step"] >= 3
    assert 0 < res["conv_path_frac_of_own_roof"] < res["conv_path_frac_of_fp32_mfma_peak"]
