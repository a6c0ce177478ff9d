#!/usr/bin/env python3
"""Headline benchmark: CelebA 64x64 beta-VAE-GAN training iteration (encoder + decoder +
discriminator, three optimizer steps; experiments/new_betavaegan.py:87-193 of the
reference) in images/s on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 128]

One process per GPU; per-GPU batch 128 (BASELINE.json configs[1] at N=1; configs[2]
= global batch 1024 at N=8: weak scaling).  ``--gpus N`` with N > 1 STARTS the N ranks
itself: before anything touches the GPU this process launches
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
bench.py --gpus N ...`` as a child, relays rank 0's JSON line and exits with the child's
status (replaces nn.DataParallel of new_betavaegan.py:42,44).  When the driver has already
launched the ranks (WORLD_SIZE in the environment) the process is one of them.
Synthetic data (U(-1,1) images, N(0,1) noise; each rank its own shard) resident in HBM
before the timed region; weights from the reference's seed recipe.  Prints ONE JSON line
on rank 0.

On one GPU the timed steps are replays of ONE HIP graph of the whole iteration (trainer._CapturedIteration: captured
during untimed preparation, bit-identical to eager stepping; ``launch_mode`` in the line says which it was; VG_GRAPH=0
forces eager).  Launches inside a replayed graph cannot be bracketed by events, so the dominant kernel of ``roofline`` is
timed in an eager leg of the same K iterations right after the timed region (``roofline.launches_timed_in``).  Under
N > 1 the iteration stays eager (the gradient exchange is launched from autograd hooks) and the line carries
``data_parallel``: ranks the transport connected, bytes all-reduced and time the compute stream waited per step.

``--rehearse-launch``: the same launch path without a GPU -- every rank joins a gloo group,
the world is checked with an all-reduce and rank 0 prints a line marked "rehearsal" (no
throughput is measured or reported).  ``VG_DIST_BACKEND=gloo`` runs the real benchmark with
gloo as the transport (ranks may then share one GPU: a 2-rank rehearsal on a one-GPU box).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak (spec)
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 / f16 MFMA (spec); fp16x3 issues 3 of them per fp32 multiply, bf16x6 6, bf16x3 3
ALG_GFLOP_PER_IMAGE = 20.075       # BASELINE.md section 2 (necessary passes only)
ALG_CONV_GFLOP_PER_IMAGE = 18.488
MFMAS_PER_PRODUCT = {"fp16x3": 3, "bf16x6": 6, "bf16x3": 3}
GRAPH_PREP_STEPS = 3               # trainer.GRAPH_WARM_STEPS eager iterations + the one that captures and replays
ARITH_NOTE = {
    "fp16x3": "fp32-equivalent: every fp32 operand, scaled by an exact power of two from a device-side bound of the "
              "tensor's largest magnitude, split into fp16 hi + lo (11 + 11 significand bits, residual <= 2^-24), "
              "3 f16 MFMAs per multiply (hi*hi, hi*lo, lo*hi), fp32 accumulate; conv rel. error vs fp64 4e-7..6e-7 "
              "(bf16x6: 4e-7..9e-7; the fp32-input MFMA: 5e-7..1e-6); the 3-channel edge layers in bf16x6",
    "fp32": "exact fp32-input MFMA (v_mfma_f32_32x32x2_f32)",
    "bf16x6": "fp32-equivalent: every fp32 operand split exactly into 3 bf16 planes (8+8+8 mantissa bits), 6 bf16 "
              "MFMAs per multiply (all plane pairs whose index sum < 3), fp32 accumulate; conv rel. error vs fp64 "
              "4e-7..9e-7 (the fp32-input MFMA: 5e-7..1e-6)",
    "bf16x3": "operands split into 2 bf16 planes (hi/lo), 3 bf16 MFMAs per multiply, fp32 accumulate; 4.5e-6 per "
              "convolution",
}


def conv_flops(key):
    """Algorithmic FLOPs of one launch: 2 * MACs (5x5 taps; transposed conv counts its
    real taps, i.e. the same MACs as the convolution it transposes)."""
    op, B, Cin, H, W, Cout, s = key
    if op == "convT_fwd":        # x (B,Cin,H,W) -> (B,Cout,sH,sW): every input pixel meets 25 taps
        return 2.0 * B * H * W * Cin * Cout * 25
    oh, ow = (H - 1) // s + 1, (W - 1) // s + 1
    return 2.0 * B * oh * ow * Cin * Cout * 25   # conv_fwd and conv_wgrad


def pmc_traffic(dominant, arith):
    """HBM bytes per launch of the dominant kernel.  Hardware counters cannot be read from inside
    the process being measured: they come from separate `rocprofv3 --pmc` passes over THIS command
    (scripts/pmc_bench.sh: FETCH_SIZE x2 -- the gfx950 correction of MI355X_MICROARCH.md section HBM --
    and WRITE_SIZE, one pass each), whose per-launch averages are committed under profiles/.  The
    figure is reported only when a committed pass exists for exactly this launch shape and arithmetic."""
    pdir = os.path.join(ROOT, "profiles")
    try:
        names = sorted(f for f in os.listdir(pdir) if f.endswith(".json") and "_pmc_" in f)
    except OSError:
        return None, None
    for name in reversed(names):                       # newest round first
        try:
            with open(os.path.join(pdir, name)) as f:
                pmc = json.load(f)
            L = pmc["launch"]
            key = (L["op"], L["B"], L["Cin"], L["H"], L["W"], L["Cout"], L["stride"])
        except (OSError, KeyError, ValueError):
            continue
        if tuple(dominant) == key and pmc.get("arith", "fp32") == arith:
            return pmc["hbm_bytes_per_launch"], "profiles/" + name
    return None, None


def measured_mfma_rate(arith="bf16x6"):
    """16-bit MFMA TFLOP/s of a bare register-operand loop of the instruction the kernels use (v_mfma_f32_32x32x16_f16
    for fp16x3, _bf16 otherwise) on RANDOM operands, as measured on an MI355X of this pool by scripts/mfma_f16.hip /
    scripts/mfma_peak.hip (committed: profiles/*_mfma_f16.jsonl, *_mfma_peak.jsonl): what the matrix pipe delivers at the
    clock the chip holds under that load.  Informational -- `roofline.peak` stays the nominal dense peak."""
    pdir = os.path.join(ROOT, "profiles")
    f16 = arith == "fp16x3"
    try:
        names = sorted(f for f in os.listdir(pdir) if f.endswith("_mfma_f16.jsonl" if f16 else "_mfma_peak.jsonl"))
    except OSError:
        return None, None
    for name in reversed(names):
        try:
            with open(os.path.join(pdir, name)) as f:
                rows = [json.loads(l) for l in f if l.strip()]
            if f16:
                best = max(r["tflops"] for r in rows if r.get("operands") == "random" and r.get("mfma") == "32x32x16_f16")
            else:
                best = max(r["tflops"] for r in rows if r.get("operands") == "random" and r.get("shape") == "32x32x16")
            return best, "profiles/" + name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def _median_time(fn, warm, timed):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def cpu_baseline(batch, beta):
    """The oracle (CPU restatement of the reference's path) timed on this host's cores on a bounded
    sample (SURVEY.md section 8d protocol): 1 warm-up + 5 timed iterations, median -- the beta-VAE-GAN
    iteration at the benchmark's per-GPU batch (BASELINE config 2) and the `new_vae` iteration at
    batch 16 (config 1, the reference's own CPU-runnable case)."""
    import torch
    from oracle import steps as osteps
    # the GPU box gives one job a 16-core share of the host; more threads only oversubscribe
    threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 16)
    torch.set_num_threads(threads)
    eg, d, oeg, od = osteps.build_nets()
    b = osteps.synthetic_batch(batch)
    med, ts = _median_time(lambda: osteps.betavaegan_step(eg, d, oeg, od, b["data"], b["noise"], b["eps2"], b["eps3"],
                                                          beta=beta), 1, 5)
    vae, _, ovae, _ = osteps.build_nets()
    b16 = osteps.synthetic_batch(16)
    med_v, ts_v = _median_time(lambda: osteps.vae_step(vae, ovae, b16["data"], b16["eps2"], beta=1.0), 1, 5)
    return {"value": round(batch / med, 3), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"beta-VAE-GAN iteration at batch {batch}: 1 warm-up + 5 timed iterations, median {med:.3f} s "
                      f"(min {min(ts):.3f}, max {max(ts):.3f}); torch CPU fp32, {threads} threads",
            "config1_new_vae_b16": {"value": round(16 / med_v, 2), "unit": "images/s", "s_per_iteration": round(med_v, 4),
                                    "sample": f"new_vae.py beta=1 VAE iteration at batch 16, 1 warm-up + 5 timed, median "
                                              f"(min {min(ts_v):.3f} s, max {max(ts_v):.3f} s)"}}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """Start `args.gpus` ranks as children (this process has not touched the GPU and never will),
    relay their output and return the exit status."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def rehearse(args, world, rank):
    """Launch-path rehearsal on CPU ranks (gloo): proves that N ranks start, rendezvous and agree on
    the world; measures nothing."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.ones(1)
    if world > 1:
        dist.all_reduce(t)
    assert int(t.item()) == world == args.gpus, (t.item(), world, args.gpus)
    if rank == 0:
        print(json.dumps({"rehearsal": True, "note": "launch path only (gloo, CPU ranks): nothing was measured",
                          "n_gpus": world, "ranks_counted_by_all_reduce": int(t.item()),
                          "config": {"parallelism": "dp%d" % world, "global_batch": args.batch * world}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--beta", type=float, default=25.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-opt-in", action="store_true", help="skip the informational legs in the other arithmetics")
    ap.add_argument("--rehearse-launch", action="store_true", help="exercise the N-rank launch on CPU (gloo); no measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))       # nothing above touched the GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.rehearse_launch:
        return rehearse(args, world, rank)

    import torch
    import torch.distributed as dist

    if world > 1:
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))   # (rehearsals: ranks may share a GPU)
        backend = os.environ.get("VG_DIST_BACKEND", "nccl")     # "nccl" IS RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
        if os.environ.get("VG_DP_ALONE", "0") == "1":
            # one GPU, the data-parallel code path: a process group of ONE rank over RCCL, the gradient exchange forced on
            # (every all-reduce runs, over one rank) -- what a one-GPU box can show of the N > 1 iteration: that it is
            # captured and replayed like the single-GPU one (`data_parallel.launch_mode`) and what it costs next to it
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", torch.cuda.current_device())

    from disentangle_mlp_amd import ops
    from disentangle_mlp_amd.trainer import BetaVAEGANTrainer

    B = args.batch
    arith = ops.CONV_ARITH               # the product default unless VG_CONV_ARITH was exported
    linear_own = bool(ops.LINEAR_SPLIT and arith == "fp16x3")     # ops.linear_split_ok: the big Linear layers leave the vendor GEMMs
    alone = world == 1 and dist.is_initialized()
    if alone:
        from disentangle_mlp_amd.trainer import FlatGrads
        FlatGrads.exchange_when_alone = True
    tr = BetaVAEGANTrainer(device=dev, seed=999, beta=args.beta, data_parallel=True if alone else None)
    ck0 = tr.checkpoint(0)               # the initial state (the reference's seed recipe), cloned: reloaded before the warm-up
    ck0 = {k: ({n: t.clone() for n, t in v.items()} if k.endswith("_model") else __import__("copy").deepcopy(v))
           for k, v in ck0.items()}
    g = torch.Generator().manual_seed(1234 + rank)            # each rank its own shard
    data = (torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).to(dev)
    noise = [torch.randn(B, 128, generator=g).to(dev) for _ in range(3)]

    def one_step():
        return tr.step(data, noise[0], noise[1], noise[2], real_label=0.9, fake_label=0.1)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_steps(n):
        """n iterations; wall time of all of them + the per-iteration GPU times between HIP events
        recorded on the launch stream (no synchronisation inside the loop)."""
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        fence()
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(n):
            o = one_step()
            marks[i + 1].record()
        timed_steps.host_ms = (time.perf_counter() - t0) * 1e3 / n      # host time to ENQUEUE a step (no waiting)
        fence()
        wall = time.perf_counter() - t0
        return wall, [marks[i].elapsed_time(marks[i + 1]) for i in range(n)], o

    # ---- untimed preparation: one instrumented (eager) iteration times every convolution launch to find the dominant
    # one; a single-process trainer captures the iteration in a HIP graph on its third iteration of this shape
    # (trainer.GRAPH_WARM_STEPS eager ones first) -- all of that happens here, before the W warm-up steps
    prep_steps = GRAPH_PREP_STEPS if tr.graph else 1
    for _ in range(prep_steps):
        one_step()
    graphed = bool(tr.graph and tr._graphs)
    ops.start_timing()                              # (an instrumented iteration is always eager)
    one_step()
    prep_steps += 1
    per_key = ops.stop_timing()
    totals = {k: sum(v) for k, v in per_key.items()}
    dominant = max(totals, key=totals.get) if totals else None
    conv_ms_profiled = sum(totals.values())
    # The preparation above has already trained for prep_steps iterations on this one synthetic batch, and on it the
    # discriminator saturates within a few iterations (the fake path's gradients then underflow and their launches
    # run on near-zero operands, which the chip clocks higher): go back to the initial weights and optimizer state, in
    # place (the captured graph stays valid), so that warm-up and timed steps are iterations 0 .. W + K - 1 of training.
    tr.load_in_place(ck0)
    for i in range(args.warmup):
        one_step()

    # ---- timed region: exactly K steps (graph replays when captured, eager launches otherwise)
    flats = [f for f in (tr.flat_d, tr.flat_eg) if f is not None]
    for f in flats:
        f.reset_stats()
        f.time_finish = True
    if not graphed:
        ops.start_timing(only=dominant)             # eager launches: the dominant kernel is bracketed in place
    elapsed, step_ms, out = timed_steps(args.steps)
    host_ms = timed_steps.host_ms
    dom_ms = ops.stop_timing().get(dominant, []) if not graphed else []
    out = {k: v.clone() for k, v in out.items()}
    comm = None
    if world > 1 or alone:
        ranks = torch.ones(1, device=dev)
        dist.all_reduce(ranks)                  # how many ranks the transport really connects
        comm = {"ranks_counted_by_all_reduce": int(ranks.item()),
                "bytes_all_reduced_per_step": sum(f.bytes_reduced for f in flats) // args.steps,
                "collectives_per_step": sum(f.collectives for f in flats) / args.steps,
                # measured around the waits of FlatGrads.finish() in eager iterations; inside a replayed graph the join
                # is a graph edge and cannot be bracketed by events
                "exposed_comm_ms_per_step": None if graphed else round(sum(f.exposed_ms() for f in flats) / args.steps, 3),
                "backend": dist.get_backend(),
                "transport": "own RCCL communicator (rccl.py), all-reduces on a forked side stream"
                             if all(f.capturable for f in flats) else "torch.distributed async all-reduce",
                "launch_mode": "HIP graph replay (the all-reduces are nodes of the captured iteration)" if graphed
                               else "eager (collectives launched from autograd hooks)"}
    for f in flats:
        f.time_finish = False
    # what regime the last timed step ran in: mean D(x), and how much of the gradient that D sends back into the decoder
    # through the generated batch is exactly zero (one probed -- eager -- iteration right after the timed region)
    regime = {"D_x_mean_last_timed_step": round(float(out["D_x_sum"]) / B, 6),
              "iterations_since_initial_weights": args.warmup + args.steps}
    def _probe(name, gten):
        regime["fake_path_gy_zero_fraction"] = round(float((gten == 0).float().mean()), 6)
        regime["fake_path_gy_absmax"] = float(gten.abs().max())
    tr.probe = _probe
    one_step()
    tr.probe = None

    # ---- launches inside a replayed graph cannot be bracketed by events: the dominant kernel's K x n launches are timed
    # in an eager leg of the same K iterations right after the timed region (same process, same kernels, same inputs)
    dom_leg = "timed region"
    if graphed:
        ops.start_timing(only=dominant)
        timed_steps(args.steps)
        dom_ms = ops.stop_timing().get(dominant, [])
        dom_leg = "eager leg of the same K iterations right after the timed region (launches of a replayed HIP graph cannot be bracketed by events)"

    # ---- informational: the same K steps in the other arithmetics (N = 1 only; NOT `value`)
    other = None
    if world == 1 and not args.no_opt_in:
        other = {}
        try:
            for mode in ("fp32", "bf16x6", "bf16x3", "fp16x3"):
                if mode == arith:
                    continue
                ops.CONV_ARITH = mode
                for _ in range(1 + (GRAPH_PREP_STEPS if tr.graph else 1)):     # captures this arithmetic's iteration too
                    one_step()
                e2, ms2, out2 = timed_steps(args.steps)
                other[mode] = {"value": round(B * args.steps / e2, 2), "unit": "images/s",
                               "ms_per_step": round(e2 / args.steps * 1e3, 3),
                               "ms_per_step_median": round(statistics.median(ms2), 3), "arithmetic": ARITH_NOTE[mode],
                               "losses_finite": all(bool(torch.isfinite(v).all()) for v in out2.values())}
        finally:
            ops.CONV_ARITH = arith
        other["note"] = "ops.CONV_ARITH / VG_CONV_ARITH; informational, not the headline `value`"

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    finite = all(bool(torch.isfinite(v).all()) for v in out.values())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        roof = None
        if dominant and dom_ms:
            avg_ms = sum(dom_ms) / len(dom_ms)
            ach = conv_flops(dominant) / (avg_ms * 1e-3) / 1e12
            traffic, tsrc = pmc_traffic(dominant, arith)
            split = arith != "fp32" and ops.conv_runs_split(dominant[0], dominant[2])
            nprod = MFMAS_PER_PRODUCT.get(arith, 1) if split else 1
            # the roof of the split arithmetics: each fp32 multiply-add costs `nprod` bf16 MFMA multiply-adds,
            # so the dense bf16 peak / nprod is what a perfect kernel of this arithmetic would reach
            peak = PEAK_BF16_MFMA_TFLOPS / nprod if split else PEAK_FP32_MFMA_TFLOPS
            if not split:
                kname = "conv5x5_igemm_kernel" if dominant[0] != "conv_wgrad" else "conv5x5_wgrad_kernel"
            elif dominant[0] == "conv_wgrad":
                kname = "conv5x5_wgrad_split8_kernel"
            else:      # stride 2 runs on the 8-wave ring kernel (conv_ring.hip), the rest on conv_bf16split.hip
                kname = "conv5x5_ring_kernel" if ops.conv_fusable(dominant[0] == "convT_fwd", dominant[2], dominant[5],
                                                                  dominant[6]) else "conv5x5_bf16split_kernel"
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": "bytes/launch",
                    "traffic_source": tsrc, "kernel": kname,
                    "peak_note": ("dense 16-bit MFMA 2500 TFLOP/s / %d MFMAs per fp32 multiply" % nprod) if split
                    else "fp32-input MFMA = fp32 vector peak",
                    "launch": {"op": dominant[0], "B": dominant[1], "Cin": dominant[2], "H": dominant[3],
                               "W": dominant[4], "Cout": dominant[5], "stride": dominant[6]},
                    "avg_launch_ms": round(avg_ms, 4), "launches_timed": len(dom_ms), "launches_timed_in": dom_leg,
                    "alg_gflop_per_launch": round(conv_flops(dominant) / 1e9, 3)}
            mrate, msrc = measured_mfma_rate(arith)
            if split and mrate:
                # informational: the same achieved rate against what a bare MFMA loop reaches on random operands
                roof["measured_mfma_rate_random_operands"] = {
                    "mfma_tflops": mrate, "per_fp32_multiply": round(mrate / nprod, 1),
                    "frac_of_it": round(ach / (mrate / nprod), 4), "source": msrc}
        res = {
            "metric": "celeba64_betavaegan_train_images_per_sec", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "ms_per_step_median": round(statistics.median(step_ms), 3),
            "host_enqueue_ms_per_step": round(host_ms, 3),     # Python + launch calls only: below ms_per_step = GPU-bound
            "launch_mode": "HIP graph replay (whole iteration captured: one graph launch per step)" if graphed
                           else "eager (one launch per kernel)",
            "untimed_preparation_steps": prep_steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp32" if arith == "fp32" else
                     "fp32-equivalent (%s split of fp32 operands on the %s MFMA, fp32 accumulate, every convolution%s; "
                     "%sBatchNorm, losses, Adam in fp32)" % (
                         arith, "f16" if arith == "fp16x3" else "bf16",
                         " and the Linear layers with >= 2^20 weights (vg_gemm_nt_f16x3)" if linear_own else "",
                         "the small Linear layers (vendor GEMMs), " if linear_own else "Linear layers (vendor GEMMs), ")
                     if arith in ("bf16x6", "fp16x3") else
                     arith + " (split-bf16 operands on the bf16 MFMA, fp32 accumulate) + fp32 elsewhere",
            "arithmetic": ARITH_NOTE[arith],
            "linear_gemm": "own fp16x3 GEMM (csrc/gemm_split.hip) for layers with >= 2^20 weights, vendor fp32 GEMMs for the rest"
                           if linear_own else "vendor fp32 GEMMs (hipBLASLt / rocBLAS, tuned table)",
            "data": "synthetic",
            "config": {"workload": "new_betavaegan.py beta=25 VAE-GAN iteration (D + decoder + encoder phases, "
                                   "3 Adam steps), CelebA 64x64, per-GPU batch %d" % B,
                       "global_batch": B * world, "per_gpu_batch": B, "beta": args.beta,
                       "parallelism": "dp%d" % world},
            "roofline": roof,
            "step_tflops_algorithmic": round(ALG_GFLOP_PER_IMAGE * value / world / 1e3, 2),
            "conv_path_frac_of_fp32_mfma_peak": round(ALG_CONV_GFLOP_PER_IMAGE * value / world / 1e3
                                                      / PEAK_FP32_MFMA_TFLOPS, 4),
            # the same conv-path rate against the roof of the arithmetic the kernels actually run in (the dense bf16
            # MFMA peak / MFMAs per fp32 multiply for the split modes): iteration level, everything else included
            "conv_path_frac_of_own_roof": round(ALG_CONV_GFLOP_PER_IMAGE * value / world / 1e3 / (
                PEAK_BF16_MFMA_TFLOPS / MFMAS_PER_PRODUCT[arith] if arith in MFMAS_PER_PRODUCT else PEAK_FP32_MFMA_TFLOPS), 4),
            "conv_ms_per_step_profiled": round(conv_ms_profiled, 3),
            "losses_finite": finite,
            "regime": regime,
        }
        if comm is not None:
            res["data_parallel"] = comm
        if other is not None:
            res["other_arithmetics"] = other
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(B, args.beta)
        print(json.dumps(res), flush=True)
    if world > 1 or alone:
        dist.barrier()
        from disentangle_mlp_amd import rccl
        rccl.shutdown()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
