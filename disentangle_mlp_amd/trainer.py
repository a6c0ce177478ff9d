"""Training-step drivers: the counterparts of the reference's ``train(epoch)`` loop
bodies, one process per GPU.

  BetaVAEGANTrainer.step  <->  experiments/new_betavaegan.py:87-193
  VAETrainer.step         <->  experiments/new_vae.py:53-59
  GANTrainer.step         <->  experiments/new_gan.py:66-141

Observable results follow the reference (three optimizer steps per iteration, all of
EG moves in both EG phases, BatchNorm running statistics updated by every forward
in the reference's order, train-mode BN everywhere).  Provably dead work is
skipped:
  * the separate ``backward()`` calls of a phase are one backward of the summed loss;
  * in the decoder phase the discriminator only relays gradients: its parameters are
    frozen (no weight gradients) and ``sim_real`` is computed without a graph.
No host synchronisation happens inside ``step``; loss scalars stay on the device.

Data parallelism: every rank holds a replica and a shard of the batch; BatchNorm
statistics are replica-local (as under the reference's nn.DataParallel); before each
optimizer step the gradients, laid out in one flat fp32 buffer per network, are
summed with a single RCCL all-reduce (SUM, not mean: sum-reduced losses use local
sums, BCE is divided by the *global* batch), which reproduces the global-batch
gradient.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist
from torch import optim

from . import functional as F
from . import ops
from .optim import HipAdam
from .model import VAE, Discriminator_celeba, Generator_celeba, weights_init, shadowed_bias_params


@dataclass
class ModelOpt:
    """The fields of the reference's argparse namespace the models read
    (utils/envsetter.py:41-42,45)."""
    input_channels: int = 3
    n_hidden: int = 128
    n_z: List[int] = field(default_factory=lambda: [256, 8, 8])


class FlatGrads:
    """Gradient exchange of one network for data parallelism: RCCL all-reduces (SUM) launched from
    autograd hooks so that they overlap the rest of the backward pass.

    * LARGE parameters (>= `direct_bytes`; the 134 MB Linear weights and the big conv filters,
      > 95 % of the payload) are reduced IN PLACE on the gradient tensor autograd produced, as
      soon as it is final -- no flattening copy, no zero-fill, no extra accumulate pass.  The
      Linear-weight gradients are the first ones ready in backward, so their exchange hides
      behind the convolution backward.
    * SMALL parameters live as views in one flat fp32 buffer, cut into buckets of ~`bucket_bytes`
      from the END of the parameter list (ready first); a bucket is reduced when its last
      gradient has been accumulated.
    `finish()` reduces whatever did not fire and makes the compute stream wait for every
    exchange.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): few, large messages keep
    all the rings / trees RCCL builds busy."""

    exchange_when_alone = False      # tests: run the collectives even in a process group of one
    own_rccl = __import__("os").environ.get("VG_OWN_RCCL", "1") != "0"     # 0: c10d's collectives on the GPU too (not capturable)

    def __init__(self, params, bucket_bytes=8 << 20, direct_bytes=1 << 20, overlap=True, silent=()):
        """``silent``: parameters that never receive a gradient from autograd (the biases whose gradient is defined as
        zero, model.shadowed_bias_params): their zeroed views travel with their bucket, which does not wait for them."""
        self.params = [p for p in params]
        self._silent = {id(p) for p in silent}
        self.direct = [p.numel() * 4 >= direct_bytes for p in self.params]
        small = [i for i, d in enumerate(self.direct) if not d]
        n = sum(self.params[i].numel() for i in small)
        dev = self.params[0].device
        self.flat = torch.zeros(max(n, 1), dtype=torch.float32, device=dev)
        self.views, self.offsets = {}, {}
        off = 0
        for i in small:
            p = self.params[i]
            self.views[i] = self.flat[off:off + p.numel()].view_as(p)
            self.offsets[i] = off
            off += p.numel()
        self.buckets = []          # dicts: start, end (element offsets into flat), members (param indices)
        cap = max(1, bucket_bytes // 4)
        cur = None
        for i in reversed(small):
            if cur is None or ((cur["end"] - self.offsets[i]) > cap and cur["members"]):
                cur = dict(start=self.offsets[i], end=self.offsets[i] + self.params[i].numel(), members=[])
                self.buckets.append(cur)
            cur["start"] = self.offsets[i]
            cur["members"].append(i)
        self.bucket_of = {i: b_i for b_i, b in enumerate(self.buckets) for i in b["members"]}
        self.overlap = overlap
        self._pending, self._launched, self._direct_done, self._handles = [], [], set(), []
        self._armed = False
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(self.params)]
        # Transport.  On the GPU with the "nccl" (= RCCL) backend: our own RCCL communicator (rccl.py), all-reduces
        # enqueued on a side stream forked from / joined to the compute stream by events -- plain kernel launches, so an
        # iteration containing them can be captured in a HIP graph (c10d's collectives cannot: rccl.py).  Otherwise
        # (gloo: CPU tests, two ranks sharing one GPU) torch.distributed's async all-reduce.  Creating the communicator
        # is collective: every rank constructs its trainers in the same order.
        self._comm = self._side = None
        if (self.own_rccl and dev.type == "cuda" and dist.is_available() and dist.is_initialized()
                and dist.get_backend() == "nccl"):
            from . import rccl
            self._comm = rccl.communicator()          # None: not available on every rank (warned once) -> c10d, eager
            self._side = torch.cuda.Stream(device=dev) if self._comm is not None else None
        self._forked = False
        self._capturing = False
        # bookkeeping for bench.py: bytes handed to all-reduce since the last reset, and -- when `time_finish` is set --
        # HIP-event pairs around the waits of finish() (how long the compute stream stood still for the exchange)
        self.bytes_reduced, self.collectives = 0, 0
        self.time_finish, self.finish_events = False, []

    def _make_hook(self, i):
        def hook(p):
            if not self._armed:
                return
            if self.direct[i]:
                if self.overlap:
                    self._reduce_direct(i)
                return
            if id(p) in self._silent:      # (autograd may or may not run AccumulateGrad for a gradient it was not given)
                return
            b = self.bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0 and self.overlap:
                self._launch(b)
        return hook

    def _reduce_direct(self, i):
        g = self.params[i].grad
        if g is None or i in self._direct_done:
            return
        self._direct_done.add(i)
        self.bytes_reduced += g.numel() * 4
        self.collectives += 1
        self._all_reduce(g)

    def _launch(self, b):
        bk = self.buckets[b]
        self._launched[b] = True
        self.bytes_reduced += (bk["end"] - bk["start"]) * 4
        self.collectives += 1
        self._all_reduce(self.flat[bk["start"]:bk["end"]])

    @property
    def capturable(self):
        """Whether an iteration that exchanges through this object may be captured in a HIP graph."""
        return self._comm is not None

    def _all_reduce(self, t):
        if self._comm is None:
            self._handles.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))
            return
        # own RCCL: fork the side stream from the stream this gradient is final on (the hook runs on the autograd
        # thread, whose current stream the engine has set to the producing node's), enqueue, join in finish()
        cur = torch.cuda.current_stream()
        if torch.cuda.is_current_stream_capturing() != self._capturing:
            raise RuntimeError("FlatGrads: a gradient hook ran on a stream outside the capture that armed the exchange")
        self._side.wait_stream(cur)
        self._comm.all_reduce_sum_(t, self._side)
        self._forked = True

    def zero_and_attach(self):
        """Start of a phase: small gradients -> zeroed views of the flat buffer (autograd
        accumulates into them in place); large gradients -> None (autograd installs its result)."""
        self.flat.zero_()
        for i, p in enumerate(self.params):
            p.grad = None if self.direct[i] else self.views[i]
        self._pending = [sum(1 for i in b["members"] if id(self.params[i]) not in self._silent) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._direct_done = set()
        self._handles = []
        self._forked = False
        self._capturing = self.flat.is_cuda and torch.cuda.is_current_stream_capturing()
        # a 1-rank group still exchanges (sum over one rank): lets a single GPU exercise the RCCL path end to end
        self._armed = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FlatGrads.exchange_when_alone)

    def finish(self):
        """Reduce what did not fire from the hooks (or everything, without overlap), then make
        the current stream wait for every exchange."""
        if not self._armed:
            return
        for i, d in enumerate(self.direct):
            if d:
                self._reduce_direct(i)
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        timed = self.time_finish and self.flat.is_cuda and not self._capturing     # (events inside a capture cannot be timed)
        if timed:
            a = torch.cuda.Event(enable_timing=True)
            a.record()
        for h in self._handles:
            h.wait()
        if self._forked:                         # own RCCL: the compute stream joins the side stream
            torch.cuda.current_stream().wait_stream(self._side)
            self._forked = False
        if timed:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self.finish_events.append((a, b))
        self._handles = []
        self._armed = False

    def reset_stats(self):
        self.bytes_reduced, self.collectives, self.finish_events = 0, 0, []

    def exposed_ms(self):
        """Sum over the recorded finish() calls of the time between the compute stream reaching the waits and passing
        them (synchronises)."""
        if self.finish_events:
            torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.finish_events)

    def gathered(self):
        """All gradients flattened in parameter order (tests / diagnostics; copies)."""
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                          for p in self.params])


def _zero_grads(net):
    """``net.zero_grad(set_to_none=True)``, except for the convolution biases whose gradient is defined as exactly zero
    (model.shadowed_bias_params): they keep ONE persistent all-zero ``.grad`` -- their backward returns nothing, so it
    is never written -- instead of a zero fill and an accumulate launch per pass (35 launches per iteration)."""
    keep = getattr(net, "_vg_zero_bias_grads", None)
    if keep is None:
        keep = net._vg_zero_bias_grads = {id(p) for p in shadowed_bias_params(net)}
    for p in net.parameters():
        if id(p) in keep:
            if p.grad is None or p.grad.shape != p.shape or p.grad.device != p.device:
                p.grad = torch.zeros_like(p)
        else:
            p.grad = None


_ONES = {}


def _backward(losses):
    """One backward of the summed scalar losses, seeded with a cached 1.0 per device (autograd otherwise fills a fresh
    ones_like for every loss: 8 tiny launches per iteration)."""
    dev = losses[0].device
    one = _ONES.get(dev)
    if one is None:
        one = _ONES[dev] = torch.ones((), dtype=torch.float32, device=dev)
    torch.autograd.backward(losses, grad_tensors=[one] * len(losses))


def _make_adam(params, lr, fused, capturable=False):
    """Adam with the reference's defaults (new_betavaegan.py:49-50).  On the GPU the step runs on the
    hand-written kernel (optim.HipAdam, a torch.optim.Adam subclass: identical state_dict); ``capturable``:
    its scalars are formed on the device so that a whole iteration can be captured in a HIP graph.  CPU
    construction uses torch's own implementation."""
    if fused:
        return HipAdam(params, lr=lr, capturable=capturable)
    return optim.Adam(params, lr=lr, capturable=capturable)


def _dist_world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _dist_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _latent_generator(device, seed, rank):
    """Generator of the latent draws this replica makes itself (``noise`` of new_betavaegan.py:111,
    ``eps`` of model.py:534).  Weights come from ``torch.manual_seed(seed)`` on every rank (replicas
    start identical, and the CPU generator stays shared: the loader's permutation must agree on all
    ranks); the latents must NOT: the reference draws one independent sample per image of the global
    batch, so each rank seeds its own stream with seed + 1 + rank."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) + 1 + int(rank))
    return g


def sample_labels(rng=None):
    """The per-iteration label draw of new_betavaegan.py:89-90 (new_gan.py:68-69 alike): soft labels
    0.9 / 0.1, each flipped with probability 5 %.  Draw order as in the reference -- the fake label
    first, then the real one -- from ``rng.choice`` (default: NumPy's global ``np.random``, the stream
    the reference consumes; pass a ``np.random.RandomState`` for a reproducible run).
    Returns (real_label, fake_label) as Python floats."""
    rng = np.random if rng is None else rng
    fake_label = rng.choice(a=[0.1, 0.9], p=[0.95, 0.05])
    real_label = rng.choice(a=[0.1, 0.9], p=[0.05, 0.95])
    return float(real_label), float(fake_label)


def _shared_label_rng(seed):
    """Label stream of a data-parallel run.  The reference is ONE process: its draw of new_betavaegan.py:89-90 applies
    to the whole global batch.  With one process per GPU every rank must make the same draw, so each trainer owns a
    ``np.random.RandomState`` seeded alike on all ranks (NumPy's global stream, which the reference consumes and a
    single-process run here still uses by default, is seeded nowhere and shared by nobody)."""
    return np.random.RandomState((int(seed) + 0x5EED) % (2 ** 32))


def _loader_global_batch(loader, local_batch, world):
    """Images of the current GLOBAL batch: what nn.BCELoss's mean runs over under the reference's
    DataParallel.  data.DeviceLoader publishes it (a short last batch is split unevenly over the
    ranks); for a foreign loader every rank is assumed to hold an equal share."""
    gb = getattr(loader, "last_global_batch", None)
    return int(gb) if gb else int(local_batch) * world


GRAPH_DEFAULT = __import__("os").environ.get("VG_GRAPH", "1") != "0"     # 0: trainers never capture (every step eager)
GRAPH_WARM_STEPS = 2      # eager iterations of a shape before it is captured (workspaces, packs, GEMM plans exist then)


class _CapturedIteration:
    """One training iteration captured in a HIP graph (torch.cuda.CUDAGraph) and replayed: ~390 kernel launches become
    one graph launch, so the host no longer paces the GPU.

    What the graph freezes and how it stays correct from replay to replay:
      * inputs / latents: static device buffers the caller's tensors are copied into before a replay;
      * the two label scalars (new_betavaegan.py:89-90): a device tensor the BCE kernels read (vg_bce_loss_dev);
      * Adam's step count: a device counter advanced inside the graph (optim.HipAdam(capturable=True));
      * Python-side bookkeeping a replay does not execute -- BatchNorm ``num_batches_tracked`` (counted lazily by the
        modules), Adam's host step counts, the trainer's iteration counter -- is re-applied after each replay from what
        the capture pass recorded.
    The capture pass only records (nothing executes); the replay that follows it IS that iteration, so capturing has
    no side effect on the training state.  Outputs are static tensors, overwritten by the next replay."""

    def __init__(self, trainer, run, inputs, optimizers, bn_modules):
        self.inputs = {k: torch.empty_like(v) for k, v in inputs.items()}
        self.labels = torch.zeros(2, dtype=torch.float32, device=trainer.device)     # [real, fake]
        self.label_values = None
        self.optimizers, self.bn = optimizers, bn_modules
        self.graph = torch.cuda.CUDAGraph()
        for k, v in inputs.items():
            self.inputs[k].copy_(v)
        before_nbt = [m._nbt_pending for m in bn_modules]
        # a data-parallel iteration: the exchange's counters (bench.py's data_parallel block) advance per replay as well
        self.flats = [f for f in (getattr(trainer, n, None) for n in ("flat_d", "flat_eg", "flat_g", "flat")) if f is not None]
        before_flat = [(f.bytes_reduced, f.collectives) for f in self.flats]
        host_steps = [{p: float(o.state[p]["step"]) for g in o.param_groups for p in g["params"] if len(o.state[p])}
                      for o in optimizers]
        for o in optimizers:
            o.prepare_capture()
        try:
            with ops.amax_capture_scope(), torch.cuda.graph(self.graph):
                self.out = run(self.inputs, self.labels[0:1], self.labels[1:2])
        except Exception:
            # nothing executed: take back what the pass did on the host side
            for m, n in zip(bn_modules, before_nbt):
                m._nbt_pending = n
            for o, hs in zip(optimizers, host_steps):
                for p, v in hs.items():
                    o.state[p]["step"].fill_(v)
                o._captured = []
            raise
        # the graph holds raw pointers into the pack cache and the scratch buffers of ops: they live as long as it does
        self._buffers = ops.buffers_in_use()
        self.nbt_delta = [m._nbt_pending - n for m, n in zip(bn_modules, before_nbt)]
        self.flat_delta = [(f.bytes_reduced - b0, f.collectives - c0) for f, (b0, c0) in zip(self.flats, before_flat)]
        self.fresh = True          # the capture pass already did the host-side bookkeeping of the first replay

    def replay(self, inputs, real_label, fake_label):
        for k, v in inputs.items():
            if v is not self.inputs[k]:
                self.inputs[k].copy_(v)
        if (real_label, fake_label) != self.label_values:
            self.labels[0].fill_(real_label)
            self.labels[1].fill_(fake_label)
            self.label_values = (real_label, fake_label)
        self.graph.replay()
        if self.fresh:
            self.fresh = False
        else:
            for m, d in zip(self.bn, self.nbt_delta):
                m._nbt_pending += d
            for o in self.optimizers:
                o.replayed()
            for f, (db, dc) in zip(self.flats, self.flat_delta):
                f.bytes_reduced += db
                f.collectives += dc
        return self.out


def _use_tuned_gemms(device):
    """The Linear layers' vendor GEMMs with the measured algorithm table (tuned_gemms.py), on the GPU."""
    if torch.device(device).type == "cuda":
        from . import tuned_gemms
        tuned_gemms.enable()


class _GraphedSteps:
    """What the three trainers share to replay their iteration as a HIP graph: `_graph_usable` (may this call be a
    replay at all) and `_run_graphed` (warm-up count per shape, capture, replay, fallback)."""
    graph = False
    iteration = 0
    probe = None      # callable(name, tensor): taps of an (eager) iteration, see BetaVAEGANTrainer._phases

    def _graph_init(self, graph, on_gpu, fused_adam, dp):
        """A data-parallel iteration is captured too when its exchange is capturable (FlatGrads on our own RCCL
        communicator); over torch.distributed's collectives (gloo, VG_OWN_RCCL=0) it stays eager."""
        dp_ok = (not dp) or (on_gpu and FlatGrads.own_rccl and dist.is_available() and dist.is_initialized()
                             and dist.get_backend() == "nccl")      # (+ FlatGrads.capturable, checked per step)
        self.graph = (GRAPH_DEFAULT if graph is None else bool(graph)) and on_gpu and fused_adam and dp_ok
        self._graphs, self._shape_steps = {}, {}
        return self.graph

    _pack_plans = None

    def _run_with_pack_plan(self, body, nets):
        """``body(prepack)`` is one iteration; ``prepack(name)`` re-packs, in ONE launch, every split-bf16 filter of network
        ``name`` (None: of all networks) that the trainer's convolutions are known to ask for -- to be called at the start
        of the iteration and after each optimizer step.  The first iteration only records what is asked for (the filters
        are packed one by one on first use, as without a plan)."""
        if self._pack_plans is None:
            with ops.record_pack_requests() as rec:
                out = body(lambda name=None: None)
            owner = {p.data_ptr(): name for name, net in nets.items() for p in net.parameters()}
            seen, plans = set(), {name: [] for name in nets}
            for r in rec.requests:
                k = (r[0].data_ptr(), r[3], r[4])
                if k not in seen and r[0].data_ptr() in owner:
                    seen.add(k)
                    plans[owner[r[0].data_ptr()]].append(r)
            self._pack_plans = plans
            return out
        plans = self._pack_plans
        return body(lambda name=None: ops.prepack_filters(plans[name] if name is not None
                                                          else [r for v in plans.values() for r in v]))

    def _graph_usable(self, optimizers, data, grad_hook):
        flats = [f for f in (getattr(self, n, None) for n in ("flat_d", "flat_eg", "flat_g", "flat")) if f is not None]
        if self.graph and self._any_module_hooks():      # a replay would never fire them (and a capture would run their
            return False                                 # Python body once): hooked networks step eagerly
        return (self.graph and grad_hook is None and self.probe is None and ops._timing is None and data.is_cuda
                and all(f.capturable for f in flats)
                and all(isinstance(o, HipAdam) and o.device_scalars for o in optimizers)
                and not torch.cuda.is_current_stream_capturing())

    def _nets(self):
        return [n for n in (getattr(self, a, None) for a in ("netEG", "netD", "netG", "model")) if n is not None]

    def _any_module_hooks(self):
        from .model import _has_hooks
        return _has_hooks([m for net in self._nets() for m in net.modules()])

    def _host_state_key(self):
        """Host-side switches a capture freezes besides the shapes: part of every capture key, so that flipping one
        captures anew instead of replaying the old launches."""
        from . import model as M
        return (M.FUSE_CONV_BN, M.FUSE_HEAD_BCE, F.DEFER_WGRAD, ops.THIN_SPLIT, ops.USE_PACKED_FILTERS,
                tuple(net.training for net in self._nets()),
                tuple(p.requires_grad for net in self._nets() for p in net.parameters()))

    def _draw_into(self, cap, name, batch):
        """A latent the caller left to the trainer: drawn from this replica's stream straight into the capture's static
        buffer (same generator, same order as the eager path's torch.randn)."""
        buf = cap.inputs[name] if cap is not None else torch.empty(batch, self.opt.n_hidden, device=self.device)
        return buf.normal_(generator=self.latent_generator)

    def _run_graphed(self, key, make_inputs, labels, run, optimizers, nets, eager):
        """``key``: everything the capture freezes (shapes, divisors, learning rates, arithmetic).  ``make_inputs(cap)``
        -> dict of device tensors; ``run(inputs, real_dev, fake_dev)`` -> the iteration on static tensors;
        ``eager(inputs=None)``: the same iteration launched kernel by kernel (the first GRAPH_WARM_STEPS iterations of a
        shape, and for good after a failed capture -- then on the inputs ``make_inputs`` has already produced)."""
        key = (key, self._host_state_key())
        cap = self._graphs.get(key)
        if cap is None and self._shape_steps.get(key, 0) < GRAPH_WARM_STEPS:
            self._shape_steps[key] = self._shape_steps.get(key, 0) + 1
            return eager()
        inputs = make_inputs(cap)
        if cap is None:
            if len(self._graphs) >= 4:               # each capture keeps its own memory pool: bound them
                self._graphs.pop(next(iter(self._graphs)))
            bns = [m for net in nets for m in net.modules() if hasattr(m, "_nbt_pending")]
            it0 = self.iteration
            try:
                cap = _CapturedIteration(self, run, inputs, optimizers, bns)
            except Exception as e:                   # stay correct: this trainer goes on eagerly
                import warnings
                warnings.warn(f"HIP-graph capture of the training iteration failed ({type(e).__name__}: {e}); "
                              "continuing with eager launches")
                self.iteration = it0
                self.graph = False                   # for good: a capture that fails once is not retried
                return eager(inputs)                 # (with the latents already drawn: the RNG stream stays an eager run's)
            self._graphs[key] = cap
            self.iteration = it0                     # counted below, once per executed iteration
        out = cap.replay(inputs, *labels)
        self.iteration += 1
        return out


class BetaVAEGANTrainer(_GraphedSteps):
    """One replica of the beta-VAE-GAN (new_betavaegan.py:36-53 construction recipe).

    ``graph`` (default: on for a single-process CUDA trainer, VG_GRAPH=0 turns it off): from the third iteration of a
    batch shape on, `step` replays a HIP graph of the whole iteration (`_CapturedIteration`) instead of launching its
    ~390 kernels one by one.  Iterations that need the host in the loop stay eager: a ``grad_hook``, data parallelism
    (the gradient exchange runs from autograd hooks), launch timing (bench.py's instrumented step)."""

    def __init__(self, device="cuda", seed=999, beta=25.0, lr=1e-3, opt: Optional[ModelOpt] = None,
                 data_parallel: Optional[bool] = None, fused_adam: bool = True, capturable: Optional[bool] = None,
                 graph: Optional[bool] = None):
        self.opt = opt or ModelOpt()
        self.device = torch.device(device)
        _use_tuned_gemms(self.device)
        self.beta = float(beta)
        self.world = _dist_world()
        self.dp = (self.world > 1) if data_parallel is None else data_parallel
        self._graph_init(graph, self.device.type == "cuda", fused_adam, self.dp)
        if capturable is None:
            capturable = self.graph
        torch.manual_seed(seed)                       # new_betavaegan.py:36
        net_eg = VAE(self.opt)                        # :41  (constructed on CPU: same RNG stream
        net_d = Discriminator_celeba(self.opt)        # :43   as the reference => identical weights)
        net_eg.apply(weights_init)                    # :46
        net_d.apply(weights_init)                     # :47
        self.netEG = net_eg.to(self.device)
        self.netD = net_d.to(self.device)
        fused = fused_adam and self.device.type == "cuda"
        self.optimizerEG = _make_adam(self.netEG.parameters(), lr, fused, capturable)   # :49 (hard-coded 1e-3 there)
        self.optimizerD = _make_adam(self.netD.parameters(), lr, fused, capturable)     # :50
        self.flat_eg = FlatGrads(self.netEG.parameters(), silent=shadowed_bias_params(self.netEG)) if self.dp else None
        self.flat_d = FlatGrads(self.netD.parameters(), silent=shadowed_bias_params(self.netD)) if self.dp else None
        self.netEG.train()
        self.netD.train()
        self.iteration = 0
        self._eg_params = [p for p in self.netEG.parameters() if p.dim() == 4]   # convolution filters
        self._d_params = [p for p in self.netD.parameters() if p.dim() == 4]
        self.rank = _dist_rank()
        self.latent_generator = _latent_generator(self.device, seed, self.rank)
        self.label_rng = _shared_label_rng(seed)          # used by train_epoch when world > 1: one draw per GLOBAL batch

    def draw_latents(self, batch):
        """One N(0,1) draw of shape (batch, n_hidden) from this replica's own stream."""
        return torch.randn(batch, self.opt.n_hidden, device=self.device, generator=self.latent_generator)

    # -- gradient plumbing ------------------------------------------------------
    def _zero(self, net, flat):
        if flat is not None:
            flat.zero_and_attach()
        else:
            _zero_grads(net)

    def _exchange(self, flat):
        if flat is not None and (self.world > 1 or FlatGrads.exchange_when_alone):
            flat.finish()

    def _set_d_frozen(self, frozen):
        for p in self.netD.parameters():
            p.requires_grad_(not frozen)

    # -- one iteration ------------------------------------------------------------
    def step(self, data, noise=None, eps2=None, eps3=None, real_label=0.9, fake_label=0.1,
             global_batch: Optional[int] = None, grad_hook=None) -> Dict[str, torch.Tensor]:
        """One iteration; returns the nine loss scalars as device tensors (no host sync).  From the third iteration of a
        batch shape on the iteration is a HIP-graph replay and the returned tensors are the capture's STATIC outputs: the
        next replay overwrites them -- ``.clone()`` (or ``float()``) what is kept across steps."""
        if self._graph_usable((self.optimizerD, self.optimizerEG), data, grad_hook):
            return self._step_graphed(data, noise, eps2, eps3, float(real_label), float(fake_label), global_batch)
        # weights change only at the three optimizer steps below: packed filters are reused between them
        with ops.packed_filter_scope():
            return self._step(data, noise, eps2, eps3, real_label, fake_label, global_batch, grad_hook)

    def _step_graphed(self, data, noise, eps2, eps3, real_label, fake_label, global_batch):
        B = data.size(0)
        gb = global_batch if global_batch is not None else B * self.world
        key = (tuple(data.shape), int(gb), self.beta, self.optimizerEG.param_groups[0]["lr"],
               self.optimizerD.param_groups[0]["lr"], ops.CONV_ARITH, self.optimizerEG.state_generation,
               self.optimizerD.state_generation)

        def make_inputs(cap):                        # latents left to the trainer: drawn in the eager path's order
            lat = {name: (t if t is not None else self._draw_into(cap, name, B))
                   for name, t in (("noise", noise), ("eps2", eps2), ("eps3", eps3))}
            return dict(data=data.contiguous(), **lat)

        def run(inp, real_dev, fake_dev):
            with ops.packed_filter_scope():
                return self._step(inp["data"], inp["noise"], inp["eps2"], inp["eps3"], real_dev, fake_dev, gb, None)

        def eager(inp=None):
            lat = (inp["noise"], inp["eps2"], inp["eps3"]) if inp is not None else (noise, eps2, eps3)
            with ops.packed_filter_scope():
                return self._step(data, *lat, real_label, fake_label, global_batch, None)
        return self._run_graphed(key, make_inputs, (real_label, fake_label), run, [self.optimizerD, self.optimizerEG],
                                 (self.netEG, self.netD), eager)

    def _step(self, data, noise, eps2, eps3, real_label, fake_label, global_batch, grad_hook):
        """data (B,3,64,64) in [-1,1]; noise / eps2 / eps3 (B, n_hidden) ~ N(0,1) are drawn on
        the device when omitted (new_betavaegan.py:111, model.py:534).  Labels are the
        per-iteration scalars of :89-90.  Returns device scalars (no sync).
        ``grad_hook(phase, net)`` (tests) is called right before each optimizer step."""
        netEG, netD = self.netEG, self.netD
        B = data.size(0)
        if noise is None:
            noise = self.draw_latents(B)
        if eps2 is None:
            eps2 = self.draw_latents(B)
        if eps3 is None:
            eps3 = self.draw_latents(B)
        # Packed filters: all filters an optimizer step has changed are re-packed in ONE launch (32 launches -> 3)
        return self._run_with_pack_plan(
            lambda prepack: self._step_body(data, noise, eps2, eps3, real_label, fake_label, global_batch, grad_hook, prepack),
            {"d": netD, "eg": netEG})

    def _step_body(self, data, noise, eps2, eps3, real_label, fake_label, global_batch, grad_hook, prepack):
        # layers applied twice before one backward add their second parameter gradient inside its own kernel
        with F.accumulate_param_grads() as acc:
            return self._phases(data, noise, eps2, eps3, real_label, fake_label, global_batch, grad_hook, prepack, acc)

    def _phases(self, data, noise, eps2, eps3, real_label, fake_label, global_batch, grad_hook, prepack, acc):
        netEG, netD = self.netEG, self.netD
        B = data.size(0)
        prepack()
        # BCE is a mean over the GLOBAL batch (DataParallel gathers the outputs before the loss); equal
        # shards are assumed unless the caller says otherwise (train_epoch passes the loader's count)
        gb = global_batch if global_batch is not None else B * self.world
        out = {}

        # ---- phase 1: discriminator (:95-123)
        self._zero(netD, self.flat_d)
        fake = netEG.decode(noise)                           # graph kept for phase 2 (:113)
        if self.probe is not None and fake.requires_grad:    # diagnostics (bench.py `regime`): such an iteration is eager
            fake.register_hook(lambda g: self.probe("grad_wrt_fake", g))
        with F.deferred_wgrad():                             # D runs twice, one backward: big Linear weight gradient once
            p_real, _, err_real = netD.forward_with_bce(data, real_label, gb)
            p_fake, _, err_fake = netD.forward_with_bce(fake.detach(), fake_label, gb)
            _backward([err_real, err_fake])
        self._exchange(self.flat_d)
        if grad_hook:
            grad_hook("D", netD)
        self.optimizerD.step()
        ops.invalidate_packed_filters(self._d_params)
        prepack("d")
        acc.reset()
        out["errD_real"], out["errD_fake"] = err_real.detach(), err_fake.detach()
        out["D_x_sum"] = p_real.detach().sum()

        # ---- phase 2: "decoder" -- every EG parameter moves (:127-164)
        self._zero(netEG, self.flat_eg)
        self._set_d_frozen(True)
        with torch.no_grad():
            _, sim_real = netD(data)                         # D fwd #3: BN statistics still update
        recon, mu, logvar = netEG(data, eps2)
        _, _, err_g_fake = netD.forward_with_bce(fake, real_label, gb)
        _, sim_rec, err_g_rec = netD.forward_with_bce(recon, real_label, gb)
        sim = F.sim_loss(sim_rec, sim_real)
        mse2 = F.reconstruction_loss(recon, data)
        _backward([err_g_fake, err_g_rec, sim, mse2])
        self._set_d_frozen(False)
        self._exchange(self.flat_eg)
        if grad_hook:
            grad_hook("EG2", netEG)
        self.optimizerEG.step()
        ops.invalidate_packed_filters(self._eg_params)
        prepack("eg")
        acc.reset()
        out.update(errG_fake=err_g_fake.detach(), errG_recon=err_g_rec.detach(), sim=sim.detach(),
                   mse_dec=mse2.detach())

        # ---- phase 3: "encoder" -- again every EG parameter moves (:167-193)
        self._zero(netEG, self.flat_eg)
        recon, mu, logvar, kld = netEG.forward_with_kl(data, eps3, self.beta)
        mse3 = F.reconstruction_loss(recon, data)
        _backward([kld, mse3])
        self._exchange(self.flat_eg)
        if grad_hook:
            grad_hook("EG3", netEG)
        self.optimizerEG.step()
        ops.invalidate_packed_filters(self._eg_params)
        out.update(kld=kld.detach(), mse_enc=mse3.detach())
        self.iteration += 1
        return out

    # -- one epoch (new_betavaegan.py:77-201) ---------------------------------------------
    def train_epoch(self, loader, label_rng=None, max_iterations=None):
        """``train(epoch)`` of new_betavaegan.py:77-201 over ``for data, _ in loader``: per iteration
        the label draw of :89-90 (`sample_labels`) and one `step`; returns the reference's tuple
        ``(avg_recon_enc_loss, avg_recon_dec_loss, avg_dis_loss, avg_Dx)`` of :196-201.

        The reference's bookkeeping is kept as it is: both
        reconstruction averages accumulate the *encoder-phase* pixel MSE (:188-189 add the same
        ``loss``), ``avg_dis_loss`` and ``avg_Dx`` both accumulate the batch mean of D(x) (:105-107,
        :190), and all four sums -- per-batch sums of squared errors, and per-batch *means* of D(x) --
        are divided by ``len(loader.dataset)``.  The sums live on the device; the host reads them once
        per epoch (the reference synchronises four times per iteration with ``.item()``).
        Under data parallelism the sums are all-reduced once at the end, so every rank returns the
        global-batch values the reference's DataParallel run would log, and the labels come -- unless
        ``label_rng`` is given -- from the trainer's own stream, seeded alike on every rank: one label pair
        per global batch, as in the reference's single process."""
        acc = torch.zeros(2, dtype=torch.float64, device=self.device)     # [sum of mse_enc, sum of mean D(x)]
        n_it = 0
        if label_rng is None and self.world > 1:
            label_rng = self.label_rng                    # every rank the same label pair (see _shared_label_rng)
        for data, _ in loader:
            real_label, fake_label = sample_labels(label_rng)
            gb = _loader_global_batch(loader, data.size(0), self.world)
            out = self.step(data, real_label=real_label, fake_label=fake_label, global_batch=gb)
            acc[0] += out["mse_enc"]
            acc[1] += out["D_x_sum"] / gb
            n_it += 1
            if max_iterations is not None and n_it >= max_iterations:
                break
        if self.world > 1:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        mse_sum, dx_sum = acc.tolist()                                    # the epoch's one device -> host read
        n = len(loader.dataset)
        return mse_sum / n, mse_sum / n, dx_sum / n, dx_sum / n

    # -- the experiment script's ``__main__`` (new_betavaegan.py:211-267) -----------------
    def fit(self, loader, epochs, start_epoch=0, model_path=None, label_rng=None, calc_fid=False, n_samples=1000,
            fid_path_recons=None, fid_path_pretrained=None, get_fid=None, log=None, max_iterations=None, verbose=True):
        """The training half of the reference's ``__main__`` (new_betavaegan.py:218-246): per epoch ``train(epoch)``
        (`train_epoch`), the checkpoint ``{model_path}/model_{epoch+1}.tar`` (:222-228), optionally
        ``generate_fid_samples`` + ``get_fid`` (:231-235), the printed line (:237-238) and the logger row (:241-246:
        ``log(row)``, e.g. the reference's ``Logger.log``).  ``get_fid`` defaults to this package's
        (`fid.get_fid`; it needs Inception weights on disk, so ``calc_fid`` is off unless asked for).  Under data
        parallelism every rank trains, rank 0 writes.  Returns the rows (with ``Dx`` added)."""
        import os
        from . import image_io
        rows = []
        for epoch in range(start_epoch, epochs):
            enc_loss, dec_loss, dis_loss, dx = self.train_epoch(loader, label_rng=label_rng, max_iterations=max_iterations)
            fid = "N/A"
            if self.rank == 0:
                with torch.no_grad():
                    if model_path is not None:
                        self.save(os.path.join(model_path, f"model_{epoch + 1}.tar"), epoch + 1)
                    if calc_fid:
                        if get_fid is None:
                            from .fid import get_fid
                        image_io.generate_fid_samples(self.netEG.decode, epoch, n_samples, self.opt.n_hidden,
                                                      fid_path_recons, device=self.device)
                        fid = get_fid(fid_path_recons, fid_path_pretrained)
                if verbose:
                    print("====> Epoch: {} Avg Encoder Loss: {:.4f} Avg Decoder Loss: {:.4f} Avg Discriminator Loss: {:.4f} "
                          "FID: {} Dx: {:.4f}".format(epoch, enc_loss, dec_loss, dis_loss, fid, dx), flush=True)
                row = {"Epoch": epoch, "Avg Eec Loss": enc_loss, "Avg Dnc Loss": dec_loss, "Avg Dis Loss": dis_loss,
                       "FID": fid}
                if log is not None:
                    log(row)
                rows.append(dict(row, Dx=dx))
        return rows

    def evaluate(self, load_paths, test_loader=None, start_epoch=0, calc_fid=False, n_samples=1000, fid_path_samples=None,
                 fid_path_pretrained=None, get_fid=None, test_recons=False, test_results_path_recons=None,
                 test_results_path_originals="", test_samples=False, test_results_path_samples=None):
        """The evaluation half (new_betavaegan.py:248-267): for every checkpoint of ``load_paths`` -- load it, renumber
        its epoch the way the reference does so that files of several checkpoints do not overwrite each other (:252-254),
        then FID samples + score (:256-259), one grid of test reconstructions with ``nrow=1`` (+ the originals, :260-263)
        and five samples named after ``start_epoch`` (:264-267: the reference passes ``start_epoch`` there, so several
        checkpoints write the same file; kept).  Train-mode BatchNorm throughout: the reference never calls ``.eval()``
        (SURVEY.md section 3.1 item 5).  Returns one dict per checkpoint."""
        from . import image_io
        out, tmp_epoch = [], 0
        for m in load_paths:
            epoch = self.load(m)
            epoch = epoch if epoch != tmp_epoch and tmp_epoch < epoch else tmp_epoch + 1
            tmp_epoch = epoch
            res = {"path": m, "epoch": epoch, "FID": "N/A"}
            with torch.no_grad():
                if calc_fid:
                    if get_fid is None:
                        from .fid import get_fid
                    image_io.generate_fid_samples(self.netEG.decode, epoch, n_samples, self.opt.n_hidden, fid_path_samples,
                                                  device=self.device)
                    res["FID"] = get_fid(fid_path_samples, fid_path_pretrained)
                if test_recons:
                    image_io.gen_reconstructions(lambda x: self.netEG(x.to(self.device))[0], test_loader, epoch,
                                                 test_results_path_recons, nrow=1,
                                                 path_for_originals=test_results_path_originals, device=self.device)
                if test_samples:
                    image_io.generate_samples(self.netEG.decode, start_epoch, 5, self.opt.n_hidden,
                                              test_results_path_samples, nrow=1, device=self.device)
            out.append(res)
        return out

    # -- checkpoint (new_betavaegan.py:203-209, 222-228) --------------------------------
    def checkpoint(self, epoch):
        """The reference's dict.  ``discriminator_model`` keys carry the ``module.`` prefix
        because the reference saves the DataParallel-wrapped netD (:44, :225)."""
        return {
            "epoch": epoch,
            "encoder_decoder_model": self.netEG.state_dict(),
            "discriminator_model": {"module." + k: v for k, v in self.netD.state_dict().items()},
            "encoder_decoder_optimizer": self.optimizerEG.state_dict(),
            "discriminator_optimizer": self.optimizerD.state_dict(),
        }

    def save(self, path, epoch, legacy_format=False):
        """torch.save of the reference's dict (new_betavaegan.py:222-228).  ``legacy_format=True``
        writes the pre-1.6 (non-zip) pickle that the reference's pinned torch 1.3.1 can read."""
        torch.save(self.checkpoint(epoch), path, _use_new_zipfile_serialization=not legacy_format)

    def load(self, path_or_dict):
        ck = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location=self.device)
        self.netEG.load_state_dict(ck["encoder_decoder_model"])
        d_sd = ck["discriminator_model"]
        if all(k.startswith("module.") for k in d_sd):       # both layouts accepted
            d_sd = {k[len("module."):]: v for k, v in d_sd.items()}
        self.netD.load_state_dict(d_sd)
        self.optimizerEG.load_state_dict(ck["encoder_decoder_optimizer"])
        self.optimizerD.load_state_dict(ck["discriminator_optimizer"])
        return ck["epoch"]

    @torch.no_grad()
    def load_in_place(self, ck):
        """`load` of a checkpoint dict of THIS trainer's shapes that copies into the existing parameter, buffer and moment
        tensors: a captured iteration stays valid and the next `step` replays it from the loaded state (bench.py times
        from the initial state this way).  The packs a replay does not rewrite itself are rewritten here."""
        d_sd = ck["discriminator_model"]
        if all(k.startswith("module.") for k in d_sd):
            d_sd = {k[len("module."):]: v for k, v in d_sd.items()}
        for net, sd in ((self.netEG, ck["encoder_decoder_model"]), (self.netD, d_sd)):
            for k, v in net.state_dict().items():            # (flushes the lazily counted num_batches_tracked)
                v.copy_(sd[k])
        if isinstance(self.optimizerEG, HipAdam) and isinstance(self.optimizerD, HipAdam):
            self.optimizerEG.load_state_in_place(ck["encoder_decoder_optimizer"])
            self.optimizerD.load_state_in_place(ck["discriminator_optimizer"])
        else:
            self.optimizerEG.load_state_dict(ck["encoder_decoder_optimizer"])
            self.optimizerD.load_state_dict(ck["discriminator_optimizer"])
        ops.invalidate_packed_filters()
        if self._pack_plans is not None:
            with ops.packed_filter_scope():
                ops.prepack_filters([r for v in self._pack_plans.values() for r in v])
        for opt in (self.optimizerEG, self.optimizerD):      # (the bounds the captured GEMMs read beside the weights)
            if isinstance(opt, HipAdam):
                opt.refresh_weight_bounds()
        return ck["epoch"]


class VAETrainer(_GraphedSteps):
    """new_vae.py:33-37 construction, :39-48 loss, :53-59 step.  ``graph``: as BetaVAEGANTrainer."""

    def __init__(self, device="cuda", seed=999, beta=1.0, lr=3e-3, opt: Optional[ModelOpt] = None,
                 fused_adam: bool = True, capturable: Optional[bool] = None, graph: Optional[bool] = None):
        self.opt = opt or ModelOpt()
        self.device = torch.device(device)
        _use_tuned_gemms(self.device)
        self.beta = float(beta)
        torch.manual_seed(seed)
        m = VAE(self.opt)
        m.apply(weights_init)
        self.model = m.to(self.device)
        self.world = _dist_world()
        self._graph_init(graph, self.device.type == "cuda", fused_adam, self.world > 1)
        self.optimizer = _make_adam(self.model.parameters(), lr, fused_adam and self.device.type == "cuda",
                                    self.graph if capturable is None else capturable)
        self.flat = FlatGrads(self.model.parameters(), silent=shadowed_bias_params(self.model)) if self.world > 1 else None
        self.latent_generator = _latent_generator(self.device, seed, _dist_rank())
        self.model.train()

    def step(self, data, eps=None):
        def eager(inp=None):
            with ops.packed_filter_scope():       # one optimizer step at the end: packs live for the iteration
                return self._step(data, inp["eps"] if inp is not None else eps)
        if not self._graph_usable((self.optimizer,), data, None):
            return eager()

        def run(inp, real_dev, fake_dev):
            with ops.packed_filter_scope():
                return self._step(inp["data"], inp["eps"])
        key = (tuple(data.shape), self.beta, self.optimizer.param_groups[0]["lr"], ops.CONV_ARITH,
               self.optimizer.state_generation)
        return self._run_graphed(key, lambda cap: dict(data=data.contiguous(),
                                                       eps=eps if eps is not None else self._draw_into(cap, "eps", data.size(0))),
                                 (0.0, 0.0), run, [self.optimizer], (self.model,), eager)

    def _step(self, data, eps):
        if eps is None:                       # model.py:534, from this replica's own stream
            eps = torch.randn(data.size(0), self.opt.n_hidden, device=self.device, generator=self.latent_generator)
        return self._run_with_pack_plan(lambda prepack: self._step_body(data, eps, prepack), {"vae": self.model})

    def _step_body(self, data, eps, prepack):
        prepack()                             # the filters the previous optimizer step changed: one pack launch
        if self.flat is not None:
            self.flat.zero_and_attach()
        else:
            _zero_grads(self.model)
        recon, mu, logvar, kld = self.model.forward_with_kl(data, eps, self.beta)
        mse = F.reconstruction_loss(recon, data)
        _backward([mse, kld])
        if self.flat is not None:
            self.flat.finish()
        self.optimizer.step()
        return dict(mse=mse.detach(), kld=kld.detach())

    def train_epoch(self, loader, max_iterations=None):
        """``train(epoch)`` of new_vae.py:48-68: returns ``avg_loss`` = sum over the epoch of
        (MSE + KLD) / len(dataset); one device -> host read per epoch."""
        acc = torch.zeros((), dtype=torch.float64, device=self.device)
        n_it = 0
        for data, _ in loader:
            out = self.step(data)
            acc += out["mse"] + out["kld"]
            n_it += 1
            if max_iterations is not None and n_it >= max_iterations:
                break
        if self.world > 1:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        return float(acc.item()) / len(loader.dataset)

    def checkpoint(self, epoch):
        return {"epoch": epoch, "VAE_model": self.model.state_dict(), "optimizer": self.optimizer.state_dict()}


class GANTrainer(_GraphedSteps):
    """new_gan.py:47-61 construction, :66-141 step.  Data parallel like the beta-VAE-GAN driver (the
    reference wraps both nets in nn.DataParallel, new_gan.py:51-53): replica-local BatchNorm, one
    gradient exchange (SUM) per optimizer step, BCE divided by the global batch.  ``graph``: as
    BetaVAEGANTrainer."""

    def __init__(self, device="cuda", seed=999, lr=3e-3, opt: Optional[ModelOpt] = None, fused_adam: bool = True,
                 data_parallel: Optional[bool] = None, graph: Optional[bool] = None):
        self.opt = opt or ModelOpt()
        self.device = torch.device(device)
        _use_tuned_gemms(self.device)
        torch.manual_seed(seed)
        g = Generator_celeba(self.opt)
        d = Discriminator_celeba(self.opt)
        g.apply(weights_init)
        d.apply(weights_init)
        self.netG, self.netD = g.to(self.device), d.to(self.device)
        fused = fused_adam and self.device.type == "cuda"
        self.world = _dist_world()
        self.dp = (self.world > 1) if data_parallel is None else data_parallel
        self._graph_init(graph, self.device.type == "cuda", fused_adam, self.dp)
        self.optimizerG = _make_adam(self.netG.parameters(), lr, fused, self.graph)
        self.optimizerD = _make_adam(self.netD.parameters(), lr, fused, self.graph)
        self.flat_g = FlatGrads(self.netG.parameters(), silent=shadowed_bias_params(self.netG)) if self.dp else None
        self.flat_d = FlatGrads(self.netD.parameters(), silent=shadowed_bias_params(self.netD)) if self.dp else None
        self.latent_generator = _latent_generator(self.device, seed, _dist_rank())
        self.label_rng = _shared_label_rng(seed)
        self.netG.train()
        self.netD.train()

    def _zero(self, net, flat):
        if flat is not None:
            flat.zero_and_attach()
        else:
            _zero_grads(net)

    def _exchange(self, flat):
        if flat is not None and (self.world > 1 or FlatGrads.exchange_when_alone):
            flat.finish()

    def step(self, data, noise=None, real_label=0.9, fake_label=0.1, global_batch: Optional[int] = None,
             grad_hook=None):
        def eager(inp=None):
            with ops.packed_filter_scope():
                return self._step(data, inp["noise"] if inp is not None else noise, real_label, fake_label, global_batch,
                                  grad_hook)
        if not self._graph_usable((self.optimizerD, self.optimizerG), data, grad_hook):
            return eager()
        B = data.size(0)
        gb = global_batch if global_batch is not None else B * self.world

        def run(inp, real_dev, fake_dev):
            with ops.packed_filter_scope():
                return self._step(inp["data"], inp["noise"], real_dev, fake_dev, gb, None)
        key = (tuple(data.shape), int(gb), self.optimizerG.param_groups[0]["lr"], self.optimizerD.param_groups[0]["lr"],
               ops.CONV_ARITH, self.optimizerG.state_generation, self.optimizerD.state_generation)
        return self._run_graphed(key, lambda cap: dict(data=data.contiguous(),
                                                       noise=noise if noise is not None else self._draw_into(cap, "noise", B)),
                                 (float(real_label), float(fake_label)), run, [self.optimizerD, self.optimizerG],
                                 (self.netG, self.netD), eager)

    def _step(self, data, noise, real_label, fake_label, global_batch, grad_hook):
        if noise is None:
            noise = torch.randn(data.size(0), self.opt.n_hidden, device=self.device, generator=self.latent_generator)
        return self._run_with_pack_plan(
            lambda prepack: self._step_body(data, noise, real_label, fake_label, global_batch, grad_hook, prepack),
            {"d": self.netD, "g": self.netG})

    def _step_body(self, data, noise, real_label, fake_label, global_batch, grad_hook, prepack):
        with F.accumulate_param_grads() as acc:              # D runs twice before its backward (as BetaVAEGANTrainer)
            return self._phases(data, noise, real_label, fake_label, global_batch, grad_hook, prepack, acc)

    def _phases(self, data, noise, real_label, fake_label, global_batch, grad_hook, prepack, acc):
        B = data.size(0)
        gb = global_batch if global_batch is not None else B * self.world
        prepack()
        self._zero(self.netD, self.flat_d)
        fake = self.netG(noise)
        with F.deferred_wgrad():                             # as BetaVAEGANTrainer's discriminator phase
            p_real, _, err_real = self.netD.forward_with_bce(data, real_label, gb)
            p_fake, _, err_fake = self.netD.forward_with_bce(fake.detach(), fake_label, gb)
            _backward([err_real, err_fake])
        self._exchange(self.flat_d)
        if grad_hook:
            grad_hook("D", self.netD)
        self.optimizerD.step()
        ops.invalidate_packed_filters([p for p in self.netD.parameters() if p.dim() == 4])
        prepack("d")
        acc.reset()
        self._zero(self.netG, self.flat_g)
        for p in self.netD.parameters():
            p.requires_grad_(False)
        _, _, err_g = self.netD.forward_with_bce(fake, real_label, gb)
        _backward([err_g])
        for p in self.netD.parameters():
            p.requires_grad_(True)
        self._exchange(self.flat_g)
        if grad_hook:
            grad_hook("G", self.netG)
        self.optimizerG.step()
        return dict(errD_real=err_real.detach(), errD_fake=err_fake.detach(), errG=err_g.detach(),
                    D_x_sum=p_real.detach().sum())

    def train_epoch(self, loader, label_rng=None, max_iterations=None):
        """``train()`` of new_gan.py:66-141: label draw per iteration (:68-69), returns the reference's
        ``(avg_loss_G, avg_loss_D)`` -- including its bookkeeping: ``avg_loss_G`` is the sum of the
        per-batch generator losses over ``len(dataset)``, and the second value is *that average* divided
        by ``len(dataset)`` once more (new_gan.py:138 reads ``avg_loss_G`` where ``avg_loss_D`` was
        meant).  The sum of errD over the epoch is available as ``self.last_epoch_sums["errD"]``."""
        acc = torch.zeros(2, dtype=torch.float64, device=self.device)
        n_it = 0
        if label_rng is None and self.world > 1:
            label_rng = self.label_rng                    # as BetaVAEGANTrainer.train_epoch
        for data, _ in loader:
            real_label, fake_label = sample_labels(label_rng)
            gb = _loader_global_batch(loader, data.size(0), self.world)
            out = self.step(data, real_label=real_label, fake_label=fake_label, global_batch=gb)
            acc[0] += out["errG"]
            acc[1] += out["errD_real"] + out["errD_fake"]
            n_it += 1
            if max_iterations is not None and n_it >= max_iterations:
                break
        if self.world > 1:                       # local BCE terms are already divided by the global batch
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        g_sum, d_sum = acc.tolist()
        n = len(loader.dataset)
        self.last_epoch_sums = {"errG": g_sum, "errD": d_sum}
        avg_loss_g = g_sum / n
        return avg_loss_g, avg_loss_g / n

    def checkpoint(self, epoch):
        return {"epoch": epoch, "netG": self.netG.state_dict(), "netD": self.netD.state_dict(),
                "G_trainer": self.optimizerG.state_dict(), "D_trainer": self.optimizerD.state_dict()}
