"""``optim.Adam`` of /root/reference/experiments/new_betavaegan.py:49-50 with its step on the
hand-written HIP kernel (``vg_adam_step``).

``HipAdam`` IS a ``torch.optim.Adam``: same constructor defaults, ``param_groups``, per-parameter
``state`` (``step``, ``exp_avg``, ``exp_avg_sq``) and ``state_dict`` / ``load_state_dict`` -- the
reference's optimizer checkpoints load into it and its own load into ``torch.optim.Adam``.  Only
``step()`` is replaced: one multi-tensor launch per 24 parameters instead of torch's fused
``multi_tensor_apply``.  Configurations the kernel does not implement (weight decay, amsgrad,
maximize, a closure, non-fp32 / non-contiguous / CPU tensors) take the inherited ``step()``.
"""
import ctypes
import math

import torch
from torch import optim

from . import _lib, ops
from ._lib import check


class _AdamTensor(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("n", ctypes.c_size_t), ("amax", ctypes.c_void_p)]


class HipAdam(optim.Adam):
    """``capturable=True``: the step's scalars (lr / bias_correction1, sqrt(bias_correction2)) are formed on the DEVICE
    by ``vg_adam_prepare`` -- from the host's step count in an eager step, from a device counter in a step that is being
    captured in a HIP graph (kernel arguments are frozen at capture, the step count is not) -- so a whole iteration
    including its optimizer steps can be captured and replayed (trainer.BetaVAEGANTrainer(graph=True)), and an eager step
    and a replayed one give the same bits.  The ``state_dict`` stays torch.optim.Adam's (``step`` as a CPU tensor): the
    host mirrors the device counter (`prepare_capture` / `replayed`)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, capturable=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad,
                         foreach=False, fused=False, capturable=False)
        self.device_scalars = bool(capturable)
        self._dev = {}            # group index -> (device step counter float64[1], scalars float32[2])
        self._captured = []       # per captured step() call: the parameters it stepped (host bookkeeping of a replay)
        self._debt = 0            # replays whose host-side step counts have not been added yet (flushed lazily)
        self.state_generation = 0 # bumped by load_state_dict: captured graphs point at the moment tensors of one generation
        # max |w| of the big Linear weights, emitted by the step itself (VgAdamTensor.amax) for the fp16x3 GEMMs that read
        # the weight next: one persistent fp32 word per weight (persistent: a replayed step writes where the next replay's
        # GEMMs read), zeroed in front of every step
        self._bound_of = {}       # parameter -> index into self._bounds
        self._bounds = None
        self.register_state_dict_pre_hook(lambda opt: opt._flush_replays())

    def _native_ok(self, group):
        if group["weight_decay"] != 0 or group["amsgrad"] or group.get("maximize", False) \
                or group.get("capturable", False) or group.get("differentiable", False):
            return False
        for p in group["params"]:
            if p.grad is None:
                continue
            g = p.grad
            if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous()
                    and g.is_contiguous() and not g.is_sparse):
                return False
        return True

    # ---- HIP-graph support -------------------------------------------------------------------------------
    def _device_state(self, gi, device):
        d = self._dev.get(gi)
        if d is None:
            d = self._dev[gi] = (torch.zeros(1, dtype=torch.float64, device=device),
                                 torch.zeros(2, dtype=torch.float32, device=device))
        return d

    def _flush_replays(self):
        """Add the step counts of the replays made since the last flush to the host-side ``state[p]["step"]``."""
        if self._debt:
            for params in self._captured:
                for p in params:
                    self.state[p]["step"] += self._debt
            self._debt = 0

    def prepare_capture(self):
        """Before a capture that contains step() calls: every parameter of a group must be at the same step count (the
        captured kernels share one device counter per group), which the device counter is set to."""
        if not self.device_scalars:
            raise RuntimeError("HipAdam: construct with capturable=True to capture its step in a HIP graph")
        self._flush_replays()
        self._captured = []
        for gi, group in enumerate(self.param_groups):
            steps = {float(self.state[p]["step"]) for p in group["params"] if len(self.state[p])}
            unborn = [p for p in group["params"] if not len(self.state[p])]
            if len(steps) > 1 or (steps and unborn):
                raise RuntimeError("HipAdam.prepare_capture: parameters of one group are at different step counts")
            dev = group["params"][0].device
            self._device_state(gi, dev)[0].fill_(steps.pop() if steps else 0.0)

    def replayed(self, times=1):
        """A captured iteration was replayed: the device counters advanced inside the graph, the host's follow (lazily)."""
        self._debt += times

    def load_state_dict(self, state_dict):
        """(torch replaces the moment tensors: a graph captured before holds the old ones -- `state_generation` tells the
        trainers to capture anew.)"""
        self._debt, self._captured = 0, []
        self.state_generation += 1
        return super().load_state_dict(state_dict)

    @torch.no_grad()
    def load_state_in_place(self, state_dict):
        """`load_state_dict` that COPIES into the existing moment tensors (same parameters, same shapes) instead of
        replacing them: an iteration captured in a HIP graph stays valid (`state_generation` does not change).  Parameters
        absent from ``state_dict`` (a checkpoint taken before the first step) go back to step 0 with zero moments."""
        self._flush_replays()
        src = state_dict["state"]
        i = 0
        for gi, group in enumerate(self.param_groups):
            steps = set()
            for p in group["params"]:
                mine, theirs = self.state[p], src.get(i)
                i += 1
                if not len(mine):
                    if theirs:
                        raise RuntimeError("HipAdam.load_state_in_place: no state to copy into (use load_state_dict)")
                    continue
                if theirs:
                    mine["step"].fill_(float(theirs["step"]))
                    mine["exp_avg"].copy_(theirs["exp_avg"])
                    mine["exp_avg_sq"].copy_(theirs["exp_avg_sq"])
                else:
                    mine["step"].zero_()
                    mine["exp_avg"].zero_()
                    mine["exp_avg_sq"].zero_()
                steps.add(float(mine["step"]))
            if gi in self._dev and len(steps) == 1:
                self._dev[gi][0].fill_(steps.pop())          # the device counter a replayed step advances

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or not all(self._native_ok(g) for g in self.param_groups):
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("HipAdam: this configuration takes torch's step, which cannot be captured here")
            self._flush_replays()
            return super().step(closure)
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing and not self.device_scalars:
            raise RuntimeError("HipAdam: construct with capturable=True to capture its step in a HIP graph")
        if not capturing:
            self._flush_replays()
        bounds = self._weight_bounds()
        if bounds is not None:
            bounds.zero_()
        for gi, group in enumerate(self.param_groups):
            beta1, beta2 = group["betas"]
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:                                   # torch.optim.Adam._init_group
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if st["step"].is_cuda:                             # a checkpoint written by a fused / capturable Adam
                    st["step"] = st["step"].cpu()
                st["step"] += 1
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous()):
                    raise RuntimeError("HipAdam: optimizer state must be contiguous")
                by_step.setdefault(float(st["step"]), []).append((p, p.grad, m, v))
            if capturing and len(by_step) > 1:
                raise RuntimeError("HipAdam: a captured step needs every parameter of a group at the same step count")
            for step, items in by_step.items():
                arr = (_AdamTensor * len(items))()
                for i, (p, g, m, v) in enumerate(items):
                    bi = self._bound_of.get(p) if bounds is not None else None
                    arr[i] = _AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                         None if bi is None else bounds[bi:bi + 1].data_ptr())
                if self.device_scalars:
                    step_dev, scalars = self._device_state(gi, items[0][0].device)
                    # an eager step also stores its count in the device counter: replays may follow it
                    check(lib.vg_adam_prepare(float(step), step_dev.data_ptr(), 1 if capturing else 0, float(group["lr"]),
                                              float(beta1), float(beta2), scalars.data_ptr(), stream), "vg_adam_prepare")
                    check(lib.vg_adam_step_dev(arr, len(items), float(beta1), float(beta2), float(group["eps"]),
                                               scalars.data_ptr(), stream), "vg_adam_step_dev")
                    if capturing:
                        self._captured.append([it[0] for it in items])
                    continue
                bc1 = 1.0 - beta1 ** step
                bc2_sqrt = math.sqrt(1.0 - beta2 ** step)
                check(lib.vg_adam_step(arr, len(items), float(group["lr"]), float(beta1), float(beta2), float(group["eps"]),
                                       bc1, bc2_sqrt, stream), "vg_adam_step")
        if bounds is not None:
            for p, bi in self._bound_of.items():
                slot = bounds[bi:bi + 1]
                if p.grad is None:      # not stepped: its word was zeroed with the others -- measured again (rare: a frozen layer)
                    check(lib.vg_absmax(p.data_ptr(), p.numel(), slot.data_ptr(), stream), "vg_absmax")
                ops.set_weight_bound(p, slot)
        return None

    @torch.no_grad()
    def refresh_weight_bounds(self):
        """Measure max |w| of the big Linear weights again, into the same device words (after weights were written by
        something other than `step` -- a checkpoint copied in place under a captured iteration, whose GEMMs read these
        words)."""
        bounds = self._weight_bounds()
        if bounds is None:
            return
        bounds.zero_()
        lib, stream = _lib.load(), torch.cuda.current_stream().cuda_stream
        for p, bi in self._bound_of.items():
            slot = bounds[bi:bi + 1]
            check(lib.vg_absmax(p.data_ptr(), p.numel(), slot.data_ptr(), stream), "vg_absmax")
            ops.set_weight_bound(p, slot)

    def _weight_bounds(self):
        """The bounds tensor (one word per Linear weight the fp16x3 GEMM takes), or None when there is none."""
        if self._bounds is None:
            big = [p for g in self.param_groups for p in g["params"]
                   if p.dim() == 2 and p.is_cuda and p.numel() >= ops.LINEAR_SPLIT_MIN_WEIGHTS]
            if not big:
                return None
            self._bound_of = {p: i for i, p in enumerate(big)}
            self._bounds = torch.zeros(len(big), dtype=torch.float32, device=big[0].device)
        return self._bounds
