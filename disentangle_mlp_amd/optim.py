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

from . import _lib
from ._lib import check


class _AdamTensor(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("n", ctypes.c_size_t)]


class HipAdam(optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad,
                         foreach=False, fused=False, capturable=False)

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

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or not all(self._native_ok(g) for g in self.param_groups):
            return super().step(closure)
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        for group in self.param_groups:
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
            for step, items in by_step.items():
                arr = (_AdamTensor * len(items))()
                for i, (p, g, m, v) in enumerate(items):
                    arr[i] = _AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel())
                bc1 = 1.0 - beta1 ** step
                bc2_sqrt = math.sqrt(1.0 - beta2 ** step)
                check(lib.vg_adam_step(arr, len(items), float(group["lr"]), float(beta1), float(beta2), float(group["eps"]),
                                       bc1, bc2_sqrt, stream), "vg_adam_step")
        return None
