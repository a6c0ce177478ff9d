"""RCCL bound directly (ctypes) for the gradient exchange of the data-parallel trainers.

Why not ``torch.distributed``'s collectives: an iteration that is replayed as ONE HIP graph must contain its all-reduces,
and a c10d collective launched from an autograd hook inside a capture leaves a ``WorkNCCL`` whose event the process
group's watchdog thread later queries -- "operation not permitted on an event last recorded in a capturing stream", the
process aborts (profiles/r03_logs/r3_dpgraph.log).  RCCL itself is capturable: ``ncclAllReduce(..., stream)`` is a kernel
launch on the stream it is given.  So the communicator here is our own -- ``ncclGetUniqueId`` on rank 0, the 128 bytes
broadcast through the EXISTING ``torch.distributed`` group (whatever its backend; outside any capture),
``ncclCommInitRank`` on every rank -- and an exchange is ``ncclAllReduce`` on a side stream forked from / joined to the
compute stream by events (trainer.FlatGrads): inside a capture the fork and join become graph edges, the all-reduce a
graph node.  ``torch.distributed`` stays what launches the ranks, carries the rendezvous and the few host-side scalars
(replaces nn.DataParallel of /root/reference/experiments/new_betavaegan.py:42,44).

The library is the ``librccl.so`` PyTorch ships (already mapped into the process: c10d's "nccl" backend IS RCCL on ROCm).
"""
import ctypes
import os

import torch

_NCCL_FLOAT32, _NCCL_SUM = 7, 0
_lib = None


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_ubyte * 128)]      # (c_ubyte: a c_char array reads back truncated at the first NUL)


def _load():
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so", "librccl.so"]
    err = None
    for c in cands:
        try:
            lib = ctypes.CDLL(c)
            break
        except OSError as e:
            err = e
    else:
        raise ImportError(f"librccl.so not found ({err}): the data-parallel exchange has no other transport on the GPU")
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
    lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_void_p, ctypes.c_void_p]
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    lib.ncclGetErrorString.argtypes = [ctypes.c_int]
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclCommDestroy):
        f.restype = ctypes.c_int
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: RCCL error {rc} ({_load().ncclGetErrorString(rc).decode()})")


class Communicator:
    """One RCCL communicator over the ranks of the default ``torch.distributed`` group, on the current device."""

    def __init__(self):
        import torch.distributed as dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self._comm = ctypes.c_void_p()
        uid, box = _UniqueId(), [None]
        try:
            lib = _load()
            if self.rank == 0:
                _check(lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
                box = [ctypes.string_at(ctypes.byref(uid), 128)]
        except Exception as e:                               # rank 0 tells the others (None) instead of leaving them waiting
            self._error = e
        if self.world > 1:
            dist.broadcast_object_list(box, src=0)          # rendezvous over the group that already exists
        err = getattr(self, "_error", None)
        if box[0] is None:
            err = err or RuntimeError("rank 0 could not create an RCCL unique id")
        else:
            try:
                ctypes.memmove(ctypes.byref(uid), box[0], 128)
                _check(_load().ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")
            except Exception as e:
                err = e
        # every rank learns whether EVERY rank has a communicator: one that failed alone must not leave the others on a
        # transport it does not have
        ok = torch.tensor([0 if err is not None else 1], device="cuda")
        if self.world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            self.close()
            raise RuntimeError(f"RCCL communicator not available on every rank ({err or 'another rank failed'})")
        self.device = torch.cuda.current_device()

    def all_reduce_sum_(self, t, stream):
        """In-place SUM over the ranks of a contiguous fp32 CUDA tensor, enqueued on ``stream`` (a torch.cuda.Stream)."""
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("rccl.all_reduce_sum_: contiguous fp32 CUDA tensors only")
        _check(_load().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _NCCL_FLOAT32, _NCCL_SUM, self._comm,
                                     stream.cuda_stream), "ncclAllReduce")

    def close(self):
        if self._comm:
            _load().ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()


_comms = {}


def communicator():
    """The process's communicator for the current device (created on first use: a collective call -- every rank must
    reach it), or None when it could not be created on every rank (decided once, by all ranks together: the callers
    then stay on torch.distributed's collectives)."""
    dev = torch.cuda.current_device()
    if dev not in _comms:
        try:
            _comms[dev] = Communicator()
        except Exception as e:
            import warnings
            warnings.warn(f"own RCCL communicator unavailable ({e}); the gradient exchange uses torch.distributed's "
                          "collectives and data-parallel iterations stay eager")
            _comms[dev] = None
    return _comms[dev]


def shutdown():
    for c in _comms.values():
        if c is not None:
            c.close()
    _comms.clear()
