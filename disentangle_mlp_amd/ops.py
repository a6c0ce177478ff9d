"""Tensor-level wrappers over the C-ABI HIP library (include/vaegan_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every
arithmetic op below runs in libvaegan_hip.so.  Inputs must be CUDA (ROCm) fp32
tensors -- there is deliberately no CPU path.
"""
import os

import torch

from . import _lib
from ._lib import check

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
EW_LRELU, EW_TANH, EW_SIGMOID = 0, 1, 2

_workspaces = {}

# Optional launch timing (bench.py): when set to a dict, every convolution launch whose
# key passes `_timing_filter` is bracketed by HIP events on the launch stream.
_timing = None
_timing_filter = None


def start_timing(only=None):
    """Collect (start, end) HIP-event pairs per convolution launch key; ``only`` restricts
    collection to one key (dominant-kernel timing inside the timed region)."""
    global _timing, _timing_filter
    _timing, _timing_filter = {}, only


def stop_timing():
    """Returns {key: [ms, ...]} (synchronises)."""
    global _timing, _timing_filter
    t, _timing, _timing_filter = _timing, None, None
    if not t:
        return {}
    torch.cuda.synchronize()
    return {k: [a.elapsed_time(b) for a, b in v] for k, v in t.items()}


class _timed:
    __slots__ = ("key", "a")

    def __init__(self, key):
        self.key = key if (_timing is not None and (_timing_filter is None or _timing_filter == key)) else None

    def __enter__(self):
        if self.key is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.key is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _timing.setdefault(self.key, []).append((self.a, b))
        return False


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: disentangle_mlp_amd ops need CUDA/ROCm tensors (no CPU fallback)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: tensor must be contiguous")
    return t


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def workspace(nbytes, device):
    """Scratch per (device, stream), grown on demand: kernels launched on one stream are ordered, so
    one buffer serves every two-stage reduction / split-K slab of that stream in turn; work on
    another stream (a side stream, a second trainer thread) gets its own."""
    key = (device.type, device.index, _stream())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        nbytes = max(int(nbytes), 1 << 20)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def reserve_workspace(nbytes, device):
    """Pre-size the scratch buffer (call before HIP-graph capture)."""
    return workspace(nbytes, device)


# ---------------------------------------------------------------- convolutions
# Packed filters (vg_conv5x5_pack): the implicit-GEMM kernels read the filter as
# [class][ci][tap][cout].  Outside a `packed_filter_scope` every launch re-packs its weight
# (a few microseconds); inside one -- the trainer opens it around an iteration, where it
# alone decides when weights change -- a pack is reused until `invalidate_packed_filters()`.
USE_PACKED_FILTERS = True
FP32_CONV_STATS = False     # see conv5x5_fwd
# Arithmetic of the three convolution kernels (forward, transposed = data gradient, weight gradient):
#   "fp16x3"  the product DEFAULT: every fp32 operand, times an exact power of two taken from a bound of the tensor's
#             largest magnitude, split into fp16 hi + lo (11 + 11 significand bits, residual <= 2^-24), the 3 plane
#             products hi*hi, hi*lo, lo*hi on the f16 MFMA, fp32 accumulation, scales undone on the accumulators:
#             fp32-equivalent (4e-7..6e-7 vs fp64 per convolution) at half the matrix work of bf16x6.  The bounds live in
#             device memory (`amax_of`, producers emit them: bn_act_bwd, bn_finalize_stats, affine_act);
#   "bf16x6"  OPT-IN: 3 bf16 planes (8 + 8 + 8 mantissa bits: exact, full fp32 range), 6 plane products:
#             fp32-equivalent at any dynamic range (4e-7..9e-7 vs fp64), the default of rounds 2-3;
#   "fp32"    OPT-IN: exact fp32-input MFMA (v_mfma_f32_32x32x2_f32), bit-for-bit a k-ordered fmaf chain;
#   "bf16x3"  OPT-IN: 2 bf16 planes (hi/lo), 3 MFMAs per multiply, ~4.5e-6 relative error per convolution.
# The 3-channel edge layers (conv_thin_*.hip) run bf16x6 under fp16x3 too: they are bound by HBM and issue slots.
# Set here, or with VG_CONV_ARITH in the environment.  DESIGN.md section 2.
CONV_ARITH = __import__("os").environ.get("VG_CONV_ARITH", "fp16x3")
WGRAD_SPLIT = True      # within the split modes: False keeps the weight gradient on the exact-fp32 kernel
if CONV_ARITH not in ("fp32", "bf16x3", "bf16x6", "fp16x3"):
    raise ImportError(f"VG_CONV_ARITH={CONV_ARITH!r}: expected 'fp16x3', 'bf16x6', 'bf16x3' or 'fp32'")


THIN_SPLIT = os.environ.get("VG_THIN_SPLIT", "1") != "0"   # 0: the 3-channel edge layers stay on the fp32 VALU / fp32-MFMA kernels in every arithmetic
PLANES_F16 = 0x100      # VG_PLANES_F16 of include/vaegan_hip.h


def _planes():
    """`planes` argument of the split kernels for the active arithmetic: 0 (exact fp32 MFMA), 2 (bf16x3), 3 (bf16x6) or
    2 | VG_PLANES_F16 (fp16x3)."""
    return {"fp32": 0, "bf16x3": 2, "bf16x6": 3, "fp16x3": 2 | PLANES_F16}[CONV_ARITH]


def _f16():
    return CONV_ARITH == "fp16x3"


def _thin_planes():
    """The 3-channel edge kernels take bf16 planes only: bf16x6 under fp16x3."""
    return 3 if _f16() else _planes()


# ---- bounds of max |tensor| for the fp16 planes (csrc/absmax.hip) ---------------------------------------------------
# A bound is a one-element fp32 device tensor.  Slots come zeroed from an arena (one fill launch per 512 of them); the
# producing kernel adds its maximum with an atomic.  A tensor object remembers its bound (`_vg_amax`: the tensor's version
# counter, what was applied on load, the slot), so that a gradient used by the data gradient AND the weight gradient, or
# an input used forward and again by the weight gradient, is measured once.
_AMAX_CHUNK = 512
_amax_arenas = {}        # (device index, capturing) -> [chunk, next free]


def _amax_slot(device):
    key = (device.index, torch.cuda.is_current_stream_capturing())
    a = _amax_arenas.get(key)
    if a is None or a[1] >= _AMAX_CHUNK:
        a = _amax_arenas[key] = [torch.zeros(_AMAX_CHUNK, dtype=torch.float32, device=device), 0]
    a[1] += 1
    return a[0][a[1] - 1:a[1]]


class amax_capture_scope:
    """Around a HIP-graph capture: the slots handed out inside come from chunks allocated -- and zero-filled -- INSIDE
    the capture (a replay zeroes them again before its kernels add their maxima), and no slot of those chunks is handed
    out once the capture has ended (a replay would zero it under an eager consumer)."""

    def __enter__(self):
        for k in [k for k in _amax_arenas if k[1]]:
            del _amax_arenas[k]
        return self

    def __exit__(self, *exc):
        for k in [k for k in _amax_arenas if k[1]]:
            del _amax_arenas[k]
        return False


_last_amax = None      # (data_ptr, shape, slot) of the latest plain set_amax: see adopt_amax


def set_amax(t, slot, in_affine=None):
    """Remember ``slot`` as the bound of ``t`` (as read through ``in_affine``); returns ``slot``."""
    global _last_amax
    t._vg_amax = (t._version, None if in_affine is None else id(in_affine[0]), slot)
    if in_affine is None:
        _last_amax = (t.data_ptr(), tuple(t.shape), slot)
    return slot


def adopt_amax(out):
    """``out`` is what a torch.autograd.Function just returned: autograd hands back a NEW tensor object for the tensor its
    forward produced, without the Python attribute the producing kernel's wrapper attached.  If ``out`` is that tensor
    (same memory, same shape, set just now), it inherits the bound."""
    global _last_amax
    la, _last_amax = _last_amax, None
    if la is not None and la[0] == out.data_ptr() and la[1] == tuple(out.shape) and not hasattr(out, "_vg_amax"):
        out._vg_amax = (out._version, None, la[2])
    return out


def keep_amax(src, view):
    """``view`` is a reshape of ``src`` (same elements): it inherits the bound a producer attached to ``src``."""
    known = getattr(src, "_vg_amax", None)
    if known is not None and known[0] == src._version and known[1] is None:
        view._vg_amax = (view._version, None, known[2])
    return view


def amax_of(t, in_affine=None):
    """Bound of max |t| -- of max |act(t * scale[c] + shift[c])| with ``in_affine`` = (scale, shift, act[, bound]) -- as a
    one-element device tensor: the one a producer attached, the 4th element of ``in_affine``, or one pass over t."""
    if in_affine is not None and len(in_affine) > 3 and in_affine[3] is not None:
        return in_affine[3]
    known = getattr(t, "_vg_amax", None)
    tag = None if in_affine is None else id(in_affine[0])
    if known is not None and known[0] == t._version and known[1] == tag:
        return known[2]
    lib = _lib.load()
    slot = _amax_slot(t.device)
    if in_affine is None:
        check(lib.vg_absmax(t.data_ptr(), t.numel(), slot.data_ptr(), _stream()), "vg_absmax")
    else:
        B, C = t.shape[0], t.shape[1]
        check(lib.vg_absmax_affine(t.data_ptr(), in_affine[0].data_ptr(), in_affine[1].data_ptr(), int(in_affine[2]), B, C,
                                   t.numel() // (B * C), slot.data_ptr(), _stream()), "vg_absmax_affine")
    return set_amax(t, slot, in_affine)


def conv_runs_split(op, cin, cout=None, stride=None):
    """Whether a convolution launch of kind ``op`` ("conv_fwd" | "convT_fwd" | "conv_wgrad") with
    ``cin`` input channels runs on the split-bf16 kernels under the active arithmetic (bench.py:
    which roofline a launch is priced against).  Mirrors the dispatch conditions below."""
    planes = _planes()
    if not planes:
        return False
    if cin <= 3 and op in ("conv_fwd", "conv_wgrad"):
        return THIN_SPLIT          # conv_thin_fwd.hip / conv_thin_wgrad.hip (shapes they take: output width % 32 / % 16)
    if op == "conv_wgrad":
        return WGRAD_SPLIT and cin >= (16 if (planes & 0xff) == 2 else 32)
    if op == "convT_fwd" and stride == 1 and cout is not None and cout <= 4:
        return THIN_SPLIT and cin == 32 and cout <= 3          # conv_thin_mfma.hip
    return cin % 16 == 0


_pack_scope_depth = 0
_wbound_cache = {}    # Linear weights: data_ptr -> (version, shape, bound slot); lives and dies with the pack cache's entries
_pack_cache = {}      # (data_ptr, transposed, stride, shape) -> [valid, version, packed tensor]
_pack_scratch = {}    # (device, stream, numel) -> tensor, for un-cached packs
_PACK_CACHE_MAX = 64  # entries (a beta-VAE-GAN iteration uses 21); beyond it the cache is rebuilt


class packed_filter_scope:
    """Within the scope packed filters are cached per weight storage; the owner of the scope
    promises to call `invalidate_packed_filters()` after every in-place weight update."""

    def __enter__(self):
        global _pack_scope_depth
        if _pack_scope_depth == 0:
            if len(_pack_cache) > _PACK_CACHE_MAX:     # weights of trainers that no longer exist
                _pack_cache.clear()
            invalidate_packed_filters()
        _pack_scope_depth += 1
        return self

    def __exit__(self, *exc):
        global _pack_scope_depth
        _pack_scope_depth -= 1
        if _pack_scope_depth == 0:
            invalidate_packed_filters()
        return False


def buffers_in_use():
    """Every cached pack buffer and scratch buffer that exists now, as a list of tensors.  A HIP graph captured over
    launches that read or write them holds raw device pointers: its owner keeps this list alive for as long as the graph
    may be replayed, so that a cache rebuilt (`_PACK_CACHE_MAX`) or a workspace regrown by somebody else never hands the
    memory a replay still writes to another tensor."""
    return [ent[2] for ent in _pack_cache.values()] + list(_workspaces.values()) + list(_pack_scratch.values())


def invalidate_packed_filters(params=None):
    """Drop cached packs -- all of them, or those of the given weight tensors."""
    if params is None:
        for ent in _pack_cache.values():
            ent[0] = False
        _wbound_cache.clear()
        return
    ptrs = {p.data_ptr() for p in params if p.dim() == 4}
    for key, ent in _pack_cache.items():
        if key[0] in ptrs:
            ent[0] = False
    for p in params:
        if p.dim() == 2:
            _wbound_cache.pop(p.data_ptr(), None)


def _packed_filter(lib, w, cout, cin, transposed, stride):
    bf16x3 = transposed >= 2                      # 2 / 3: split-bf16 pack of the opt-in modes (conv / transposed conv)
    planes = _planes()
    if bf16x3 and _pack_log is not None:
        _pack_log.append((w, cout, cin, transposed, stride))
    n = lib.vg_conv5x5_packed_bf16split_bytes(cout, cin, planes) // 4 if bf16x3 else lib.vg_conv5x5_packed_floats(cout, cin)
    if _pack_scope_depth > 0:
        key = (w.data_ptr(), transposed, stride, tuple(w.shape), planes if bf16x3 else 0)
        ent = _pack_cache.get(key)
        if ent is not None and ent[0] and ent[1] == w._version:
            return ent[2]
        if ent is None:
            ent = _pack_cache[key] = [False, -1, torch.empty(n, dtype=torch.float32, device=w.device)]
        buf = ent[2]
    else:
        ent = None
        skey = (w.device.index, _stream(), n)
        buf = _pack_scratch.get(skey)
        if buf is None:
            buf = _pack_scratch[skey] = torch.empty(n, dtype=torch.float32, device=w.device)
    if bf16x3:
        wmax = None
        if planes & PLANES_F16:          # a fresh (zeroed) slot per pack: the bound follows the weights down as well as up
            wmax = _amax_slot(w.device)
            check(lib.vg_absmax(w.data_ptr(), w.numel(), wmax.data_ptr(), _stream()), "vg_absmax")
        check(lib.vg_conv5x5_pack_bf16split(w.data_ptr(), buf.data_ptr(), cout, cin, transposed - 2, stride, planes,
                                         _ptr(wmax), _stream()), "vg_conv5x5_pack_bf16split")
    else:
        check(lib.vg_conv5x5_pack(w.data_ptr(), buf.data_ptr(), cout, cin, transposed, stride, _stream()),
              "vg_conv5x5_pack")
    if ent is not None:
        ent[0], ent[1] = True, w._version
    return buf


_pack_log = None      # while a list: every split-bf16 pack request appends (weight tensor, cout, cin, kind, stride)


class record_pack_requests:
    """Context: collects which (weight, layout) pairs the convolutions inside it asked for -- a trainer records its
    first iteration and afterwards re-packs all filters an optimizer step has changed in ONE launch
    (`prepack_filters`) instead of one launch per filter on first use."""

    def __enter__(self):
        global _pack_log
        self._prev, _pack_log = _pack_log, []
        self.requests = _pack_log
        return self

    def __exit__(self, *exc):
        global _pack_log
        _pack_log = self._prev
        return False


def prepack_filters(requests):
    """Pack (split-bf16 layout) every listed filter whose cached pack is stale, all in one launch.  ``requests``:
    (weight, cout, cin, kind, stride) tuples as `record_pack_requests` collects them.  Only inside a
    `packed_filter_scope` (outside it nothing is cached)."""
    if _pack_scope_depth <= 0 or not requests:
        return
    lib = _lib.load()
    planes = _planes()
    if not planes:
        return
    todo = []
    for (w, cout, cin, kind, stride) in requests:
        key = (w.data_ptr(), kind, stride, tuple(w.shape), planes)
        ent = _pack_cache.get(key)
        if ent is not None and ent[0] and ent[1] == w._version:
            continue
        if ent is None:
            n = lib.vg_conv5x5_packed_bf16split_bytes(cout, cin, planes) // 4
            ent = _pack_cache[key] = [False, -1, torch.empty(n, dtype=torch.float32, device=w.device)]
        todo.append((w, ent, cout, cin, kind, stride))
    if not todo:
        return
    arr = (_lib.PackEntry * len(todo))()
    wmax = {}
    if planes & PLANES_F16:              # the filters' bounds first, all in one launch (one per weight, not per layout)
        for (w, *_r) in todo:
            if w.data_ptr() not in wmax:
                wmax[w.data_ptr()] = (w, _amax_slot(w.device))
        am = (_lib.AbsmaxEntry * len(wmax))()
        for i, (w, slot) in enumerate(wmax.values()):
            am[i] = _lib.AbsmaxEntry(w.data_ptr(), w.numel(), slot.data_ptr())
        check(lib.vg_absmax_multi(am, len(wmax), _stream()), "vg_absmax_multi")
    for i, (w, ent, cout, cin, kind, stride) in enumerate(todo):
        slot = wmax[w.data_ptr()][1].data_ptr() if wmax else None
        arr[i] = _lib.PackEntry(w.data_ptr(), ent[2].data_ptr(), cout, cin, kind - 2, stride, slot)
    check(lib.vg_conv5x5_pack_bf16split_multi(arr, len(todo), planes, _stream()), "vg_conv5x5_pack_bf16split_multi")
    for (w, ent, *_rest) in todo:
        ent[0], ent[1] = True, w._version


def conv_fusable(transposed, cin, cout, stride):
    """Whether the kernel of this layer (under the active arithmetic) applies a producer's BatchNorm + activation
    while it loads its input and can leave output statistics (include/vaegan_hip.h, vg_conv_fusion)."""
    return bool(_planes()) and bool(_lib.load().vg_conv5x5_bf16split_fusable(1 if transposed else 0, cin, cout, stride))


def _fusion_struct(x, in_affine, stats):
    """ctypes vg_conv_fusion (or None) + the tensors it points at (kept alive by the caller).  fp16 planes: always, with
    the bound of the input as the kernel reads it."""
    if in_affine is None and stats is None and not _f16():
        return None
    f = _lib.ConvFusion()
    if _f16():
        f._amax = amax_of(x, in_affine)          # kept alive with the struct
        f.in_amax = f._amax.data_ptr()
    if in_affine is not None:
        scale, shift, act = in_affine[:3]
        _req(scale, "in_scale"), _req(shift, "in_shift")
        if scale.numel() != x.shape[1] or shift.numel() != x.shape[1]:
            raise RuntimeError("in_affine: one coefficient per input channel")
        f.in_scale, f.in_shift, f.in_act = scale.data_ptr(), shift.data_ptr(), int(act)
    if stats is not None:
        f.stats, f.stats_floats = stats.data_ptr(), stats.numel()
    return f


def _materialize(x, in_affine):
    """act(x * scale[c] + shift[c]) as a tensor: the fallback for kernels that cannot apply it on load."""
    return x if in_affine is None else affine_act(x, *in_affine[:3])


def conv5x5_fwd(x, w, bias, stride, in_affine=None, want_stats=False):
    """``in_affine`` = (scale, shift, act): the input is act(x * scale[c] + shift[c]) -- the producing layer's
    train-mode BatchNorm + activation -- applied on load where the kernel can, materialised first where it cannot.
    ``want_stats``: returns (y, stats) with per-channel partial sums of y for `bn_finalize_stats`, or (y, None)
    when this layer's kernel cannot emit them."""
    lib = _lib.load()
    _req(x, "x"), _req(w, "w")
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    if w.shape != (Cout, Cin, 5, 5):
        raise RuntimeError(f"conv5x5_fwd: weight {tuple(w.shape)} does not match input channels {Cin}")
    if bias is not None:
        _req(bias, "bias")
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((B, Cout, OH, OW), dtype=torch.float32, device=x.device)
    stats = None
    if _planes() and Cin % 16 == 0:
        fus = conv_fusable(False, Cin, Cout, stride)
        if in_affine is not None and not fus:
            x, in_affine = _materialize(x, in_affine), None
        if want_stats and fus:
            n = lib.vg_conv5x5_fwd_bf16split_stats_floats(B, Cin, H, W, Cout, stride, _planes())
            stats = torch.empty(n, dtype=torch.float32, device=x.device) if n else None
        f = _fusion_struct(x, in_affine, stats)
        pk = _packed_filter(lib, w, Cout, Cin, 2, stride)      # the stride-2 kernel has its own step order
        need = lib.vg_conv5x5_fwd_bf16split_workspace_bytes(B, Cin, H, W, Cout, stride, _planes())    # split-K slabs, deep-K layers only
        ws = workspace(need, x.device) if need else None
        with _timed(("conv_fwd", B, Cin, H, W, Cout, stride)):
            check(lib.vg_conv5x5_fwd_bf16split(x.data_ptr(), pk.data_ptr(), _ptr(bias), y.data_ptr(), B, Cin, H, W,
                                            Cout, stride, _planes(), _ptr(ws), ws.numel() if need else 0,
                                            __import__("ctypes").byref(f) if f is not None else None, _stream()),
                  "vg_conv5x5_fwd_bf16split")
        return (y, stats) if want_stats else y
    x = _materialize(x, in_affine)
    if _planes() and THIN_SPLIT and Cin <= 3 and lib.vg_conv5x5_thin_bf16split_ok(Cin, H, W, Cout, stride):
        # <= 3 input channels: filter resident in registers, one write pass over y, statistics on the way out
        n = lib.vg_conv5x5_thin_bf16split_stats_floats(B, Cin, H, W, Cout, stride) if want_stats else 0
        stats = torch.empty(n, dtype=torch.float32, device=x.device) if n else None
        with _timed(("conv_fwd", B, Cin, H, W, Cout, stride)):
            check(lib.vg_conv5x5_thin_bf16split(x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), B, Cin, H, W, Cout,
                                                stride, _thin_planes(), _ptr(stats), n, _stream()), "vg_conv5x5_thin_bf16split")
        return (y, stats) if want_stats else y
    if USE_PACKED_FILTERS:
        pk = _packed_filter(lib, w, Cout, Cin, 0, stride)
        # The exact-fp32 kernel can leave the next BatchNorm's statistics too (vg_conv5x5_fwd_packed_stats), but on the
        # 3-channel first layers -- 16 384 slots for 32 channels at B = 128 -- writing and reducing the slots costs more
        # than the one pass over the output it saves (measured: +0.26 ms per iteration): opt-in only.
        n = lib.vg_conv5x5_fwd_packed_stats_floats(B, Cin, H, W, Cout, stride) if (want_stats and FP32_CONV_STATS) else 0
        if n:
            stats = torch.empty(n, dtype=torch.float32, device=x.device)
            with _timed(("conv_fwd", B, Cin, H, W, Cout, stride)):
                check(lib.vg_conv5x5_fwd_packed_stats(x.data_ptr(), pk.data_ptr(), _ptr(bias), y.data_ptr(), B, Cin, H,
                                                      W, Cout, stride, stats.data_ptr(), n, _stream()),
                      "vg_conv5x5_fwd_packed_stats")
            return y, stats
        with _timed(("conv_fwd", B, Cin, H, W, Cout, stride)):
            check(lib.vg_conv5x5_fwd_packed(x.data_ptr(), pk.data_ptr(), _ptr(bias), y.data_ptr(), B, Cin, H, W,
                                            Cout, stride, _stream()), "vg_conv5x5_fwd_packed")
        return (y, None) if want_stats else y
    with _timed(("conv_fwd", B, Cin, H, W, Cout, stride)):
        check(lib.vg_conv5x5_fwd(x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), B, Cin, H, W, Cout,
                                 stride, _stream()), "vg_conv5x5_fwd")
    return (y, None) if want_stats else y


def convT5x5_fwd(x, w, bias, stride, in_affine=None, want_stats=False):
    """w is (Cin, Cout, 5, 5); output is (B, Cout, stride*H, stride*W).  ``in_affine`` / ``want_stats``: see
    `conv5x5_fwd`."""
    lib = _lib.load()
    _req(x, "x"), _req(w, "w")
    B, Cin, H, W = x.shape
    Cout = w.shape[1]
    if w.shape != (Cin, Cout, 5, 5):
        raise RuntimeError(f"convT5x5_fwd: weight {tuple(w.shape)} does not match input channels {Cin}")
    if bias is not None:
        _req(bias, "bias")
    y = torch.empty((B, Cout, H * stride, W * stride), dtype=torch.float32, device=x.device)
    # stride 1 with <= 4 output channels runs the direct VALU kernel on the plain layout
    thin = stride == 1 and Cout <= 4
    stats = None
    if _planes() and Cin % 16 == 0 and not thin:
        fus = conv_fusable(True, Cin, Cout, stride)
        if in_affine is not None and not fus:
            x, in_affine = _materialize(x, in_affine), None
        if want_stats and fus:
            n = lib.vg_convT5x5_fwd_bf16split_stats_floats(B, Cin, H, W, Cout, stride, _planes())
            stats = torch.empty(n, dtype=torch.float32, device=x.device) if n else None
        f = _fusion_struct(x, in_affine, stats)
        pk = _packed_filter(lib, w, Cout, Cin, 3, stride)
        need = lib.vg_convT5x5_fwd_bf16split_workspace_bytes(B, Cin, H, W, Cout, stride, _planes())   # split-K slabs, small grids only
        ws = workspace(need, x.device) if need else None
        with _timed(("convT_fwd", B, Cin, H, W, Cout, stride)):
            check(lib.vg_convT5x5_fwd_bf16split(x.data_ptr(), pk.data_ptr(), _ptr(bias), y.data_ptr(), B, Cin, H, W,
                                             Cout, stride, _planes(), _ptr(ws), ws.numel() if need else 0,
                                             __import__("ctypes").byref(f) if f is not None else None, _stream()),
                  "vg_convT5x5_fwd_bf16split")
        return (y, stats) if want_stats else y
    if thin and _planes() and THIN_SPLIT and lib.vg_convT5x5_s1_thin_bf16split_ok(Cin, H, W, Cout):
        # 32 -> (<= 3) channels: filter resident in registers, one pass over x (BatchNorm + activation applied on load)
        scale, shift, act = in_affine[:3] if in_affine is not None else (None, None, ACT_NONE)
        if scale is not None:
            _req(scale, "in_scale"), _req(shift, "in_shift")
        with _timed(("convT_fwd", B, Cin, H, W, Cout, stride)):
            check(lib.vg_convT5x5_s1_thin_bf16split(x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), B, Cin, H, W, Cout,
                                                    _thin_planes(), _ptr(scale), _ptr(shift), int(act), _stream()),
                  "vg_convT5x5_s1_thin_bf16split")
        return (y, None) if want_stats else y
    x = _materialize(x, in_affine)
    if USE_PACKED_FILTERS and not thin:
        pk = _packed_filter(lib, w, Cout, Cin, 1, stride)
        with _timed(("convT_fwd", B, Cin, H, W, Cout, stride)):
            check(lib.vg_convT5x5_fwd_packed(x.data_ptr(), pk.data_ptr(), _ptr(bias), y.data_ptr(), B, Cin, H, W,
                                             Cout, stride, _stream()), "vg_convT5x5_fwd_packed")
        return (y, None) if want_stats else y
    with _timed(("convT_fwd", B, Cin, H, W, Cout, stride)):
        check(lib.vg_convT5x5_fwd(x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), B, Cin, H, W, Cout,
                                  stride, _stream()), "vg_convT5x5_fwd")
    return (y, None) if want_stats else y


def conv5x5_wgrad(x, gy, stride, out=None, in_affine=None, affine_on_gy=False, accumulate=False):
    """dw[Cout,Cin,5,5] for y = conv(x, w, stride); x (B,Cin,H,W), gy (B,Cout,OH,OW).  ``in_affine`` = (scale, shift,
    act): the operand x -- or gy when ``affine_on_gy`` (the weight gradient of a transposed convolution passes the
    layer's input there) -- is read as act(v * scale[c] + shift[c]), on load where the kernel can.  ``accumulate``: the
    result is added to what ``out`` holds (inside the kernel's final slab sum)."""
    if accumulate and out is None:
        raise RuntimeError("conv5x5_wgrad: accumulate needs the tensor to accumulate into (out=)")
    acc = 1 if accumulate else 0
    lib = _lib.load()
    _req(x, "x"), _req(gy, "gy")
    B, Cin, H, W = x.shape
    Cout = gy.shape[1]
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    if gy.shape != (B, Cout, OH, OW):
        raise RuntimeError(f"conv5x5_wgrad: gy {tuple(gy.shape)} does not match x {tuple(x.shape)} stride {stride}")
    dw = out if out is not None else torch.empty((Cout, Cin, 5, 5), dtype=torch.float32, device=x.device)
    # thin inputs stay on the exact-fp32 kernel: the re-layout of gy costs more than the split arithmetic saves
    # (measured: 2 planes pay off from 16 input channels, 3 planes from 32)
    if _planes() and WGRAD_SPLIT and Cin >= (16 if (_planes() & 0xff) == 2 else 32):
        need = lib.vg_conv5x5_wgrad_bf16split_workspace_bytes(B, Cin, H, W, Cout, stride, _planes())    # 0: shape not taken
        if need:
            ws = workspace(need, x.device)
            sc, sh, act = in_affine[:3] if in_affine is not None else (None, None, 0)
            xmax = gmax = None
            if _f16():       # bounds of the two operands as the kernel reads them
                xmax = amax_of(x, None if affine_on_gy else in_affine)
                gmax = amax_of(gy, in_affine if affine_on_gy else None)
            with _timed(("conv_wgrad", B, Cin, H, W, Cout, stride)):
                check(lib.vg_conv5x5_wgrad_bf16split(x.data_ptr(), gy.data_ptr(), dw.data_ptr(), B, Cin, H, W, Cout, stride,
                                                  _planes(), ws.data_ptr(), ws.numel(), _ptr(sc), _ptr(sh), int(act),
                                                  1 if affine_on_gy else 0, _ptr(xmax), _ptr(gmax), acc, _stream()),
                      "vg_conv5x5_wgrad_bf16split")
            return dw
    if _planes() and THIN_SPLIT and Cin <= 3 and (in_affine is None or affine_on_gy):
        # <= 3 input channels: one read pass over gy (a producer's BatchNorm + activation applied to it on load: the weight
        # gradient of the decoder's last layer), x split once per workgroup into shifted plane copies in LDS
        need = lib.vg_conv5x5_thin_wgrad_bf16split_workspace_bytes(B, Cin, H, W, Cout, stride, _thin_planes())   # 0: shape not taken
        if need:
            ws = workspace(need, x.device)
            sc, sh, act = in_affine[:3] if in_affine is not None else (None, None, 0)
            if sc is not None:
                _req(sc, "in_scale"), _req(sh, "in_shift")
            with _timed(("conv_wgrad", B, Cin, H, W, Cout, stride)):
                check(lib.vg_conv5x5_thin_wgrad_bf16split(x.data_ptr(), gy.data_ptr(), dw.data_ptr(), B, Cin, H, W, Cout,
                                                          stride, _thin_planes(), ws.data_ptr(), ws.numel(), _ptr(sc), _ptr(sh),
                                                          int(act), acc, _stream()), "vg_conv5x5_thin_wgrad_bf16split")
            return dw
    if in_affine is not None:      # the kernels below take the operand as a tensor
        if affine_on_gy:
            gy = _materialize(gy, in_affine)
        else:
            x = _materialize(x, in_affine)
    if _planes() and THIN_SPLIT and Cin <= 3:
        need = lib.vg_conv5x5_thin_wgrad_bf16split_workspace_bytes(B, Cin, H, W, Cout, stride, _thin_planes())
        if need:
            ws = workspace(need, x.device)
            with _timed(("conv_wgrad", B, Cin, H, W, Cout, stride)):
                check(lib.vg_conv5x5_thin_wgrad_bf16split(x.data_ptr(), gy.data_ptr(), dw.data_ptr(), B, Cin, H, W, Cout,
                                                          stride, _thin_planes(), ws.data_ptr(), ws.numel(), None, None, 0, acc,
                                                          _stream()), "vg_conv5x5_thin_wgrad_bf16split")
            return dw
    need = lib.vg_conv5x5_wgrad_workspace_bytes(B, Cin, H, W, Cout, stride)
    ws = workspace(need, x.device)
    with _timed(("conv_wgrad", B, Cin, H, W, Cout, stride)):
        check(lib.vg_conv5x5_wgrad(x.data_ptr(), gy.data_ptr(), dw.data_ptr(), B, Cin, H, W, Cout, stride,
                                   ws.data_ptr(), ws.numel(), acc, _stream()), "vg_conv5x5_wgrad")
    return dw


# ------------------------------------------------------------------- Linear layers
# The Linear GEMMs of the big layers (16384 <-> 2048 / 512, 128 -> 16384: model.py:460-471, 402-408, 490-492) on this
# package's fp16x3 GEMM (csrc/gemm_split.hip) under the default arithmetic; under the opt-in arithmetics, for small
# layers and for reductions that are not a multiple of 32 (the weight gradient at such batches) they stay on the vendor
# fp32 GEMM.  VG_LINEAR_SPLIT=0 (or ops.LINEAR_SPLIT = False): vendor GEMMs everywhere.
LINEAR_SPLIT = os.environ.get("VG_LINEAR_SPLIT", "1") != "0"
LINEAR_SPLIT_MIN_WEIGHTS = 1 << 20


def linear_split_ok(reduction, nweights):
    return LINEAR_SPLIT and _f16() and reduction % 32 == 0 and nweights >= LINEAR_SPLIT_MIN_WEIGHTS


_wbound_emitted = {}      # id(weight) -> (weak reference, version, bound): what the optimizer step that last wrote it emitted


def set_weight_bound(w, bound):
    """``bound`` (one-element device tensor) holds max |w| as of now -- HipAdam's step emits it (VgAdamTensor.amax).
    Valid for THIS tensor object (a weak reference: an address or id re-used by another tensor never matches) until the
    next torch-side in-place write (the version counter) or the next call for this weight."""
    import weakref
    if len(_wbound_emitted) > 256:                     # weights of trainers that no longer exist
        for k in [k for k, e in _wbound_emitted.items() if e[0]() is None]:
            del _wbound_emitted[k]
    _wbound_emitted[id(w)] = (weakref.ref(w), w._version, bound)


def weight_bound(w):
    """Bound of max |w| of a Linear weight: the one the optimizer step emitted when it wrote the weight; failing that,
    inside a `packed_filter_scope`, measured once per weight version (the scope's owner invalidates after optimizer
    steps, as for the packed filters), otherwise per call."""
    ent = _wbound_emitted.get(id(w))
    if ent is not None and ent[0]() is w and ent[1] == w._version:
        return ent[2]
    if _pack_scope_depth > 0:
        ent = _wbound_cache.get(w.data_ptr())
        if ent is not None and ent[0] == w._version and ent[1] == tuple(w.shape):
            return ent[2]
    slot = _amax_slot(w.device)          # a fresh (zeroed) slot: the bound follows the weights down as well as up
    check(_lib.load().vg_absmax(w.data_ptr(), w.numel(), slot.data_ptr(), _stream()), "vg_absmax")
    if _pack_scope_depth > 0:
        _wbound_cache[w.data_ptr()] = (w._version, tuple(w.shape), slot)
    return slot


def _gemm_nt(A, B, bias, C, M, N, K, ars, aks, brs, bks, a_amax, b_amax):
    lib = _lib.load()
    need = lib.vg_gemm_nt_f16x3_workspace_bytes(M, N, K)
    ws = workspace(need, A.device) if need else None
    check(lib.vg_gemm_nt_f16x3(A.data_ptr(), B.data_ptr(), _ptr(bias), C.data_ptr(), M, N, K, ars, aks, brs, bks,
                               a_amax.data_ptr(), b_amax.data_ptr(), _ptr(ws), ws.numel() if need else 0, _stream()),
          "vg_gemm_nt_f16x3")
    return C


def linear_fwd(x, w, bias):
    """y = x W^T + bias (nn.Linear forward, model.py:460-471): x (M, K), w (N, K)."""
    _req(x, "x"), _req(w, "w")
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    return _gemm_nt(x, w, bias, y, M, N, K, K, 1, K, 1, amax_of(x), weight_bound(w))


def linear_dgrad(gy, w):
    """gx = gy W: gy (M, N), w (N, K) -> (M, K); the reduction runs over N, W is read with its row index contiguous."""
    _req(gy, "gy"), _req(w, "w")
    M, N = gy.shape
    K = w.shape[1]
    gx = torch.empty((M, K), dtype=torch.float32, device=gy.device)
    return _gemm_nt(gy, w, None, gx, M, K, N, N, 1, 1, K, amax_of(gy), weight_bound(w))


def linear_wgrad(gy, x):
    """gW = gy^T x: gy (M, N), x (M, K) -> (N, K); the reduction runs over the batch M (both operands strided)."""
    _req(gy, "gy"), _req(x, "x")
    M, N = gy.shape
    K = x.shape[1]
    gw = torch.empty((N, K), dtype=torch.float32, device=gy.device)
    return _gemm_nt(gy, x, None, gw, N, K, M, 1, N, 1, K, amax_of(gy), amax_of(x))


def channel_sum(g):
    lib = _lib.load()
    _req(g, "g")
    B, C = g.shape[0], g.shape[1]
    HW = g.numel() // (B * C)
    out = torch.empty(C, dtype=torch.float32, device=g.device)
    ws = workspace(lib.vg_bn_workspace_bytes(C), g.device)
    check(lib.vg_channel_sum(g.data_ptr(), out.data_ptr(), B, C, HW, ws.data_ptr(), ws.numel(), _stream()),
          "vg_channel_sum")
    return out


# ------------------------------------------------------------------- BatchNorm
def bn_act_fwd(x, gamma, beta, running_mean, running_var, eps, momentum, act):
    lib = _lib.load()
    _req(x, "x"), _req(gamma, "gamma"), _req(beta, "beta")
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = workspace(lib.vg_bn_workspace_bytes(C), x.device)
    slot = _amax_slot(x.device) if (_f16() and HW > 1) else None      # max |y| on the way out (as affine_act)
    check(lib.vg_bn_act_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _ptr(running_mean),
                            _ptr(running_var), mean.data_ptr(), invstd.data_ptr(), B, C, HW, eps, momentum, act, _ptr(slot),
                            ws.data_ptr(), ws.numel(), _stream()), "vg_bn_act_fwd")
    if slot is not None:
        set_amax(y, slot)
    return y, mean, invstd


def bn_finalize_stats(stats, count, gamma, beta, running_mean, running_var, eps, momentum, want_bound=False):
    """Coefficients of a train-mode BatchNorm from a convolution's statistics slots (``stats`` from
    conv5x5_fwd / convT5x5_fwd with want_stats): (mean, invstd, scale, shift); running statistics updated in place.
    ``want_bound``: a 5th result, the bound of max |act(BN(x))| an fp16-plane consumer needs (None in other arithmetics) --
    from the coefficients alone, no pass over x."""
    lib = _lib.load()
    _req(stats, "stats"), _req(gamma, "gamma"), _req(beta, "beta")
    C = gamma.numel()
    nslots = stats.numel() // (2 * C)
    out = torch.empty((4, C), dtype=torch.float32, device=gamma.device)
    mean, invstd, scale, shift = out[0], out[1], out[2], out[3]
    bound = _amax_slot(gamma.device) if (want_bound and _f16()) else None
    ws = workspace(lib.vg_bn_workspace_bytes(C), gamma.device)
    check(lib.vg_bn_finalize_stats(stats.data_ptr(), nslots, C, float(count), gamma.data_ptr(), beta.data_ptr(),
                                   _ptr(running_mean), _ptr(running_var), mean.data_ptr(), invstd.data_ptr(),
                                   scale.data_ptr(), shift.data_ptr(), eps, momentum, _ptr(bound), ws.data_ptr(), ws.numel(),
                                   _stream()), "vg_bn_finalize_stats")
    return (mean, invstd, scale, shift, bound) if want_bound else (mean, invstd, scale, shift)


def bn_stats(x, gamma, beta, running_mean, running_var, eps, momentum, want_bound=False):
    """The same coefficients from a pass over x (layers whose producer leaves no statistics)."""
    lib = _lib.load()
    _req(x, "x"), _req(gamma, "gamma"), _req(beta, "beta")
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    out = torch.empty((4, C), dtype=torch.float32, device=x.device)
    mean, invstd, scale, shift = out[0], out[1], out[2], out[3]
    bound = _amax_slot(x.device) if (want_bound and _f16()) else None
    ws = workspace(lib.vg_bn_workspace_bytes(C), x.device)
    check(lib.vg_bn_stats(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(running_mean), _ptr(running_var),
                          mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, C, HW, eps, momentum,
                          _ptr(bound), ws.data_ptr(), ws.numel(), _stream()), "vg_bn_stats")
    return (mean, invstd, scale, shift, bound) if want_bound else (mean, invstd, scale, shift)


def affine_act(x, scale, shift, act):
    """act(x * scale[c] + shift[c]) (the normalise pass of a BatchNorm whose coefficients are known)."""
    lib = _lib.load()
    _req(x, "x"), _req(scale, "scale"), _req(shift, "shift")
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x)
    slot = _amax_slot(x.device) if _f16() else None          # max |y| on the way out: y feeds a convolution
    check(lib.vg_affine_act(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), B, C, HW, int(act), _ptr(slot),
                            _stream()), "vg_affine_act")
    if slot is not None:
        set_amax(y, slot)
    return y


def bn_act_bwd(gy, x, gamma, beta, mean, invstd, act, need_param_grads=True, accumulate_into=None):
    """``accumulate_into`` = (dgamma, dbeta) tensors the parameter gradients are ADDED to (the layer's second use before
    one backward); they are then returned as they are."""
    lib = _lib.load()
    _req(gy, "gy"), _req(x, "x")
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    gx = torch.empty_like(x)
    if accumulate_into is not None:
        dgamma, dbeta = accumulate_into
    else:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if need_param_grads else None
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if need_param_grads else None
    ws = workspace(lib.vg_bn_workspace_bytes(C), x.device)
    slot = _amax_slot(x.device) if (_f16() and HW > 1) else None   # max |gx| on the way out: gx feeds a data / weight gradient
    check(lib.vg_bn_act_bwd(gy.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                            invstd.data_ptr(), gx.data_ptr(), _ptr(dgamma), _ptr(dbeta), B, C, HW, act,
                            1 if accumulate_into is not None else 0, _ptr(slot), ws.data_ptr(), ws.numel(), _stream()),
          "vg_bn_act_bwd")
    if slot is not None:
        set_amax(gx, slot)
    return gx, dgamma, dbeta


# ----------------------------------------------------------------- elementwise
def bias_act_fwd(x, bias, kind):
    lib = _lib.load()
    _req(x, "x")
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x)
    check(lib.vg_bias_act_fwd(x.data_ptr(), _ptr(bias), y.data_ptr(), B, C, HW, kind, _stream()),
          "vg_bias_act_fwd")
    return y


def act_bwd(gy, y, kind):
    lib = _lib.load()
    _req(gy, "gy"), _req(y, "y")
    gx = torch.empty_like(y)
    slot = _amax_slot(y.device) if (_f16() and LINEAR_SPLIT and y.dim() == 2) else None      # feeds a Linear layer's backward GEMMs
    check(lib.vg_act_bwd(gy.data_ptr(), y.data_ptr(), gx.data_ptr(), y.numel(), kind, _ptr(slot), _stream()), "vg_act_bwd")
    if slot is not None:
        set_amax(gx, slot)
    return gx


def scale_by_scalar(g, s):
    """g * s[0] with s a 0-dim device tensor (no host read)."""
    lib = _lib.load()
    _req(g, "g"), _req(s, "s")
    out = torch.empty_like(g)
    check(lib.vg_scale_by_scalar(g.data_ptr(), s.data_ptr(), out.data_ptr(), g.numel(), _stream()),
          "vg_scale_by_scalar")
    return out


# ---------------------------------------------------------------------- losses
def reparam_kl_fwd(mu, logvar, eps, beta, want_rows=False):
    lib = _lib.load()
    _req(mu, "mu"), _req(logvar, "logvar"), _req(eps, "eps")
    B, D = mu.shape
    z = torch.empty_like(mu)
    kl = torch.empty((), dtype=torch.float32, device=mu.device)
    rows = torch.empty(B, dtype=torch.float32, device=mu.device) if want_rows else None
    check(lib.vg_reparam_kl_fwd(mu.data_ptr(), logvar.data_ptr(), eps.data_ptr(), z.data_ptr(), kl.data_ptr(),
                                _ptr(rows), B, D, float(beta), _stream()), "vg_reparam_kl_fwd")
    return z, kl, rows


def reparam_kl_bwd(gz, mu, logvar, eps, gkl, beta):
    """gz: tensor or None; gkl: 0-dim device tensor (upstream grad of the KL scalar) or None."""
    lib = _lib.load()
    B, D = mu.shape
    gmu, glv = torch.empty_like(mu), torch.empty_like(mu)
    if gz is not None:
        _req(gz, "gz")
    if gkl is not None:
        _req(gkl, "gkl")
    check(lib.vg_reparam_kl_bwd(_ptr(gz), mu.data_ptr(), logvar.data_ptr(), eps.data_ptr(), _ptr(gkl), float(beta),
                                gmu.data_ptr(), glv.data_ptr(), B, D, _stream()), "vg_reparam_kl_bwd")
    return gmu, glv


def sqdiff_loss(a, b, scale, gscale=1.0, want_grad=True):
    lib = _lib.load()
    _req(a, "a"), _req(b, "b")
    if a.shape != b.shape:
        raise RuntimeError("sqdiff_loss: shape mismatch")
    loss = torch.empty((), dtype=torch.float32, device=a.device)
    ga = torch.empty_like(a) if want_grad else None
    ws = workspace(lib.vg_sqdiff_workspace_bytes(a.numel()), a.device)
    check(lib.vg_sqdiff_loss(a.data_ptr(), b.data_ptr(), loss.data_ptr(), _ptr(ga), a.numel(), float(scale),
                             float(gscale), ws.data_ptr(), ws.numel(), _stream()), "vg_sqdiff_loss")
    return loss, ga


def bce_loss(p, target, divisor=None, gscale=1.0, want_grad=True):
    """``target``: a Python float, or a one-element fp32 DEVICE tensor (a label that changes between replays of a
    captured iteration: the kernel reads it from memory)."""
    lib = _lib.load()
    _req(p, "p")
    B = p.numel()
    loss = torch.empty((), dtype=torch.float32, device=p.device)
    gp = torch.empty_like(p) if want_grad else None
    div = float(divisor if divisor is not None else B)
    if isinstance(target, torch.Tensor):
        _req(target, "target")
        if target.numel() != 1:
            raise RuntimeError("bce_loss: a device label is one fp32 value")
        check(lib.vg_bce_loss_dev(p.data_ptr(), target.data_ptr(), loss.data_ptr(), _ptr(gp), B, div, float(gscale),
                                  _stream()), "vg_bce_loss_dev")
    else:
        check(lib.vg_bce_loss(p.data_ptr(), float(target), loss.data_ptr(), _ptr(gp), B, div, float(gscale), _stream()),
              "vg_bce_loss")
    return loss, gp


def dot_sigmoid_bce_fwd(feat, w, bias, target, divisor=None, want_grad=True):
    """The discriminator's head + its BCE in one launch: p = sigmoid(feat @ w + bias) (B,), the mean BCE of
    p against ``target`` (float or one-element device tensor) over ``divisor``, and dlogit = d loss / d logit."""
    lib = _lib.load()
    _req(feat, "feat"), _req(w, "w")
    B, K = feat.shape
    if w.numel() != K:
        raise RuntimeError("dot_sigmoid_bce: weight does not match the features")
    p = torch.empty(B, dtype=torch.float32, device=feat.device)
    loss = torch.empty((), dtype=torch.float32, device=feat.device)
    dlogit = torch.empty(B, dtype=torch.float32, device=feat.device) if want_grad else None
    dev = isinstance(target, torch.Tensor)
    if dev:
        _req(target, "target")
    ws = workspace(lib.vg_dot_sigmoid_bce_workspace_bytes(B), feat.device)
    check(lib.vg_dot_sigmoid_bce_fwd(feat.data_ptr(), w.data_ptr(), _ptr(bias), 0.0 if dev else float(target),
                                     target.data_ptr() if dev else None, p.data_ptr(), loss.data_ptr(), _ptr(dlogit), B, K,
                                     float(divisor if divisor is not None else B), ws.data_ptr(), ws.numel(), _stream()),
          "vg_dot_sigmoid_bce_fwd")
    return p, loss, dlogit


def dot_sigmoid_bce_bwd(dlogit, gloss, feat, w, need_feat=True, need_w=True, need_b=True, accumulate_into=None):
    """``accumulate_into`` = (gw, gb) tensors the parameter gradients are added to (see bn_act_bwd)."""
    lib = _lib.load()
    B, K = feat.shape
    gfeat = torch.empty_like(feat) if need_feat else None
    if accumulate_into is not None:
        gw, gb = accumulate_into
    else:
        gw = torch.empty(w.shape, dtype=torch.float32, device=feat.device) if need_w else None
        gb = torch.empty(1, dtype=torch.float32, device=feat.device) if need_b else None
    check(lib.vg_dot_sigmoid_bce_bwd(dlogit.data_ptr(), _ptr(gloss), feat.data_ptr(), w.data_ptr(), _ptr(gfeat), _ptr(gw),
                                     _ptr(gb), B, K, 1 if accumulate_into is not None else 0, _stream()),
          "vg_dot_sigmoid_bce_bwd")
    return gfeat, gw, gb


# ---------------------------------------------------------------- image I/O
def u8_gather_normalize(images_u8, index, mean=0.5, std=0.5):
    """images_u8 [N,H,W,C] uint8 (device), index int64 (device) -> fp32 [B,C,H,W] =
    (u8 / 255 - mean) / std  (ToTensor + Normalize, dataset.py:37-43)."""
    lib = _lib.load()
    if not (isinstance(images_u8, torch.Tensor) and images_u8.is_cuda and images_u8.dtype == torch.uint8
            and images_u8.dim() == 4 and images_u8.is_contiguous()):
        raise RuntimeError("u8_gather_normalize: image cache must be a contiguous CUDA/ROCm uint8 [N,H,W,C] tensor")
    if not (isinstance(index, torch.Tensor) and index.is_cuda and index.dtype == torch.int64 and index.dim() == 1):
        raise RuntimeError("u8_gather_normalize: index must be a 1-D CUDA/ROCm int64 tensor")
    index = index.contiguous()
    N, H, W, C = images_u8.shape
    B = index.numel()
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=images_u8.device)
    for s in range(0, B, 65535):                      # grid.y limit
        n = min(65535, B - s)
        check(lib.vg_u8_gather_normalize(images_u8.data_ptr(), index.data_ptr() + 8 * s, out[s:].data_ptr(), n, C, H, W,
                                         float(mean), float(std), _stream()), "vg_u8_gather_normalize")
    return out


def minmax(x):
    """Device tensor [min(x), max(x)] (no host sync)."""
    lib = _lib.load()
    _req(x, "x")
    n = x.numel()
    nbytes = lib.vg_minmax_workspace_bytes(n)
    ws = workspace(nbytes, x.device)
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    check(lib.vg_minmax(x.data_ptr(), n, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "vg_minmax")
    return out


def image_grid_u8(x, nrow=8, padding=2, normalize=False, pad_value=0.0):
    """torchvision 0.2.1 make_grid + save_image quantisation on the device: x [B,C,H,W] (C = 1 or 3)
    -> uint8 [GH,GW,3]."""
    import ctypes
    lib = _lib.load()
    _req(x, "x")
    if x.dim() == 3:
        x = x.unsqueeze(0)
    B, C, H, W = x.shape
    gh, gw = ctypes.c_int(), ctypes.c_int()
    check(lib.vg_image_grid_shape(B, H, W, nrow, padding, ctypes.byref(gh), ctypes.byref(gw)), "vg_image_grid_shape")
    mm = minmax(x) if normalize else None
    grid = torch.empty((gh.value, gw.value, 3), dtype=torch.uint8, device=x.device)
    check(lib.vg_image_grid_u8(x.data_ptr(), _ptr(mm), grid.data_ptr(), B, C, H, W, nrow, padding, float(pad_value),
                               _stream()), "vg_image_grid_u8")
    return grid
