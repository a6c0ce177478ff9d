"""ctypes binding of libvaegan_hip.so (include/vaegan_hip.h).

The library is the product's only compute path for the convolutions, BatchNorm,
activations and losses.  There is NO CPU or PyTorch fallback: if the shared
object is missing or a symbol is absent, import fails loudly.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_size_t, c_void_p

# torch must be imported first: its bundled HIP runtime (libamdhip64) then serves this
# library too, so kernels launch in the same HIP context that owns the tensors' memory.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvaegan_hip.so")
ABI_VERSION = 5      # VG_ABI_VERSION of include/vaegan_hip.h this binding was written against

_P, _I, _F, _Z = c_void_p, c_int, c_float, c_size_t

# name -> (restype, argtypes); mirrors include/vaegan_hip.h one to one
SIGNATURES = {
    "vg_version": (_I, []),
    "vg_conv5x5_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_convT5x5_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_conv5x5_packed_floats": (_Z, [_I, _I]),
    "vg_conv5x5_pack": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "vg_conv5x5_fwd_packed": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_convT5x5_fwd_packed": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_conv5x5_fwd_packed_stats_floats": (_Z, [_I, _I, _I, _I, _I, _I]),
    "vg_conv5x5_fwd_packed_stats": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "vg_conv5x5_packed_bf16split_bytes": (_Z, [_I, _I, _I]),
    "vg_conv5x5_pack_bf16split": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "vg_gemm_nt_f16x3_workspace_bytes": (_Z, [_I, _I, _I]),
    "vg_gemm_nt_f16x3": (_I, [_P, _P, _P, _P, _I, _I, _I, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_long, _P, _P,
                              _P, _Z, _P]),
    "vg_absmax": (_I, [_P, _Z, _P, _P]),
    "vg_absmax_affine": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "vg_absmax_multi": (_I, [_P, _I, _P]),
    "vg_conv5x5_pack_bf16split_multi": (_I, [_P, _I, _I, _P]),
    "vg_convT5x5_fwd_bf16split_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "vg_convT5x5_fwd_bf16split": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _P]),
    "vg_conv5x5_fwd_bf16split_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "vg_conv5x5_bf16split_fusable": (_I, [_I, _I, _I, _I]),
    "vg_conv5x5_fwd_bf16split_stats_floats": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "vg_convT5x5_fwd_bf16split_stats_floats": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "vg_convT5x5_s1_thin_bf16split_ok": (_I, [_I, _I, _I, _I]),
    "vg_conv5x5_thin_wgrad_bf16split_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "vg_conv5x5_thin_wgrad_bf16split": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _P, _I, _I, _P]),
    "vg_conv5x5_thin_bf16split_ok": (_I, [_I, _I, _I, _I, _I]),
    "vg_conv5x5_thin_bf16split_stats_floats": (_Z, [_I, _I, _I, _I, _I, _I]),
    "vg_conv5x5_thin_bf16split": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "vg_convT5x5_s1_thin_bf16split": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "vg_conv5x5_fwd_bf16split": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _P]),
    "vg_conv5x5_wgrad_bf16split_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "vg_conv5x5_wgrad_bf16split": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _P, _I, _I, _P, _P, _I, _P]),
    "vg_conv5x5_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "vg_conv5x5_wgrad": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _I, _P]),
    "vg_channel_sum": (_I, [_P, _P, _I, _I, _I, _P, _Z, _P]),
    "vg_bn_workspace_bytes": (_Z, [_I]),
    "vg_bn_act_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _P, _P, _Z, _P]),
    "vg_bn_act_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P]),
    "vg_bn_finalize_stats": (_I, [_P, _I, _I, ctypes.c_double, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _Z, _P]),
    "vg_bn_stats": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _Z, _P]),
    "vg_affine_act": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "vg_bias_act_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "vg_act_bwd": (_I, [_P, _P, _P, _Z, _I, _P, _P]),
    "vg_reparam_kl_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "vg_reparam_kl_bwd": (_I, [_P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _P]),
    "vg_scale_by_scalar": (_I, [_P, _P, _P, _Z, _P]),
    "vg_sqdiff_workspace_bytes": (_Z, [_Z]),
    "vg_sqdiff_loss": (_I, [_P, _P, _P, _P, _Z, _F, _F, _P, _Z, _P]),
    "vg_bce_loss": (_I, [_P, _F, _P, _P, _I, _F, _F, _P]),
    "vg_adam_step": (_I, [_P, _I] + [ctypes.c_double] * 6 + [_P]),
    "vg_adam_prepare": (_I, [ctypes.c_double, _P, _I, ctypes.c_double, ctypes.c_double, ctypes.c_double, _P, _P]),
    "vg_adam_step_dev": (_I, [_P, _I] + [ctypes.c_double] * 3 + [_P, _P]),
    "vg_bce_loss_dev": (_I, [_P, _P, _P, _P, _I, _F, _F, _P]),
    "vg_dot_sigmoid_bce_workspace_bytes": (_Z, [_I]),
    "vg_dot_sigmoid_bce_fwd": (_I, [_P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _F, _P, _Z, _P]),
    "vg_dot_sigmoid_bce_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vg_u8_gather_normalize": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _F, _P]),
    "vg_minmax_workspace_bytes": (_Z, [_Z]),
    "vg_minmax": (_I, [_P, _Z, _P, _P, _Z, _P]),
    "vg_image_grid_shape": (_I, [_I, _I, _I, _I, _I, _P, _P]),
    "vg_image_grid_u8": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
}

# the tuning build (-DVG_TUNING) adds the process-global knobs of include/vaegan_hip.h's last section
TUNING_LIB_PATH = os.path.join(_HERE, "libvaegan_hip_tuning.so")
TUNING_SIGNATURES = {
    "vg_debug_set_conv_tile": (_I, [_I, _I]),
    "vg_debug_set_wgrad": (_I, [_I, _I]),
    "vg_debug_set_conv_bf16split_tile": (_I, [_I]),
    "vg_debug_set_conv_ring_tile": (_I, [_I]),
}

_lib = None          # the library ops.py calls: the product one unless a test / script switched to the tuning build
_product = None
_tuning = None


def _open(path, signatures):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -m disentangle_mlp_amd.build` "
            "(hipcc --offload-arch=gfx950). disentangle_mlp_amd has no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in signatures.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    built = lib.vg_version()
    if built != ABI_VERSION:      # a stale build product: same symbols, shifted arguments / old pack layouts
        raise ImportError(f"{path} was built for ABI {built}, this package binds ABI {ABI_VERSION}: rebuild it with "
                          "`python -m disentangle_mlp_amd.build`")
    return lib


def load():
    """The active HIP library (the product build unless `use_tuning(True)`); raises -- never falls back -- when
    it is absent."""
    global _lib, _product
    if _lib is not None:
        return _lib
    if _product is None:
        _product = _open(LIB_PATH, SIGNATURES)
    _lib = _product
    return _lib


def load_tuning():
    """libvaegan_hip_tuning.so: same kernels plus the vg_debug_* knobs.  Tests and tuning scripts only."""
    global _tuning
    if _tuning is None:
        _tuning = _open(TUNING_LIB_PATH, {**SIGNATURES, **TUNING_SIGNATURES})
    return _tuning


class use_tuning:
    """Context manager: route every op through the tuning build (so that a forced tile variant takes effect), and
    back.  ``with use_tuning() as lib: lib.vg_debug_set_conv_tile(0, 3); ...``"""

    def __enter__(self):
        global _lib
        self._prev = load()
        _lib = load_tuning()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._prev
        return False


class ConvFusion(ctypes.Structure):
    """vg_conv_fusion of include/vaegan_hip.h."""
    _fields_ = [("in_scale", c_void_p), ("in_shift", c_void_p), ("in_act", c_int), ("stats", c_void_p),
                ("stats_floats", c_size_t), ("in_amax", c_void_p)]


class PackEntry(ctypes.Structure):
    """VgPackEntry of include/vaegan_hip.h."""
    _fields_ = [("w", c_void_p), ("packed", c_void_p), ("Cout", c_int), ("Cin", c_int), ("transposed", c_int),
                ("stride", c_int), ("w_amax", c_void_p)]


class AbsmaxEntry(ctypes.Structure):
    """VgAbsmaxEntry of include/vaegan_hip.h."""
    _fields_ = [("x", c_void_p), ("n", c_size_t), ("amax", c_void_p)]


class HipKernelError(RuntimeError):
    pass


def check(status, name):
    if status != 0:
        what = {-1: "bad argument", -2: "workspace too small"}.get(status, f"hipError_t {status}")
        raise HipKernelError(f"{name} failed: {what}")
