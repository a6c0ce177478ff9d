"""Vendor-GEMM algorithm table for the Linear layers (SURVEY.md K7 leaves them to hipBLASLt / rocBLAS).

PyTorch's TunableOp picks, per GEMM shape, the fastest of the algorithms hipBLASLt and rocBLAS offer.  The table for the
shapes of the beta-VAE-GAN iteration at the benchmark's batch (16384 <-> 2048 forward / data gradient / weight gradient
at M = 128 and 256, 128 -> 16384, ...) was measured once on an MI355X (`scripts/tune_gemms.py`, which also checks every
chosen algorithm for run-to-run bit reproducibility) and is shipped as ``tuned/gfx950_gemm.csv``: weight gradient 97 -> 70
us, forward 86 -> 69 us, -0.36 ms per iteration.  `enable()` switches TunableOp on in LOOK-UP mode only (no tuning at run
time, nothing written): shapes that are not in the table, or a table whose validators (PyTorch / ROCm library versions,
gfx arch) do not match this installation, take the libraries' default choice -- results stay correct either way.
``VG_TUNED_GEMMS=0`` leaves TunableOp alone."""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
TABLE = os.path.join(_HERE, "tuned", "gfx950_gemm.csv")
_state = {"done": False, "active": False}


def enable(path=None):
    """Idempotent.  Returns True when the table is in use."""
    if _state["done"]:
        return _state["active"]
    _state["done"] = True
    if os.environ.get("VG_TUNED_GEMMS", "1") == "0" or os.environ.get("PYTORCH_TUNABLEOP_ENABLED"):
        return False           # switched off, or the user drives TunableOp themselves
    import torch
    path = path or TABLE
    if not (torch.cuda.is_available() and os.path.exists(path)):
        return False
    try:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(False)
        if hasattr(tunable, "record_untuned_enable"):
            tunable.record_untuned_enable(False)
        ok = bool(tunable.read_file(path))
        if not ok:
            tunable.enable(False)
        _state["active"] = ok
    except Exception:          # an installation without TunableOp: the default algorithms
        _state["active"] = False
    return _state["active"]


def disable():
    """Switch TunableOp off again (it is process-wide: a user whose other GEMMs should not go through the look-up)."""
    if _state["active"]:
        import torch.cuda.tunable as tunable
        tunable.enable(False)
    _state["done"], _state["active"] = True, False
