"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/vaegan_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib_path():
    from disentangle_mlp_amd import build
    return build.build(verbose=False)


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vaegan_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vg_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from disentangle_mlp_amd import _lib
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_symbols():
        assert getattr(lib, name) is not None, name


def test_pure_host_entry_points(lib_path):
    """Workspace-size queries and argument validation run without a device."""
    from disentangle_mlp_amd import _lib
    lib = _lib.load()
    assert lib.vg_version() == 1
    assert lib.vg_conv5x5_wgrad_workspace_bytes(128, 128, 32, 32, 256, 2) > 0
    assert lib.vg_conv5x5_wgrad_workspace_bytes(128, 128, 32, 32, 256, 3) == 0     # bad stride
    assert lib.vg_bn_workspace_bytes(256) >= 256 * 64 * 16
    assert lib.vg_sqdiff_workspace_bytes(10) > 0
    # rejected arguments return VG_ERR_BAD_ARG before any launch
    assert lib.vg_conv5x5_fwd(None, None, None, None, 1, 1, 8, 8, 1, 2, None) == -1
    assert lib.vg_convT5x5_fwd(None, None, None, None, 1, 1, 8, 8, 1, 2, None) == -1
    assert lib.vg_bce_loss(None, 0.9, None, None, 4, 4.0, 1.0, None) == -1


def test_ops_refuse_cpu_tensors():
    import torch
    from disentangle_mlp_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.bn_act_fwd(torch.zeros(2, 3, 4, 4), torch.ones(3), torch.zeros(3), None, None, 1e-5, 0.1, 1)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from disentangle_mlp_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
