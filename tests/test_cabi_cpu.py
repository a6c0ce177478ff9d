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


def declared_symbols(tuning=False):
    """Entry points the header declares: the product ABI, or the `#ifdef VG_TUNING` section (the process-global
    knobs that exist only in libvaegan_hip_tuning.so)."""
    text = open(os.path.join(ROOT, "include", "vaegan_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    m = re.search(r"#ifdef VG_TUNING(.*?)#endif", text, flags=re.S)
    assert m, "the header keeps its tuning-only declarations in one #ifdef VG_TUNING block"
    text = m.group(1) if tuning else text[:m.start()] + text[m.end():]
    return sorted(set(re.findall(r"\b(vg_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from disentangle_mlp_amd import _lib
    assert declared_symbols() == sorted(_lib.SIGNATURES)
    assert declared_symbols(tuning=True) == sorted(_lib.TUNING_SIGNATURES)
    assert all(n.startswith("vg_debug_") for n in _lib.TUNING_SIGNATURES)


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_symbols():
        assert getattr(lib, name) is not None, name


def test_product_library_has_no_tuning_knobs(lib_path):
    """include/vaegan_hip.h promises "no global mutable state": the vg_debug_* setters (and the globals behind
    them) are compiled only into the tuning twin, which exports the whole product ABI as well."""
    from disentangle_mlp_amd import _lib
    lib = ctypes.CDLL(lib_path)
    for name in declared_symbols(tuning=True):
        assert not hasattr(lib, name), name
    tun = ctypes.CDLL(_lib.TUNING_LIB_PATH)
    for name in declared_symbols() + declared_symbols(tuning=True):
        assert getattr(tun, name) is not None, name


def header_abi_version():
    import re
    with open(os.path.join(ROOT, "include", "vaegan_hip.h")) as f:
        return int(re.search(r"#define\s+VG_ABI_VERSION\s+(\d+)", f.read()).group(1))


def test_binding_refuses_a_library_of_another_abi_version(monkeypatch):
    """The .so files are build products that travel with the working tree: one built before an incompatible change
    still exports every symbol, so the binding compares vg_version() with the version it was written against."""
    from disentangle_mlp_amd import _lib
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(ImportError, match="rebuild"):
        _lib._open(_lib.LIB_PATH, _lib.SIGNATURES)


def test_pure_host_entry_points(lib_path):
    """Workspace-size queries and argument validation run without a device."""
    from disentangle_mlp_amd import _lib
    lib = _lib.load()
    assert lib.vg_version() == _lib.ABI_VERSION == header_abi_version()
    assert lib.vg_conv5x5_wgrad_workspace_bytes(128, 128, 32, 32, 256, 2) > 0
    assert lib.vg_conv5x5_wgrad_workspace_bytes(128, 128, 32, 32, 256, 3) == 0     # bad stride
    assert lib.vg_bn_workspace_bytes(256) >= 256 * 64 * 16
    assert lib.vg_sqdiff_workspace_bytes(10) > 0
    # rejected arguments return VG_ERR_BAD_ARG before any launch
    assert lib.vg_conv5x5_fwd(None, None, None, None, 1, 1, 8, 8, 1, 2, None) == -1
    assert lib.vg_convT5x5_fwd(None, None, None, None, 1, 1, 8, 8, 1, 2, None) == -1
    assert lib.vg_bce_loss(None, 0.9, None, None, 4, 4.0, 1.0, None) == -1
    # split-bf16 convolutions: K-split workspaces are sized on the host (conv_ring.hip's plan)
    assert lib.vg_conv5x5_fwd_bf16split_workspace_bytes(128, 128, 32, 32, 256, 2, 3) == 0          # 256 tiles: no split
    assert lib.vg_conv5x5_fwd_bf16split_workspace_bytes(128, 256, 16, 16, 256, 2, 3) == 4 * 128 * 256 * 8 * 8 * 4
    assert lib.vg_conv5x5_fwd_bf16split_workspace_bytes(128, 256, 16, 16, 256, 2, 2 | 0x100) == 4 * 128 * 256 * 8 * 8 * 4
    assert lib.vg_convT5x5_fwd_bf16split_workspace_bytes(128, 256, 16, 16, 128, 2, 3) == 0
    # packs: chunks * 25 steps + 6 spare zero steps (the DMA ring's run-ahead), [plane][k-block][cout] x 16 bytes;
    # fp16 planes (2 | VG_PLANES_F16): + the 16-byte trailer with the inverse of the filter's scale
    assert lib.vg_conv5x5_packed_bf16split_bytes(256, 128, 3) == (8 * 25 + 6) * 2 * 3 * 256 * 16
    assert lib.vg_conv5x5_packed_bf16split_bytes(256, 128, 2 | 0x100) == (8 * 25 + 6) * 2 * 2 * 256 * 16 + 16
    assert lib.vg_conv5x5_packed_bf16split_bytes(256, 128, 3 | 0x100) == 0                     # fp16 planes: two of them
    assert lib.vg_conv5x5_wgrad_bf16split_workspace_bytes(128, 128, 32, 32, 256, 2, 2 | 0x100) > 0
    assert lib.vg_absmax(None, 16, None, None) == -1
    # Linear GEMM: the K split (and with it the slab workspace) is decided on the host -- 128 x 128 tiles, as many splits
    # as bring the grid to about 256 workgroups while every split keeps >= 8 stages of 32
    assert lib.vg_gemm_nt_f16x3_workspace_bytes(128, 2048, 16384) == 16 * 128 * 2048 * 4        # 16 tiles x 16 splits
    assert lib.vg_gemm_nt_f16x3_workspace_bytes(128, 16384, 2048) == 2 * 128 * 16384 * 4        # data gradient: 128 tiles x 2
    assert lib.vg_gemm_nt_f16x3_workspace_bytes(2048, 16384, 128) == 0                          # weight gradient: 2048 tiles
    assert lib.vg_gemm_nt_f16x3_workspace_bytes(128, 2048, 100) == 0                            # K % 32: not taken
    assert lib.vg_gemm_nt_f16x3(None, None, None, None, 128, 2048, 16384, 16384, 1, 16384, 1, None, None, None, 0, None) == -1


def test_ops_refuse_cpu_tensors():
    import torch
    from disentangle_mlp_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.bn_act_fwd(torch.zeros(2, 3, 4, 4), torch.ones(3), torch.zeros(3), None, None, 1e-5, 0.1, 1)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from disentangle_mlp_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_product", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
