"""CPU-only: the C-ABI library loads, exports every symbol include/bevops.h
declares, and validates arguments before touching a device."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from bevformer_tensorrt_amd.utils import load_library
    return load_library()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bevops.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bevops_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    syms = _declared_symbols()
    assert "bevops_msda_forward" in syms
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/bevops.h but not exported"


def test_ctypes_signatures_cover_header():
    from bevformer_tensorrt_amd.utils.lib import SIGNATURES
    assert sorted(SIGNATURES) == _declared_symbols()


def test_version_and_status_strings(lib):
    assert b"gfx950" in lib.bevops_version()
    assert lib.bevops_status_string(0) == b"success"
    assert lib.bevops_status_string(2) == b"bad parameter"


def test_query_table(lib):
    for name in (b"bevops_msda_forward", b"MultiScaleDeformableAttnTRT",
                 b"MultiScaleDeformableAttnTRT2"):
        assert lib.bevops_query(name), name
    assert not lib.bevops_query(b"NoSuchPlugin")
    assert not lib.bevops_query(None)
    # every operator entry of the header resolves through the registry to the exported symbol itself
    # (identification / A-B switches are not operators)
    skip = {"bevops_version", "bevops_status_string", "bevops_query"}
    for s in _declared_symbols():
        if s in skip or s.endswith("_set_variant"):
            continue
        addr = lib.bevops_query(s.encode())
        assert addr, f"{s} missing from bevops_query's table"
        assert addr == ctypes.cast(getattr(lib, s), ctypes.c_void_p).value, s
    # all ten reference plugin type names (both versions) resolve
    for plugin in ("MultiScaleDeformableAttnTRT", "RotateTRT", "GridSampler2DTRT", "GridSampler3DTRT",
                   "BEVPoolV2TRT", "ModulatedDeformableConv2dTRT"):
        for v in ("", "2"):
            assert lib.bevops_query((plugin + v).encode()), plugin + v


def test_msda_rejects_bad_params_without_gpu(lib):
    f = ctypes.c_float
    # null pointers -> BAD_PARAM (2), mirrors helper.h:22 STATUS_BAD_PARAM
    st = lib.bevops_msda_forward(0, None, None, None, None, 0, None, None, None,
                                 1, 1, 1, 1, 1, 1, 1, 1, f(1), f(1), f(1), f(1), None)
    assert st == 2
    buf = (ctypes.c_char * 64)()
    p = ctypes.addressof(buf)
    st = lib.bevops_msda_forward(0, p, p, None, p, 0, p, p, p,
                                 0, 1, 1, 1, 1, 1, 1, 1, f(1), f(1), f(1), f(1), None)
    assert st == 2  # bs == 0
    host_shapes = (ctypes.c_int32 * 2)(3, 3)
    st = lib.bevops_msda_forward(0, p, p, ctypes.addressof(host_shapes), p, 0, p, p, p,
                                 1, 10, 1, 32, 1, 1, 4, 1, f(1), f(1), f(1), f(1), None)
    assert st == 2  # nk != sum(h*w)


def test_registry_mirrors_reference_names():
    import bevformer_tensorrt_amd as bev
    for name in ("multi_scale_deformable_attn", "multi_scale_deformable_attn2"):
        assert name in bev.TRT_FUNCTIONS
        assert bev.TRT_FUNCTIONS.get(name) is getattr(bev, name)
    with pytest.raises(KeyError):
        bev.TRT_FUNCTIONS.register_module(module=bev.multi_scale_deformable_attn)


def test_ops_require_gpu_tensor():
    import torch
    import bevformer_tensorrt_amd as bev
    v = torch.zeros(1, 4, 1, 32)
    with pytest.raises(AssertionError):
        bev.multi_scale_deformable_attn(v, torch.tensor([[2, 2]]), torch.zeros(1, 1, 1, 2),
                                        torch.zeros(1, 1, 1, 8), torch.zeros(1, 1, 1, 4))


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from bevformer_tensorrt_amd.utils import lib as L
    monkeypatch.setattr(L, "_LIB", None)
    monkeypatch.setenv("BEVOPS_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no fallback"):
        L.load_library()
