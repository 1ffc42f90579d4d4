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


def test_round5_entries_reject_bad_params_without_gpu(lib):
    """Argument checks of the round-5 entries (visibility plan, one-kernel stem) that return before any device call:
    status codes as helper.h:19-25 (2 = BAD_PARAM, 3 = NOT_SUPPORTED), sizes as plain arithmetic."""
    f = ctypes.c_float
    lib.bevops_sca_plan_size.restype = ctypes.c_size_t
    lib.bevops_stem_packed_size.restype = ctypes.c_size_t
    assert lib.bevops_sca_plan_size(6, 40000) == 64 + 6 * 40000 * 4 + 6 * 1024    # counts + 32-bit entries + builder scratch
    assert lib.bevops_sca_plan_size(6, 100) == 64 + 6 * 128 * 4 + 6 * 1024        # lists padded to 64 entries
    assert lib.bevops_sca_plan_size(17, 100) == 0 and lib.bevops_sca_plan_size(6, 65536) == 0 and lib.bevops_sca_plan_size(0, 5) == 0
    assert lib.bevops_stem_packed_size() == 11 * 2 * 64 * 8 * 2
    buf = (ctypes.c_char * 256)()
    p = ctypes.addressof(buf)
    assert lib.bevops_sca_plan_build(1, None, 6, 100, p, ctypes.c_size_t(4096), None) == 2          # no mask
    assert lib.bevops_sca_plan_build(1, p, 0, 100, p, ctypes.c_size_t(4096), None) == 2             # no cameras
    assert lib.bevops_sca_plan_build(0, p, 6, 100, p, ctypes.c_size_t(4096), None) == 3             # fp32 mask
    assert lib.bevops_stem_pack(1, None, None, p, None) == 2
    assert lib.bevops_stem_pack(0, p, None, p, None) == 3
    assert lib.bevops_stem_conv_pool(1, 1, None, p, p, 1, 16, 16, f(0), None) == 2
    assert lib.bevops_stem_conv_pool(1, 1, p, p, p, 1, 16, 15, f(0), None) == 3                     # odd width
    assert lib.bevops_stem_conv_pool(1, 0, p, p, p, 1, 16, 16, f(0), None) == 3                     # fp32 output
    assert lib.bevops_stem_conv_pool(0, 1, p, p, p, 1, 16, 16, f(0), None) == 3                     # fp32 input
    assert lib.bevops_stem_conv_pool(1, 2, p, p, p, 1, 16, 16, f(0), None) == 2                     # int8 out, no scale
    assert lib.bevops_stem_conv_pool(1, 1, p, p, p, -1, 16, 16, f(0), None) == 2
    assert lib.bevops_stem_set_variant(1) == 0 and lib.bevops_stem_set_variant(0) == 1              # returns the previous value
    host_shapes = (ctypes.c_int32 * 8)(116, 200, 58, 100, 29, 50, 15, 25)
    st = lib.bevops_sca_forward_planned(1, p, ctypes.c_size_t(64), ctypes.addressof(host_shapes), p, p, p, p, None,
                                        ctypes.c_size_t(0), p, 6, 30825, 8, 32, 4, 40000, 8, 4, p, ctypes.c_size_t(0), None)
    assert st == 2      # no plan
    # round 6: the camera projection of the BEV pillars (bevops_point_sampling)
    assert lib.bevops_point_sampling(1, None, p, p, p, 6, 100, 4, f(928), f(1600), None) == 2      # no pillars
    assert lib.bevops_point_sampling(1, p, p, p, p, 0, 100, 4, f(928), f(1600), None) == 2         # no cameras
    assert lib.bevops_point_sampling(1, p, p, p, p, 6, 100, 4, f(0), f(1600), None) == 2           # empty image
    assert lib.bevops_point_sampling(1, p, p, p, p, 6, 100, 3, f(928), f(1600), None) == 3         # three anchors per pillar
    assert lib.bevops_point_sampling(2, p, p, p, p, 6, 100, 4, f(928), f(1600), None) == 3         # int8 output
    assert lib.bevops_point_sampling(1, p + 4, p, p, p, 6, 100, 4, f(928), f(1600), None) == 2     # misaligned pillars
    # round 6: the GEMM with the LayerNorm in its epilogue
    ll = ctypes.c_longlong
    assert lib.bevops_tsgemm_f16_ln(p, p, None, None, None, p, f(1e-5), p, ll(64), 256, 64, None) == 2    # no norm weight
    assert lib.bevops_tsgemm_f16_ln(p, p, None, None, p, p, f(1e-5), p, ll(64), 512, 64, None) == 3       # N != 256
    assert lib.bevops_tsgemm_f16_ln(p, p, None, None, p, p, f(1e-5), p, ll(64), 256, 48, None) == 3       # K % 64
    assert lib.bevops_tsgemm_f16_ln(p, p, None, None, p, p, f(-1.0), p, ll(64), 256, 64, None) == 2       # negative eps
    # round 6: the decoder's self-attention
    lib.bevops_mha_selfattn_max_queries.restype = ctypes.c_size_t
    assert lib.bevops_mha_selfattn_max_queries() == 1024
    assert lib.bevops_mha_selfattn_f16(None, p, 900, 8, 32, f(0.17), None) == 2
    assert lib.bevops_mha_selfattn_f16(p, p, 900, 8, 64, f(0.17), None) == 3          # head width
    assert lib.bevops_mha_selfattn_f16(p, p, 2000, 8, 32, f(0.17), None) == 3         # more keys than LDS holds
    assert lib.bevops_mha_selfattn_f16(p, p, 900, 8, 32, f(0.0), None) == 2           # no scale


def test_sca_knobs_do_not_disturb_the_kernel_family_selection(lib):
    """The 30xx values of bevops_msda_set_variant are knobs of the fused SCA op, not kernel-family selectors: after one of
    them, a save / restore pair around a family switch (what multi_scale_deformable_attn_local does with variant 10)
    must leave the default family in force -- the base SCA shape keeps its head-major workspace (round-5 advisor: the
    knob used to be handed back as "previous" and the restore then left variant 10 behind for the whole thread)."""
    lib.bevops_msda_workspace_size.restype = ctypes.c_size_t
    base_sca_i8 = (2, 6, 30825, 8, 32, 4, 40000, 8)           # BEVOPS_I8: its size is 0 under "never head-major"
    assert lib.bevops_msda_set_variant(0) in (0, 10, 19)       # whatever an earlier test of this process left
    assert lib.bevops_msda_workspace_size(*base_sca_i8) > 0
    for knob in (3002, 3010, 3012):
        assert lib.bevops_msda_set_variant(knob) == 0          # knobs report the family request, and do not become it
        prev = lib.bevops_msda_set_variant(10)
        assert prev == 0
        assert lib.bevops_msda_workspace_size(*base_sca_i8) == 0
        assert lib.bevops_msda_set_variant(prev) == 10
        assert lib.bevops_msda_workspace_size(*base_sca_i8) > 0, "variant 10 survived the restore after knob %d" % knob
    assert lib.bevops_msda_set_variant(19) == 0 and lib.bevops_msda_set_variant(0) == 19   # family values round-trip


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
