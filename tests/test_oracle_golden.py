"""Pin the CPU oracle against golden vectors produced by the reference's own
Python code (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import golden

MSDA_CASES = ["sca_like", "tsa_like", "tiny_sca", "edges", "odd_lp"]


@pytest.mark.parametrize("case", MSDA_CASES)
def test_msda_oracle_matches_reference_fp32(oracle_mod, case):
    g = golden("msda_" + case)
    out = oracle_mod.msda_f32(g["value"], g["shapes"], g["ref"], g["off"], g["logit"])
    # reference test tolerance is mean-abs 1e-5 (test_multi_scale_deformable_attn.py:139-141);
    # we hold the oracle to an element-wise bound
    np.testing.assert_allclose(out, g["out_fp32"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("case", MSDA_CASES)
def test_msda_oracle_vs_reference_fp16_eager(oracle_mod, case):
    """The reference's fp16 eager path does its pre-processing (locations, softmax)
    in half before upcasting (functions/multi_scale_deformable_attn.py:58-101); the
    oracle computes in fp32 from the same fp16-rounded inputs.  They agree within
    the reference's own fp16 tolerance (mean abs err <= 0.01,
    test_multi_scale_deformable_attn.py:142-144)."""
    g = golden("msda_" + case)
    h = lambda a: a.astype(np.float16)
    out = oracle_mod.msda_f32(h(g["value"]), g["shapes"], h(g["ref"]), h(g["off"]),
                              h(g["logit"]))
    err = np.abs(out - g["out_fp16_eager"].astype(np.float32))
    assert err.mean() <= 0.01


@pytest.mark.parametrize("case", MSDA_CASES)
def test_torch_port_matches_reference(case):
    """oracle/torch_ref.py (the CPU-baseline port) against the reference's output."""
    import torch
    from oracle import torch_ref
    g = golden("msda_" + case)
    t = lambda k: torch.from_numpy(g[k])
    out = torch_ref.msda(t("value"), t("shapes").long(), t("ref"), t("off"), t("logit"))
    np.testing.assert_allclose(out.numpy(), g["out_fp32"], rtol=1e-5, atol=2e-6)


MODES = ("bilinear", "nearest", "bicubic")
PADS = ("zeros", "border", "reflection")


@pytest.mark.parametrize("mode", range(3))
@pytest.mark.parametrize("pad", range(3))
@pytest.mark.parametrize("align", [False, True])
def test_grid_sampler_2d_oracle(oracle_mod, mode, pad, align):
    g = golden("grid_sampler_2d")
    out = oracle_mod.grid_sampler(g["input"], g["grid"], mode, pad, align)
    want = g[f"{MODES[mode]}_{PADS[pad]}_{int(align)}"]
    if mode == 1:
        # nearest: identical pixels except possible .5 ties (reference test allows 0.1 mean)
        assert (out != want).mean() <= 1e-3
    else:
        np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("mode", range(2))
@pytest.mark.parametrize("pad", range(3))
@pytest.mark.parametrize("align", [False, True])
def test_grid_sampler_3d_oracle(oracle_mod, mode, pad, align):
    g = golden("grid_sampler_3d")
    out = oracle_mod.grid_sampler(g["input"], g["grid"], mode, pad, align)
    want = g[f"{MODES[mode]}_{PADS[pad]}_{int(align)}"]
    if mode == 1:
        assert (out != want).mean() <= 1e-3
    else:
        np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", ["sq_small", "rect", "offcenter", "bev_like"])
def test_rotate_oracle(oracle_mod, case):
    g = golden("rotate_" + case)
    ob = oracle_mod.rotate(g["img"], g["angle"], g["center"], 0)
    np.testing.assert_allclose(ob, g["bilinear"], rtol=1e-4, atol=1e-4)  # test_rotate.py fp32 1e-4
    on = oracle_mod.rotate(g["img"], g["angle"], g["center"], 1)
    assert (on != g["nearest"]).mean() <= 2e-3


@pytest.mark.parametrize("case", ["caffe", "torch"])
def test_image_ref_against_float64_definition(case):
    """oracle/image_ref.py (mmcv's float32 pipeline: two roundings per pixel) against the published
    definition of mmcv.imnormalize + impad_to_multiple evaluated in float64
    (tests/golden/make_image_golden.py): within 2 float32 ulps of the result, padding exactly zero,
    BGR->RGB order and the pad geometry as the definition says."""
    import os
    from oracle.image_ref import image_normalize_pad
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "image_norm.npz"))
    got = image_normalize_pad(g["img"], tuple(g[case + "_mean"]), tuple(g[case + "_std"]), bool(g[case + "_to_rgb"]))
    want = g[case + "_out64"]
    assert got.shape == want.shape and got.dtype == np.float32
    assert (got[:, :, 37:, :] == 0).all() and (got[:, :, :, 45:] == 0).all()
    # float32(x - float32(mean)) * float32(1/std): the subtraction's rounding (<= ulp(256)/2 = 1.5e-5) scaled by 1/std,
    # plus the product's own rounding
    inv = 1.0 / g[case + "_std"].reshape(1, 3, 1, 1)
    tol = 1.6e-5 * inv + 2 * np.spacing(np.abs(want).astype(np.float32)).astype(np.float64) + np.abs(want) * 1.2e-7
    assert (np.abs(got.astype(np.float64) - want) <= tol).all()
