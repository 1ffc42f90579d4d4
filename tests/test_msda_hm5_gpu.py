"""GPU parity for the re-scheduled head-major fp16 MSDA path with the exact visibility pre-pass
(msda_hm5.hip): the base SCA call shape (4 levels x 8 points x 4 anchors).  fp16 tolerance as in
tests/test_msda_gpu.py: 1e-2 element-wise vs the fp32 oracle on fp16-rounded inputs; items all of whose
samples fail the reference's range gate (multiScaleDeformableAttnKernel.cu:673) must be exactly 0."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LEVELS = [[116, 200], [58, 100], [29, 50], [15, 25]]
VARIANTS = {"hm5": 1000, "hm5_no_prepass": 1001}


@pytest.fixture(scope="module")
def ctx():
    import bevformer_tensorrt_amd as b
    from bevformer_tensorrt_amd.utils import load_library
    return b, load_library()


def gen(bs, nq, seed=0, mode="edge", levels=LEVELS, off_std=1.5):
    heads, C, P, ppg = 8, 32, 8, 4
    g = torch.Generator().manual_seed(seed)
    L = len(levels)
    nk = sum(h * w for h, w in levels)
    value = torch.randn(bs, nk, heads, C, generator=g)
    ref = torch.rand(bs, nq, 1, 2 * ppg, generator=g)
    if mode == "edge":          # anchors a little outside the image too
        ref = ref * 1.2 - 0.1
    elif mode == "nonfinite":   # anchors of pillars behind a camera overflow binary16 (+-inf), some NaN
        ref = ref * 1.2 - 0.1
        kill = torch.rand(bs, nq, 1, 2 * ppg, generator=g)
        ref = torch.where(kill < 0.15, torch.full_like(ref, float("inf")), ref)
        ref = torch.where((kill >= 0.15) & (kill < 0.3), torch.full_like(ref, -float("inf")), ref)
        ref = torch.where((kill >= 0.3) & (kill < 0.33), torch.full_like(ref, 7.0e4), ref)
    elif mode == "oov":         # most (batch, query) pairs far out of view, some just outside
        far = (torch.rand(bs, nq, 1, 1, generator=g) < 0.7).float()
        near = (torch.rand(bs, nq, 1, 1, generator=g) < 0.5).float()
        ref = ref + far * (near * 1.02 + (1 - near) * 3.0)
    off = torch.randn(bs, nq, heads, L * P * 2, generator=g) * off_std
    logit = torch.randn(bs, nq, heads, L * P, generator=g)
    sh = torch.tensor(levels, dtype=torch.int32)
    return [value.half().cuda(), sh.cuda(), ref.half().cuda(), off.half().cuda(), logit.half().cuda()]


def run(ctx, args, variant, poison=False):
    bev, lib = ctx
    lib.bevops_msda_set_variant(variant)
    try:
        out = bev.multi_scale_deformable_attn(*args)
        torch.cuda.synchronize()
    finally:
        lib.bevops_msda_set_variant(0)
    return out


def oracle(oracle_mod, args):
    v, sh, r, o, w = (a.float().cpu().numpy() if a.is_floating_point() else a.cpu().numpy() for a in args)
    return oracle_mod.msda_f32(v, sh, r, o, w)


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("mode,nq", [("edge", 4000), ("oov", 4000), ("edge", 2049), ("oov", 2100), ("nonfinite", 2304)])
def test_hm5_vs_oracle(ctx, oracle_mod, variant, mode, nq):
    # (the head-major planes are only built for calls with >= 2048 queries; smaller calls stay on the
    # layout-preserving kernel)
    args = gen(6 if nq > 3000 else 2, nq, seed=nq, mode=mode)
    out = run(ctx, args, VARIANTS[variant]).float().cpu().numpy()
    want = oracle(oracle_mod, args)
    assert np.isfinite(out).all()
    assert np.abs(out - want).max() <= 1e-2
    # exact zeros where the reference produces exact zeros (every sample out of range)
    dead = np.abs(want).reshape(want.shape[0], want.shape[1], -1).max(-1) == 0
    assert (out.reshape(dead.shape[0], dead.shape[1], -1)[dead] == 0).all()


def test_hm5_other_level_sizes(ctx, oracle_mod):
    """Another 4-level pyramid whose two small levels stage (odd widths, a 1-pixel-high level)."""
    levels = [[90, 161], [45, 81], [23, 41], [1, 21]]
    args = gen(3, 2600, seed=5, mode="edge", levels=levels)
    for variant in ("hm5", "hm5_no_prepass"):
        out = run(ctx, args, VARIANTS[variant]).float().cpu().numpy()
        assert np.abs(out - oracle(oracle_mod, args)).max() <= 1e-2


def test_hm5_is_default_and_overwrites_stale_output(ctx):
    """The default dispatch takes hm5 for the base SCA shape and every output element is written (the
    zeros of invisible items by the pre-pass), even when the output buffer held garbage."""
    bev, lib = ctx
    args = gen(6, 40000, seed=1, mode="oov")
    a = run(ctx, args, 0)
    b = run(ctx, args, VARIANTS["hm5"])
    assert torch.equal(a, b)
    c = run(ctx, args, 16).float()      # hm3: same arithmetic per sample, another schedule
    assert (a.float() - c).abs().max().item() <= 2e-3


@pytest.mark.parametrize("mode", ["uniform", "oov"])
def test_hm5_full_size_matches_layout_preserving_kernel(ctx, mode):
    args = gen(6, 40000, seed=0, mode=mode, off_std=1.0)
    base = run(ctx, args, 10).float()
    for name, v in VARIANTS.items():
        o = run(ctx, args, v).float()
        assert torch.isfinite(o).all(), name
        assert (o - base).abs().max().item() <= 6e-3, name
    # determinism
    assert torch.equal(run(ctx, args, 1000), run(ctx, args, 1000))


@pytest.mark.parametrize("variant", [0, 11, 15, 16, 17])
@pytest.mark.parametrize("shape", ["tiny_sca", "base_sca_q4k", "tsa_like"])
def test_nonfinite_reference_points_give_zero_not_nan(ctx, oracle_mod, variant, shape):
    """Reference points of pillars behind a camera overflow binary16 (point_sampling divides by
    max(z, 1e-5)): such samples fail the reference's range gate and contribute exactly 0 there
    (multiScaleDeformableAttnKernel.cu:673).  Every head-major generation must do the same -- their
    front ends used to turn an infinite location into NaN fractions and NaN * 0 reached the output."""
    bs, levels, nq, P, ppg = {"tiny_sca": (6, [[15, 25]], 2500, 8, 4),
                              "base_sca_q4k": (6, LEVELS, 4000, 8, 4),
                              "tsa_like": (2, [[50, 50]], 2500, 4, 1)}[shape]
    heads, C = 8, 32
    g = torch.Generator().manual_seed(3)
    L = len(levels)
    nk = sum(h * w for h, w in levels)
    value = torch.randn(bs, nk, heads, C, generator=g)
    ref = torch.rand(bs, nq, 1, 2 * ppg, generator=g)
    kill = torch.rand(bs, nq, 1, 2 * ppg, generator=g)
    ref = torch.where(kill < 0.2, torch.full_like(ref, float("inf")), ref)
    ref = torch.where((kill >= 0.2) & (kill < 0.4), torch.full_like(ref, -float("inf")), ref)
    off = torch.randn(bs, nq, heads, L * P * 2, generator=g)
    logit = torch.randn(bs, nq, heads, L * P, generator=g)
    args = [value.half().cuda(), torch.tensor(levels, dtype=torch.int32).cuda(), ref.half().cuda(),
            off.half().cuda(), logit.half().cuda()]
    out = run(ctx, args, variant).float().cpu().numpy()
    assert np.isfinite(out).all()
    assert np.abs(out - oracle(oracle_mod, args)).max() <= 1e-2
