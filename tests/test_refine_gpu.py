"""bevops_refine_reference_points (csrc/refine.hip) against the framework's op sequence of the decoder's reference-point
refinement (geometry.refine_reference_points = det2trt/models/modules/decoder.py:24-40, 93-103): BIT-EXACT -- the refined
points are the next layer's sampling locations."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _all_half(lo, hi):
    bits = torch.arange(0, 1 << 16, dtype=torch.int32).to(torch.int16)
    v = bits.view(torch.float16)
    keep = torch.isfinite(v.float()) & (v.float() >= lo) & (v.float() <= hi)
    return v[keep]


def _check(tmp3, ref):
    """tmp3 [n, 3] (the three regression values that are used), ref [n, 3]"""
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd import geometry as G
    n = ref.shape[0]
    tmp = torch.zeros(1, n, 10, dtype=torch.float16, device="cuda")
    tmp[0, :, 0], tmp[0, :, 1], tmp[0, :, 4] = tmp3[:, 0], tmp3[:, 1], tmp3[:, 2]
    tmp[0, :, 2] = 7.0      # (columns the refinement must not read)
    r = ref.cuda().view(1, n, 3)
    want = G.refine_reference_points(tmp, r)
    new, xy = bev.refine_reference_points(tmp, r)
    assert new.shape == (1, n, 3) and xy.shape == (1, n, 1, 2) and xy.is_contiguous()
    same = (new.view(torch.int16) == want.view(torch.int16)) | (torch.isnan(new) & torch.isnan(want))
    assert bool(same.all()), "differs at %d of %d" % (int((~same).sum()), same.numel())
    assert torch.equal(xy.view(n, 2).view(torch.int16), new[0, :, :2].contiguous().view(torch.int16))


def test_every_reference_point_value_and_every_regression_value():
    """(a) every binary16 reference point in [0, 1] (15 361 values: the inverse sigmoid incl. both clamps) with a few
    regression values; (b) every finite binary16 regression value against reference points that make the inverse sigmoid
    0 (0.5) and +- large: the sum and the sigmoid on every input they can see."""
    refs = _all_half(0.0, 1.0)
    n = refs.numel()
    for t in (0.0, 0.37, -2.5, 11.0):
        _check(torch.full((n, 3), t, dtype=torch.float16).cuda(), torch.stack([refs, refs.flip(0), refs], 1))
    ts = _all_half(-65504.0, 65504.0)
    m = ts.numel()
    for rv in (0.5, 0.25, 0.999, 1e-4):
        _check(torch.stack([ts, ts.flip(0), ts], 1).cuda(), torch.full((m, 3), rv, dtype=torch.float16))


def test_random_points_beyond_the_unit_interval_and_the_model_shapes():
    """Reference points outside [0, 1] (clamped), NaN / inf regression values, the decoder's 900 queries."""
    g = torch.Generator().manual_seed(0)
    ref = (torch.rand(900, 3, generator=g) * 1.4 - 0.2).half()
    tmp3 = (torch.randn(900, 3, generator=g) * 3).half()
    tmp3[5, 0], tmp3[6, 1], tmp3[7, 2] = float("inf"), float("-inf"), float("nan")
    _check(tmp3.cuda(), ref)
    ref[11, 0] = float("nan")
    _check(tmp3.cuda(), ref)


def _decode_reference(regs, refs):
    """the framework's op sequence (bevformer.py, the per-level loop of the reference head, batched)"""
    from bevformer_tensorrt_amd import bevformer as B
    crd = regs.clone()
    reference = B.inverse_sigmoid(refs)
    crd[..., 0:2] = (crd[..., 0:2] + reference[..., 0:2]).sigmoid()
    crd[..., 4:5] = (crd[..., 4:5] + reference[..., 2:3]).sigmoid()
    P = B.PC_RANGE
    crd[..., 0:1] = crd[..., 0:1] * (P[3] - P[0]) + P[0]
    crd[..., 1:2] = crd[..., 1:2] * (P[4] - P[1]) + P[1]
    crd[..., 4:5] = crd[..., 4:5] * (P[5] - P[2]) + P[2]
    return crd


def test_decode_boxes_is_bit_exact_on_every_reference_value_and_every_regression_value():
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd import bevformer as B

    def check(regs, refs):
        want = _decode_reference(regs, refs)
        got = bev.decode_boxes(regs, refs, B.PC_RANGE)
        same = (got.view(torch.int16) == want.view(torch.int16)) | (torch.isnan(got) & torch.isnan(want))
        assert bool(same.all()), "differs at %d of %d" % (int((~same).sum()), same.numel())

    g = torch.Generator().manual_seed(3)
    refs_all = _all_half(-0.25, 1.25)                 # incl. values the first clamp cuts
    n = refs_all.numel()
    for t in (0.0, -1.3, 4.0):
        regs = (torch.randn(n, 10, generator=g) * 2).half()
        regs[:, 0], regs[:, 1], regs[:, 4] = t, t, t
        check(regs.cuda(), torch.stack([refs_all, refs_all.flip(0), refs_all], 1).cuda())
    ts = _all_half(-65504.0, 65504.0)
    m = ts.numel()
    for rv in (0.5, 0.031, 0.97):
        regs = torch.zeros(m, 10, dtype=torch.float16)
        regs[:, 0], regs[:, 1], regs[:, 4] = ts, ts.flip(0), ts
        check(regs.cuda(), torch.full((m, 3), rv, dtype=torch.float16).cuda())
    regs = (torch.randn(6, 1, 900, 10, generator=g) * 3).half().cuda()     # the model's stacked shape, NaN / inf inside
    regs[0, 0, 3, 0], regs[1, 0, 4, 4] = float("nan"), float("inf")
    refs = torch.rand(6, 1, 900, 3, generator=g).half().cuda()
    refs[2, 0, 5, 1] = float("nan")
    check(regs, refs)
