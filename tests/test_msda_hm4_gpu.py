"""hm4 (csrc/msda_hm4.hip): the software-pipelined head-major kernels, fp16 and both int8 flavours.
  fp16 : vs the oracle, |err| <= 1e-2 (north_star), and vs hm3 (same arithmetic, other schedule);
  int8 : BIT-IDENTICAL to the layout-preserving int8 kernels of msda.hip -- the exact integer
         requantisation (saturating v_mad_i32_i24 + top byte) must reproduce the float formula
         T2int8(tsum / 127) resp. / 255 for every input -- and vs the C oracle with the int8 LSB
         budget of tests/test_msda_int8_gpu.py.
Every (L*P, big batches) instantiation is exercised: staged / not staged, 1 level and 4 levels."""
import numpy as np
import pytest
import torch

from test_msda_gpu import gen, oracle_of
from test_msda_int8_gpu import make, quantize

pytestmark = pytest.mark.gpu

BASE = [[116, 200], [58, 100], [29, 50], [15, 25]]
# name: (shape, expected kernel instance -- documentation only)
SHAPES = {
    "base_sca_q4k": ((6, BASE, 4096, 8, 4), "<32,4> two levels staged"),
    "base_sca_q1k": ((6, BASE, 1000, 8, 4), "<32,8> nothing staged (few queries)"),
    "sca_3lvl_q3k": ((2, [[160, 260], [80, 130], [40, 65], [20, 33]], 3000, 8, 4), "<32,6> one level staged"),
    "small_sca": ((6, [[23, 40]], 22500, 8, 4), "<8,0> whole plane staged"),
    "sca_1lvl_q1k": ((6, [[23, 40]], 1500, 8, 4), "<8,2> not staged"),
    "tsa_like": ((2, [[120, 120]], 9000, 4, 1), "<4,1>"),
    "tsa_staged": ((2, [[40, 40]], 9000, 4, 1), "<4,0>"),
    "ragged_tail": ((3, BASE, 2049, 8, 4), "<32,4> last chunk of one query"),
}


@pytest.fixture(scope="module")
def ctx():
    import bevformer_tensorrt_amd as b
    from bevformer_tensorrt_amd.utils import load_library
    return b, load_library()


def run(ctx, args, variant, int8_scales=None):
    bev, lib = ctx
    lib.bevops_msda_set_variant(variant)
    try:
        if int8_scales is None:
            out = bev.multi_scale_deformable_attn(*args)
        else:
            out = bev.multi_scale_deformable_attn_int8(*args, *int8_scales)
        torch.cuda.synchronize()
    finally:
        lib.bevops_msda_set_variant(0)
    return out


@pytest.mark.parametrize("name", list(SHAPES))
def test_fp16_vs_oracle_and_hm3(ctx, oracle_mod, name):
    args = gen(SHAPES[name][0], dtype=torch.float16)
    out = run(ctx, args, 17)
    want = oracle_of(oracle_mod, args)
    err = np.abs(out.float().cpu().numpy() - want)
    assert err.max() <= 1e-2, (name, err.max())
    ref = run(ctx, args, 10)                      # layout-preserving quad kernel
    assert (out.float() - ref.float()).abs().max().item() <= 6e-3
    assert torch.equal(out, run(ctx, args, 17))   # deterministic


def test_fp16_out_of_view_and_dirty_workspace(ctx):
    big = gen(SHAPES["small_sca"][0], dtype=torch.float16)
    run(ctx, big, 17)                              # leaves a large, dirty workspace behind
    args = gen(SHAPES["base_sca_q4k"][0], dtype=torch.float16)
    args[2] = args[2] + 7.0                        # every sample out of range
    assert torch.count_nonzero(run(ctx, args, 17)).item() == 0
    args = gen(SHAPES["base_sca_q4k"][0], dtype=torch.float16, ref_lo=-0.3, ref_hi=1.3)
    a, b = run(ctx, args, 17), run(ctx, args, 10)
    assert torch.isfinite(a).all() and (a.float() - b.float()).abs().max().item() <= 6e-3


@pytest.mark.parametrize("dtype", [torch.float16])
def test_fp16_camera_shared_offsets(ctx, dtype):
    args = gen(SHAPES["base_sca_q4k"][0], dtype=dtype)
    one = [args[0], args[1], args[2], args[3][:1], args[4][:1]]
    rep = [args[0], args[1], args[2], one[3].repeat(6, 1, 1, 1), one[4].repeat(6, 1, 1, 1)]
    shared = [args[0], args[1], args[2], one[3].expand(6, -1, -1, -1), one[4].expand(6, -1, -1, -1)]
    assert torch.equal(run(ctx, rep, 17), run(ctx, shared, 17))


def same_as_quad(a, b, ref_dtype, name):
    """x127 flavour (fp32 reference points): every step is integer or a single rounding -> bit
    identical.  x255 flavour: S is a float sum of 32 un-quantised weights whose association
    differs between the two kernels (8 lanes x 4 points vs 4 lanes x 8 points), which moves the
    final requantisation by one LSB on a few outputs per million (the reference's own kernel sums
    them in binary16)."""
    if ref_dtype == torch.float32:
        assert torch.equal(a, b), (name, (a.int() - b.int()).abs().max().item(), (a != b).float().mean().item())
    else:
        d = (a.int() - b.int()).abs()
        assert d.max().item() <= 1 and (d > 0).float().mean().item() <= 2e-5, \
            (name, d.max().item(), (d > 0).float().mean().item())


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("ref_dtype", [torch.float32, torch.float16], ids=["s8w_f32ref", "u8w_f16ref"])
def test_int8_bit_identical_to_layout_preserving_kernel(ctx, oracle_mod, name, ref_dtype):
    value, sh, ref, off, logit = make(SHAPES[name][0])
    if name == "ragged_tail":                      # saturating inputs: the clamp of T2int8 must bite
        value = value * 3.0
    qv, s_v = quantize(value); qo, s_o = quantize(off); qw, s_w = quantize(logit)
    if name == "ragged_tail":
        qv = torch.where(torch.rand(qv.shape) < 0.3, torch.full_like(qv, 127), qv)
        qv = torch.where(torch.rand(qv.shape) < 0.2, torch.full_like(qv, -128), qv)
    ref_in = ref.to(ref_dtype)
    args = (qv.cuda(), sh.cuda(), ref_in.cuda(), qo.cuda(), qw.cuda())
    scales = (s_v, s_o, s_w, 0.02)
    a = run(ctx, args, 17, scales)
    b = run(ctx, args, 10, scales)
    same_as_quad(a, b, ref_dtype, name)
    if SHAPES[name][0][2] <= 4096:
        want = oracle_mod.msda_s8(qv.numpy(), s_v, sh.numpy(), ref_in.float().numpy(), qo.numpy(), s_o,
                                  qw.numpy(), s_w, 0.02, u8_weights=(ref_dtype == torch.float16)).astype(np.int32)
        d = np.abs(a.cpu().numpy().astype(np.int32) - want)
        assert d.max() <= 1 and (d > 0).mean() <= 0.01


@pytest.mark.parametrize("name", ["base_sca_q4k", "base_sca_q1k", "sca_3lvl_q3k", "ragged_tail"])
@pytest.mark.parametrize("ref_dtype", [torch.float32, torch.float16], ids=["s8w_f32ref", "u8w_f16ref"])
def test_int8_block_plans_bit_identical(ctx, name, ref_dtype):
    """int8, L*P = 32: the default two-blocks-per-CU plan (variant 17: one level staged, <= 128 registers) against the
    one-block plan (19: two levels staged): same arithmetic in the same order -> equal bits; and against the
    layout-preserving kernel (10).  Saturating inputs on the ragged shape.  A packed value is sampled under the plan
    it was made for (the plan is part of the variant)."""
    value, sh, ref, off, logit = make(SHAPES[name][0])
    if name == "ragged_tail":
        value = value * 3.0
    qv, s_v = quantize(value); qo, s_o = quantize(off); qw, s_w = quantize(logit)
    if name == "ragged_tail":
        qv = torch.where(torch.rand(qv.shape) < 0.3, torch.full_like(qv, 127), qv)
        qv = torch.where(torch.rand(qv.shape) < 0.2, torch.full_like(qv, -128), qv)
    args = (qv.cuda(), sh.cuda(), ref.to(ref_dtype).cuda(), qo.cuda(), qw.cuda())
    scales = (s_v, s_o, s_w, 0.02)
    quad = run(ctx, args, 10, scales)
    a, b = run(ctx, args, 17, scales), run(ctx, args, 19, scales)
    assert torch.equal(a, b), (name, (a != b).float().mean().item())
    same_as_quad(a, quad, ref_dtype, name)
    bev, lib = ctx
    lib.bevops_msda_set_variant(19)
    try:
        packed = bev.msda_pack_value(args[0], args[1], SHAPES[name][0][2], SHAPES[name][0][3], reference_dtype=ref_dtype)
        c = bev.multi_scale_deformable_attn_prepacked(packed, args[2], args[3], args[4], scales)
    finally:
        lib.bevops_msda_set_variant(0)
    assert torch.equal(c, b)


@pytest.mark.parametrize("ref_dtype", [torch.float32, torch.float16], ids=["s8w_f32ref", "u8w_f16ref"])
def test_int8_default_choice_at_full_base_size(ctx, ref_dtype):
    """6 x 40 000 queries: the default dispatch (hm4) against the layout-preserving kernel."""
    value, sh, ref, off, logit = make((6, BASE, 40000, 8, 4))
    qv, s_v = quantize(value); qo, s_o = quantize(off); qw, s_w = quantize(logit)
    args = (qv.cuda(), sh.cuda(), ref.to(ref_dtype).cuda(), qo.cuda(), qw.cuda())
    a = run(ctx, args, 0, (s_v, s_o, s_w, 0.02))
    b = run(ctx, args, 10, (s_v, s_o, s_w, 0.02))
    same_as_quad(a, b, ref_dtype, "base_sca")


@pytest.mark.parametrize("flavour", ["fp16", "s8w_f32ref", "u8w_f16ref"])
def test_prepacked_equals_one_call(ctx, flavour):
    """bevops_msda_pack_value + bevops_msda_forward_prepacked == the forced-hm4 single call, bit for
    bit; one packed value serves several sampling calls."""
    bev, lib = ctx
    shape = SHAPES["base_sca_q4k"][0]
    if flavour == "fp16":
        args = gen(shape, dtype=torch.float16)
        packed = bev.msda_pack_value(args[0], args[1], shape[2], shape[3])
        for seed in (0, 1):
            other = gen(shape, seed=seed, dtype=torch.float16)
            a = bev.multi_scale_deformable_attn_prepacked(packed, other[2], other[3], other[4])
            b = run(ctx, [args[0], args[1], other[2], other[3], other[4]], 17)
            assert torch.equal(a, b)
        return
    rdt = torch.float32 if flavour.startswith("s8w") else torch.float16
    value, sh, ref, off, logit = make(shape)
    qv, s_v = quantize(value); qo, s_o = quantize(off); qw, s_w = quantize(logit)
    packed = bev.msda_pack_value(qv.cuda(), sh.cuda(), shape[2], shape[3], reference_dtype=rdt)
    scales = (s_v, s_o, s_w, 0.02)
    a = bev.multi_scale_deformable_attn_prepacked(packed, ref.to(rdt).cuda(), qo.cuda(), qw.cuda(), scales)
    b = run(ctx, (qv.cuda(), sh.cuda(), ref.to(rdt).cuda(), qo.cuda(), qw.cuda()), 17, scales)
    assert torch.equal(a, b)
    with pytest.raises(TypeError):   # packed for one flavour, sampled with the other
        bev.multi_scale_deformable_attn_prepacked(packed, ref.to(torch.float16 if rdt == torch.float32 else torch.float32).cuda(),
                                                  qo.cuda(), qw.cuda(), scales)


@pytest.mark.parametrize("seed", range(8))
def test_random_pyramids_head_major_vs_layout_preserving(ctx, seed):
    """Random (not halving, odd-sized, sometimes 1-row) pyramids, batch sizes and query counts: wherever a
    head-major kernel accepts the call (forced variants 16 = hm3, 17 = hm4), fp16 agrees with the
    layout-preserving generic kernel (variant 99) within the fp16 bar, and the int8 hm4 result equals the
    layout-preserving int8 kernel (variant 10) -- the padded re-layout, the staging plan and the chunking see
    shapes no model produces."""
    from bevformer_tensorrt_amd.utils.lib import BevopsError
    rng = np.random.default_rng(100 + seed)
    L = 4 if seed % 3 else 1
    levels = [[int(rng.integers(1 if seed == 5 else 3, 70)), int(rng.integers(3, 90))] for _ in range(L)]
    if L == 4:
        levels.sort(key=lambda hw: -hw[0] * hw[1])
    shape = (int(rng.integers(1, 5)), levels, int(rng.integers(700, 5000)), 8 if L == 4 else int(rng.choice([4, 8])), 4 if L == 4 else 1)
    args = gen(shape, dtype=torch.float16, ref_lo=-0.2, ref_hi=1.2)
    want = run(ctx, args, 99).float()
    took = []
    for v in (16, 17, 0):
        try:
            got = run(ctx, args, v).float()
        except BevopsError as exc:          # this plan has no instantiation in that family
            assert exc.status == 3, exc
            continue
        took.append(v)
        assert (got - want).abs().max().item() <= 1e-2 * max(1.0, want.abs().max().item()), (v, shape)
    assert 0 in took
    value, sh, ref, off, logit = make(shape)
    qv, s_v = quantize(value); qo, s_o = quantize(off); qw, s_w = quantize(logit)
    for rdt in (torch.float32, torch.float16):
        iargs = (qv.cuda(), sh.cuda(), ref.to(rdt).cuda(), qo.cuda(), qw.cuda())
        b = run(ctx, iargs, 10, (s_v, s_o, s_w, 0.02))
        try:
            a = run(ctx, iargs, 17, (s_v, s_o, s_w, 0.02))
        except BevopsError as exc:
            assert exc.status == 3, exc
            continue
        same_as_quad(a, b, rdt, f"random{seed}")
