"""GPU parity: DCNv2 HIP path vs the CPU oracle.
  fp32: element-wise 1e-4 relative to the output scale (reference test: mean abs 1e-5 at
        K = 1152; fp32 sums of 1152..4608 products reorder between implementations)
  fp16: element-wise 1e-2 relative to the output scale, mean abs <= 0.05 (reference fp16
        tolerance, test_modulated_deformable_conv2d.py:101-104)
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bev():
    import bevformer_tensorrt_amd as b
    return b


def make(B, Cin, Cout, H, W, K, stride, pad, dil, g, dg, seed=0, off_std=1.0):
    gen = torch.Generator().manual_seed(seed)
    Ho = (H + 2 * pad - (dil * (K - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (K - 1) + 1)) // stride + 1
    x = torch.randn(B, Cin, H, W, generator=gen)
    off = torch.randn(B, dg * 2 * K * K, Ho, Wo, generator=gen) * off_std
    mask = torch.rand(B, dg * K * K, Ho, Wo, generator=gen)
    w = torch.randn(Cout, Cin // g, K, K, generator=gen) / (Cin // g * K * K) ** 0.5
    b = torch.randn(Cout, generator=gen)
    return x, off, mask, w, b


CASES = {
    # reference test (test_modulated_deformable_conv2d.py:6-11,36): x [8,256,256,256] is 2 GB of
    # columns per image on the CPU oracle -> same channels/groups, smaller image
    "ref_test_like": dict(B=2, Cin=256, Cout=256, H=40, W=44, K=3, stride=1, pad=1, dil=1, g=2, dg=2),
    "r101_stage3": dict(B=2, Cin=256, Cout=256, H=58, W=100, K=3, stride=1, pad=1, dil=1, g=1, dg=1),
    "r101_stage4": dict(B=2, Cin=512, Cout=512, H=29, W=50, K=3, stride=1, pad=1, dil=1, g=1, dg=1),
    "stride2": dict(B=1, Cin=64, Cout=96, H=31, W=45, K=3, stride=2, pad=1, dil=1, g=1, dg=1),
    "dilated_groups": dict(B=2, Cin=32, Cout=48, H=17, W=19, K=3, stride=1, pad=2, dil=2, g=4, dg=2),
    "odd_channels": dict(B=1, Cin=6, Cout=10, H=9, W=11, K=3, stride=1, pad=1, dil=1, g=1, dg=3),
    "k1": dict(B=1, Cin=16, Cout=8, H=8, W=8, K=1, stride=1, pad=0, dil=1, g=1, dg=1),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("with_bias", [True, False])
def test_mdconv_vs_oracle(bev, oracle_mod, name, dtype, with_bias):
    c = CASES[name]
    x, off, mask, w, b = (t.to(dtype) for t in make(**c))
    bias = b if with_bias else None
    out = bev.modulated_deformable_conv2d(x.cuda(), off.cuda(), mask.cuda(), w.cuda(),
                                          bias.cuda() if with_bias else None, c["stride"], c["pad"],
                                          c["dil"], c["g"], c["dg"])
    torch.cuda.synchronize()
    want = oracle_mod.mdconv(x.float().numpy(), off.float().numpy(), mask.float().numpy(),
                             w.float().numpy(), bias.float().numpy() if with_bias else None,
                             (c["stride"],) * 2, (c["pad"],) * 2, (c["dil"],) * 2, c["g"], c["dg"])
    got = out.float().cpu().numpy()
    assert got.shape == want.shape
    scale = max(1.0, float(np.abs(want).max()))
    if dtype == torch.float32:
        assert np.abs(got - want).max() <= 1e-4 * scale
    else:
        assert np.abs(got - want).max() <= 1e-2 * scale
        assert np.abs(got - want).mean() <= 0.05


def test_zero_offsets_equal_conv2d_full_stage3(bev):
    """Full base stage-3 size (6 cams x 256 ch x 58 x 100): zero offsets + unit mask
    must reproduce torch's conv2d (size-independent property, fp16)."""
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(6, 256, 58, 100, generator=gen).half().cuda()
    w = (torch.randn(256, 256, 3, 3, generator=gen) / 48).half().cuda()
    b = torch.randn(256, generator=gen).half().cuda()
    off = torch.zeros(6, 18, 58, 100).half().cuda()
    mask = torch.ones(6, 9, 58, 100).half().cuda()
    out = bev.modulated_deformable_conv2d(x, off, mask, w, b, 1, 1, 1, 1, 1).float()
    want = F.conv2d(x.float(), w.float(), b.float(), 1, 1)
    assert (out - want).abs().max().item() <= 2e-2
    assert torch.equal(out.half(), bev.modulated_deformable_conv2d2(x, off, mask, w, b, 1, 1, 1, 1, 1))


def test_mdconv_bad_shapes_raise(bev):
    x, off, mask, w, b = (t.cuda() for t in make(1, 8, 8, 6, 6, 3, 1, 1, 1, 1, 1))
    with pytest.raises(ValueError):
        bev.modulated_deformable_conv2d(x, off[:, :4], mask, w, b, 1, 1, 1, 1, 1)
    from bevformer_tensorrt_amd.utils.lib import BevopsError
    with pytest.raises(BevopsError):   # Cin not divisible by groups
        bev.modulated_deformable_conv2d(x, off, mask, w[:, :3], b, 1, 1, 1, 3, 1)


@pytest.mark.parametrize("name", ["ref_test_like", "r101_stage3", "r101_stage4", "stride2"])
def test_fused_matches_im2col_pipeline(bev, name):
    """fp16: the fused implicit-GEMM kernel (default) vs the im2col + GEMM pipeline (variant 1)."""
    from bevformer_tensorrt_amd.utils import load_library
    lib = load_library()
    c = CASES[name]
    x, off, mask, w, b = (t.half().cuda() for t in make(**c))
    args = (x, off, mask, w, b, c["stride"], c["pad"], c["dil"], c["g"], c["dg"])
    fused = bev.modulated_deformable_conv2d(*args).float()
    try:
        lib.bevops_mdconv_set_variant(1)
        two = bev.modulated_deformable_conv2d(*args).float()
    finally:
        lib.bevops_mdconv_set_variant(0)
    scale = max(1.0, two.abs().max().item())
    assert (fused - two).abs().max().item() <= 1e-2 * scale


@pytest.mark.parametrize("shape", [(6, 256, 256, 58, 100), (6, 512, 512, 29, 50), (5, 64, 128, 61, 93)])
def test_lds_dma_kernel_and_split_k_tail_match_register_staged_kernel(bev, shape):
    """Full-size fp16 calls: the default (weights by LDS-DMA; leftover tiles of a sparse last round
    split along K, partials summed in a fixed order by a finish kernel) vs the same kernel
    without the tail split (variant 4) vs the register-staged kernel of r01c (variant 2).  Same
    fp16 blend; the tail only changes the fp32 summation order.  Deterministic run to run."""
    from bevformer_tensorrt_amd.utils import load_library
    lib = load_library()
    B, Cin, Cout, H, W = shape
    x, off, mask, w, b = (t.half().cuda() for t in make(B, Cin, Cout, H, W, 3, 1, 1, 1, 1, 1, off_std=2.0))
    args = (x, off, mask, w, b, 1, 1, 1, 1, 1)
    outs = {}
    for v in (0, 4, 5, 2):
        try:
            lib.bevops_mdconv_set_variant(v)
            outs[v] = bev.modulated_deformable_conv2d(*args)
        finally:
            lib.bevops_mdconv_set_variant(0)
    scale = max(1.0, outs[2].abs().max().item())
    assert (outs[0].float() - outs[2].float()).abs().max().item() <= 4e-3 * scale
    assert (outs[4].float() - outs[2].float()).abs().max().item() <= 4e-3 * scale
    assert (outs[5].float() - outs[2].float()).abs().max().item() <= 4e-3 * scale   # 1024-thread / 128-pixel tiles
    for _ in range(3):
        assert torch.equal(bev.modulated_deformable_conv2d(*args), outs[0])
    # the wave orders of the LDS-DMA kernel (default since round 6: the lower half of a block's waves issues all the
    # weight DMA, the upper half runs its matrix segment first; 13: one order for all waves; 7: the round-2 rotation)
    # differ in WHEN a wave does what, never in what is summed in which order: the same bits
    for v in (13, 7):
        try:
            lib.bevops_mdconv_set_variant(v)
            assert torch.equal(bev.modulated_deformable_conv2d(*args), outs[0]), v
        finally:
            lib.bevops_mdconv_set_variant(0)


def _q(x, s=None):
    s = float(x.abs().max()) / 127.0 if s is None else s
    return torch.clamp(torch.round(x / s), -127, 127).to(torch.int8), s


@pytest.mark.parametrize("name", ["ref_test_like", "r101_stage3", "stride2", "dilated_groups"])
@pytest.mark.parametrize("with_bias", [True, False])
def test_mdconv_int8_vs_oracle(bev, oracle_mod, name, with_bias):
    """INT8 flavour: bit-level agreement with the C restatement of the reference's integer
    pipeline (+-1 LSB on <1 % of outputs: fp32 rounding of coordinates), and within the
    reference test's int8 tolerance of the fp32 op (mean abs 1.5 at K = 1152,
    test_modulated_deformable_conv2d.py:105-108, here scaled to the output range)."""
    c = CASES[name]
    x, off, mask, w, b = make(**c)
    qx, s_x = _q(x); qo, s_o = _q(off); qm, s_m = _q(mask); qw, s_w = _q(w)
    ref = oracle_mod.mdconv(x.numpy(), off.numpy(), mask.numpy(), w.numpy(), b.numpy() if with_bias else None,
                            (c["stride"],) * 2, (c["pad"],) * 2, (c["dil"],) * 2, c["g"], c["dg"])
    s_out = float(np.abs(ref).max()) / 127.0
    out = bev.modulated_deformable_conv2d_int8(qx.cuda(), qo.cuda(), qm.cuda(), qw.cuda(),
                                               b.cuda() if with_bias else None, s_x, s_o, s_m, s_w, s_out,
                                               c["stride"], c["pad"], c["dil"], c["g"], c["dg"])
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.int32)
    want = oracle_mod.mdconv_s8(qx.numpy(), s_x, qo.numpy(), s_o, qm.numpy(), s_m, qw.numpy(), s_w,
                                b.numpy() if with_bias else None, s_out, (c["stride"],) * 2, (c["pad"],) * 2,
                                (c["dil"],) * 2, c["g"], c["dg"]).astype(np.int32)
    d = np.abs(got - want)
    assert d.max() <= 1 and (d > 0).mean() <= 0.01, (d.max(), (d > 0).mean())
    assert np.abs(got * s_out - ref).mean() <= 0.05 * float(np.abs(ref).max())


def test_packed_weight_cache_tracks_weight_updates(bev):
    """The functional wrapper caches the re-laid-out weights per weight tensor; an in-place
    update of the weights (new version) must not serve the stale image."""
    c = CASES["r101_stage3"]
    x, off, mask, w, b = (t.half().cuda() for t in make(**c))
    args = lambda ww: (x, off, mask, ww, b, c["stride"], c["pad"], c["dil"], c["g"], c["dg"])
    a1 = bev.modulated_deformable_conv2d(*args(w))
    a2 = bev.modulated_deformable_conv2d(*args(w))          # cache hit
    assert torch.equal(a1, a2)
    w.mul_(0.5)                                             # same object, new version
    a3 = bev.modulated_deformable_conv2d(*args(w))
    fresh = bev.modulated_deformable_conv2d(*args(w.clone()))
    assert torch.equal(a3, fresh)
    assert not torch.equal(a3, a1)


@pytest.mark.parametrize("shape", [(6, 256, 58, 100), (6, 512, 29, 50), (2, 64, 13, 17), (1, 128, 9, 9), (3, 256, 7, 5)])
def test_conv_offset_nhwc_matches_conv2d(bev, shape):
    """The pack's offset convolution as our LDS-resident-weight implicit GEMM (27 channels padded to
    32, bias in the epilogue) vs the library convolution in fp32."""
    import torch.nn.functional as F
    B, Cin, H, W = shape
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Cin, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(27, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).half().cuda()
    b = torch.randn(27, generator=g).half().cuda()
    out = bev.conv_offset_nhwc(x, w, b)
    assert out.shape == (B, 32, H, W) and out.is_contiguous(memory_format=torch.channels_last)
    want = F.conv2d(x.float(), w.float(), b.float(), 1, 1)
    err = (out[:, :27].float() - want).abs().max().item()
    assert err <= 4e-3 * max(1.0, want.abs().max().item()), err
    assert not out[:, 27:].any()
    assert torch.equal(out, bev.conv_offset_nhwc(x, w, b))      # cached packed weights, deterministic
    out2 = bev.conv_offset_nhwc(x, w, None)                       # bias is optional
    want2 = F.conv2d(x.float(), w.float(), None, 1, 1)
    assert (out2[:, :27].float() - want2).abs().max().item() <= 4e-3 * max(1.0, want2.abs().max().item())


@pytest.mark.parametrize("shape", [(6, 256, 58, 100), (6, 512, 29, 50), (2, 64, 13, 17), (1, 128, 9, 9), (3, 256, 7, 5),
                                   (6, 256, 46, 80), (1, 256, 116, 200), (1, 64, 1, 1), (6, 128, 92, 160)])
def test_conv_offset_variants_match(bev, shape):
    import torch.nn.functional as F
    from bevformer_tensorrt_amd.utils import load_library
    lib = load_library()
    B, Cin, H, W = shape
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Cin, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(27, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).half().cuda()
    b = torch.randn(27, generator=g).half().cuda()
    try:
        lib.bevops_conv3x3_c32_set_variant(2)     # tile kernel, one wave per 32-pixel tile (rounds 1-4)
        ref = bev.conv_offset_nhwc(x, w, b)
        lib.bevops_conv3x3_c32_set_variant(1)     # rows-in-LDS kernel
        got = bev.conv_offset_nhwc(x, w, b)
        lib.bevops_conv3x3_c32_set_variant(3)     # tile kernel, three waves per tile (one per kernel row; round 5's default)
        split = bev.conv_offset_nhwc(x, w, b)
    finally:
        lib.bevops_conv3x3_c32_set_variant(0)
    want = F.conv2d(x.float(), w.float(), b.float(), 1, 1)
    for o in (got, split):     # same products, other sum order
        assert (o[:, :27].float() - want).abs().max().item() <= 4e-3 * max(1.0, want.abs().max().item())
        assert (o.float() - ref.float()).abs().max().item() <= 2e-3 * max(1.0, want.abs().max().item())
        assert not o[:, 27:].any()
    assert torch.equal(split, bev.conv_offset_nhwc(x, w, b))      # deterministic


@pytest.mark.parametrize("shape", [(6, 256, 58, 100), (6, 256, 46, 80), (1, 256, 58, 100), (3, 256, 7, 5), (1, 256, 1, 1),
                                   (2, 256, 8, 8), (1, 256, 116, 200), (2, 256, 17, 9), (5, 256, 24, 40)])
@pytest.mark.parametrize("with_bias", [True, False])
def test_conv_offset_resident_build_is_bit_identical_to_the_tile_kernel(bev, shape, with_bias):
    """Round 6: at Cin = 256 the default is the build with the weights in registers and 8 x 8-pixel image tiles in LDS
    (double-buffered LDS-DMA by a fourth wave).  Same k order per kernel row and the same (p0 + p1) + p2 + bias as the
    tile kernel with three waves per tile (variant 3): equal bit for bit -- whole tiles, ragged edges, images smaller
    than a tile, more tiles than blocks (the persistent loop and both buffers), one tile (no second buffer)."""
    import torch.nn.functional as F
    from bevformer_tensorrt_amd.utils import load_library
    lib = load_library()
    B, Cin, H, W = shape
    g = torch.Generator().manual_seed(B + H)
    x = torch.randn(B, Cin, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(27, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).half().cuda()
    b = torch.randn(27, generator=g).half().cuda() if with_bias else None
    try:
        lib.bevops_conv3x3_c32_set_variant(3)
        ref = bev.conv_offset_nhwc(x, w, b).clone()
    finally:
        lib.bevops_conv3x3_c32_set_variant(0)
    got = bev.conv_offset_nhwc(x, w, b)
    assert torch.equal(got, ref)
    assert torch.equal(got, bev.conv_offset_nhwc(x, w, b))
    want = F.conv2d(x.float(), w.float(), None if b is None else b.float(), 1, 1)
    assert (got[:, :27].float() - want).abs().max().item() <= 4e-3 * max(1.0, want.abs().max().item())
    assert not got[:, 27:].any()


def test_packed_weight_cache_survives_dtype_conversion():
    """nn.Module.half() swaps the storage of the SAME Parameter object without bumping its
    version counter: the packed-weight cache must notice (dtype / data pointer are part of its
    stamp) instead of handing the fp32-packed image to the fp16 kernel."""
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(3)
    conv = torch.nn.Conv2d(64, 64, 3, padding=1).cuda()
    x = torch.randn(2, 64, 20, 24, generator=g).cuda()
    off = (torch.randn(2, 18, 20, 24, generator=g) * 1.5).cuda()
    mask = torch.rand(2, 9, 20, 24, generator=g).cuda()
    y32 = bev.modulated_deformable_conv2d(x, off, mask, conv.weight, conv.bias, 1, 1, 1, 1, 1)
    w_obj = conv.weight
    conv.half()
    assert conv.weight is w_obj and conv.weight.dtype == torch.float16
    y16 = bev.modulated_deformable_conv2d(x.half(), off.half(), mask.half(), conv.weight, conv.bias, 1, 1, 1, 1, 1)
    y16b = bev.modulated_deformable_conv2d(x.half(), off.half(), mask.half(), conv.weight, conv.bias, 1, 1, 1, 1, 1)
    torch.cuda.synchronize()
    assert torch.isfinite(y16).all() and torch.equal(y16, y16b)
    assert (y16.float() - y32).abs().max().item() <= 3e-2 * max(1.0, y32.abs().max().item())
    conv.float()
    y32b = bev.modulated_deformable_conv2d(x, off, mask, conv.weight, conv.bias, 1, 1, 1, 1, 1)
    assert (y32b - y32).abs().max().item() <= 1e-2   # weights went through fp16 once


@pytest.mark.parametrize("shape", [(6, 256, 256, 58, 100), (6, 512, 512, 29, 50), (2, 64, 96, 33, 47), (3, 128, 64, 9, 140)])
def test_int8_fused_kernel_is_bit_identical_to_im2col_gemm(bev, shape):
    """dcn_fused_s8_kernel (column elements produced into LDS, exact integer T2int8(t / 255), u8-biased
    image copy) against the im2col + GEMM pair it replaces (variant 6): same integers everywhere ->
    bit-identical, at the two ResNet-101 DCN shapes and two ragged ones; saturated inputs included."""
    from bevformer_tensorrt_amd.utils import load_library
    lib = load_library()
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-128, 128, (B, Cin, H, W), generator=g, dtype=torch.int8)
    x = torch.where(torch.rand(x.shape, generator=g) < 0.1, torch.full_like(x, 127), x).cuda()
    off = torch.randint(-127, 128, (B, 18, H, W), generator=g, dtype=torch.int8).cuda()
    mask = torch.randint(0, 128, (B, 9, H, W), generator=g, dtype=torch.int8).cuda()
    w = torch.randint(-127, 128, (Cout, Cin, 3, 3), generator=g, dtype=torch.int8).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    args = (x, off, mask, w, b, 0.02, 0.03, 1 / 127, 0.004, 0.6, 1, 1, 1, 1, 1)
    try:
        lib.bevops_mdconv_set_variant(8)        # force the fused kernel (small calls default to the pair)
        a = bev.modulated_deformable_conv2d_int8(*args)
        lib.bevops_mdconv_set_variant(6)
        ref = bev.modulated_deformable_conv2d_int8(*args)
    finally:
        lib.bevops_mdconv_set_variant(0)
    torch.cuda.synchronize()
    assert torch.equal(a, ref), ((a.int() - ref.int()).abs().max().item(), (a != ref).float().mean().item())
    assert a.float().abs().mean().item() > 1.0      # the output scale leaves real signal


@pytest.mark.parametrize("shape", [
    # B, Cin, Cout, H, W, stride, groups, deform_groups
    (6, 256, 256, 58, 100, 1, 1, 1),     # base stage 3: 128-pixel tiles + split-K tail (272 tiles on 256 CUs)
    (6, 512, 512, 29, 50, 1, 1, 1),      # base stage 4: 64-pixel tiles, two Cout tiles, 4 chunks per tap
    (2, 128, 192, 33, 47, 1, 1, 1),      # one chunk per tap, ragged pixel count, Cout below one tile
    (3, 256, 320, 19, 23, 2, 2, 2),      # stride 2, two groups of 128 channels with their own deform group
    (1, 256, 64, 5, 7, 1, 1, 1),         # less than one pixel tile
])
@pytest.mark.parametrize("variant", [0, 4, 9])
def test_int8_lds_dma_kernel_is_bit_identical_to_im2col_gemm(bev, shape, variant):
    """dcn_glds_s8_kernel (weights by LDS-DMA, two LDS buffers, gathers a step ahead, split-K tail with int32
    partials; variant 4: without the tail split; variant 9: forced also where the default keeps another
    kernel, i.e. the 64-pixel-tile instantiation, two Cout tiles, groups) against the im2col + GEMM pair
    (variant 6)."""
    from bevformer_tensorrt_amd.utils import load_library
    lib = load_library()
    B, Cin, Cout, H, W, stride, G, DG = shape
    g = torch.Generator().manual_seed(7)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    x = torch.randint(-128, 128, (B, Cin, H, W), generator=g, dtype=torch.int8)
    x = torch.where(torch.rand(x.shape, generator=g) < 0.1, torch.full_like(x, 127), x).cuda()
    off = torch.randint(-127, 128, (B, DG * 18, Ho, Wo), generator=g, dtype=torch.int8).cuda()
    mask = torch.randint(-8, 128, (B, DG * 9, Ho, Wo), generator=g, dtype=torch.int8).cuda()   # a few negative masks too
    w = torch.randint(-127, 128, (Cout, Cin // G, 3, 3), generator=g, dtype=torch.int8).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    args = (x, off, mask, w, b, 0.02, 0.03, 1 / 127, 0.004, 0.6, stride, 1, 1, G, DG)
    try:
        lib.bevops_mdconv_set_variant(variant)
        a = bev.modulated_deformable_conv2d_int8(*args)
        a2 = bev.modulated_deformable_conv2d_int8(*args)
        lib.bevops_mdconv_set_variant(6)
        ref = bev.modulated_deformable_conv2d_int8(*args)
    finally:
        lib.bevops_mdconv_set_variant(0)
    torch.cuda.synchronize()
    assert torch.equal(a, a2)
    assert torch.equal(a, ref), ((a.int() - ref.int()).abs().max().item(), (a != ref).float().mean().item())
    assert a.float().abs().mean().item() > 1.0
