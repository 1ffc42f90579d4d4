"""The one-kernel ResNet stem (bevops_stem_conv_pool, csrc/stem.hip: conv 7x7 / 2 / 3 from 3 to 64 channels + shift ->
ReLU -> max_pool 3 / 2 / 1, from planar images to the pooled channels-last activation) against the framework's fp32
evaluation of the same three layers (backbones/resnet.py: conv1 -> norm1 (folded) -> relu -> maxpool).  The kernel
accumulates fp16 products in fp32 and rounds once: the bar is the binary16 rounding of the result plus the fp32
summation order (2e-3 relative + 2e-3 absolute on values of order 1-10)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(n, h, w, seed, bias=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, h, w, generator=g).half().cuda()
    wt = (torch.randn(64, 3, 7, 7, generator=g) / 12).half().cuda()
    b = torch.randn(64, generator=g).half().cuda() if bias else None
    return x, wt, b


def _want(x, wt, b):
    y = F.conv2d(x.float(), wt.float(), None if b is None else b.float(), 2, 3)
    return F.max_pool2d(F.relu(y), 3, 2, 1)


# (images, height, width): odd heights, widths that are not a multiple of the 60-column block or the 15-column strip,
# a 2 x 2 result, one base camera image, the six tiny images
@pytest.mark.parametrize("n,h,w", [(1, 20, 26), (2, 37, 130), (2, 8, 8), (1, 66, 250), (3, 7, 2), (1, 928, 1600),
                                   (6, 480, 800), (1, 129, 482)])
@pytest.mark.parametrize("bias", [True, False])
def test_stem_matches_fp32_conv_relu_pool(n, h, w, bias):
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.utils import lib as L
    x, wt, b = _case(n, h, w, seed=h * 7 + w + n, bias=bias)
    want = _want(x, wt, b)
    handle = L.load_library()
    outs = []
    try:
        for variant in (0, 1):      # pooling neighbours through DPP wave shifts / through ds_bpermute
            handle.bevops_stem_set_variant(variant)
            got = bev.stem_conv_pool(x, wt, b)
            torch.cuda.synchronize()
            assert got.shape == want.shape and got.dtype == torch.float16
            assert got.is_contiguous(memory_format=torch.channels_last)
            err = (got.float() - want).abs()
            assert bool((err <= 2e-3 * want.abs() + 2e-3).all()), (variant, float(err.max()))
            outs.append(got)
    finally:
        handle.bevops_stem_set_variant(0)
    assert torch.equal(outs[0], outs[1])


def test_stem_equals_the_two_pass_form_to_its_intermediate_rounding():
    """Against the form it replaces in the frame (library convolution -> bevops_bias_relu_maxpool_nhwc): that form
    rounds the convolution's output to binary16 before the shift; the results agree to that rounding."""
    import bevformer_tensorrt_amd as bev
    x, wt, b = _case(2, 232, 400, seed=5)
    y = F.conv2d(x.contiguous(memory_format=torch.channels_last), wt, None, 2, 3).contiguous(memory_format=torch.channels_last)
    two_pass = bev.bias_relu_maxpool_nhwc(y, b)
    got = bev.stem_conv_pool(x, wt, b)
    err = (got.float() - two_pass.float()).abs()
    assert bool((err <= 4e-3 * two_pass.float().abs() + 4e-3).all()), float(err.max())


@pytest.mark.parametrize("n,h,w", [(2, 37, 130), (1, 232, 400)])
def test_stem_int8_output_is_the_quantised_fp32_result(n, h, w):
    """out_dtype int8: q = min(rne(v / s), 127) of the fp32 pooled value.  Against the fp32 reference quantised the
    same way: equal except where v / s sits within the summation-order noise of a rounding boundary (<= 1 LSB there)."""
    import bevformer_tensorrt_amd as bev
    x, wt, b = _case(n, h, w, seed=11)
    want = _want(x, wt, b)
    s = float(want.max()) / 150.0          # (the largest values clamp at 127)
    q = bev.stem_conv_pool(x, wt, b, s)
    assert q.dtype == torch.int8 and q.is_contiguous(memory_format=torch.channels_last)
    ref = torch.clamp(torch.round(want / s), max=127)
    d = (q.float() - ref).abs()
    assert float(d.max()) <= 1.0 and float((d == 0).float().mean()) >= 0.995, (float(d.max()), float((d == 0).float().mean()))
    assert int(q.max()) == 127 and int(q.min()) == 0


def test_stem_entry_validates_its_arguments():
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.utils import lib as L
    h = L.load_library()
    x, wt, b = _case(1, 16, 16, seed=1)
    with pytest.raises(ValueError):
        bev.stem_conv_pool(x[..., :15].contiguous(), wt, b)          # odd width
    with pytest.raises(ValueError):
        bev.stem_conv_pool(x, wt[:32], b)
    assert h.bevops_stem_packed_size() == 11 * 2 * 64 * 8 * 2
    packed = torch.empty(h.bevops_stem_packed_size(), dtype=torch.uint8, device="cuda")
    out = torch.empty(1, 4, 4, 64, dtype=torch.half, device="cuda")
    st = L.current_stream_ptr(x.device)
    assert h.bevops_stem_pack(L.F32, wt.data_ptr(), b.data_ptr(), packed.data_ptr(), st) == L.NOT_SUPPORTED
    assert h.bevops_stem_pack(L.F16, wt.data_ptr(), None, packed.data_ptr(), st) == 0
    assert h.bevops_stem_conv_pool(L.F16, L.F32, x.data_ptr(), packed.data_ptr(), out.data_ptr(), 1, 16, 16, 0.0, st) == L.NOT_SUPPORTED
    assert h.bevops_stem_conv_pool(L.F16, L.F16, x.data_ptr(), packed.data_ptr(), out.data_ptr(), 1, 16, 15, 0.0, st) == L.NOT_SUPPORTED
    assert h.bevops_stem_conv_pool(L.F16, L.I8, x.data_ptr(), packed.data_ptr(), out.data_ptr(), 1, 16, 16, 0.0, st) == L.BAD_PARAM
    assert h.bevops_stem_conv_pool(L.F16, L.F16, x.data_ptr(), packed.data_ptr(), out.data_ptr(), 0, 16, 16, 0.0, st) == 0
    torch.cuda.synchronize()


def test_backbone_with_the_fused_stem_matches_the_library_stem():
    """ResNet.forward_nhwc with the one-kernel stem against the same network with BEVOPS_STEM_FUSED off (library
    convolution + pooling pass): stage outputs agree to fp16 noise."""
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd import bevformer as B
    from bevformer_tensorrt_amd.functions import conv as Cv
    torch.manual_seed(0)
    net = B.ResNet(50, (False, False, False, False), (1, 2, 3), bev).cuda().half()
    img = torch.randn(2, 3, 96, 160, device="cuda").half()
    assert net._stem_fused(img, bev) is not None          # the network takes the one-kernel stem for this input
    fused = net.forward_nhwc(img, bev)
    Cv.STEM_FUSED["enabled"] = False
    try:
        plain = net.forward_nhwc(img, bev)
    finally:
        Cv.STEM_FUSED["enabled"] = True
    for a, b in zip(fused, plain):
        assert a.shape == b.shape
        scale = float(b.float().abs().max())
        assert scale > 0 and scale < 6e4
        assert float((a.float() - b.float()).abs().max()) <= 2e-2 * scale
