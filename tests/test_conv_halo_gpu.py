"""bevops_conv3x3_c64_f16 (csrc/conv_halo.hip): the 64-channel 3x3 convolution of ResNet stage 1 with both operands in
LDS, against the oracle's convolution (fp32) and -- bit for bit -- against the tiled implicit GEMM it replaces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(B, H, W, seed, bias=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 64, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).half().cuda()
    b = torch.randn(64, generator=g).half().cuda() if bias else None
    return x, w, b


@pytest.mark.parametrize("B,H,W", [(1, 16, 16), (2, 16, 32), (1, 17, 19), (3, 5, 7), (1, 1, 1), (2, 33, 40), (1, 48, 16),
                                   (6, 58, 100)])
@pytest.mark.parametrize("relu", [False, True])
def test_matches_the_tiled_implicit_gemm_bit_for_bit_and_the_fp32_convolution(B, H, W, relu):
    """Whole tiles, ragged right / bottom edges, images smaller than one tile, several images per launch (more tiles
    than one block's share on small grids: the persistent loop and its register prefetch)."""
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.functions import conv as CV
    x, w, b = _case(B, H, W, 7 * B + H + W)
    got = CV.conv3x3_c64(x, w, b, relu)
    tile = CV.conv_nhwc(x, w, b, relu)
    assert got.shape == tile.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, tile)
    want = F.conv2d(x.float(), w.float(), b.float(), 1, 1)
    want = F.relu(want) if relu else want
    err = (got.float() - want).abs()
    assert float(err.max()) <= 2e-3 * max(1.0, float(want.abs().max())) + 1e-3     # one binary16 rounding of |y| <= ~8
    assert bev is not None


def test_full_size_layer_without_bias_and_error_codes():
    """The base model's layer (6 x 232 x 400) against the tiled kernel, no bias; other layers are NOT_SUPPORTED (status
    3) through the C ABI: channel counts, a stride, identity rows."""
    from bevformer_tensorrt_amd.functions import conv as CV
    from bevformer_tensorrt_amd.utils import lib as _lib
    x, w, _ = _case(6, 232, 400, 3, bias=False)
    assert torch.equal(CV.conv3x3_c64(x, w, None, True), CV.conv_nhwc(x, w, None, True))
    h = _lib.load_library()
    st = h.bevops_conv3x3_c64_f16(x.data_ptr(), w.data_ptr(), None, x.data_ptr(), 1, 8, 8, 128, 128, 0, None)
    assert st == _lib.NOT_SUPPORTED
    assert h.bevops_conv3x3_c64_f16(None, w.data_ptr(), None, x.data_ptr(), 1, 8, 8, 64, 64, 0, None) == _lib.BAD_PARAM
    with pytest.raises(_lib.BevopsError) as e:
        CV.conv3x3_c64(x, w, None, True, residual=x)
    assert e.value.status == _lib.NOT_SUPPORTED
    with pytest.raises(_lib.BevopsError):
        CV.conv3x3_c64(x, w, None, True, stride=2)

