"""bevops_bottleneck_c256_64_f16 (csrc/bottleneck.hip): a whole ResNet stage-1 bottleneck (256 -> 64 -> 64 -> 256, identity
shortcut) as one kernel, against the fp32 evaluation and -- bit for bit -- against the three hand-written kernels it fuses."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(B, H, W, seed, bias=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 256, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    w1 = (torch.randn(64, 256, 1, 1, generator=g) / 16).half().cuda()
    w2 = (torch.randn(64, 64, 3, 3, generator=g) / 24).half().cuda()
    w3 = (torch.randn(256, 64, 1, 1, generator=g) / 8).half().cuda()
    bs = [torch.randn(c, generator=g).half().cuda() * 0.5 if bias else None for c in (64, 64, 256)]
    return x, w1, w2, w3, bs


def _chain(x, w1, w2, w3, bs):
    """the same block as three launches of the hand-written kernels (what the model ran before)"""
    from bevformer_tensorrt_amd.functions import conv as CV, linear as Ln
    B, C, H, W = x.shape
    rows = x.permute(0, 2, 3, 1).reshape(-1, C)
    y1 = Ln.tile_gemm(rows, w1.view(64, 256), bs[0], None, True)
    y1 = y1.view(B, H, W, 64).permute(0, 3, 1, 2)
    y2 = CV.conv_nhwc(y1, w2, bs[1], True)
    r2 = y2.permute(0, 2, 3, 1).reshape(-1, 64)
    out = Ln.tile_gemm(r2, w3.view(256, 64), bs[2], rows, True)
    return out.view(B, H, W, 256).permute(0, 3, 1, 2)


@pytest.mark.parametrize("B,H,W", [(1, 8, 8), (2, 16, 24), (1, 17, 19), (3, 5, 7), (1, 1, 1), (2, 33, 40), (6, 58, 100)])
@pytest.mark.parametrize("bias", [True, False])
def test_matches_the_three_kernels_bit_for_bit_and_the_fp32_block(B, H, W, bias):
    """Whole tiles, ragged edges, images smaller than a tile, more tiles than blocks; with and without biases."""
    from bevformer_tensorrt_amd.functions import conv as CV
    x, w1, w2, w3, bs = _case(B, H, W, 5 * B + H + W, bias)
    got = CV.bottleneck_c256_64(x, w1, bs[0], w2, bs[1], w3, bs[2])
    assert got.shape == x.shape and got.is_contiguous(memory_format=torch.channels_last)
    want = _chain(x, w1, w2, w3, bs)
    assert torch.equal(got, want)
    assert torch.equal(got, CV.bottleneck_c256_64(x, w1, bs[0], w2, bs[1], w3, bs[2]))
    f = [None if b is None else b.float() for b in bs]
    y1 = F.relu(F.conv2d(x.float(), w1.float(), f[0])).half().float()
    y2 = F.relu(F.conv2d(y1, w2.float(), f[1], 1, 1)).half().float()
    ref = F.relu(F.conv2d(y2, w3.float(), f[2]) + x.float())
    assert float((got.float() - ref).abs().max()) <= 4e-3 * max(1.0, float(ref.abs().max()))


def test_full_size_block_and_error_codes():
    from bevformer_tensorrt_amd.functions import conv as CV
    from bevformer_tensorrt_amd.utils import lib as _lib
    x, w1, w2, w3, bs = _case(6, 232, 400, 1)
    assert torch.equal(CV.bottleneck_c256_64(x, w1, bs[0], w2, bs[1], w3, bs[2]), _chain(x, w1, w2, w3, bs))
    h = _lib.load_library()
    a = (x.data_ptr(), w1.data_ptr(), None, w2.data_ptr(), None, w3.data_ptr(), None)
    assert h.bevops_bottleneck_c256_64_f16(*a, x.data_ptr(), 1, 8, 8, 256, 64, None) == _lib.BAD_PARAM      # out aliases x
    out = torch.empty_like(x)
    assert h.bevops_bottleneck_c256_64_f16(*a, out.data_ptr(), 1, 8, 8, 512, 128, None) == _lib.NOT_SUPPORTED
    with pytest.raises(_lib.BevopsError):
        CV.bottleneck_c256_64(x, w1, None, w2[:, :32], None, w3, None)
