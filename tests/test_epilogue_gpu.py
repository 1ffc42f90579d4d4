"""GPU parity of the small fused epilogue passes of the re-hosted neck / encoder input (not reference plugins):
bit-equal to the framework op sequences they replace."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [((6, 256, 116, 200), (58, 100)), ((2, 64, 7, 9), (4, 5)), ((1, 8, 5, 3), (2, 1))])
def test_upsample_add_matches_interpolate_plus_add(shape):
    import bevformer_tensorrt_amd as bev
    (n, c, h, w), (hb, wb) = shape
    g = torch.Generator().manual_seed(0)
    a = torch.randn(n, c, h, w, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(n, c, hb, wb, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    want = a + torch.nn.functional.interpolate(b, size=(h, w), mode="nearest")
    got = bev.upsample_add_nhwc_(a.clone(memory_format=torch.channels_last), b)
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)


def test_feat_embed_matches_adds_and_cat():
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(1)
    levels = [(6, 7), (3, 4), (2, 2)]
    feats = [torch.randn(6, h * w, 256, generator=g).half().cuda() for h, w in levels]
    cam = (torch.randn(6, 256, generator=g) * 0.1).half().cuda()
    lvl = (torch.randn(3, 256, generator=g) * 0.1).half().cuda()
    want = torch.cat([f + cam[:, None, :] + lvl[i][None, None, :] for i, f in enumerate(feats)], dim=1)
    out = torch.full_like(want, float("nan"))
    row = 0
    for i, f in enumerate(feats):
        bev.feat_embed_nhwc(f, cam, lvl[i], out[:, row:row + f.shape[1], :])
        row += f.shape[1]
    assert torch.equal(out, want)


@pytest.mark.parametrize("n,c,h,w", [(6, 64, 464, 800), (2, 64, 7, 9), (1, 8, 1, 1), (3, 16, 2, 5)])
def test_bias_relu_maxpool_equals_two_pass_form(n, c, h, w):
    """bevops_bias_relu_maxpool_nhwc == max_pool2d(bias_act(x), 3, 2, 1), bit for bit (odd sizes, borders)."""
    import torch.nn.functional as F
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(n + c + h + w)
    x = torch.randn(n, c, h, w, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(c, generator=g).half().cuda()
    got = bev.bias_relu_maxpool_nhwc(x, b)
    want = F.max_pool2d(bev.bias_act_nhwc_(x.clone(memory_format=torch.channels_last), b, None, True), 3, 2, 1)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)


@pytest.mark.parametrize("nq", [40000, 22500, 2500, 37, 1])
def test_tsa_split_and_queue_mean_equal_the_framework_ops(nq):
    """bevops_tsa_split == the two view / permute / contiguous copies of temporal_self_attention.py:409-424 (pure data
    movement), bevops_queue_mean2 == torch.mean over the two BEV-queue entries (fp32 sum, one rounding): both bit for bit."""
    import bevformer_tensorrt_amd as bev
    heads, points = 8, 4
    g = torch.Generator().manual_seed(nq)
    both = torch.randn(nq, heads * 2 * points * 3, generator=g).half().cuda()
    n_off = 2 * heads * points * 2
    off = both[:, :n_off].view(1, nq, heads, 2, 1, points, 2).permute(0, 3, 1, 2, 4, 5, 6).contiguous().view(2, nq, heads, -1)
    w = both[:, n_off:].view(1, nq, heads, 2, 1, points).permute(0, 3, 1, 2, 4, 5).contiguous().view(2, nq, heads, -1)
    o2, w2 = bev.tsa_split(both, heads, points)
    assert torch.equal(o2, off) and torch.equal(w2, w)
    x = (torch.randn(2, nq, 256, generator=g) * 3).half().cuda()
    x[0, 0, :4] = torch.tensor([65504.0, -65504.0, 6.1e-5, 5.96e-8]).half()      # extremes: the sum is formed in fp32
    x[1, 0, :4] = torch.tensor([65504.0, 65504.0, 6.1e-5, 5.96e-8]).half()
    assert torch.equal(bev.queue_mean2(x), torch.mean(x, dim=0, keepdim=True))
