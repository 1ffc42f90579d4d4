"""bevops_tile_gemm_f16 (csrc/tile_gemm.hip, the fp16 flavour of the tiled GEMM skeleton) against
torch.nn.functional.linear evaluated in fp32 on the same fp16 operands: one rounding to fp16 is the only
difference (bar: 1 fp16 ulp of the result + fp32 summation-order noise), over the dense-layer shapes of the
re-hosted model and the edge cases of the tiling (row / column / k tails, N not a multiple of 8, tiny M);
and the measured dispatch (functions/linear.py: dense_auto) over its candidates."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(34800, 256, 1024), (34800, 1024, 256), (139200, 128, 512), (40000, 512, 256), (900, 256, 256),
          (333, 100, 64), (1, 8, 8), (129, 136, 72), (8700, 2048, 512)]


def _ref(x, w, b, r, relu):
    y = torch.nn.functional.linear(x.float(), w.float(), None if b is None else b.float())
    if r is not None:
        y = y + r.float()
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("epi", ["plain", "bias_relu", "bias_res", "bias_res_relu"])
def test_tile_gemm_matches_fp32_linear(M, N, K, epi):
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda() if epi != "plain" else None
    r = torch.randn(M, N, generator=g).half().cuda() if "res" in epi else None
    relu = "relu" in epi
    out = bev.tile_gemm(x, w, b, r, relu)
    rows = slice(0, min(M, 4096))
    want = _ref(x[rows], w, b, None if r is None else r[rows], relu)
    err = (out[rows].float() - want).abs()
    tol = 1e-3 * want.abs().clamp_min(1.0) + 2e-3       # fp16 rounding of the result + summation order
    assert bool((err <= tol).all()), float(err.max())
    if M > 4096:                                          # the last row tile (row tail) too
        tail = slice(M - 300, M)
        want = _ref(x[tail], w, b, None if r is None else r[tail], relu)
        assert bool(((out[tail].float() - want).abs() <= 1e-3 * want.abs().clamp_min(1.0) + 2e-3).all())


def test_tile_gemm_rejects_unsupported_k():
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.utils.lib import BevopsError, NOT_SUPPORTED
    x = torch.randn(64, 12).half().cuda()
    w = torch.randn(16, 12).half().cuda()
    with pytest.raises(BevopsError) as e:
        bev.tile_gemm(x, w)
    assert e.value.status == NOT_SUPPORTED


def test_dense_auto_measures_once_and_matches():
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.functions import linear as L
    g = torch.Generator().manual_seed(11)
    M, N, K = 40008, 256, 512          # (a row count no model poses: never measured by a test that ran before)
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    r = torch.randn(M, N, generator=g).half().cuda()
    n_log = len(L.DENSE_LOG)
    y1 = bev.dense_auto(x, w, b, r, False)
    y2 = bev.dense_auto(x, w, b, r, False)
    assert len(L.DENSE_LOG) == n_log + 1                 # one measurement for the problem
    key, times = L.DENSE_LOG[-1]
    assert {"tsgemm", "tile", "blaslt"} <= set(times) and "torch" not in times   # addmm has no identity term
    assert L._DENSE_CHOICE[key] == min(times, key=times.get)
    assert torch.equal(y1, y2)
    want = _ref(x[:2048], w, b, r[:2048], False)
    assert bool(((y1[:2048].float() - want).abs() <= 1e-3 * want.abs().clamp_min(1.0) + 2e-3).all())
    y3 = bev.dense_auto(x, w, b, None, True)             # no identity: the framework path is a candidate too
    assert "torch" in L.DENSE_LOG[-1][1]
    want = _ref(x[:2048], w, b, None, True)
    assert bool(((y3[:2048].float() - want).abs() <= 1e-3 * want.abs().clamp_min(1.0) + 2e-3).all())


@pytest.mark.parametrize("B,C,H,W,Cout", [(2, 64, 37, 53, 64), (1, 128, 16, 20, 128), (6, 64, 232, 400, 64),
                                         (3, 32, 5, 3, 20), (1, 256, 29, 50, 256), (1, 96, 1, 1, 8)])
@pytest.mark.parametrize("epi", ["plain", "bias_relu", "bias_res_relu"])
def test_conv3x3_tile_matches_fp32_conv(B, C, H, W, Cout, epi):
    """bevops_conv3x3_tile_f16 (implicit GEMM: taps as k, zero padding by the buffer range check) vs
    F.conv2d in fp32 on the same fp16 operands; image borders, single-pixel images, Cout not a multiple of 8."""
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(B * C + H * W + Cout)
    x = torch.randn(B, C, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (9 * C) ** 0.5).half().cuda()
    b = torch.randn(Cout, generator=g).half().cuda() if epi != "plain" else None
    r = torch.randn(B, Cout, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last) \
        if "res" in epi else None
    relu = "relu" in epi
    nb = min(B, 2)
    out = bev.conv3x3_nhwc(x, w, b, relu, r)
    assert out.shape == (B, Cout, H, W) and out.is_contiguous(memory_format=torch.channels_last)
    want = torch.nn.functional.conv2d(x[:nb].float(), w.float(), None if b is None else b.float(), 1, 1)
    if r is not None:
        want = want + r[:nb].float()
    if relu:
        want = torch.relu(want)
    err = (out[:nb].float() - want).abs()
    assert bool((err <= 1e-3 * want.abs().clamp_min(1.0) + 2e-3).all()), float(err.max())
    if B > nb:     # the last image (row tail of the pixel tiling)
        want = torch.nn.functional.conv2d(x[-1:].float(), w.float(), None if b is None else b.float(), 1, 1)
        if r is not None:
            want = want + r[-1:].float()
        if relu:
            want = torch.relu(want)
        assert bool(((out[-1:].float() - want).abs() <= 1e-3 * want.abs().clamp_min(1.0) + 2e-3).all())


def test_conv3x3_auto_picks_a_measured_winner(monkeypatch):
    """A problem the shipped dispatch table does not list (the table is emptied for the test) is timed once, both
    candidates, and the faster one is kept for the process."""
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.functions import conv as Cv, linear as Ln
    monkeypatch.setattr(Ln, "_TABLE", {"loaded": True, "dense": {}, "conv": {}})
    monkeypatch.setattr(Cv, "_CHOICE", {})
    monkeypatch.setattr(Cv, "CONV_LOG", [])
    monkeypatch.setattr(Cv, "CONV_MISSES", [])
    g = torch.Generator().manual_seed(2)
    x = torch.randn(6, 128, 116, 200, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(128, 128, 3, 3, generator=g) / 34.0).half().cuda()
    b = torch.randn(128, generator=g).half().cuda()
    y = bev.conv3x3_auto(x, w, b, True)
    assert len(Cv.CONV_LOG) == 1 and len(Cv.CONV_MISSES) == 1
    key, times = Cv.CONV_LOG[-1]
    assert key[1:6] == (6, 116, 200, 128, 128)
    assert set(times) == {"tile", "library"} and Cv._CHOICE[key] == min(times, key=times.get)
    bev.conv3x3_auto(x, w, b, True)
    assert len(Cv.CONV_LOG) == 1      # decided once
    want = torch.relu(torch.nn.functional.conv2d(x[:1].float(), w.float(), b.float(), 1, 1))
    assert bool(((y[:1].float() - want).abs() <= 1e-3 * want.abs().clamp_min(1.0) + 2e-3).all())


@pytest.mark.parametrize("B,C,H,W,Cout,k,stride", [(2, 64, 37, 53, 128, 1, 2), (6, 256, 116, 200, 128, 1, 2),
                                                   (1, 256, 29, 50, 256, 3, 2), (2, 32, 7, 9, 24, 3, 2),
                                                   (2, 64, 8, 8, 64, 1, 1), (1, 32, 5, 4, 16, 3, 3)])
def test_conv_tile_strided_and_pointwise(B, C, H, W, Cout, k, stride):
    """bevops_conv_tile_f16 with a stride (the row addressing sub-samples: no strided copy) and as a 1x1
    convolution, vs F.conv2d in fp32; bias + identity + ReLU epilogue."""
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(B + C + H + W + Cout + k + stride)
    x = torch.randn(B, C, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, C, k, k, generator=g) / (k * k * C) ** 0.5).half().cuda()
    b = torch.randn(Cout, generator=g).half().cuda()
    want = torch.nn.functional.conv2d(x.float(), w.float(), b.float(), stride, k // 2)
    r = torch.randn(want.shape, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    want = torch.relu(want + r.float())
    out = bev.conv_nhwc(x, w, b, True, r, stride)
    assert out.shape == want.shape and out.is_contiguous(memory_format=torch.channels_last)
    err = (out.float() - want).abs()
    assert bool((err <= 1e-3 * want.abs().clamp_min(1.0) + 2e-3).all()), float(err.max())


def test_narrow_tile_layers():
    """N <= 64 takes the 64-column tile flavour (fp16 and both int8 flavours)."""
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(9)
    for M, N, K in ((556800, 64, 64), (4099, 40, 96), (300, 64, 256)):
        x = torch.randn(M, K, generator=g).half().cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
        b = torch.randn(N, generator=g).half().cuda()
        r = torch.randn(M, N, generator=g).half().cuda()
        out = bev.tile_gemm(x, w, b, r, True)
        rows = slice(max(0, M - 3000), M)
        want = _ref(x[rows], w, b, r[rows], True)
        assert bool(((out[rows].float() - want).abs() <= 1e-3 * want.abs().clamp_min(1.0) + 2e-3).all())
        s_x, s_w = float(x.abs().max()) / 127, float(w.abs().max()) / 127
        wq = torch.clamp(torch.round(w.float() / s_w), -127, 127).to(torch.int8)
        q = bev.quantize_rows(x[rows].contiguous(), s_x)
        o8 = bev.linear_int8(q, s_x, wq, s_w, b.float(), r[rows].contiguous(), relu=True)
        acc = q.cpu().long() @ wq.cpu().long().t()
        want8 = torch.relu(acc.double() * (s_x * s_w) + b.cpu().double() + r[rows].cpu().double())
        assert (o8.cpu().double() - want8).abs().max().item() <= 2e-3 * max(1.0, want8.abs().max().item())
