"""bevops_small_gemm_f16 (csrc/small_gemm.hip: the no-pipeline GEMM for layers with few rows -- the decoder's 900
queries) against torch.nn.functional.linear in fp32 on the same fp16 operands: one rounding to fp16 is the only
difference (bar: 1 fp16 ulp of the result + fp32 summation-order noise).  Shapes: every decoder / head layer of the
re-hosted model, row / column tails, N not a multiple of 8, every supported K."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(900, 768, 256), (900, 256, 256), (900, 64, 256), (900, 32, 256), (900, 10, 256), (900, 512, 256),
          (900, 256, 512), (1, 8, 64), (33, 72, 128), (31, 100, 192), (4096, 512, 1024), (2500, 256, 320)]


def _ref(x, w, b, r, relu):
    y = torch.nn.functional.linear(x.float(), w.float(), None if b is None else b.float())
    if r is not None:
        y = y + r.float()
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("epi", ["plain", "bias_relu", "bias_res", "bias_res_relu"])
def test_small_gemm_matches_fp32_linear(M, N, K, epi):
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda() if epi != "plain" else None
    r = torch.randn(M, N, generator=g).half().cuda() if "res" in epi else None
    relu = "relu" in epi
    out = bev.small_gemm(x, w, b, r, relu)
    want = _ref(x, w, b, r, relu)
    err = (out.float() - want).abs()
    tol = 1e-3 * want.abs().clamp_min(1.0) + 2e-3
    assert out.shape == (M, N) and bool((err <= tol).all()), float(err.max())
    assert torch.equal(out, bev.small_gemm(x, w, b, r, relu))        # fixed summation order


def test_small_gemm_domain():
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.utils.lib import NOT_SUPPORTED, BevopsError
    x = torch.randn(900, 96).half().cuda()
    w = torch.randn(64, 96).half().cuda()
    with pytest.raises(BevopsError) as e:
        bev.small_gemm(x, w)                       # K % 64 != 0
    assert e.value.status == NOT_SUPPORTED
    with pytest.raises(BevopsError):
        bev.small_gemm(torch.randn(9000, 64).half().cuda(), torch.randn(64, 64).half().cuda())   # too many rows


def test_dense_auto_offers_the_small_kernel_to_decoder_shapes():
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.functions import linear as L
    g = torch.Generator().manual_seed(5)
    M, N, K = 904, 256, 256          # (a row count no model poses: measured here, whatever ran before)
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / 16).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    y = bev.dense_auto(x, w, b, None, True)
    key, times = L.DENSE_LOG[-1]
    assert key[1:4] == (M, N, K) and "small" in times and "tile" in times
    want = _ref(x, w, b, None, True)
    assert bool(((y.float() - want).abs() <= 1e-3 * want.abs().clamp_min(1.0) + 2e-3).all())
