"""GPU parity of the hand-written tall-skinny MFMA GEMM (csrc/tsgemm.hip) against the fp32 evaluation of the
same fp16 operands: one rounding of the fp32 result (<= half an fp16 ulp of the value + accumulation-order
noise), for the dense-layer shapes of BEVFormer-base and ragged row counts."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # M, N, K, bias, residual, relu
    (34800, 256, 1024, True, False, True),      # ResNet stage 3 conv1
    (34800, 1024, 256, True, True, True),       # stage 3 conv3 + identity
    (8700, 2048, 512, True, True, True),        # stage 4 conv3
    (184950, 256, 256, True, False, False),     # SCA value_proj
    (40000, 512, 256, True, False, True),       # FFN fc1
    (40000, 256, 512, True, True, False),       # FFN fc2 + identity
    (1, 256, 64, False, False, False), (31, 256, 128, True, True, False), (33, 512, 64, True, False, True),
    (161, 256, 192, False, True, True), (5 * 32 * 256 + 7, 256, 64, True, False, False),
]


@pytest.mark.parametrize("M,N,K,has_bias,has_res,relu", SHAPES)
def test_tsgemm_matches_fp32_reference(M, N, K, has_bias, has_res, relu):
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda() if has_bias else None
    r = torch.randn(M, N, generator=g).half().cuda() if has_res else None
    got = bev.tsgemm(x, w, b, r, relu).float()
    want = x.float() @ w.float().t()
    if b is not None:
        want = want + b.float()
    if r is not None:
        want = want + r.float()
    if relu:
        want = torch.relu(want)
    err = (got - want).abs()
    tol = 1e-3 * want.abs() + 2e-3          # fp16 rounding of the result + fp32 summation-order noise
    assert bool((err <= tol).all()), (err.max().item(), (err / (want.abs() + 1e-3)).max().item())


def test_tsgemm_rejects_shapes_outside_its_domain():
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.utils import lib
    x = torch.zeros(64, 96, dtype=torch.float16, device="cuda")
    with pytest.raises(lib.BevopsError) as e:
        bev.tsgemm(x, torch.zeros(256, 96, dtype=torch.float16, device="cuda"))
    assert e.value.status == lib.NOT_SUPPORTED
    with pytest.raises(lib.BevopsError):
        bev.tsgemm(torch.zeros(64, 128, dtype=torch.float16, device="cuda"), torch.zeros(100, 128, dtype=torch.float16, device="cuda"))
