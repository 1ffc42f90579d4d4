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


LN_SHAPES = [  # M, K, bias, residual
    (40000, 256, True, True),        # encoder output_proj + identity -> norm
    (40000, 512, True, True),        # FFN fc2 + identity -> norm
    (900, 256, True, True),          # decoder (object queries)
    (900, 512, True, True),
    (37, 64, False, False), (161, 192, True, False), (5 * 32 * 256 + 7, 128, False, True), (32, 256, True, True),
]


@pytest.mark.parametrize("M,K,has_bias,has_res", LN_SHAPES)
def test_tsgemm_with_layer_norm_epilogue(M, K, has_bias, has_res):
    """bevops_tsgemm_f16_ln (the block's LayerNorm in the epilogue of the block's last GEMM) against (i) the unfused pair
    it replaces -- bevops_tsgemm_f16, then bevops_layer_norm: same binary16 sums, so only the last bit of the
    normalisation may differ -- and (ii) the fp32 evaluation of the same fp16 operands, rounded as the pair rounds
    (the GEMM's result to binary16, then the norm)."""
    import torch.nn.functional as F
    import bevformer_tensorrt_amd as bev
    N = 256
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda() if has_bias else None
    r = (torch.randn(M, N, generator=g) * 2 + 0.3).half().cuda() if has_res else None
    gam = (1 + 0.2 * torch.randn(N, generator=g)).half().cuda()
    bet = (0.1 * torch.randn(N, generator=g)).half().cuda()
    got = bev.tsgemm_ln(x, w, b, r, gam, bet, 1e-5)
    assert got.shape == (M, N) and got.dtype == torch.float16
    pair = bev.layer_norm(bev.tsgemm(x, w, b, r, False), gam, bet, 1e-5)
    d = (got.float() - pair.float()).abs()
    assert d.max().item() <= 4e-3 and d.mean().item() <= 1e-4, (d.max().item(), d.mean().item())
    y = x.float() @ w.float().t()
    if b is not None:
        y = y + b.float()
    if r is not None:
        y = y + r.float()
    want = F.layer_norm(y.half().float(), (N,), gam.float(), bet.float(), 1e-5)
    err = (got.float() - want).abs()
    assert err.max().item() <= 2e-2 and err.mean().item() <= 6e-4, (err.max().item(), err.mean().item())
    # 3-d operands keep their leading dimensions
    if M % 4 == 0:
        got3 = bev.tsgemm_ln(x.view(4, M // 4, K), w, b, None if r is None else r.view(4, M // 4, N), gam, bet, 1e-5)
        assert got3.shape == (4, M // 4, N) and torch.equal(got3.view(M, N), got)


def test_tsgemm_ln_rejects_other_widths():
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.utils import lib as L
    x = torch.zeros(64, 64, dtype=torch.half, device="cuda")
    for n, k in ((512, 64), (256, 48)):
        w = torch.zeros(n, k, dtype=torch.half, device="cuda")
        with pytest.raises(L.BevopsError) as e:
            bev.tsgemm_ln(x[:, :k].contiguous(), w, None, None, torch.ones(n).half().cuda(), torch.zeros(n).half().cuda())
        assert e.value.status == L.NOT_SUPPORTED


def test_model_with_fused_norms_equals_model_with_separate_norms():
    """BEVFormer-tiny, two frames: every encoder / decoder block's LayerNorm in the epilogue of the block's last GEMM
    (default) against the same blocks with GEMM and norm as two launches."""
    import bevformer_tensorrt_amd.functions as hip_ops
    from bevformer_tensorrt_amd import bevformer as B
    from test_model_gpu import run_sequence
    a = run_sequence("tiny", hip_ops, torch.float16, n=2)
    B._LN_FUSED["enabled"] = False
    try:
        b = run_sequence("tiny", hip_ops, torch.float16, n=2)
    finally:
        B._LN_FUSED["enabled"] = True
    for fa, fb in zip(a, b):
        for x, y in zip(fa, fb):
            scale = max(1.0, y.abs().max().item())
            assert (x - y).abs().max().item() <= 4e-2 * scale and (x - y).abs().mean().item() <= 4e-3 * scale
