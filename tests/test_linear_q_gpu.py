"""INT8 dense layers: bevops_quantize_rows / bevops_linear_int8 and the LinearQ module
against the QuantLinear formula they stand for -- F.linear(dq(q(x)), dq(q(w))) + bias with per-tensor
symmetric scales (pytorch_quantization semantics, det2trt/models/utils/register.py:78-84).  The integer
GEMM is exact, the fake-quant reference sums fp32 products: bar 2e-3 relative to the output scale
(fp16 output rounding), and bit-equality against an int64 evaluation of the same integers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(40000, 256, 256), (40000, 512, 256), (900, 256, 256), (333, 64, 100), (184950, 256, 256)]


@pytest.mark.parametrize("M,K,N", SHAPES)
@pytest.mark.parametrize("per_channel", [False, True])
def test_linear_int8_vs_integer_reference(M, K, N, per_channel):
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g).half().cuda()
    s_x = float(x.abs().max()) / 127
    q = bev.quantize_rows(x, s_x)
    # (on the HOST: the device's tensor / python-scalar division multiplies by the rounded reciprocal)
    want_q = torch.clamp(torch.round(x.float().cpu() / s_x), -127, 127).to(torch.int8)
    assert torch.equal(q.cpu(), want_q)
    if per_channel:
        s_w = (w.abs().amax(1) / 127).clamp_min(1e-12)
        wq = torch.clamp(torch.round(w / s_w[:, None]), -127, 127).to(torch.int8)
        sw_arg = s_w.cuda()
    else:
        s_w = float(w.abs().max()) / 127
        wq = torch.clamp(torch.round(w / s_w), -127, 127).to(torch.int8)
        sw_arg = s_w
    out = bev.linear_int8(q, s_x, wq.cuda(), sw_arg, b.cuda(), r, relu=True)
    rows = slice(0, min(M, 2048))
    acc = q[rows].cpu().long() @ wq.long().t()                      # exact integers
    scale = (s_x * s_w)[None, :] if per_channel else s_x * s_w
    want = torch.relu(acc.double() * scale + b.double() + r[rows].cpu().double())
    err = (out[rows].cpu().double() - want).abs().max().item()
    assert err <= 2e-3 * max(1.0, want.abs().max().item()), err
    # the fp16 activation quantised inside the GEMM's operand load (bevops_linear_int8_fused):
    # q = clamp(rne(x * fl(1 / s_x))) with product and rounding in one fma -- emulated here in float64 (the
    # product of an fp16 and an fp32 value is exact there); the GEMM on those integers must be bit-identical
    r32 = np.float32(1.0) / np.float32(s_x)
    q_f = np.clip(np.rint(x.cpu().numpy().astype(np.float64) * np.float64(r32)), -127, 127).astype(np.int8)
    flips = float((q_f != q.cpu().numpy()).mean())
    assert flips <= 1e-3, flips                      # the two quantisers differ on near-ties of x / s_x only
    out_f = bev.linear_int8(x, s_x, wq.cuda(), sw_arg, b.cuda(), r, relu=True)
    out_q = bev.linear_int8(torch.from_numpy(q_f).cuda(), s_x, wq.cuda(), sw_arg, b.cuda(), r, relu=True)
    assert torch.equal(out_f, out_q)
    # int8 output for a following int8 layer
    o8 = bev.linear_int8(q, s_x, wq.cuda(), sw_arg, b.cuda(), None, relu=False, out_dtype=torch.int8, scale_out=0.05)
    want8 = torch.clamp(torch.round((acc.double() * scale + b.double()).float() / 0.05), -127, 127)
    d = (o8[rows].cpu().float() - want8).abs()
    assert d.max().item() <= 1 and (d > 0).float().mean().item() <= 1e-3


def test_linearq_module_three_phases():
    from bevformer_tensorrt_amd.quantization import EntropyCalibrator, LinearQ, MinMaxCalibrator
    g = torch.Generator().manual_seed(1)
    lin = torch.nn.Linear(256, 512).cuda().half()
    for cal_cls, tol in ((MinMaxCalibrator, 3e-2), (EntropyCalibrator, 8e-2)):
        cal = cal_cls()
        m = LinearQ.from_linear(lin, cal, "site")
        x = torch.randn(4096, 256, generator=g).half().cuda()
        y0 = m(x)
        assert torch.equal(y0, lin(x))
        m.calibrate()
        for _ in range(3):
            m(torch.randn(4096, 256, generator=g).half().cuda())
        m.freeze()
        y = m(x)
        ref = m.fake_quant_reference(x)
        assert (y.float() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())
        rel = (y.float() - lin(x).float()).abs().mean().item() / lin(x).float().abs().mean().item()
        assert rel <= tol, rel      # 8-bit per-tensor quantisation noise of a 256-deep dot product


def test_linear_int8_fused_quantiser_is_the_row_quantiser_off_ties():
    """x on a grid that has no near-ties (integers times the scale, +- 0.25 step): the fused operand load and
    bevops_quantize_rows must produce the same integers, hence bit-identical outputs; and saturation."""
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(5)
    M, K, N = 3000, 192, 136
    s_x = 0.03125                                            # a power of two: k * s_x is exact in fp16
    k = torch.randint(-200, 201, (M, K), generator=g).float()           # beyond +-127: clamps
    x = ((k + 0.25 * torch.randint(-1, 2, (M, K), generator=g)) * s_x).half().cuda()
    wq = torch.randint(-127, 128, (N, K), generator=g).to(torch.int8).cuda()
    b = torch.randn(N, generator=g).cuda()
    q = bev.quantize_rows(x, s_x)
    assert int(q.max()) == 127 and int(q.min()) == -127
    a = bev.linear_int8(q, s_x, wq, 0.01, b, None, relu=False)
    f = bev.linear_int8(x, s_x, wq, 0.01, b, None, relu=False)
    assert torch.equal(a, f)
    a8 = bev.linear_int8(q, s_x, wq, 0.01, b, None, relu=True, out_dtype=torch.int8, scale_out=0.5)
    f8 = bev.linear_int8(x, s_x, wq, 0.01, b, None, relu=True, out_dtype=torch.int8, scale_out=0.5)
    assert torch.equal(a8, f8)
    want = torch.relu((q.cpu().long() @ wq.cpu().long().t()).double() * (s_x * 0.01) + b.cpu().double())
    d = (a8.cpu().double() - torch.clamp(torch.round(want / 0.5), -127, 127)).abs()
    assert d.max().item() <= 1 and (d > 0).float().mean().item() <= 1e-3


def test_dequantize_rows_is_the_two_pass_formula():
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(6)
    q = torch.randint(-127, 128, (3, 4001, 8), generator=g).to(torch.int8)
    for s in (0.0371, 1.0, 3.1e-4):
        got = bev.dequantize_rows(q.cuda(), s)
        want = (q.float() * s).half()          # product in fp32, one rounding
        assert got.shape == q.shape and torch.equal(got.cpu(), want)


@pytest.mark.parametrize("B,C,H,W,Cout,k,stride", [(2, 64, 37, 53, 64, 3, 1), (6, 128, 116, 200, 128, 3, 1),
                                                   (1, 256, 29, 50, 256, 3, 2), (2, 256, 20, 30, 128, 1, 2),
                                                   (1, 64, 3, 2, 24, 3, 1)])
def test_conv_int8_matches_integer_reference(B, C, H, W, Cout, k, stride):
    """bevops_conv_tile_int8_fused (ConvTapsQ / strided Conv2dQ): the integers of the in-kernel quantiser (emulated
    in float64) convolved exactly (int64) then de-quantised, bias + identity + ReLU; zero padding, stride, tails."""
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(B + C + H + W + Cout + k + stride)
    x = torch.randn(B, C, H, W, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, C, k, k, generator=g) / (k * k * C) ** 0.5
    b = torch.randn(Cout, generator=g)
    s_x, s_w = float(x.abs().max()) / 127, float(w.abs().max()) / 127
    wq = torch.clamp(torch.round(w / s_w), -127, 127).to(torch.int8)
    r32 = np.float32(1.0) / np.float32(s_x)
    q = np.clip(np.rint(x.cpu().numpy().astype(np.float64) * np.float64(r32)), -127, 127)
    acc = torch.nn.functional.conv2d(torch.from_numpy(q).double(), wq.double(), None, stride, k // 2)   # exact integers
    r = torch.randn(acc.shape, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    want = torch.relu(acc * (s_x * s_w) + b.double().view(1, -1, 1, 1) + r.cpu().double())
    out = bev.conv_int8_nhwc(x, s_x, wq.permute(0, 2, 3, 1).contiguous().cuda(), s_w, b.cuda(), True, r, stride)
    assert out.shape == want.shape and out.is_contiguous(memory_format=torch.channels_last)
    err = (out.cpu().double() - want).abs().max().item()
    assert err <= 2e-3 * max(1.0, want.abs().max().item()), err


def test_convtapsq_module_three_phases():
    from bevformer_tensorrt_amd.quantization import ConvTapsQ, MinMaxCalibrator
    g = torch.Generator().manual_seed(8)
    conv = torch.nn.Conv2d(128, 64, 3, 1, 1).cuda().half()
    m = ConvTapsQ(conv, MinMaxCalibrator(), "site").cuda().half()
    x = torch.randn(2, 128, 24, 40, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    # float phase = the library convolution on the shared weights (two calls of it may pick different algorithms)
    assert (m(x).float() - conv(x).float()).abs().max().item() <= 4e-3
    m.calibrate()
    m(x)
    m.freeze()
    y = m(x)
    ref = m.fake_quant_reference(x)
    assert (y.float() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())
    rel = (y.float() - conv(x).float()).abs().mean().item() / conv(x).float().abs().mean().item()
    assert rel <= 3e-2, rel
