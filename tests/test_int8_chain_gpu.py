"""The INT8 engine's int8 activation chain (functions/int8_chain.py, quantization.Int8ChainBackbone): each operator
against an integer evaluation of the same layer in float64 (the integer sums are exact; the requantisation
q = clamp(rne(v / s_out)) may differ from the kernel's fp32 epilogue by one step on near-ties only), the
channels-last INT8 DCNv2 block against the INT8 PLUGIN entry on the operands it quantises internally (same
arithmetic: bit-identical up to the sigmoid's last ulp), and the chained backbone / the engine against the fp16
model."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _q(t, s):
    return torch.clamp(torch.round(t.float() / s), -127, 127).to(torch.int8)


def _close_int8(got, want, frac=2e-3):
    d = (got.cpu().float() - want.cpu().float()).abs()
    assert d.max().item() <= 1, d.max().item()
    assert (d > 0).float().mean().item() <= frac, (d > 0).float().mean().item()


@pytest.mark.parametrize("M,K,N", [(34800, 1024, 256), (5000, 256, 1024), (777, 64, 64), (4096, 512, 2048)])
@pytest.mark.parametrize("out8", [False, True])
def test_linear_int8_chain_int8_identity(M, K, N, out8):
    from bevformer_tensorrt_amd.functions import int8_chain as C
    g = torch.Generator().manual_seed(M + N)
    a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8)
    w = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8)
    res = torch.randint(-127, 128, (M, N), generator=g, dtype=torch.int8)
    b = torch.randn(N, generator=g)
    s_a, s_w, s_r, s_o = 0.021, 0.0031 / K ** 0.5, 0.043, 0.05
    out = C.linear_int8_chain(a.cuda(), s_a, w.cuda(), s_w, b.cuda(), res.cuda(), s_r, True,
                              torch.int8 if out8 else torch.float16, s_o)
    rows = slice(0, min(M, 3000))
    acc = a[rows].long() @ w.long().t()
    want = torch.relu(acc.double() * (np.float32(s_a) * np.float32(s_w)) + b.double() + res[rows].double() * np.float32(s_r))
    if out8:
        assert out.dtype == torch.int8
        _close_int8(out[rows], torch.clamp(torch.round(want / s_o), -127, 127))
    else:
        err = (out[rows].cpu().double() - want).abs().max().item()
        assert err <= 2e-3 * max(1.0, want.abs().max().item()), err
    # no identity rows, and an fp16 activation through the same entry (quantised in the operand load)
    out2 = C.linear_int8_chain(a.cuda(), s_a, w.cuda(), s_w, b.cuda(), None, 1.0, False, torch.float16)
    want2 = acc.double() * (np.float32(s_a) * np.float32(s_w)) + b.double()
    assert (out2[rows].cpu().double() - want2).abs().max().item() <= 2e-3 * max(1.0, want2.abs().max().item())
    x16 = (a.float() * s_a).half().cuda()
    out3 = C.linear_int8_chain(x16, s_a, w.cuda(), s_w, b.cuda(), None, 1.0, False, torch.float16)
    assert (out3[rows].cpu().double() - want2).abs().max().item() <= 4e-3 * max(1.0, want2.abs().max().item())




@pytest.mark.parametrize("B,Cin,H,W,Cout,k,stride", [(2, 64, 40, 56, 64, 3, 1), (6, 128, 29, 50, 128, 3, 1),
                                                     (2, 256, 24, 30, 32, 3, 1), (2, 256, 30, 44, 512, 1, 2),
                                                     (1, 64, 33, 47, 64, 3, 2)])
@pytest.mark.parametrize("out8", [False, True])
def test_conv_int8_chain(B, Cin, H, W, Cout, k, stride, out8):
    from bevformer_tensorrt_amd.functions import int8_chain as C
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = torch.randint(-127, 128, (B, Cin, H, W), generator=g, dtype=torch.int8)
    w = torch.randint(-127, 128, (Cout, Cin, k, k), generator=g, dtype=torch.int8)
    b = torch.randn(Cout, generator=g)
    sw = (torch.rand(Cout, generator=g) + 0.5) * 0.002 / (Cin * k * k) ** 0.5
    s_a, s_o = 0.017, 0.04
    xq = x.cuda().contiguous(memory_format=torch.channels_last)
    out = C.conv_int8_chain_nhwc(xq, s_a, w.permute(0, 2, 3, 1).contiguous().cuda(), sw.cuda(), b.cuda(), True, stride,
                                 torch.int8 if out8 else torch.float16, s_o)
    acc = F.conv2d(x.double(), w.double(), None, stride, k // 2)              # exact integers in float64
    want = torch.relu(acc * (np.float32(s_a) * sw.double()).view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1))
    assert out.shape == want.shape and out.is_contiguous(memory_format=torch.channels_last)
    if out8:
        _close_int8(out, torch.clamp(torch.round(want / s_o), -127, 127))
    else:
        assert (out.cpu().double() - want).abs().max().item() <= 2e-3 * max(1.0, want.abs().max().item())


def test_stem_pool_int8():
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.functions import int8_chain as C
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 45, 62, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(64, generator=g).half().cuda()
    s = 0.03
    q = C.bias_relu_maxpool_nhwc_int8(x, b, s)
    ref = bev.bias_relu_maxpool_nhwc(x, b)
    want = F.max_pool2d(torch.relu(x.float() + b.float().view(1, -1, 1, 1)), 3, 2, 1)
    assert q.shape == ref.shape and q.is_contiguous(memory_format=torch.channels_last)
    _close_int8(q, torch.clamp(torch.round(want / s), -127, 127))


@pytest.mark.parametrize("B,Ch,H,W,relu", [(6, 256, 58, 100, True), (2, 128, 20, 30, False), (6, 512, 29, 50, True)])
def test_dcn_int8_nhwc_is_the_int8_plugin_on_channels_last(B, Ch, H, W, relu):
    """bevops_mdconv_forward_int8_nhwc against bevops_mdconv_forward_int8 (the INT8 plugin entry) fed the operands
    the channels-last entry quantises internally: offsets / sigmoid(mask logits) with rne, input as is."""
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.functions import int8_chain as C
    g = torch.Generator().manual_seed(B * Ch + H)
    x = torch.randint(-127, 128, (B, Ch, H, W), generator=g, dtype=torch.int8).cuda()
    w = torch.randint(-127, 128, (Ch, Ch, 3, 3), generator=g, dtype=torch.int8).cuda()
    bias = torch.randn(Ch, generator=g).cuda()
    om = torch.zeros(B, 32, H, W)
    om[:, :18] = torch.randn(B, 18, H, W, generator=g) * 2.0
    om[:, 18:27] = torch.randn(B, 9, H, W, generator=g) * 1.5
    om = om.half().cuda().contiguous(memory_format=torch.channels_last)
    s_in, s_off, s_mask, s_w, s_out = 0.02, 6.0 / 127, 1.0 / 127, 0.01 / (Ch * 9) ** 0.5, 0.06
    got = C.modulated_deformable_conv2d_int8_nhwc(x.contiguous(memory_format=torch.channels_last), s_in, om, s_off,
                                                  s_mask, w, s_w, bias, s_out, relu, exact=True)
    fast = C.modulated_deformable_conv2d_int8_nhwc(x.contiguous(memory_format=torch.channels_last), s_in, om, s_off,
                                                   s_mask, w, s_w, bias, s_out, relu)
    # (the divisions on the HOST: the device's tensor / python-scalar division multiplies by the rounded reciprocal)
    off_q = _q(om[:, :18].cpu(), s_off).contiguous().cuda()
    mask_q = _q(torch.sigmoid(om[:, 18:27]).cpu(), s_mask).contiguous().cuda()   # fp16 sigmoid, as the fp16 block's tensor
    want = bev.modulated_deformable_conv2d_int8(x.contiguous(), off_q, mask_q, w, bias, s_in, s_off, s_mask, s_w, s_out,
                                                1, 1, 1, 1, 1)
    if relu:
        want = torch.clamp(want, min=0)
    assert got.shape == want.shape and got.dtype == torch.int8
    d = (got.float() - want.float()).abs()
    # identical integer pipeline; a mask value whose fp16 sigmoid differs in the last ulp between the device's
    # exp and the framework's can move one quantised mask step, i.e. a few outputs by a step or two
    assert (d > 0).float().mean().item() <= 2e-3, (d > 0).float().mean().item()
    assert d.max().item() <= 3, d.max().item()
    # the engine's default flavour (mask folded into the quantised area weights, ONE requantisation per column
    # element): not the plugin's bits, but within its quantisation noise -- a column element moves by at most a step,
    # an output (a 9 * Ch-term dot product of them, requantised) by a few
    df = (fast.float() - want.float()).abs()
    assert df.mean().item() <= 0.6, df.mean().item()
    assert df.max().item() <= 8, df.max().item()


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_chain_backbone_tracks_the_fp16_backbone(name):
    """Int8ChainBackbone (calibrated on 2 frames) against the fp16 channels-last backbone + FPN on an unseen frame."""
    from bevformer_tensorrt_amd import bevformer as B
    from bevformer_tensorrt_amd.quantization import EntropyCalibrator, Int8ChainBackbone
    dev = torch.device("cuda")
    model = B.BEVFormer(name, seed=0).to(dev, torch.float16)
    H, W = B.CONFIGS[name]["image"]
    g = torch.Generator().manual_seed(5)
    imgs = [torch.randn(1, 6, 3, H, W, generator=g).to(dev, torch.float16) for _ in range(3)]
    want = [f.float() for f in model.extract_feat(imgs[2])]
    chain = Int8ChainBackbone(model, EntropyCalibrator())
    for im in imgs[:2]:
        model.extract_feat(im)
    chain.freeze()
    assert chain.ready
    got = model.extract_feat(imgs[2])
    for a, b in zip(got, want):
        assert a.shape == b.shape and a.dtype == torch.float16
        rel = ((a.float() - b).abs().mean() / b.abs().mean()).item()
        assert rel <= 0.08, rel          # 8-bit per-tensor noise through 16 (R50) / 33 (R101) bottlenecks


def test_int8_engine_tiny_runs_and_tracks_fp16():
    from bevformer_tensorrt_amd import bevformer as B, geometry as G
    from bevformer_tensorrt_amd.quantization import build_int8_engine
    dev, dtype = torch.device("cuda"), torch.float16
    H, W = B.CONFIGS["tiny"]["image"]
    l2i = G.synthetic_lidar2img((H, W)).to(dev)
    g = torch.Generator().manual_seed(1)

    def frame(i):
        can = torch.zeros(18)
        can[0], can[1], can[-1] = 0.4 * i, -0.1 * i, 1.0 * i
        return torch.randn(1, 6, 3, H, W, generator=g).to(dev, dtype), can, l2i

    model, qops, note = build_int8_engine(B, "tiny", dev, [frame(i) for i in range(3)])
    assert note["activation_chain"] and note["int8_dense_layers"] > 0
    ref = B.BEVFormer("tiny", seed=0).to(dev, dtype)
    rq, rf = B.FrameRunner(model, dev, dtype), B.FrameRunner(ref, dev, dtype)
    for i in range(2):
        f = frame(10 + i)
        cq, bq = rq.step(*f, "s")
        cf, bf = rf.step(*f, "s")
    assert torch.isfinite(cq.float()).all() and torch.isfinite(bq.float()).all()
    rel = ((rq.prev_bev.float() - rf.prev_bev.float()).abs().mean() / rf.prev_bev.float().std()).item()
    assert rel <= 0.1, rel
    # the same engine replays from a HIP graph
    rg = B.FrameRunner(model, dev, dtype, graph=True)
    f = frame(20)
    a = rg.step(*f, "g")
    b = rg.step(*f, "g")
    assert torch.isfinite(a[0].float()).all() and torch.isfinite(b[0].float()).all()


def test_conv_taps_q_on_a_prequantised_input():
    """ConvTapsQ(prequant=True) -- the FPN's 3x3 convolutions in the INT8 engine: one quantise pass over the fp16 input,
    then the int8 implicit GEMM on the int8 copy -- against its own fake-quant reference (QuantConv2d's formula) and
    against the fused-quantise form (same integers up to rounding ties of x / s vs x * (1 / s))."""
    from bevformer_tensorrt_amd.quantization import ConvTapsQ, MinMaxCalibrator
    g = torch.Generator().manual_seed(4)
    conv = torch.nn.Conv2d(256, 256, 3, 1, 1).cuda().half()
    cal = MinMaxCalibrator()
    m = ConvTapsQ(conv, cal, "site").cuda().half()
    x = torch.randn(2, 256, 29, 50, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    m.calibrate()
    m(x)
    m.freeze()
    ref = m.fake_quant_reference(x)
    m.prequant = True
    y_pre = m.int8_nhwc(x)
    m.prequant = False
    y_fused = m.int8_nhwc(x)
    assert y_pre.shape == ref.shape and y_pre.is_contiguous(memory_format=torch.channels_last)
    assert (y_pre.float() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())
    assert (y_pre.float() - y_fused.float()).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    # stride 2 (the FPN's extra level)
    conv2 = torch.nn.Conv2d(256, 256, 3, 2, 1).cuda().half()
    m2 = ConvTapsQ(conv2, cal, "site2").cuda().half()
    m2.calibrate(); m2(x); m2.freeze()
    m2.prequant = True
    y2 = m2.int8_nhwc(x)
    r2 = m2.fake_quant_reference(x)
    assert y2.shape == r2.shape and (y2.float() - r2).abs().max().item() <= 4e-3 * max(1.0, r2.abs().max().item())


@pytest.mark.parametrize("M,K,N", [(34800, 1024, 256), (34800, 256, 1024), (5000, 128, 512), (8700, 2048, 512), (777, 256, 256)])
@pytest.mark.parametrize("out8", [False, True])
@pytest.mark.parametrize("res", ["none", "int8", "fp16"])
def test_tsgemm_s8_matches_tiled_int8_gemm(M, K, N, out8, res):
    """bevops_tsgemm_s8 (persistent, LDS-DMA operands, 128 k-values per step; functions/int8_chain._TS_S8) against the
    tiled int8 GEMM on the same operands: the int32 sums are exact in both, the fp32 epilogues may contract their
    multiply-adds differently -> int8 outputs equal up to one step on near-ties; and against the integer evaluation in
    float64 on the first rows."""
    from bevformer_tensorrt_amd.functions import int8_chain as C
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8)
    w = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8)
    b = torch.randn(N, generator=g)
    sw = (torch.rand(N, generator=g) + 0.5) * 0.003 / K ** 0.5
    r = {"none": None, "int8": torch.randint(-127, 128, (M, N), generator=g, dtype=torch.int8),
         "fp16": torch.randn(M, N, generator=g).half()}[res]
    s_a, s_r, s_o = 0.021, 0.043, 0.05
    args = (a.cuda(), s_a, w.cuda(), sw.cuda(), b.cuda(), r.cuda() if r is not None else None, s_r, True,
            torch.int8 if out8 else torch.float16, s_o)
    prev = C._TS_S8["enabled"]
    C._TS_S8["enabled"] = False
    try:
        want = C.linear_int8_chain(*args)
        C._TS_S8["enabled"] = True
        got = C.linear_int8_chain(*args)
        torch.cuda.synchronize()
    finally:
        C._TS_S8["enabled"] = prev
    if out8:
        _close_int8(got, want)
    else:
        assert (got.float() - want.float()).abs().max().item() <= 2e-3 * max(1.0, want.float().abs().max().item())
    # exact integer evaluation on the FIRST and on the LAST rows (a persistent kernel's tail partition is where its
    # row bookkeeping can go wrong)
    for rows in (slice(0, min(M, 2000)), slice(max(0, M - 2000), M)):
        acc = a[rows].long() @ w.long().t()
        ref = acc.double() * (np.float32(s_a) * sw.double()) + b.double()
        if r is not None:
            ref = ref + (r[rows].double() * np.float32(s_r) if res == "int8" else r[rows].double())
        ref = torch.relu(ref)
        if out8:
            _close_int8(got[rows], torch.clamp(torch.round(ref / s_o), -127, 127))
        else:
            assert (got[rows].cpu().double() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
