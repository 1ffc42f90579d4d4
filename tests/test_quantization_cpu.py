"""CPU tests of the native PTQ calibrators."""
import math

import torch

from bevformer_tensorrt_amd import quantization as Q


def test_minmax_tracks_running_max():
    c = Q.get_calibrator("minmax")()
    c.collect("a", torch.tensor([0.5, -2.0]))
    c.collect("a", torch.tensor([1.0, 3.0]))
    assert abs(c.scale("a") - 3.0 / 127) < 1e-9
    q = c.quantize(torch.tensor([3.0, -3.0, 100.0, 0.011]), c.scale("a"))
    assert q.tolist() == [127, -127, 127, 0]


def test_histogram_range_growth_keeps_counts():
    c = Q.PercentileCalibrator(100.0)
    g = torch.Generator().manual_seed(0)
    c.collect("x", torch.rand(10000, generator=g))            # range ~1
    c.collect("x", torch.rand(10000, generator=g) * 7.5)      # forces 3 doublings
    st = c._stats["x"]
    assert st["range"] >= 7.5 and abs(float(st["hist"].sum()) - 20000) < 1e-6
    assert abs(c.scale("x") * 127 - 7.5) < 0.05


def test_entropy_clips_gaussian_tail_but_keeps_uniform():
    g = torch.Generator().manual_seed(0)
    ce = Q.get_calibrator("entropy")()
    cm = Q.get_calibrator("minmax")()
    x = torch.randn(400000, generator=g)
    x[0] = 40.0                                                  # one far outlier
    for t in x.split(100000):
        ce.collect("g", t)
        cm.collect("g", t)
    thr_e, thr_m = ce.scale("g") * 127, cm.scale("g") * 127
    assert thr_m == 40.0
    assert 2.5 < thr_e < 8.0, thr_e                              # outlier ignored, bulk kept
    # a uniform distribution has no tail to clip: threshold ~ the max
    u = torch.rand(200000, generator=g) * 2 - 1
    ce.collect("u", u)
    assert 0.9 < ce.scale("u") * 127 <= 1.01


def test_entropy_threshold_minimises_kl_against_brute_force():
    g = torch.Generator().manual_seed(1)
    hist = torch.histc(torch.randn(50000, generator=g).abs(), bins=2048, min=0, max=6.0).double()
    i = Q.entropy_threshold_bin(hist)

    def kl_at(i):   # independent, loop-based statement of the same objective
        p = hist[:i].clone()
        p[i - 1] += hist[i:].sum()
        q = torch.zeros(i, dtype=torch.float64)
        for lv in range(128):
            lo, hi = int(math.ceil(lv * i / 128 - 0.5 + 1e-9)), int(math.ceil((lv + 1) * i / 128 - 0.5 + 1e-9))
            lo, hi = max(lo, 0), min(hi, i)
            seg = hist[lo:hi]
            nz = (seg > 0).sum()
            if nz > 0:
                q[lo:hi] = torch.where(seg > 0, seg.sum() / nz, torch.zeros_like(seg))
        pn, qn = p / p.sum(), q / q.sum()
        m = pn > 0
        return float((pn[m] * torch.log(pn[m] / qn[m].clamp(min=1e-12))).sum())

    best = min(range(128, 2049, 16), key=kl_at)
    assert abs(kl_at(i + 1) - kl_at(best)) <= 0.02 * max(kl_at(best), 1e-6) + 1e-4


class _FakeOps:
    """Host-only stand-ins with the operator signatures: enough to drive Int8PluginOps' plumbing."""

    def __init__(self):
        self.calls = []

    def multi_scale_deformable_attn(self, value, shapes, ref, off, w):
        self.calls.append("msda")
        return value[:, : off.shape[1]] * 0.5 + off.mean() + w.mean()

    def multi_scale_deformable_attn_int8(self, value, shapes, ref, off, w, s_v, s_o, s_w, s_out):
        self.calls.append(("msda_int8", round(s_v, 9), round(s_o, 9), round(s_w, 9), round(s_out, 9)))
        import torch
        assert value.dtype == off.dtype == w.dtype == torch.int8 and ref.dtype == torch.float16
        real = (value[:, : off.shape[1]].float() * s_v) * 0.5 + (off.float() * s_o).mean() + (w.float() * s_w).mean()
        return torch.clamp(torch.round(real / s_out), -127, 127).to(torch.int8)

    def rotate(self, img, angle, center, interpolation="nearest"):
        self.calls.append("rotate")
        return img.flip(-1)

    def rotate_int8(self, img, angle, center, s_in, s_out, interpolation="nearest"):
        self.calls.append(("rotate_int8", round(s_in, 9), round(s_out, 9)))
        return img.flip(-1)


def test_int8_plugin_ops_plumbing():
    """calibrate -> freeze -> int8: stable per-frame site names, one scale per boundary tensor, the INT8
    flavours receive exactly the calibrator's scales, results de-quantise to the fp results."""
    import torch
    from bevformer_tensorrt_amd.quantization import Int8PluginOps
    fake = _FakeOps()
    q = Int8PluginOps("minmax", fp_ops=fake)
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g)
    frames = [[(mk(1, 6, 2, 4), mk(1, 6, 2, 8), mk(1, 6, 2, 4)) for _ in range(2)] for _ in range(3)]
    img = mk(4, 5, 5)
    shapes, ref = torch.tensor([[2, 3]]), torch.rand(1, 6, 1, 2, generator=g)
    fp_out = []
    for fr in frames:                                   # three calibration frames, two MSDA sites + one rotate each
        q.begin_frame()
        outs = [q.multi_scale_deformable_attn(v, shapes, ref, o, w) for v, o, w in fr]
        outs.append(q.rotate(img, 1.0, (2.0, 2.0)))
        fp_out.append(outs)
    scales = q.freeze()
    assert sorted(scales) == sorted([f"msda#{i}.{k}" for i in (0, 1) for k in ("value", "offsets", "weights", "out")]
                                    + ["rotate#0.img"])
    # min-max over ALL calibration frames of a site
    want = max(float(fr[1][0].abs().max()) for fr in frames) / 127.0
    assert abs(scales["msda#1.value"] - want) <= 1e-12
    fake.calls.clear()
    q.begin_frame()
    v, o, w = frames[2][0]
    out = q.multi_scale_deformable_attn(v, shapes, ref, o, w)
    r = q.rotate(img, 1.0, (2.0, 2.0))
    kind, s_v, s_o, s_w, s_out = fake.calls[0]
    assert kind == "msda_int8" and s_v == round(scales["msda#0.value"], 9) and s_out == round(scales["msda#0.out"], 9)
    assert fake.calls[1][0] == "rotate_int8" and fake.calls[1][1] == fake.calls[1][2] == round(scales["rotate#0.img"], 9)
    assert out.dtype == v.dtype and (out - fp_out[2][0]).abs().max().item() <= 3 * scales["msda#0.out"] + 0.05
    assert (r - fp_out[2][2]).abs().max().item() <= scales["rotate#0.img"]
    # the counters restart with the frame: the same sites are addressed again
    q.begin_frame()
    q.multi_scale_deformable_attn(v, shapes, ref, o, w)
    assert fake.calls[-1][1] == round(scales["msda#0.value"], 9)


def test_quantize_dense_layers_sees_inputs_through_the_model_blocks():
    """The model's blocks must CALL a LinearQ (not read its float weights into the fused-GEMM entry):
    otherwise calibration collects nothing and freeze() has no scale.  The operator set here
    has a `linear_bias_act` entry that must never be reached for a swapped layer."""
    import torch
    from bevformer_tensorrt_amd import bevformer as M
    from bevformer_tensorrt_amd.quantization import MinMaxCalibrator, quantize_dense_layers, LinearQ

    class Ops:
        def linear_bias_act(self, *a):
            raise AssertionError("fused float GEMM used for a quantised layer")

    torch.manual_seed(0)
    ffn = M.FFN(64, 128)
    cal = MinMaxCalibrator()
    swapped = quantize_dense_layers(ffn, cal)
    assert len(swapped) == 2 and all(isinstance(m, LinearQ) for m in (ffn.fc1, ffn.fc2))
    x = torch.randn(1, 10, 64)
    want = x + torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x, ffn.fc1.weight, ffn.fc1.bias)),
                                          ffn.fc2.weight, ffn.fc2.bias)
    for m in swapped:
        m.calibrate()
    got = ffn(x, Ops())
    assert torch.allclose(got, want, atol=1e-6)
    for m in swapped:           # every swapped layer has seen its input -> freeze() finds its scale
        m.freeze()
        assert m.mode == "int8" and m.scale_in > 0 and m.weight_q.dtype == torch.int8
    assert abs(ffn.fc1.scale_in - float(x.abs().max()) / 127.0) < 1e-9


def test_convtapsq_phases_on_the_host():
    """ConvTapsQ (Conv2dQ for plain 3x3 / strided convolutions): float phase = the convolution itself, calibrate
    collects the input under its site, freeze quantises the weight into the taps-major int8 layout the kernel reads
    (k = [tap][Cin]) and the fake-quant reference stays within 8-bit noise of the float layer."""
    import torch
    from bevformer_tensorrt_amd.quantization import ConvTapsQ, MinMaxCalibrator
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(64, 24, 3, 2, 1)
    cal = MinMaxCalibrator()
    q = ConvTapsQ(conv, cal, "conv:site")
    x = torch.randn(2, 64, 9, 7)
    assert torch.equal(q(x), conv(x)) and q.qmode == "float"
    q.calibrate()
    q(x)
    assert abs(cal.scale("conv:site") - float(x.abs().max()) / 127) < 1e-7
    q.freeze()
    assert q.qmode == "int8" and q.weight_q.dtype == torch.int8 and tuple(q.weight_q.shape) == (24, 3, 3, 64)
    w_back = q.weight_q.permute(0, 3, 1, 2).float() * q.scale_w
    assert (w_back - conv.weight).abs().max().item() <= 0.5 * q.scale_w + 1e-7
    ref, want = q.fake_quant_reference(x), conv(x)
    assert ref.shape == want.shape
    assert (ref - want).abs().mean().item() <= 0.03 * want.abs().mean().item()


def test_quantize_backbone_convs_selection():
    """1x1 convolutions -> Conv2dQ always; plain 3x3 with Cin % 64 == 0 -> ConvTapsQ only on request; the 7x7 stem and
    the DCNv2 pack's offset convolution never."""
    import torch
    from bevformer_tensorrt_amd import bevformer as B
    from bevformer_tensorrt_amd.quantization import Conv2dQ, ConvTapsQ, MinMaxCalibrator, quantize_backbone_convs

    def kinds(conv3x3):
        model = B.BEVFormer("small", seed=0)
        swapped = quantize_backbone_convs(model, MinMaxCalibrator(), conv3x3=conv3x3)
        names = {n for n, m in model.named_modules() if isinstance(m, (Conv2dQ, ConvTapsQ))}
        return model, swapped, names

    model, swapped, names = kinds(False)
    assert swapped and all(isinstance(m, Conv2dQ) for m in swapped)
    n_1x1 = len(swapped)
    model, swapped, names = kinds(True)
    taps = [m for m in swapped if isinstance(m, ConvTapsQ)]
    assert len(swapped) - len(taps) == n_1x1 and taps
    assert all(m.kernel_size == (3, 3) and m.in_channels % 64 == 0 for m in taps)
    assert not any(n.endswith("conv_offset") or n.endswith("stem") for n in names)
    assert type(model.backbone.stem) is torch.nn.Conv2d


def test_sites_that_only_saw_empty_batches():
    """A rank of the camera-sharded path without cameras feeds its per-camera layers empty batches: the calibrators
    ignore them, and a layer without statistics freezes with a placeholder scale instead of failing."""
    import torch
    from bevformer_tensorrt_amd.quantization import EntropyCalibrator, LinearQ, MinMaxCalibrator
    for cal in (MinMaxCalibrator(), EntropyCalibrator()):
        cal.collect("site", torch.zeros(0, 256))
        assert not cal.has("site") and cal.scales() == {}
        m = LinearQ.from_linear(torch.nn.Linear(32, 16), cal, "site").calibrate()
        assert m(torch.zeros(0, 32)).shape == (0, 16)
        m.freeze()
        assert m.mode == "int8" and m.scale_in == 1.0 and m.weight_q.dtype == torch.int8
        cal.collect("site", torch.ones(4, 32))
        assert cal.has("site")


def test_entropy_calibrator_ignores_the_relu_zero_spike():
    """Behind a ReLU half (or more) of a tensor is exactly 0 -- representable at any clip point.  As one histogram
    spike it used to dominate KL(P || Q) and pull the threshold to 1.8 sigma (17 % rms error; 66 % for a shifted
    ReLU); with the zero bin neutralised (pytorch_quantization's `bins[0] = bins[1]`) the clip point is the
    tail's."""
    from bevformer_tensorrt_amd.quantization import EntropyCalibrator
    g = torch.Generator().manual_seed(0)
    for shift in (0.0, 1.0):
        x = torch.relu(torch.randn(1_000_000, generator=g) - shift)
        cal = EntropyCalibrator()
        cal.collect("a", x)
        s = cal.scale("a")
        assert s * 127 >= 3.0, s * 127                       # clip point out in the Gaussian tail
        q = torch.clamp(torch.round(x / s), -127, 127) * s
        rel = ((q - x) ** 2).mean().sqrt() / x.std()
        assert rel <= 0.03, rel
