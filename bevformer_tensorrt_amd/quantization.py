"""Per-tensor PTQ calibrators producing the scales the INT8 operators take.

In the reference the scales come from TensorRT: `get_calibrator("minmax" | "entropy" | "legacy")`
wraps `trt.IInt8MinMaxCalibrator` / `IInt8EntropyCalibrator2` / `IInt8LegacyCalibrator`
(det2trt/quantization/calibrator_trt.py:6-92) and the engine hands each plugin its tensors'
`PluginTensorDesc::scale` (multiScaleDeformableAttnPlugin.cpp:75-77); the QDQ flow uses
pytorch_quantization's max / 99.99-percentile histogram calibrators
(det2trt/quantization/calibrator_qdq.py:29-80).  Neither library exists on ROCm, so the
published algorithms are restated here as host logic (torch tensors, histograms on the
device the data lives on):

  * MinMaxCalibrator      scale = max|x| / 127 over all calibration batches
  * PercentileCalibrator  scale = percentile(|x|, p) / 127 from a running histogram (p = 99.99)
  * EntropyCalibrator     TensorRT "entropy calibration 2" (8-bit Inference with TensorRT,
                          S. Migacz, GTC 2017): 2048-bin histogram of |x|; for every candidate
                          clip bin i in [128, 2048): reference P = hist[:i] with the outliers
                          folded into the last bin, candidate Q = P merged into 128 levels and
                          expanded back over P's non-empty bins; threshold = argmin KL(P || Q).

`get_calibrator(name)` mirrors the reference's factory; `Calibrator.collect(name, tensor)` is
called per boundary tensor per calibration frame; `scales()` returns {name: scale}.
"""
import math

import os

import torch

from .functions.multi_scale_deformable_attn import _TensorCache

_NUM_BINS = 2048
_NUM_LEVELS = 128


class _Base:
    def __init__(self):
        self._stats = {}

    def collect(self, name, tensor):
        raise NotImplementedError

    def scale(self, name):
        raise NotImplementedError

    def scales(self):
        return {k: self.scale(k) for k in self._stats}

    def has(self, name):
        """False for a site that never saw data (a layer fed empty tensors only: a rank of the camera-sharded
        path without cameras runs its per-camera layers on empty batches)."""
        return name in self._stats

    @staticmethod
    def quantize(tensor, scale):
        """real -> int8 with round-to-nearest-even and saturation to [-127, 127]."""
        return torch.clamp(torch.round(tensor.float() / scale), -127, 127).to(torch.int8)


class MinMaxCalibrator(_Base):
    def collect(self, name, tensor):
        if tensor.numel() == 0:
            return
        m = float(tensor.detach().abs().max())
        self._stats[name] = max(self._stats.get(name, 0.0), m)

    def scale(self, name):
        return max(self._stats[name], 1e-12) / 127.0


class _Histogram(_Base):
    """Running histogram of |x| with a range that doubles when a batch exceeds it (bins merge
    pairwise, so earlier batches stay exactly accounted for)."""

    def collect(self, name, tensor):
        if tensor.numel() == 0:
            return
        x = tensor.detach().abs().float().flatten()
        amax = float(x.max())
        st = self._stats.get(name)
        if st is None:
            st = {"hist": torch.zeros(_NUM_BINS, dtype=torch.float64), "range": max(amax, 1e-12)}
            self._stats[name] = st
        while amax > st["range"]:
            st["hist"] = torch.cat([st["hist"].view(-1, 2).sum(1), torch.zeros(_NUM_BINS // 2, dtype=torch.float64)])
            st["range"] *= 2
        h = torch.histc(x, bins=_NUM_BINS, min=0.0, max=st["range"])
        st["hist"] += h.double().cpu()


class PercentileCalibrator(_Histogram):
    def __init__(self, percentile=99.99):
        super().__init__()
        self.percentile = percentile

    def scale(self, name):
        st = self._stats[name]
        cdf = torch.cumsum(st["hist"], 0)
        idx = int(torch.searchsorted(cdf, cdf[-1] * self.percentile / 100.0))
        return (min(idx, _NUM_BINS - 1) + 0.5) * st["range"] / _NUM_BINS / 127.0


class EntropyCalibrator(_Histogram):
    def scale(self, name):
        st = self._stats[name]
        i = entropy_threshold_bin(st["hist"])
        return (i + 0.5) * st["range"] / _NUM_BINS / 127.0


def entropy_threshold_bin(hist, num_levels=_NUM_LEVELS):
    """Clip bin minimising KL(P || Q) as described in the module docstring.  All candidate clip
    bins are evaluated at once (a [candidates, bins] batch) instead of one Python iteration each."""
    hist = hist.double().cpu()
    n = hist.numel()
    cand = torch.arange(num_levels, n + 1)                       # i = number of bins kept
    k = torch.arange(n)
    keep = k[None, :] < cand[:, None]                            # [I, n]
    csum = torch.cumsum(hist, 0)
    tail = csum[-1] - csum[cand - 1]                             # mass beyond the clip bin
    src = hist[None, :] * keep                                   # hist[:i]
    p = src.clone()
    p[torch.arange(cand.numel()), cand - 1] += tail              # outliers folded into the last bin
    # merge the i kept bins into num_levels levels, expand back over the non-empty bins
    idx = torch.ceil((k[None, :].double() + 0.5) * num_levels / cand[:, None].double()).long() - 1
    idx = idx.clamp_(0, num_levels - 1)
    nonzero = (src > 0).double()
    level_sum = torch.zeros(cand.numel(), num_levels, dtype=torch.float64).scatter_add_(1, idx, src)
    level_cnt = torch.zeros(cand.numel(), num_levels, dtype=torch.float64).scatter_add_(1, idx, nonzero)
    q = torch.gather(level_sum / level_cnt.clamp(min=1), 1, idx) * nonzero
    psum, qsum = p.sum(1, keepdim=True), q.sum(1, keepdim=True)
    ok = (psum > 0) & (qsum > 0)
    pn = p / psum.clamp(min=1e-300)
    qn = (q / qsum.clamp(min=1e-300)).clamp(min=1e-12)           # P > 0 with Q == 0: the folded bin only
    term = torch.where(pn > 0, pn * torch.log(pn.clamp(min=1e-300) / qn), torch.zeros_like(pn))
    kl = term.sum(1)
    kl[~ok[:, 0]] = math.inf
    best = int(torch.argmin(kl))                                 # first minimum, like the loop
    return int(cand[best]) - 1 if math.isfinite(float(kl[best])) else n - 1


CALIBRATORS = {"minmax": MinMaxCalibrator, "entropy": EntropyCalibrator, "percentile": PercentileCalibrator,
               "legacy": PercentileCalibrator}


def get_calibrator(calibrator):
    """Factory with the reference's names (det2trt/quantization/calibrator_trt.py:6-16)."""
    assert calibrator in CALIBRATORS, f"calibrator should be in {sorted(CALIBRATORS)}"
    return CALIBRATORS[calibrator]


class Int8PluginOps:
    """INT8 scale plumbing for the plugin call sites of a model.

    An operator namespace that can be handed to `BEVFormer(ops=...)` in place of
    `bevformer_tensorrt_amd.functions`.  Two phases, like a TensorRT PTQ build
    (tools/bevformer/onnx2trt.py:110-241 feeds calibration batches through the network, TensorRT
    records one scale per plugin tensor, the engine then hands `PluginTensorDesc::scale` to
    `enqueue`):

      calibrate : every plugin call runs the fp operator while the calibrator collects its boundary
                  tensors under a stable name  "<op>#<call index within the frame>.<tensor>";
      int8      : the same call sites quantise their inputs with the recorded scales, run the
                  INT8 flavour of the operator (`*_int8`) and de-quantise the result.

    Only the sampling operators are quantised (the scope of the custom plugins); dense layers stay
    in the model's dtype.  `begin_frame()` resets the per-frame call counters (the model's
    forward pre-hook installed by `attach`).  The fused / channels-last entries are deliberately not
    exposed, so a model built on this namespace takes the reference op sequence.
    """

    # fp entries a `channels_last=True` namespace passes through: the channels-last backbone (its DCNv2
    # block, offset convolution and epilogues stay in the model's dtype there -- the INT8 DCNv2 operator
    # exists for the plugin layout only), the fused dense / norm entries for layers that were not swapped
    # for LinearQ, and -- `fused_sca=True` -- the fused fp16 SCA sampler.  Like a TensorRT INT8 engine the
    # build is then mixed: INT8 where an INT8 implementation exists and pays, fp16 elsewhere.
    _PASS = ("bias_act_nhwc_", "conv_offset_nhwc", "modulated_deformable_conv2d_nhwc", "layer_norm",
             "linear_bias_act", "dense_auto", "conv3x3_auto", "conv_nhwc", "conv_int8_nhwc", "bias_relu_maxpool_nhwc", "image_normalize_pad")

    def __init__(self, calibrator="entropy", fp_ops=None, channels_last=False, fused_sca=False):
        from . import functions as _f
        self.fp = fp_ops if fp_ops is not None else _f
        self.cal = get_calibrator(calibrator)() if isinstance(calibrator, str) else calibrator
        self.mode = "calibrate"
        self._scales = {}
        self._n = {}
        self._wq = _TensorCache()      # float weight tensor -> (int8 weight, scale)
        self._pass = self._PASS + (("spatial_cross_attention_sample",) if fused_sca else ()) if channels_last else ()

    def __getattr__(self, name):       # only reached for names this class does not define
        if name in self.__dict__.get("_pass", ()) and hasattr(self.fp, name):
            return getattr(self.fp, name)
        raise AttributeError(name)

    # ---- bookkeeping
    def begin_frame(self):
        self._n = {}

    def attach(self, model):
        model.register_forward_pre_hook(lambda m, args: self.begin_frame())
        return self

    def freeze(self):
        """End of calibration: fix the scales and switch the call sites to their INT8 flavours."""
        self._scales = self.cal.scales()
        self._wq = _TensorCache()
        self.mode = "int8"
        return self._scales

    def _site(self, op):
        i = self._n.get(op, 0)
        self._n[op] = i + 1
        return f"{op}#{i}"

    def _q(self, name, t):
        s = self._scales[name]
        if t.is_cuda and t.dtype == torch.float16 and t.numel() % 8 == 0 and hasattr(self.fp, "quantize_rows"):
            # one pass (bevops_quantize_rows: clamp(rne(x / s)) with the division in fp32, the calibrators' host
            # formula) instead of the five framework launches and the fp32 copy of `_Base.quantize`
            return self.fp.quantize_rows(t.contiguous(), s), s
        return self.cal.quantize(t, s), s

    def _dq(self, q, s, dtype):
        """int8 result of a plugin -> the model's dtype: q * s, one pass when the operator set has the entry."""
        if q.is_cuda and dtype == torch.float16 and q.numel() % 8 == 0 and hasattr(self.fp, "dequantize_rows"):
            return self.fp.dequantize_rows(q, s)
        return q.to(dtype) * s

    # ---- call sites
    def multi_scale_deformable_attn(self, value, shapes, ref, off, w):
        site = self._site("msda")
        if self.mode == "calibrate":
            out = self.fp.multi_scale_deformable_attn(value, shapes, ref, off, w)
            for k, t in (("value", value), ("offsets", off), ("weights", w), ("out", out)):
                self.cal.collect(f"{site}.{k}", t)
            return out
        (qv, sv), (qo, so), (qw, sw) = (self._q(f"{site}.{k}", t) for k, t in
                                        (("value", value), ("offsets", off), ("weights", w)))
        s_out = self._scales[f"{site}.out"]
        # reference points stay fp16: the <__half2> flavour (multiScaleDeformableAttnPlugin.cpp:118-125)
        out = self.fp.multi_scale_deformable_attn_int8(qv, shapes, ref.to(torch.float16).contiguous(),
                                                       qo.contiguous(), qw.contiguous(), sv, so, sw, s_out)
        return self._dq(out, s_out, value.dtype)

    multi_scale_deformable_attn2 = multi_scale_deformable_attn

    def rotate(self, img, angle, center, interpolation="nearest"):
        site = self._site("rotate")
        if self.mode == "calibrate":
            out = self.fp.rotate(img, angle, center, interpolation)
            self.cal.collect(f"{site}.img", img)
            return out
        q, s = self._q(f"{site}.img", img)
        out = self.fp.rotate_int8(q.contiguous(), angle, center, s, s, interpolation)
        return self._dq(out, s, img.dtype)

    rotate2 = rotate

    def modulated_deformable_conv2d(self, x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1,
                                    groups=1, deform_groups=1):
        site = self._site("dcn")
        if self.mode == "calibrate":
            out = self.fp.modulated_deformable_conv2d(x, offset, mask, weight, bias, stride, padding, dilation,
                                                      groups, deform_groups)
            for k, t in (("x", x), ("offset", offset), ("mask", mask), ("weight", weight), ("out", out)):
                self.cal.collect(f"{site}.{k}", t)
            return out
        (qx, sx), (qo, so), (qm, sm) = (self._q(f"{site}.{k}", t) for k, t in
                                        (("x", x), ("offset", offset), ("mask", mask)))
        hit = self._wq.get(weight)           # the weights of a deployed model are quantised (and packed) once
        if hit is None:
            hit = self._wq.put(weight, self._q(f"{site}.weight", weight))
        qw, sw = hit
        s_out = self._scales[f"{site}.out"]
        out = self.fp.modulated_deformable_conv2d_int8(qx, qo, qm, qw, None if bias is None else bias.float(),
                                                       sx, so, sm, sw, s_out, stride, padding, dilation, groups,
                                                       deform_groups)
        return self._dq(out, s_out, x.dtype)

    modulated_deformable_conv2d2 = modulated_deformable_conv2d


FUSED_QUANT = {"enabled": os.environ.get("BEVOPS_FUSED_QUANT", "1") != "0"}   # A/B: LinearQ hands the fp16 activation to bevops_linear_int8_fused


class LinearQ(torch.nn.Linear):
    """`LinearQ` of the reference (det2trt/models/utils/register.py:83: pytorch_quantization's
    QuantLinear, selected through LINEAR_LAYERS at modules/spatial_cross_attention.py:58,329): a
    Linear whose input and weight are quantised per tensor (symmetric, amax / 127, round half even,
    clamp to +-127).  Three phases, like the reference's calibrator_qdq flow
    (det2trt/quantization/calibrator_qdq.py:29-80):

      "float"     plain Linear (what a freshly built or un-calibrated module does);
      "calibrate" plain Linear while the calibrator collects the input under `self.site`;
      "int8"      after `freeze()`: the fp16 input goes straight into an int8 x int8 -> int32 GEMM on the matrix
                  cores that quantises it in its operand load, q = clamp(rne(x * fl(1 / s)), +-127)
                  (bevops_linear_int8_fused; FUSED_QUANT off: bevops_quantize_rows, q = clamp(rne(x / s)), then
                  bevops_linear_int8), de-quantising epilogue, output in the input's dtype.

    Deviations from the reference's set-up (det2trt/quantization/calibrator_qdq.py:8-26), both at the level of
    rounding ties / the calibrator's choice, neither of the data flow: pytorch_quantization rounds x * (127 / amax),
    here the scale is s = amax / 127 and the quantiser multiplies by fl(1 / s) (fused) or divides by s (pass) -- the
    three forms agree except on near-ties; and `freeze()` calibrates the WEIGHT with max calibration unless a
    calibrator class is passed (the reference's init_quant_desc gives weights the same calib_method as inputs, "max"
    or "histogram": pass `weight_calibrator=type(self.cal)` for the histogram set-up).

    `fake_quant_reference(x)` is the QuantLinear formula itself -- F.linear(dq(q(x)), dq(q(w))) + b --
    which the int8 path must reproduce up to the fp32 summation order."""

    def __init__(self, in_features, out_features, bias=True, calibrator=None, site=None):
        super().__init__(in_features, out_features, bias)
        self.cal = calibrator
        self.site = site or f"linear@{id(self):x}"
        self.mode = "float"
        self.scale_in = None
        self.register_buffer("weight_q", None, persistent=False)
        self.scale_w = None
        self.bias_f32 = None      # the epilogue's fp32 shift, made once at freeze()

    @classmethod
    def from_linear(cls, lin, calibrator, site):
        m = cls(lin.in_features, lin.out_features, lin.bias is not None, calibrator, site)
        m.weight, m.bias = lin.weight, lin.bias
        return m.to(lin.weight.device, lin.weight.dtype)

    def calibrate(self):
        self.mode = "calibrate"
        return self

    def freeze(self, weight_calibrator=None):
        """Fix the input scale from the collected statistics, quantise the weight (per tensor;
        max calibration unless a calibrator class is given, as QuantDescriptor's weight default)."""
        # (a site without statistics only ever sees empty batches: any positive scale will do)
        self.scale_in = float(self.cal.scale(self.site)) if self.cal.has(self.site) else 1.0
        wc = (weight_calibrator or MinMaxCalibrator)()
        wc.collect("w", self.weight.detach())
        self.scale_w = float(wc.scale("w"))
        self.weight_q = _Base.quantize(self.weight.detach(), self.scale_w).contiguous()
        self.bias_f32 = None if self.bias is None else self.bias.detach().float().contiguous()
        self.mode = "int8"
        return self

    def fake_quant_reference(self, x):
        xq = torch.clamp(torch.round(x.float() / self.scale_in), -127, 127) * self.scale_in
        wq = self.weight_q.float() * self.scale_w
        return torch.nn.functional.linear(xq, wq, None if self.bias is None else self.bias.float())

    def forward(self, x, residual=None, relu=False):
        if self.mode != "int8":
            if self.mode == "calibrate":
                self.cal.collect(self.site, x)
            y = super().forward(x)
            if residual is not None:
                y = y + residual
            return torch.relu(y) if relu else y
        from . import functions as _f
        xh = x if x.dtype == torch.float16 else x.to(torch.float16)
        res = None if residual is None else residual.to(torch.float16)
        if self.bias_f32 is None and self.bias is not None:
            self.bias_f32 = self.bias.detach().float().contiguous()
        # the fp16 activation goes straight into the GEMM (quantised in its operand load: no quantise pass)
        a = xh if FUSED_QUANT["enabled"] and xh.shape[-1] % 16 == 0 else _f.quantize_rows(xh, self.scale_in)
        y = _f.linear_int8(a, self.scale_in, self.weight_q, self.scale_w, self.bias_f32, res, relu)
        return y if y.dtype == x.dtype else y.to(x.dtype)


class Conv2dQ(torch.nn.Conv2d):
    """`Conv2dQ` (register.py:79: QuantConv2d) for the 1x1 convolutions of the channels-last backbone:
    a 1x1 convolution on NHWC activations is the LinearQ GEMM over the [N*H*W, C] rows.  Other kernel
    sizes stay on the library convolution in the model's dtype (TensorRT is free to do the same)."""

    def __init__(self, conv, calibrator, site):
        assert conv.kernel_size == (1, 1) and conv.groups == 1 and conv.padding == (0, 0)
        super().__init__(conv.in_channels, conv.out_channels, 1, conv.stride, bias=conv.bias is not None)
        self.weight, self.bias = conv.weight, conv.bias
        lin = torch.nn.Linear(conv.in_channels, conv.out_channels, conv.bias is not None)
        lin.weight = torch.nn.Parameter(conv.weight.detach().view(conv.out_channels, conv.in_channels))
        lin.bias = conv.bias
        self.lin = LinearQ.from_linear(lin, calibrator, site)

    def calibrate(self):
        self.lin.calibrate()
        return self

    def freeze(self):
        self.lin.freeze()
        return self

    def forward(self, x):
        s = self.stride[0]
        if s > 1:
            x = x[:, :, ::s, ::s]
        n, c, h, w = x.shape
        rows = x.permute(0, 2, 3, 1).reshape(-1, c)
        y = self.lin(rows.contiguous())
        return y.view(n, h, w, -1).permute(0, 3, 1, 2)


class ConvTapsQ(torch.nn.Conv2d):
    """`Conv2dQ` (register.py:79) for the plain 3x3 convolutions of the channels-last backbone / neck (and, when
    built with a stride, for strided 1x1 convolutions): the same three phases as LinearQ, the int8 phase running
    `bevops_conv_tile_int8_fused` -- fp16 activation quantised inside the kernel's operand load, int8 weights in
    the taps-major layout, de-quantising epilogue with shift / identity / ReLU.  The channels-last data path
    (`bevformer._conv_nhwc`) reads `qmode` / `int8_nhwc`; in the float and calibrate phases it runs this module's
    weights through its ordinary fp16 path (and `collect` sees the input)."""

    def __init__(self, conv, calibrator, site):
        k = conv.kernel_size[0]
        assert conv.kernel_size == (k, k) and k in (1, 3) and conv.groups == 1 and conv.padding == (k // 2, k // 2) \
            and conv.stride[0] == conv.stride[1] and conv.dilation == (1, 1)
        super().__init__(conv.in_channels, conv.out_channels, k, conv.stride, k // 2, bias=conv.bias is not None)
        self.weight, self.bias = conv.weight, conv.bias
        self.cal, self.site, self.qmode = calibrator, site, "float"
        self.scale_in = self.scale_w = None
        self.register_buffer("weight_q", None, persistent=False)
        self.bias_f32 = None

    def calibrate(self):
        self.qmode = "calibrate"
        return self

    def collect(self, x):
        self.cal.collect(self.site, x)

    def freeze(self, weight_calibrator=None):
        self.scale_in = float(self.cal.scale(self.site)) if self.cal.has(self.site) else 1.0
        wc = (weight_calibrator or MinMaxCalibrator)()
        wc.collect("w", self.weight.detach())
        self.scale_w = float(wc.scale("w"))
        self.weight_q = _Base.quantize(self.weight.detach().permute(0, 2, 3, 1), self.scale_w).contiguous()   # taps-major
        self.bias_f32 = None if self.bias is None else self.bias.detach().float().contiguous()
        self.qmode = "int8"
        return self

    def int8_nhwc(self, x, residual=None, relu=False):
        from . import functions as _f
        return _f.conv_int8_nhwc(x, self.scale_in, self.weight_q, self.scale_w, self.bias_f32, relu, residual,
                                 self.stride[0])

    def fake_quant_reference(self, x):
        xq = torch.clamp(torch.round(x.float() / self.scale_in), -127, 127) * self.scale_in
        wq = self.weight_q.permute(0, 3, 1, 2).float() * self.scale_w
        return torch.nn.functional.conv2d(xq, wq, None if self.bias is None else self.bias.float(), self.stride,
                                          self.padding)

    def forward(self, x):
        if self.qmode == "int8" and x.is_cuda and x.dtype == torch.float16:
            return self.int8_nhwc(x.contiguous(memory_format=torch.channels_last))
        if self.qmode == "calibrate":
            self.collect(x)
        return super().forward(x)


def quantize_dense_layers(model, calibrator, select=lambda name, mod: True):
    """Swap every nn.Linear of `model` accepted by `select(name, module)` (K % 16 == 0, N % 4 == 0)
    for a LinearQ sharing its parameters and calibrator.  Returns the list of swapped modules; call
    `.calibrate()` on them, run calibration frames, then `.freeze()`."""
    swapped = []
    for name, mod in list(model.named_modules()):
        for child_name, child in list(mod.named_children()):
            full = f"{name}.{child_name}" if name else child_name
            if type(child) is torch.nn.Linear and child.in_features % 16 == 0 and child.out_features % 4 == 0 \
                    and select(full, child):
                q = LinearQ.from_linear(child, calibrator, "linear:" + full)
                setattr(mod, child_name, q)
                swapped.append(q)
    return swapped


def quantize_backbone_convs(model, calibrator, select=lambda name, mod: True, conv3x3=False):
    """Swap the 1x1 convolutions of the backbone / neck (bottleneck conv1, conv3, downsample, FPN laterals;
    `Conv2dQ` of the reference, det2trt/models/utils/register.py:79, configs/bevformer/plugin/
    bevformer_base_trt_p2_q.py) for Conv2dQ sharing their parameters: in the channels-last data path they
    are the LinearQ GEMM over the [N*H*W, C] rows with shift / identity / ReLU in its epilogue (a stride goes into
    the int8 GEMM's row addressing).  `conv3x3=True` additionally turns the plain 3x3 convolutions with Cin % 64 == 0
    (bottleneck conv2 of the stages without DCN, FPN output convolutions) into ConvTapsQ (int8 implicit GEMM) -- off
    by default: the implicit GEMM re-reads and RE-QUANTISES an input pixel once per tap, and on one box the INT8
    frame was 6 % slower with them than with the fp16 convolution kernel (18.5 vs 17.5 ms).  The 7x7 stem and the
    DCNv2 pack's offset convolution stay in the model's dtype.  Returns the swapped modules (`.calibrate()`,
    calibration frames, `.freeze()`)."""
    swapped = []
    for name, mod in list(model.named_modules()):
        if not name.startswith(("backbone", "neck")):
            continue
        for child_name, child in list(mod.named_children()):
            full = f"{name}.{child_name}"
            if type(child) is not torch.nn.Conv2d or child.groups != 1 or not select(full, child):
                continue
            q = None
            if child.kernel_size == (1, 1) and child.in_channels % 16 == 0 and child.out_channels % 4 == 0:
                q = Conv2dQ(child, calibrator, "conv:" + full)
            elif conv3x3 and child.kernel_size == (3, 3) and child.padding == (1, 1) and child.dilation == (1, 1) \
                    and child.in_channels % 64 == 0 and child_name != "conv_offset":
                # (the DCNv2 pack's offset convolution feeds sampling positions and stays in fp16)
                q = ConvTapsQ(child, calibrator, "conv:" + full)
            if q is not None:
                q = q.to(child.weight.device, child.weight.dtype)
                setattr(mod, child_name, q)
                swapped.append(q)
    return swapped
