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
                          expanded back over P's non-empty bins; threshold = argmin KL(P || Q).  The
                          zero bin is given its neighbour's count first (the spike of exact zeros behind
                          a ReLU carries no information about the clip point).

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
    hist = hist.double().cpu().clone()
    # the bin at zero takes the count of its neighbour (pytorch_quantization's histogram calibrator does the same,
    # `bins[0] = bins[1]` in _compute_amax_entropy): the exact zeros a ReLU emits -- half of the tensor or more --
    # are represented exactly at ANY threshold, yet as one spike they dominate KL(P || Q) and drag the threshold
    # towards it (a ReLU of a unit Gaussian: T = 1.8 sigma, 17 % rms error; with a shifted ReLU 66 %)
    hist[0] = hist[1]
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
             "linear_bias_act", "dense_auto", "tsgemm_ln", "self_attention_qkv", "refine_reference_points", "decode_boxes", "conv3x3_auto", "conv_nhwc", "conv_int8_nhwc", "bias_relu_maxpool_nhwc", "stem_conv_pool",
             "image_normalize_pad", "upsample_add_nhwc_", "feat_embed_nhwc", "tsa_split", "queue_mean2")
    # `engine=True` (the build bench.py times, build_int8_engine below) additionally passes the entries whose fp16
    # form is FASTER than any int8 form on MI355X: the channels-last nearest-neighbour rotate of prev_bev (pure data
    # movement: rotating commutes with quantising, the int8 plugin would only add a quantise and a de-quantise pass
    # around it) and the value projection that writes the SCA sampler's planes itself.
    # ... and the camera projection of the BEV pillars (index generation: fp32 and bit-exact in every build).
    _ENGINE_PASS = ("rotate_hwc", "spatial_cross_attention_projected", "spatial_cross_attention_plan", "point_sampling")

    def __init__(self, calibrator="entropy", fp_ops=None, channels_last=False, fused_sca=False, engine=False):
        from . import functions as _f
        self.fp = fp_ops if fp_ops is not None else _f
        self.cal = get_calibrator(calibrator)() if isinstance(calibrator, str) else calibrator
        self.mode = "calibrate"
        self._scales = {}
        self._n = {}
        self._wq = _TensorCache()      # float weight tensor -> (int8 weight, scale)
        self._pass = self._PASS + (("spatial_cross_attention_sample",) if fused_sca else ()) if channels_last else ()
        if engine:
            self._pass = self._pass + self._ENGINE_PASS
        self._site_bs = {}             # msda site -> batch dimension of its call (2: TSA, cameras: SCA, 1: decoder)
        # optional predicate site -> bool: in the int8 phase a site it rejects keeps its fp operator (a layer-precision
        # fallback, as a TensorRT build allows per layer; tools/int8_attribution.py switches groups of sites with it)
        self.site_filter = None

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

    def site_batch(self, site):
        """Batch dimension the MSDA site was calibrated with (None: not an MSDA site / never seen)."""
        return self._site_bs.get(site)

    def _fp_site(self, site):
        return self.mode == "int8" and self.site_filter is not None and not self.site_filter(site)

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
            self._site_bs[site] = int(value.shape[0])
            out = self.fp.multi_scale_deformable_attn(value, shapes, ref, off, w)
            for k, t in (("value", value), ("offsets", off), ("weights", w), ("out", out)):
                self.cal.collect(f"{site}.{k}", t)
            return out
        if self._fp_site(site):
            return self.fp.multi_scale_deformable_attn(value, shapes, ref, off, w)
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
        if self._fp_site(site):
            return self.fp.rotate(img, angle, center, interpolation)
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
        if self._fp_site(site):
            return self.fp.modulated_deformable_conv2d(x, offset, mask, weight, bias, stride, padding, dilation,
                                                       groups, deform_groups)
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

    def bev_pool_v2(self, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths,
                    out_height=128, out_width=128):
        """BEVPoolV2TRT[2] (bevPoolKernel.cu:115-149 in its INT8 flavour): depth and features int8, int32 sums,
        one requantisation with the output's calibrated scale."""
        site = self._site("bev_pool")
        args = (ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths)
        if self.mode == "calibrate" or self._fp_site(site):
            out = self.fp.bev_pool_v2_2(depth, feat, *args, out_height, out_width)
            if self.mode == "calibrate":
                for k, t in (("depth", depth), ("feat", feat), ("out", out)):
                    self.cal.collect(f"{site}.{k}", t)
            return out
        (qd, sd), (qf, sf) = self._q(f"{site}.depth", depth), self._q(f"{site}.feat", feat)
        s_out = self._scales[f"{site}.out"]
        out = self.fp.bev_pool_v2_int8(qd.contiguous(), qf.contiguous(), *args, sd, sf, s_out, out_height, out_width)
        return self._dq(out, s_out, depth.dtype)

    bev_pool_v2_2 = bev_pool_v2


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
        # prequant: quantise the fp16 input ONCE (one pass over it) and run the int8 implicit GEMM on the int8 copy,
        # instead of quantising inside the operand load -- there every input pixel is re-read and RE-QUANTISED once
        # per tap (nine times for a 3x3), which is what made the fused form lose to the fp16 kernel in round 3
        self.prequant = False

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
        if self.prequant and residual is None:
            from .functions import int8_chain as C
            q = _f.quantize_rows(x.permute(0, 2, 3, 1), self.scale_in)      # [n, h, w, c]: the channels-last bytes as they lie
            return C.conv_int8_chain_nhwc(q.permute(0, 3, 1, 2), self.scale_in, self.weight_q, self.scale_w, self.bias_f32,
                                          relu, self.stride[0], torch.float16)
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


class Int8ChainBackbone:
    """The backbone + FPN laterals of the INT8 engine as an int8 ACTIVATION CHAIN (what TensorRT builds from the
    reference's `Conv2dQ` backbone, configs/bevformer/plugin/bevformer_base_trt_p2_q.py, when it keeps the tensors
    between the layers in int8): the stem's pooled output is quantised once; from there every 1x1 / 3x3 / DCNv2
    layer of every bottleneck reads int8 and writes int8 with its consumer's calibrated input scale -- the identity
    rows included -- until the FPN lateral convolutions de-quantise into the fp16 neck.  One byte per element in and
    out of every layer instead of two; no quantise / de-quantise passes.

    Build (like a TensorRT PTQ build, three steps):
        chain = Int8ChainBackbone(model, calibrator)   # swaps the 1x1 convolutions for Conv2dQ, arms the taps
        ... calibration frames through the fp16 model ...
        chain.freeze()                                 # scales + int8 weights; model.extract_feat now runs the chain

    Layer arithmetic: `functions/int8_chain.py` (int8 x int8 -> int32 on the matrix cores, requantising epilogues;
    the DCNv2 block is the INT8 plugin's arithmetic on channels-last int8 tensors)."""

    def __init__(self, model, calibrator, weight_calibrator=None):
        self.model, self.cal, self.ready = model, calibrator, False
        self.wcal = weight_calibrator or MinMaxCalibrator
        self.convs = quantize_backbone_convs(model, calibrator)
        for m in self.convs:
            m.calibrate()
        for si, stage in enumerate(model.backbone.stages):
            for bi, blk in enumerate(stage):
                blk.chain_cal, blk.chain_site = calibrator, f"chain:{si}.{bi}"
                if hasattr(blk.conv2, "conv_offset"):
                    blk.conv2.chain_cal, blk.conv2.chain_site = calibrator, f"chain:{si}.{bi}"
        model.int8_chain = self

    # ---- build
    def _wq(self, w):
        wc = self.wcal()
        wc.collect("w", w.detach())
        s = float(wc.scale("w"))
        return _Base.quantize(w.detach(), s), s

    def _scale(self, name):
        if not self.cal.has(name):
            if not any(k.startswith("chain:") for k in self.cal._stats):
                return 1.0          # a rank without cameras fed the backbone empty batches only: placeholder, never used
            raise RuntimeError(f"int8 chain: no calibration statistics for '{name}' -- calibrate through the "
                               "channels-last chain path (the NCHW path does not visit the chain's sites)")
        return float(self.cal.scale(name))

    def freeze(self):
        m = self.model
        for c in self.convs:
            c.freeze()
        for stage in m.backbone.stages:
            for blk in stage:
                blk.chain_cal = None
                if hasattr(blk.conv2, "conv_offset"):
                    blk.conv2.chain_cal = None
        dev = m.backbone.stem.weight.device
        plan = []
        stages = m.backbone.stages
        lat_of = {}                     # stage index -> lateral convolution reading its output
        for l, si in zip(m.neck.lateral, m.backbone.out_indices):
            lat_of[si] = l
        for si, stage in enumerate(stages):
            blocks = []
            for bi, blk in enumerate(stage):
                site = f"chain:{si}.{bi}"
                e = {"s_x": blk.conv1.lin.scale_in, "s_1": self._scale(site + ".t1"), "s_2": blk.conv3.lin.scale_in,
                     "c1": blk.conv1, "c3": blk.conv3, "down": blk.downsample, "s_idt": None}
                if blk.downsample is not None:
                    e["s_idt"] = self._scale(site + ".idt")
                c2 = blk.conv2
                if hasattr(c2, "conv_offset"):       # DCNv2 pack
                    # the chain's offset convolution is the 3x3, one-deform-group form: 18 offsets + 9 mask logits,
                    # padded to 32 output channels (the channels-last INT8 entry reads exactly that layout)
                    if c2.conv_offset.out_channels != 27 or getattr(c2, "deform_groups", 1) != 1:
                        raise NotImplementedError("int8 chain: DCNv2 packs other than 3x3 / deform_groups=1")
                    wq, sw = self._wq(c2.weight)
                    ow = torch.zeros((32,) + tuple(c2.conv_offset.weight.shape[1:]), dtype=torch.float32, device=dev)
                    ow[:27] = c2.conv_offset.weight.detach().float()
                    amax = ow.abs().flatten(1).amax(1).clamp(min=1e-12)        # per output channel (zero rows: any scale)
                    osc = (amax / 127.0)
                    oq = torch.clamp(torch.round(ow / osc.view(-1, 1, 1, 1)), -127, 127).to(torch.int8)
                    ob = torch.zeros(32, dtype=torch.float32, device=dev)
                    ob[:27] = c2.conv_offset.bias.detach().float()
                    e["dcn"] = dict(wq=wq.contiguous(), sw=sw, bias=c2.bias.detach().float().contiguous(),
                                    oq=oq.permute(0, 2, 3, 1).contiguous(), osc=osc.contiguous(), ob=ob,
                                    s_off=self._scale(site + ".offset"), s_mask=self._scale(site + ".mask"),
                                    stride=c2.stride)
                else:
                    wq, sw = self._wq(c2.weight)
                    e["conv"] = dict(wq=wq.permute(0, 2, 3, 1).contiguous(), sw=sw,
                                     bias=c2.bias.detach().float().contiguous(), stride=c2.stride[0])
                blocks.append(e)
            plan.append(blocks)
        # output scale of a block = input scale of what reads it: the next block, the next stage, or the lateral
        for si, blocks in enumerate(plan):
            for bi, e in enumerate(blocks):
                if bi + 1 < len(blocks):
                    e["s_out"] = blocks[bi + 1]["s_x"]
                elif si + 1 < len(plan):
                    e["s_out"] = plan[si + 1][0]["s_x"]
                else:
                    e["s_out"] = lat_of[si].lin.scale_in
        self.plan, self.lat_of = plan, lat_of
        self.s_stem = plan[0][0]["s_x"]
        self.ready = True
        return self

    # ---- run
    @staticmethod
    def _conv1x1(x, s_x, conv, relu, out_dtype, s_out, residual=None, s_res=1.0):
        """x int8 channels-last -> conv (a Conv2dQ, frozen) as the int8 GEMM over its rows; a stride goes into the
        implicit GEMM's row addressing."""
        from .functions import int8_chain as C
        lin = conv.lin
        if conv.stride[0] > 1:
            assert residual is None
            w = lin.weight_q.view(conv.out_channels, 1, 1, conv.in_channels)
            return C.conv_int8_chain_nhwc(x, s_x, w, lin.scale_w, lin.bias_f32, relu, conv.stride[0], out_dtype, s_out)
        n, c, h, w_ = x.shape
        rows = x.permute(0, 2, 3, 1).reshape(-1, c)
        res = None if residual is None else residual.permute(0, 2, 3, 1).reshape(-1, residual.shape[1])
        y = C.linear_int8_chain(rows, s_x, lin.weight_q, lin.scale_w, lin.bias_f32, res, s_res, relu, out_dtype, s_out)
        return y.view(n, h, w_, -1).permute(0, 3, 1, 2)

    @torch.no_grad()
    def __call__(self, img):
        """img [cams, 3, H, W] fp16 -> the FPN's outputs (fp16, channels-last), as model.neck.forward_nhwc gives."""
        import torch.nn.functional as F
        from .functions import int8_chain as C
        m = self.model
        ops = m.ops
        bb = m.backbone
        from .functions.conv import STEM_FUSED, stem_conv_pool
        if (STEM_FUSED["enabled"] and img.is_contiguous() and img.shape[-1] % 2 == 0
                and tuple(bb.stem.weight.shape) == (64, 3, 7, 7) and bb.stem.stride == (2, 2) and bb.stem.padding == (3, 3)):
            x = stem_conv_pool(img, bb.stem.weight, bb.stem.bias, self.s_stem)       # 7x7 stem + pool: one kernel, int8 out
        else:
            x = img.contiguous(memory_format=torch.channels_last)
            y = F.conv2d(x, bb.stem.weight, None, bb.stem.stride, bb.stem.padding)   # 7x7 stem: library convolution
            if not y.is_contiguous(memory_format=torch.channels_last):
                y = y.contiguous(memory_format=torch.channels_last)
            x = C.bias_relu_maxpool_nhwc_int8(y, bb.stem.bias, self.s_stem)
        i8 = torch.int8
        lats = []
        for si, blocks in enumerate(self.plan):
            for e in blocks:
                if e["down"] is not None:
                    idt, s_idt = self._conv1x1(x, e["s_x"], e["down"], False, i8, e["s_idt"]), e["s_idt"]
                else:
                    idt, s_idt = x, e["s_x"]
                t = self._conv1x1(x, e["s_x"], e["c1"], True, i8, e["s_1"])
                if "dcn" in e:
                    d = e["dcn"]
                    om = C.conv_int8_chain_nhwc(t, e["s_1"], d["oq"], d["osc"], d["ob"], False, d["stride"],
                                                torch.float16)
                    t = C.modulated_deformable_conv2d_int8_nhwc(t, e["s_1"], om, d["s_off"], d["s_mask"], d["wq"],
                                                                d["sw"], d["bias"], e["s_2"], True, d["stride"], 1, 1,
                                                                1, 1)
                else:
                    c = e["conv"]
                    t = C.conv_int8_chain_nhwc(t, e["s_1"], c["wq"], c["sw"], c["bias"], True, c["stride"], i8,
                                               e["s_2"])
                x = self._conv1x1(t, e["s_2"], e["c3"], True, i8, e["s_out"], idt, s_idt)
            if si in self.lat_of:
                lats.append(self._conv1x1(x, blocks[-1]["s_out"], self.lat_of[si], False, torch.float16, 1.0))
        return m.neck.topdown_nhwc(lats, ops)


def engine_dense_select(name, mod):
    """Which nn.Linear layers of the re-hosted BEVFormer the INT8 engine runs as LinearQ: the encoder's dense layers
    except (a) TSA's sampling_offsets / attention_weights, whose fp16 form is the split projection (two K = 256 GEMMs
    on the un-concatenated operands, 40 us) where the int8 form needs the [nq, 512] concatenation copy first (50 us
    + 2 x 17 us), and (b) SCA's value_proj, which in fp16 writes the sampler's planes itself.  The decoder (900
    queries: every layer is launch-bound, INT8 buys no time) stays fp16 -- the layer-precision choice a TensorRT
    builder makes by timing; tools/int8_attribution.py has the error side of it."""
    if not name.startswith("encoder."):
        return False
    return not name.endswith(("tsa.sampling_offsets", "tsa.attention_weights", "sca.value_proj"))


def build_int8_engine(B, name, dev, frames, calibrator="entropy", chain=True, dense_select=None, decoder_int8=False,
                      sca_int8=False):
    """The PTQ build of the re-hosted model that bench.py times (`B` = the bevformer module, `frames` = an iterable
    of (image, can_bus, lidar2img) calibration frames of one scene): int8 activation chain through the backbone
    (Int8ChainBackbone; chain=False: Conv2dQ layers with fp16 tensors between them, the round-3 build), encoder
    dense layers as LinearQ, TSA's MSDA on the INT8 plugin, everything else on the fp16 operators.
    sca_int8=True: the SCA site too runs the INT8 plugin (quantise value / offsets / weights, bevops_msda_forward in
    int8, masked camera sum on the de-quantised result) as the reference's INT8 engines do
    (configs/bevformer/plugin/bevformer_base_trt_p2_q.py) -- instead of the fp16 projected sampler, which is faster
    than every int8 form of that site on MI355X; bench.py reports both builds.
    Returns (model, qops, note)."""
    dtype = torch.float16
    qops = Int8PluginOps(calibrator, channels_last=True, fused_sca=not sca_int8, engine=True)
    if sca_int8:
        qops._pass = tuple(n for n in qops._pass if not n.startswith("spatial_cross_attention"))
    model = B.BEVFormer(name, ops=qops, seed=0, backbone_layout="nhwc").to(dev, dtype)
    qops.attach(model)
    sel = dense_select or ((lambda n, m: n.startswith(("encoder.", "decoder."))) if decoder_int8 else engine_dense_select)
    q = quantize_dense_layers(model, qops.cal, sel)
    ch = None
    if chain:
        ch = Int8ChainBackbone(model, qops.cal)
        if os.environ.get("BEVOPS_INT8_FPN", "1") != "0":
            # the FPN's 3x3 output convolutions as int8 implicit GEMMs on a pre-quantised copy of their fp16 input
            # (Conv2dQ of the reference's INT8 config; 2x the matrix rate of the fp16 kernel on the frame's largest
            # convolution, 6 x 256 x 116 x 200)
            taps = [m for m in quantize_backbone_convs(model, qops.cal, lambda n, m: n.startswith("neck."), conv3x3=True)
                    if isinstance(m, ConvTapsQ)]
            for m in taps:
                m.prequant = True
            q += taps
    else:
        q += quantize_backbone_convs(model, qops.cal)
    for m in q:
        m.calibrate()
    r = B.FrameRunner(model, dev, dtype)
    n = 0
    for img, can, l2i in frames:
        r.step(img, can, l2i, "calib")
        n += 1
    scales = qops.freeze()
    for m in q:
        m.freeze()
    if ch is not None:
        ch.freeze()
    if not decoder_int8:
        qops.site_filter = lambda site: qops.site_batch(site) != 1      # the decoder's MSDA (batch 1) stays fp16
    note = {"int8_plugin_sites": sum(1 for k in scales if k.endswith(".out") and k.startswith("msda")
                                     and (decoder_int8 or qops.site_batch(k[:-4]) != 1)),
            "int8_dense_layers": len(q), "int8_backbone_layers": (len(ch.convs) + sum(len(b) for b in ch.plan)) if ch else 0,
            "activation_chain": bool(ch), "calibration_frames": n, "sca_site": "int8 plugin" if sca_int8 else "fp16 projected sampler"}
    return model, qops, note


def build_int8_bevdet(D, dev, frames, calibrator="entropy"):
    """The PTQ build of the re-hosted BEVDet-R50 (BASELINE config 5; `D` = the bevdet module, `frames` = an iterable
    of (image, ranks...) calibration inputs): ResNet-50 + FPN laterals as the int8 activation chain, bev_pool_v2 on
    its INT8 plugin flavour; the small BEV encoder / head convolutions stay fp16.  Returns (model, qops, note)."""
    qops = Int8PluginOps(calibrator, channels_last=True)
    model = D.BEVDet(ops=qops, seed=0).to(dev, torch.float16)
    model.view.ops = qops
    ch = Int8ChainBackbone(model, qops.cal)
    n = 0
    for f in frames:
        qops.begin_frame()
        model(*f)
        n += 1
    scales = qops.freeze()
    ch.freeze()
    model.register_forward_pre_hook(lambda m, args: qops.begin_frame())
    note = {"int8_plugin_sites": sum(1 for k in scales if k.startswith("bev_pool") and k.endswith(".out")),
            "int8_backbone_layers": len(ch.convs) + sum(len(b) for b in ch.plan), "activation_chain": True,
            "calibration_frames": n}
    return model, qops, note
