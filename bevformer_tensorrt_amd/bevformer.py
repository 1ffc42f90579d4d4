"""mmcv-free re-host of the reference's BEVFormer `*TRTP` inference wrappers (SURVEY.md
section 8f-1), so that the sampling hot path can be measured inside the model it serves.

Call-compatible with `BEVFormerTRT.forward_trt(image, prev_bev, use_prev_bev, can_bus,
lidar2img)` (det2trt/models/detector/bevformer.py:37-44).  Structure and tensor shapes follow

  detector   det2trt/models/detector/bevformer.py:12-44         (ResNet + FPN, forward_trt)
  head       det2trt/models/dense_heads/bevformer_head.py:211-282
  transformer det2trt/models/modules/transformer.py:245-398     (shift, rotate prev_bev, can_bus
                                                                   MLP, level/camera embeds)
  encoder    det2trt/models/modules/encoder.py:261-334,510-636  (TSA -> norm -> SCA -> norm -> FFN -> norm)
  TSA        det2trt/models/modules/temporal_self_attention.py:350-457
  SCA        det2trt/models/modules/spatial_cross_attention.py:200-273,694-768
  decoder    det2trt/models/modules/decoder.py:52-112,381-471   (MHA, CustomMSDeformableAttention, FFN,
                                                                   reference-point refinement)
  DCNv2 pack det2trt/models/modules/cnn/dcn.py:31-86
  configs    configs/bevformer/bevformer_{tiny,small,base}.py

The sampling operators come from `ops` (default: this package's HIP operators); the dense
layers (convolutions, Linear, LayerNorm, MultiheadAttention) are torch modules, i.e. MIOpen /
hipBLASLt on ROCm.  Weights are random (seeded): no checkpoints exist in this environment,
the purpose is the device-side dataflow and its speed, not detection quality.

Differences from the reference that do not change results:
  * `query.repeat(num_cams, 1, 1)` followed by per-camera Linear layers
    (spatial_cross_attention.py:254, :754-755) is computed once and passed as a stride-0
    expanded view: the MSDA operator reads the single copy (`shared_offsets`);
  * `prev_bev` stays on the device between frames (the reference round-trips it through host
    numpy, tools/bevformer/evaluate_trt.py:126,144);
  * frozen BatchNorm is folded into the preceding convolution.
"""
import math
import os
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functions as _hip_ops
from . import geometry as G
from .utils import lib as _lib

CONFIGS = {
    # name: backbone depth, DCN stages, FPN inputs/outs, padded image (h, w), BEV size, encoder layers
    # style: which convolution of a bottleneck carries the stride -- "pytorch": the 3x3, "caffe": the first
    # 1x1 (configs/bevformer/bevformer_tiny.py:62, bevformer_small.py:58, bevformer_base.py:50)
    "tiny": dict(depth=50, dcn=(False, False, False, False), fpn_in=[2048], out_indices=(3,), levels=1,
                 image=(480, 800), bev=(50, 50), enc_layers=3, style="pytorch"),
    "small": dict(depth=101, dcn=(False, False, True, True), fpn_in=[2048], out_indices=(3,), levels=1,
                  image=(736, 1280), bev=(150, 150), enc_layers=3, style="caffe"),
    "base": dict(depth=101, dcn=(False, False, True, True), fpn_in=[512, 1024, 2048], out_indices=(1, 2, 3),
                 levels=4, image=(928, 1600), bev=(200, 200), enc_layers=6, style="caffe"),
}
PC_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
EMBED, HEADS, NUM_QUERY, NUM_CAMS = 256, 8, 900, 6


inverse_sigmoid = G.inverse_sigmoid   # the head's (mmdet) form; the decoder's is in G.refine_reference_points


# --------------------------------------------------------------------------- backbone
# Two data paths through the same modules and weights:
#  * forward():      reference layout (NCHW), library convolutions + framework element-wise ops;
#  * forward_nhwc(): channels-last activations end to end (MI355X path, fp16): 1x1 convolutions are
#    row-major GEMMs with the folded-BN shift / ReLU in the hipBLASLt epilogue, 3x3 convolutions
#    run MIOpen's NHWC kernels without layout transposes, every remaining shift / residual / ReLU
#    chain is ONE bevops_bias_act_nhwc pass, DCNv2 reads and writes NHWC directly, and the FPN
#    outputs already are the [cams, keys, 256] value layout of the encoder.
_CAM_IDX = {}


def _take_cams(t, cams):
    """t[cams] along dim 0 for a Python list of camera ids, through a cached DEVICE index tensor: indexing with the
    list itself uploads a fresh index tensor every call, which a stream capture does not permit (the camera-sharded
    frame is captured into a HIP graph)."""
    cams = tuple(cams)
    if cams == tuple(range(t.shape[0])):
        return t
    key = (cams, str(t.device))
    idx = _CAM_IDX.get(key)
    if idx is None:
        idx = _CAM_IDX[key] = torch.tensor(cams, dtype=torch.long, device=t.device)
    return t.index_select(0, idx)


def _rows(x):
    """[N, C, H, W] channels-last -> [N*H*W, C] view."""
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])


def _from_rows(y, n, h, w):
    return y.view(n, h, w, y.shape[-1]).permute(0, 3, 1, 2)   # (explicit: n may be 0 on a rank without cameras)


_FUSED_LINEAR = {"enabled": os.environ.get("BEVOPS_FUSED_LINEAR", "1") != "0"}   # A/B switch
_R3 = {"enabled": os.environ.get("BEVOPS_R3_FUSIONS", "1") != "0"}   # A/B switch of the round-3 launch-count work
# TSA's MSDA on the layout-preserving quad kernel (its reference points are the BEV grid: neighbouring queries sample
# neighbouring pixels, the head-major re-layout the default dispatch makes for random points is not needed).  Round 4
# measured no gain inside the frame on one run each; round 5 repeated it interleaved under graph replay: -0.06 ..
# -0.12 ms per base frame in three pairs (profiles/r05/model_bench_tsa_local.jsonl) -> default; BEVOPS_TSA_LOCAL=0: off
_TSA_LOCAL = {"enabled": os.environ.get("BEVOPS_TSA_LOCAL", "1") == "1"}
_TSA_GLUE = {"enabled": os.environ.get("BEVOPS_TSA_GLUE", "1") == "1"}   # A/B: bevops_tsa_split / bevops_queue_mean2 vs the framework copies


def _fused_linear(ops, x, weight, bias, residual, relu):
    """out = act(x @ weight.T + bias + residual) in ONE GEMM (ops.linear_bias_act: hipBLASLt with
    the shift + identity + ReLU epilogue) -- or None when the operator set has no such entry, the
    tensors are not fp16, or the library has no algorithm for the shape (then the caller's
    two-launch path runs)."""
    fn = getattr(ops, "linear_bias_act", None)
    if fn is None or not _FUSED_LINEAR["enabled"] or x.dtype != torch.float16 or not x.is_cuda:
        return None
    auto = getattr(ops, "dense_auto", None)
    if auto is not None and _R3["enabled"]:
        # measured per problem: tsgemm / tile_gemm (hand-written MFMA GEMMs) / hipBLASLt / the framework's addmm
        try:
            return auto(x, weight, bias, residual, relu)
        except _lib.BevopsError as exc:
            if exc.status != _lib.NOT_SUPPORTED:
                raise
            return None
    try:
        return fn(x, weight, bias, residual, relu)
    except _lib.BevopsError as exc:   # only NOT_SUPPORTED (no algorithm for this shape) falls back
        if exc.status != _lib.NOT_SUPPORTED:
            raise
        return None


def _gemm_entry(ops):
    """fn(x, weight, bias, residual, relu) -> act(x @ weight.T + bias + residual): the measured dispatch when the
    operator set has it, else the hipBLASLt entry (or None)."""
    return getattr(ops, "dense_auto", None) or getattr(ops, "linear_bias_act", None)


def _dense(ops, lin, x, residual=None, relu=False):
    """act(lin(x) + residual) for an nn.Linear or a quantization.LinearQ.  A LinearQ is always CALLED
    (all three of its phases live in its forward: float, calibrate -- where it must see its input --
    and int8); a plain Linear takes the one-GEMM fused epilogue when the operator set has it."""
    if hasattr(lin, "fake_quant_reference"):        # LinearQ (kept duck-typed: quantization imports nothing from here)
        return lin(x, residual, relu)
    y = _fused_linear(ops, x, lin.weight, lin.bias, residual, relu)
    if y is not None:
        return y
    y = lin(x)
    if residual is not None:
        y = y + residual
    return F.relu(y, inplace=True) if relu else y


def _mlp(ops, seq, x):
    """nn.Sequential of Linear / ReLU modules with each ReLU in the epilogue of the GEMM in front of it."""
    mods = list(seq)
    if not _R3["enabled"] or not all(isinstance(m, (nn.Linear, nn.ReLU)) for m in mods):
        return seq(x)
    i = 0
    while i < len(mods):
        if isinstance(mods[i], nn.ReLU):
            x = F.relu(x)
            i += 1
            continue
        relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
        x = _dense(ops, mods[i], x, None, relu)
        i += 2 if relu else 1
    return x


def _layer_norm(ops, norm, x):
    """nn.LayerNorm as one streaming pass (ops.layer_norm) when the operator set has it."""
    fn = getattr(ops, "layer_norm", None)
    if fn is None or not _FUSED_LINEAR["enabled"] or x.dtype != torch.float16 or not x.is_cuda or \
            x.shape[-1] not in (64, 128, 256, 512):
        return norm(x)
    return fn(x, norm.weight, norm.bias, norm.eps)


_FUSED_REFINE = {"enabled": os.environ.get("BEVOPS_FUSED_REFINE", "1") == "1"}   # A/B: decoder refinement as one launch
_OWN_ATTN = {"enabled": os.environ.get("BEVOPS_OWN_ATTN", "1") == "1"}   # A/B: decoder self-attention on csrc/attention.hip
_LN_FUSED = {"enabled": os.environ.get("BEVOPS_LN_FUSED", "1") == "1"}   # A/B: LayerNorm in the epilogue of the GEMM in front of it
# The dense layers behind the backbone (the GEMMs that wrap the samplers, SURVEY.md 8a-5, the decoder, the heads) on the
# hand-written kernels whatever the dispatch table measured: one kernel per layer with a block-index-only summation order.
_OWN_ENCODER = {"enabled": os.environ.get("BEVOPS_OWN_ENCODER", "1") == "1"}
# camera-sharded runners: backbone on the measured dispatch (needs _OWN_ENCODER for the replicated layers)
_SHARDED_TABLE_BACKBONE = {"enabled": _OWN_ENCODER["enabled"] and os.environ.get("BEVOPS_SHARDED_RULE_DISPATCH", "0") != "1"}


def _dense_norm(ops, lin, x, residual, norm):
    """norm(lin(x) + residual): the dense layer that ends an attention / FFN block and the block's LayerNorm
    (modules/encoder.py:586-636) as ONE launch when the operator set has it (ops.tsgemm_ln: the persistent MFMA GEMM
    with the normalisation in its epilogue; 256 output columns, fp16 on the GPU, a plain Linear -- a LinearQ of the INT8
    build keeps its own path), else the GEMM followed by the one-pass norm."""
    fn = getattr(ops, "tsgemm_ln", None)
    if fn is not None and _LN_FUSED["enabled"] and _R3["enabled"] and _FUSED_LINEAR["enabled"] and type(lin) is nn.Linear \
            and x.dtype == torch.float16 and x.is_cuda and lin.out_features == EMBED and lin.in_features % 64 == 0 \
            and isinstance(norm, nn.LayerNorm) and norm.elementwise_affine and norm.normalized_shape == (EMBED,) \
            and x.numel() // x.shape[-1] >= 32:
        try:
            return fn(x, lin.weight, lin.bias, residual, norm.weight, norm.bias, norm.eps)
        except _lib.BevopsError as exc:
            if exc.status != _lib.NOT_SUPPORTED:
                raise
    return _layer_norm(ops, norm, _dense(ops, lin, x, residual, False))


def _conv1x1_nhwc(ops, x, conv, relu, residual=None):
    s = conv.stride[0]
    if s > 1:
        strided = getattr(ops, "conv_nhwc", None)
        if strided is not None and _R3["enabled"] and _FUSED_LINEAR["enabled"] and x.dtype == torch.float16 and x.is_cuda \
                and not hasattr(conv, "lin") and conv.in_channels % 32 == 0 and x.is_contiguous(memory_format=torch.channels_last):
            # the stride goes into the GEMM's row addressing: no sub-sampled copy of the activation
            return strided(x, conv.weight, conv.bias, relu, residual, s)
        lin = getattr(conv, "lin", None)
        int8_conv = getattr(ops, "conv_int8_nhwc", None)
        if lin is not None and getattr(lin, "mode", None) == "int8" and int8_conv is not None and _R3["enabled"] \
                and x.dtype == torch.float16 and x.is_cuda and conv.in_channels % 64 == 0 \
                and x.is_contiguous(memory_format=torch.channels_last):
            # Conv2dQ with a stride: the int8 GEMM sub-samples in its row addressing (no strided copy)
            return int8_conv(x, lin.scale_in, lin.weight_q.view(conv.out_channels, 1, 1, conv.in_channels), lin.scale_w,
                             lin.bias_f32, relu, residual, s)
        x = x[:, :, ::s, ::s].contiguous(memory_format=torch.channels_last)
    n, c, h, w = x.shape
    if hasattr(conv, "lin"):     # quantization.Conv2dQ: the LinearQ over the [N*H*W, C] rows (all three phases)
        y = conv.lin(_rows(x), None if residual is None else _rows(residual), relu)
        return _from_rows(y, n, h, w)
    wt = conv.weight.view(conv.out_channels, c).t()
    if residual is not None:
        # one GEMM with shift + identity + ReLU in its epilogue; else GEMM, then one fused pass (addmm
        # with the identity as its beta term would first memcpy it into the output: +21 us at layer 3)
        y = _fused_linear(ops, _rows(x), conv.weight.view(conv.out_channels, c), conv.bias, _rows(residual), relu)
        if y is None:
            y = torch.mm(_rows(x), wt)
            ops.bias_act_nhwc_(y, conv.bias, _rows(residual), relu)
    else:
        y = _fused_linear(ops, _rows(x), conv.weight.view(conv.out_channels, c), conv.bias, None, relu) \
            if _R3["enabled"] else None
        if y is None:
            y = torch._addmm_activation(conv.bias, _rows(x), wt) if relu else torch.addmm(conv.bias, _rows(x), wt)
    return _from_rows(y, n, h, w)


def _conv_nhwc(ops, x, conv, relu, residual=None):
    qmode = getattr(conv, "qmode", None)     # quantization.ConvTapsQ (kept duck-typed, as LinearQ above)
    if qmode == "calibrate":
        conv.collect(x)
    elif qmode == "int8" and x.is_contiguous(memory_format=torch.channels_last):
        return conv.int8_nhwc(x, residual, relu)
    auto = getattr(ops, "conv3x3_auto", None)
    if auto is not None and _R3["enabled"] and _FUSED_LINEAR["enabled"] and x.dtype == torch.float16 and x.is_cuda \
            and conv.kernel_size == (3, 3) and conv.stride[0] == conv.stride[1] and conv.padding == (1, 1) \
            and conv.groups == 1 and conv.dilation == (1, 1) and conv.in_channels % 32 == 0 \
            and x.is_contiguous(memory_format=torch.channels_last):
        # implicit GEMM on the tiled MFMA skeleton with the shift / ReLU in its epilogue, or the library
        # convolution + one epilogue pass: measured once per problem
        return auto(x, conv.weight, conv.bias, relu, residual, conv.stride[0])
    y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding)
    if not y.is_contiguous(memory_format=torch.channels_last):
        y = y.contiguous(memory_format=torch.channels_last)
    return ops.bias_act_nhwc_(y, conv.bias, residual, relu)


class DCNv2Pack(nn.Module):
    """ModulatedDeformConv2dPackPlugin (cnn/dcn.py:31-86): conv_offset -> (o1, o2, mask) ->
    modulated_deformable_conv2d."""

    def __init__(self, cin, cout, ops, stride=1):
        super().__init__()
        self.ops, self.stride = ops, stride
        self._c32_ok = True
        self.weight = nn.Parameter(torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9))
        self.bias = nn.Parameter(torch.zeros(cout))  # carries the folded BN shift
        self.conv_offset = nn.Conv2d(cin, 27, 3, stride, 1)
        nn.init.normal_(self.conv_offset.weight, std=0.01)   # non-zero so the sampler really deforms
        nn.init.zeros_(self.conv_offset.bias)

    def forward(self, x):
        out = self.conv_offset(x)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        return self.ops.modulated_deformable_conv2d(x, offset, torch.sigmoid(mask), self.weight,
                                                    self.bias, self.stride, 1, 1, 1, 1)

    def _chain_collect(self, om):
        """Calibration of the INT8 engine's activation chain (quantization.Int8ChainBackbone): the offsets and the
        mask (sigmoid of the logits) this pack hands its DCNv2 operator, as the INT8 plugin's two int8 inputs."""
        cal = getattr(self, "chain_cal", None)
        if cal is not None:
            cal.collect(self.chain_site + ".offset", om[:, :18])
            cal.collect(self.chain_site + ".mask", torch.sigmoid(om[:, 18:27]))

    def forward_nhwc(self, x, relu):
        # MIOpen's NHWC kernels are 3-8x slower at 27 output channels than at 32 (170 vs 52 us at
        # stage 3): run the offset conv with the filters zero-padded to 32 and drop the extra planes
        fn = getattr(self.ops, "conv_offset_nhwc", None)
        if fn is not None and self.stride == 1 and _FUSED_LINEAR["enabled"] and self._c32_ok:
            # own implicit-GEMM kernel: all 9 x Cin x 32 weights LDS-resident, bias in the epilogue
            try:
                out = fn(x, self.conv_offset.weight, self.conv_offset.bias)
                self._chain_collect(out)
                return self.ops.modulated_deformable_conv2d_nhwc(x, None, None, self.weight, self.bias, self.stride,
                                                                 1, 1, 1, 1, relu=relu, offset_mask_nhwc=out)
            except _lib.BevopsError as exc:
                if exc.status != _lib.NOT_SUPPORTED:
                    raise
                self._c32_ok = False   # unsupported channel count: library convolution below
        w = self.conv_offset.weight
        if getattr(self, "_w32", None) is None or self._w32[2] != (w._version, w.dtype, w.device):
            w32 = torch.zeros(32, *w.shape[1:], dtype=w.dtype, device=w.device)
            w32[:27] = w.detach()
            b32 = torch.zeros(32, dtype=w.dtype, device=w.device)
            b32[:27] = self.conv_offset.bias.detach()
            self._w32 = (w32.contiguous(memory_format=torch.channels_last), b32, (w._version, w.dtype, w.device))
        out = F.conv2d(x, self._w32[0], None, self.stride, 1)
        if not out.is_contiguous(memory_format=torch.channels_last):
            out = out.contiguous(memory_format=torch.channels_last)
        out = self.ops.bias_act_nhwc_(out, self._w32[1], None, False)
        self._chain_collect(out)
        # the raw [B, H, W, 32] tensor goes to the operator: (o1, o2) are its first 18 channels in the
        # order cat((o1, o2)) gives, the mask logits the next 9; sigmoid fused
        return self.ops.modulated_deformable_conv2d_nhwc(x, None, None, self.weight, self.bias, self.stride, 1, 1,
                                                         1, 1, relu=relu, offset_mask_nhwc=out)


class Bottleneck(nn.Module):
    """ResNet bottleneck (det2trt/models/backbones/resnet.py:106-260), BN folded.  style "caffe": the
    stride sits on the first 1x1, "pytorch": on the 3x3 (resnet.py:163-170)."""

    def __init__(self, cin, planes, stride, dcn, ops, downsample, style="caffe"):
        super().__init__()
        s1, s2 = (stride, 1) if style == "caffe" else (1, stride)
        self.conv1 = nn.Conv2d(cin, planes, 1, s1)
        self.conv2 = DCNv2Pack(planes, planes, ops, s2) if dcn else nn.Conv2d(planes, planes, 3, s2, 1)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1)
        self.downsample = nn.Conv2d(cin, planes * 4, 1, stride) if downsample else None

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.conv1(x), inplace=True)
        out = F.relu(self.conv2(out), inplace=True)
        return F.relu(self.conv3(out) + idt, inplace=True)

    def forward_nhwc(self, x, ops):
        idt = x if self.downsample is None else _conv1x1_nhwc(ops, x, self.downsample, False)
        out = _conv1x1_nhwc(ops, x, self.conv1, True)
        cal = getattr(self, "chain_cal", None)     # calibration of the INT8 engine's activation chain
        if cal is not None:
            cal.collect(self.chain_site + ".t1", out)
            if self.downsample is not None:
                cal.collect(self.chain_site + ".idt", idt)
        if isinstance(self.conv2, DCNv2Pack):
            out = self.conv2.forward_nhwc(out, True)
        else:
            out = _conv_nhwc(ops, out, self.conv2, True)
        return _conv1x1_nhwc(ops, out, self.conv3, True, residual=idt)


class ResNet(nn.Module):
    def __init__(self, depth, dcn, out_indices, ops, style="caffe"):
        super().__init__()
        blocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[depth]
        self.out_indices = out_indices
        self.stem = nn.Conv2d(3, 64, 7, 2, 3)
        cin, stages = 64, []
        for i, n in enumerate(blocks):
            planes, layers = 64 * 2 ** i, []
            for j in range(n):
                layers.append(Bottleneck(cin, planes, (2 if i > 0 else 1) if j == 0 else 1, dcn[i], ops, j == 0, style))
                cin = planes * 4
            stages.append(nn.Sequential(*layers))
        self.stages = nn.ModuleList(stages)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.stem(x), inplace=True), 3, 2, 1)
        outs = []
        for i, st in enumerate(self.stages):
            x = st(x)
            if i in self.out_indices:
                outs.append(x)
        return outs

    def _stem_fused(self, x, ops):
        """The one-kernel stem (functions/conv.py: stem_conv_pool) when the operator set has it and the input is in
        its domain: planar fp16 images of even width, the 7x7 / 2 / 3 convolution from 3 to 64 channels."""
        fused = getattr(ops, "stem_conv_pool", None)
        if fused is None or not (_R3["enabled"] and _FUSED_LINEAR["enabled"]):
            return None
        from .functions.conv import STEM_FUSED
        st = self.stem
        ok = (STEM_FUSED["enabled"] and x.is_cuda and x.dtype == torch.float16 and x.is_contiguous() and x.shape[-1] % 2 == 0
              and tuple(st.weight.shape) == (64, 3, 7, 7) and st.stride == (2, 2) and st.padding == (3, 3))
        return fused(x, st.weight, st.bias) if ok else None

    def forward_nhwc(self, x, ops):
        y = self._stem_fused(x, ops)
        if y is not None:
            return self._stages_nhwc(y, ops)
        x = x.contiguous(memory_format=torch.channels_last)
        pool = getattr(ops, "bias_relu_maxpool_nhwc", None)
        if pool is not None and _R3["enabled"] and _FUSED_LINEAR["enabled"] and x.dtype == torch.float16 and x.is_cuda:
            # stem: library convolution, then shift + ReLU + 3x3 / 2 max-pool as ONE pass over its output
            y = F.conv2d(x, self.stem.weight, None, self.stem.stride, self.stem.padding)
            if not y.is_contiguous(memory_format=torch.channels_last):
                y = y.contiguous(memory_format=torch.channels_last)
            x = pool(y, self.stem.bias)
        else:
            x = F.max_pool2d(_conv_nhwc(ops, x, self.stem, True), 3, 2, 1)
        return self._stages_nhwc(x, ops)

    def _stages_nhwc(self, x, ops):
        outs = []
        for i, st in enumerate(self.stages):
            for blk in st:
                x = blk.forward_nhwc(x, ops)
            if i in self.out_indices:
                outs.append(x)
        return outs


class FPN(nn.Module):
    """mmdet FPN with add_extra_convs='on_output', relu_before_extra_convs (bevformer_base.py:55-63)."""

    def __init__(self, cins, cout, num_outs):
        super().__init__()
        self.lateral = nn.ModuleList(nn.Conv2d(c, cout, 1) for c in cins)
        self.fpn = nn.ModuleList(nn.Conv2d(cout, cout, 3, 1, 1) for _ in cins)
        self.extra = nn.ModuleList(nn.Conv2d(cout, cout, 3, 2, 1) for _ in range(num_outs - len(cins)))

    def forward(self, feats):
        lat = [l(f) for l, f in zip(self.lateral, feats)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
        outs = [c(x) for c, x in zip(self.fpn, lat)]
        for e in self.extra:
            outs.append(e(F.relu(outs[-1])))
        return outs

    def forward_nhwc(self, feats, ops):
        return self.topdown_nhwc([_conv1x1_nhwc(ops, f, l, False) for l, f in zip(self.lateral, feats)], ops)

    def topdown_nhwc(self, lat, ops):
        """Everything behind the lateral 1x1 convolutions (the INT8 engine's chain evaluates those itself)."""
        up_add = getattr(ops, "upsample_add_nhwc_", None)
        for i in range(len(lat) - 1, 0, -1):
            a, b = lat[i - 1], lat[i]
            if up_add is not None and _FUSED_LINEAR["enabled"] and _R3["enabled"] and a.dtype == torch.float16 and a.is_cuda \
                    and a.is_contiguous(memory_format=torch.channels_last) and b.is_contiguous(memory_format=torch.channels_last):
                up_add(a, b)     # one in-place pass: a += nearest-up-sampled b
            else:
                lat[i - 1] = a + F.interpolate(b, size=a.shape[2:], mode="nearest")
        outs = [_conv_nhwc(ops, x.contiguous(memory_format=torch.channels_last), c, False) for c, x in zip(self.fpn, lat)]
        for e in self.extra:
            outs.append(_conv_nhwc(ops, F.relu(outs[-1]), e, False))
        return outs


# --------------------------------------------------------------------------- attention blocks
class FFN(nn.Module):
    def __init__(self, dim=EMBED, hidden=2 * EMBED):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, hidden), nn.Linear(hidden, dim)

    def forward(self, x, ops=None, norm=None):
        """`norm`: the block's LayerNorm behind the FFN (then evaluated here, in fc2's epilogue when the operator set can)."""
        h = _dense(ops, self.fc1, x, None, True)
        return _dense(ops, self.fc2, h, x, False) if norm is None else _dense_norm(ops, self.fc2, h, x, norm)


class TemporalSelfAttention(nn.Module):
    """temporal_self_attention.py:350-457 (num_bev_queue 2, 1 level, 4 points)."""

    def __init__(self, ops, points=4):
        super().__init__()
        self.ops, self.points = ops, points
        self.sampling_offsets = nn.Linear(EMBED * 2, 2 * HEADS * points * 2)
        self.attention_weights = nn.Linear(EMBED * 2, 2 * HEADS * points)
        self.value_proj, self.output_proj = nn.Linear(EMBED, EMBED), nn.Linear(EMBED, EMBED)
        self._split = self._pos_term = None

    def _split_projection(self, query, prev0, bev_pos):
        """sampling_offsets and attention_weights of cat([prev_bev, query + bev_pos]) as ONE stacked
        [192, 512] weight split along K:  prev_bev @ Wa.T + query @ Wb.T + (bev_pos @ Wb.T + b), the last
        term frame-independent (cached per bev_pos storage / version).  Two K=256 GEMMs with the running sum as
        their epilogue's identity replace the add, the [nq, 512] concatenation and two K=512 GEMMs.
        None -> the caller's module-by-module path (quantised build, fp32, CPU, no algorithm)."""
        fn = _gemm_entry(self.ops)
        so, aw = self.sampling_offsets, self.attention_weights
        if fn is None or not (_R3["enabled"] and _FUSED_LINEAR["enabled"]) or query.dtype != torch.float16 \
                or not query.is_cuda or type(so) is not nn.Linear or type(aw) is not nn.Linear:
            return None
        key = (so.weight._version, aw.weight._version, so.bias._version, aw.bias._version, so.weight.data_ptr())
        try:
            if self._split is None or self._split[0] != key:
                w = torch.cat([so.weight, aw.weight]).detach()
                self._split = (key, w[:, :EMBED].contiguous(), w[:, EMBED:].contiguous(),
                               torch.cat([so.bias, aw.bias]).detach().contiguous())
                self._pos_term = None
            _, wa, wb, b = self._split
            # keyed on the tensor's storage, not its Python identity: the model passes a fresh VIEW of the cached
            # positional encoding every frame
            pkey = (bev_pos.data_ptr(), bev_pos._version, tuple(bev_pos.shape), bev_pos.dtype)
            if self._pos_term is None or self._pos_term[0] != pkey:
                self._pos_term = (pkey, fn(bev_pos.reshape(-1, EMBED), wb, b, None, False))
            t = fn(prev0.reshape(-1, EMBED), wa, None, self._pos_term[1], False)
            return fn(query.reshape(-1, EMBED), wb, None, t, False)
        except _lib.BevopsError as exc:
            if exc.status != _lib.NOT_SUPPORTED:
                raise
            return None

    def forward(self, query, value, bev_pos, ref_2d, spatial_shapes, rows=None, norm=None):
        """`norm`: the block's LayerNorm behind this attention (evaluated in output_proj's epilogue when possible).
        `rows` = (lo, hi): query-range sharding (camera_shard.py, mode "scatter") -- `query`, `bev_pos` and `ref_2d`
        hold rows lo .. hi - 1 of the BEV queries only, `value` (the keys: prev_bev | query stack) is whole."""
        identity = query
        nq, nk = query.shape[1], value.shape[1]
        mine = value if rows is None else value[:, rows[0]:rows[1]]      # the value rows that pair with these queries
        both = self._split_projection(query, mine[0], bev_pos)
        split = getattr(self.ops, "tsa_split", None) if _TSA_GLUE["enabled"] else None
        pre = None
        if both is not None and split is not None and self.points == 4:
            pre = split(both, HEADS, self.points)      # (off [2, nq, H, 8], w [2, nq, H, 4]) in one pass
        elif both is not None:
            n_off = 2 * HEADS * self.points * 2
            off = both[:, :n_off].view(1, nq, HEADS, 2, 1, self.points, 2)
            w = both[:, n_off:].view(1, nq, HEADS, 2, 1, self.points)
        else:
            query = torch.cat([mine[:1], query + bev_pos], -1)
            off = self.sampling_offsets(query).view(1, nq, HEADS, 2, 1, self.points, 2)
            w = self.attention_weights(query).view(1, nq, HEADS, 2, 1, self.points)
        value = _dense(self.ops, self.value_proj, value).view(2, nk, HEADS, EMBED // HEADS)
        if pre is not None:
            off, w = pre
        else:
            w = w.permute(0, 3, 1, 2, 4, 5).contiguous().view(2, nq, HEADS, -1)
            off = off.permute(0, 3, 1, 2, 4, 5, 6).contiguous().view(2, nq, HEADS, -1)
        # the BEV grid's reference points have locality (neighbouring queries, neighbouring pixels): the operator set's
        # layout-preserving entry skips the head-major re-layout the default dispatch needs for random points
        msda = (getattr(self.ops, "multi_scale_deformable_attn_local", None) if _TSA_LOCAL["enabled"] else None) \
            or self.ops.multi_scale_deformable_attn
        out = msda(value, spatial_shapes, ref_2d, off, w).flatten(2)
        mean2 = getattr(self.ops, "queue_mean2", None) if _TSA_GLUE["enabled"] else None
        if mean2 is not None and out.is_cuda and out.dtype == torch.float16 and out.numel() % 16 == 0 and _R3["enabled"]:
            out = mean2(out)                            # (x0 + x1) / 2 in fp32, one rounding: torch.mean's bits
        else:
            out = torch.mean(out, keepdim=True, dim=0)
        if norm is not None:
            return _dense_norm(self.ops, self.output_proj, out, identity, norm)
        return _dense(self.ops, self.output_proj, out, identity, False)


class SpatialCrossAttention(nn.Module):
    """spatial_cross_attention.py:200-273 + MSDeformableAttention3DTRTP :694-768."""

    def __init__(self, ops, levels, points=8):
        super().__init__()
        self.ops = ops
        self._proj_ok = True
        self.sampling_offsets = nn.Linear(EMBED, HEADS * levels * points * 2)
        self.attention_weights = nn.Linear(EMBED, HEADS * levels * points)
        self.value_proj, self.output_proj = nn.Linear(EMBED, EMBED), nn.Linear(EMBED, EMBED)

    def forward(self, query, value, reference_points_cam, bev_mask, spatial_shapes, cams=None, gather=None, plan=None,
                residual=None, norm=None):
        """`plan`: the visibility plan (functions.spatial_cross_attention_plan) of the bev_mask rows of the cameras
        sampled here -- all of them, or this rank's -- or None.  `residual`: query-range sharding (exchange mode
        "scatter") -- `query` is the whole gathered tensor (every camera's sampler needs every query's offsets), the
        masked camera sum comes back reduce-SCATTERED to this rank's rows, and `residual` holds those rows of the
        layer input for the output projection."""
        inp_residual = query if residual is None else residual
        ncam, nk, nq = value.shape[0], value.shape[1], query.shape[1]   # ncam may be 0 (rank without cameras)
        mode = getattr(gather, "mode", None)
        if cams is None and gather is not None and hasattr(gather, "cams"):
            cams = gather.cams   # the exchange object knows this rank's cameras
        ref = reference_points_cam.reshape(reference_points_cam.shape[0], nq, 1, -1)
        # camera-sharded: `value` holds this rank's cameras only; their reference points / visibility weights
        ref_l, mask_l = (ref, bev_mask) if cams is None else (_take_cams(ref, cams), _take_cams(bev_mask, cams))
        # The fused forms produce the MASKED CAMERA SUM of the cameras they are given, so they serve a single GPU
        # (all cameras) and the "reduce" exchange (this rank's cameras, then ONE all-reduce of [1, nq, 256]) alike.
        local_sum = gather is None or mode in ("reduce", "scatter")
        projected = getattr(self.ops, "spatial_cross_attention_projected", None)
        fused = getattr(self.ops, "spatial_cross_attention_sample", None)
        fp16_gpu = value.dtype == torch.float16 and value.is_cuda
        slots = None
        if local_sum and ncam == 0:
            slots = torch.zeros((1, nq, EMBED), dtype=value.dtype, device=value.device)
        elif projected is not None and local_sum and _R3["enabled"] and self._proj_ok and fp16_gpu \
                and not hasattr(self.value_proj, "fake_quant_reference"):
            # value_proj's GEMM writes the sampler's planes itself; the fused sampling reads them
            off = _dense(self.ops, self.sampling_offsets, query).view(1, nq, HEADS, -1)
            w = _dense(self.ops, self.attention_weights, query).view(1, nq, HEADS, -1)
            try:
                slots = projected(value.reshape(ncam, nk, EMBED), self.value_proj.weight, self.value_proj.bias,
                                  spatial_shapes, ref_l.contiguous(), off, w, mask_l.contiguous(), HEADS,
                                  **({"plan": plan} if plan is not None else {}))
            except _lib.BevopsError as exc:
                if exc.status != _lib.NOT_SUPPORTED:
                    raise
                self._proj_ok = False   # another pyramid (tiny / small: one level): the separate projection below
        if slots is None:
            value = _dense(self.ops, self.value_proj, value.reshape(ncam, nk, EMBED)).view(ncam, nk, HEADS, EMBED // HEADS)
            # the per-camera copies of `query` are identical: project once, expand (stride 0)
            off = _dense(self.ops, self.sampling_offsets, query).view(1, nq, HEADS, -1).expand(ncam, -1, -1, -1)
            w = _dense(self.ops, self.attention_weights, query).view(1, nq, HEADS, -1).expand(ncam, -1, -1, -1)
            msda = self.ops.multi_scale_deformable_attn
            if mode == "gather":
                # camera-sharded, pipelined: camera i's all-gather overlaps the sampling of camera i + 1
                ref_c = ref_l.contiguous()
                queries = gather.gather(
                    lambda i: msda(value[i:i + 1], spatial_shapes, ref_c[i:i + 1], off[:1], w[:1]).flatten(2),
                    (nq, EMBED), value.dtype, value.device)
                slots = (queries * bev_mask).sum(0, keepdim=True)
            elif local_sum and fused is not None and (fp16_gpu or getattr(fused, "any_device", False)):
                # one call: camera-shared offsets, invisible (camera, query) pairs skipped, masked sum
                slots = fused(value, spatial_shapes, ref_l, off[:1], w[:1], mask_l)
            elif local_sum:
                queries = msda(value, spatial_shapes, ref_l.contiguous(), off, w).flatten(2)
                slots = (queries * mask_l).sum(0, keepdim=True)
            else:   # a plain callable: [cams_local, nq, 256] -> [6, nq, 256] on every rank
                queries = gather(msda(value, spatial_shapes, ref_l.contiguous(), off, w).flatten(2))
                slots = (queries * bev_mask).sum(0, keepdim=True)
        if mode == "reduce":
            slots = gather.reduce(slots)      # this rank's masked camera sum -> everyone's
        elif mode == "scatter":
            slots = gather.reduce_scatter_queries(slots, nq)      # ... -> the sum of MY rows only
        if norm is not None:    # the block's LayerNorm behind this attention, in output_proj's epilogue when possible
            return _dense_norm(self.ops, self.output_proj, slots, inp_residual, norm)
        return _dense(self.ops, self.output_proj, slots, inp_residual, False)


class BEVFormerLayer(nn.Module):
    def __init__(self, ops, levels):
        super().__init__()
        self.tsa, self.sca, self.ffn = TemporalSelfAttention(ops), SpatialCrossAttention(ops, levels), FFN()
        self.norms = nn.ModuleList(nn.LayerNorm(EMBED) for _ in range(3))

    def forward(self, query, value, bev_pos, ref_2d, ref_cam, bev_mask, spatial_shapes, bev_shapes, prev_bev,
                use_prev_bev, cams, gather, plan=None, rows=None):
        ops = self.tsa.ops
        if rows is not None:
            # query-range sharding (camera_shard.py, mode "scatter"): `query` holds rows lo .. hi - 1 only.  TSA's
            # keys are prev_bev | the layer-0 query stack (whole on every rank) -- or, without history, the CURRENT
            # query repeated, which then has to be gathered first
            lo, hi = rows
            nq = bev_pos.shape[1]
            if torch.is_tensor(use_prev_bev) or not use_prev_bev:
                full = gather.all_gather_queries(query, nq)
                prev = torch.where(use_prev_bev.to(torch.bool), prev_bev, full.expand(2, -1, -1)) \
                    if torch.is_tensor(use_prev_bev) else full.repeat(2, 1, 1)
            else:
                prev = prev_bev
            query = self.tsa(query, prev, bev_pos[:, lo:hi], ref_2d[:, lo:hi].contiguous(), bev_shapes, rows=rows,
                             norm=self.norms[0])
            full = gather.all_gather_queries(query, nq)       # every camera's sampler needs every query's offsets
            query = self.sca(full, value, ref_cam, bev_mask, spatial_shapes, cams, gather, plan, residual=query,
                             norm=self.norms[1])
            return self.ffn(query, ops, norm=self.norms[2])
        # encoder.py:586-588: use_prev_bev * prev_bev + (1 - use_prev_bev) * query.repeat(2, 1, 1) with
        # use_prev_bev in {0, 1} -- a select (one pass, exact) instead of two scalings, a copy and an add
        if torch.is_tensor(use_prev_bev):
            prev = torch.where(use_prev_bev.to(torch.bool), prev_bev, query.expand(2, -1, -1))
        else:
            prev = prev_bev if use_prev_bev else query.repeat(2, 1, 1)
        # (each block's LayerNorm rides in the epilogue of the block's last GEMM: _dense_norm)
        query = self.tsa(query, prev, bev_pos, ref_2d, bev_shapes, norm=self.norms[0])
        query = self.sca(query, value, ref_cam, bev_mask, spatial_shapes, cams, gather, plan, norm=self.norms[1])
        return self.ffn(query, ops, norm=self.norms[2])


def _static_term(cache, owner, key, pos, fn, weight, bias):
    """pos @ weight.T + bias for a frame-independent `pos` (the learned query / BEV position embedding):
    evaluated once and reused as the identity term of the layer's GEMM epilogue, so that
    lin(x + pos) = x @ W.T + (pos @ W.T + b) costs one GEMM and no add per frame.  Cached on `owner`
    under `cache`, keyed by the parameter versions in `key` and the identity of the `pos` tensor."""
    hit = getattr(owner, cache, None)
    if hit is None or hit[0] != key or hit[1] is not pos:
        hit = (key, pos, fn(pos.reshape(-1, pos.shape[-1]), weight, bias, None, False))
        setattr(owner, cache, hit)
    return hit[2]


def _fast_dense_ok(ops, x, *linears):
    return getattr(ops, "linear_bias_act", None) is not None and _R3["enabled"] and _FUSED_LINEAR["enabled"] \
        and x.dtype == torch.float16 and x.is_cuda \
        and all(isinstance(l, nn.Linear) and not hasattr(l, "fake_quant_reference") for l in linears)


def _versions(*tensors):
    return tuple(t._version for t in tensors) + tuple(t.data_ptr() for t in tensors)


class CustomMSDeformableAttention(nn.Module):
    """decoder.py:381-471 (1 level, 4 points)."""

    def __init__(self, ops, points=4):
        super().__init__()
        self.ops = ops
        self.sampling_offsets = nn.Linear(EMBED, HEADS * points * 2)
        self.attention_weights = nn.Linear(EMBED, HEADS * points)
        self.value_proj, self.output_proj = nn.Linear(EMBED, EMBED), nn.Linear(EMBED, EMBED)
        self._pos_so = self._pos_aw = None

    def forward(self, query, value, query_pos, reference_points, spatial_shapes, norm=None):
        identity = query                               # [900, 1, 256]
        ops, so, aw = self.ops, self.sampling_offsets, self.attention_weights
        n = query.shape[0]
        value = _dense(ops, self.value_proj, value.view(1, -1, EMBED)).view(1, -1, HEADS, EMBED // HEADS)
        off = w = None
        if _fast_dense_ok(ops, query, so, aw):
            # lin(query + query_pos) with the query_pos term frame-independent: no add, bias folded
            fn = _gemm_entry(ops)
            try:
                t_so = _static_term("_pos_so", self, _versions(so.weight, so.bias), query_pos, fn, so.weight, so.bias)
                t_aw = _static_term("_pos_aw", self, _versions(aw.weight, aw.bias), query_pos, fn, aw.weight, aw.bias)
                off = fn(query.view(n, EMBED), so.weight, None, t_so, False).view(1, n, HEADS, -1)
                w = fn(query.view(n, EMBED), aw.weight, None, t_aw, False).view(1, n, HEADS, -1)
            except _lib.BevopsError as exc:
                if exc.status != _lib.NOT_SUPPORTED:
                    raise
                off = None
        if off is None:
            q = (query + query_pos).view(1, -1, EMBED)
            off = so(q).view(1, n, HEADS, -1)
            w = aw(q).view(1, n, HEADS, -1)
        out = ops.multi_scale_deformable_attn(value, spatial_shapes, reference_points, off, w).flatten(2)
        # [1, 900, 256] -> [900, 1, 256] is the same memory: the identity goes into the GEMM epilogue
        if norm is not None:
            return _dense_norm(ops, self.output_proj, out.view(n, 1, EMBED), identity, norm)
        return _dense(ops, self.output_proj, out.view(n, 1, EMBED), identity, False)


class DecoderLayer(nn.Module):
    def __init__(self, ops):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(EMBED, HEADS)
        self.cross_attn, self.ffn = CustomMSDeformableAttention(ops), FFN()
        self.norms = nn.ModuleList(nn.LayerNorm(EMBED) for _ in range(3))
        self._pos_qkv = None

    def _self_attention(self, ops, query, query_pos, norm=None):
        """query + MultiheadAttention(q = k = query + query_pos, v = query) (mmcv MultiheadAttention
        with its identity, decoder layer operation order) as: ONE [900, 768] in-projection whose
        query_pos part is a cached identity term, the fused attention kernel on head-strided views,
        and the out-projection with the residual in its epilogue.  None -> nn.MultiheadAttention."""
        mha = self.self_attn
        if not _fast_dense_ok(ops, query, mha.out_proj) or mha.in_proj_weight is None:
            return None
        fn, n = _gemm_entry(ops), query.shape[0]
        w, b = mha.in_proj_weight, mha.in_proj_bias
        key = _versions(w, b)
        hit = self._pos_qkv
        try:
            if hit is None or hit[0] != key or hit[1] is not query_pos:
                # q and k see query_pos, v does not: its rows of the static term are the bias alone
                term = torch.empty(n, 3 * EMBED, dtype=query.dtype, device=query.device)
                term[:, :2 * EMBED] = fn(query_pos.reshape(n, EMBED), w[:2 * EMBED], b[:2 * EMBED], None, False)
                term[:, 2 * EMBED:] = b[2 * EMBED:]
                hit = self._pos_qkv = (key, query_pos, term)
            qkv = fn(query.view(n, EMBED), w, None, hit[2], False).view(1, n, 3, HEADS, EMBED // HEADS)
        except _lib.BevopsError as exc:
            if exc.status != _lib.NOT_SUPPORTED:
                raise
            return None
        own = getattr(ops, "self_attention_qkv", None) if _OWN_ATTN["enabled"] else None
        o = None
        if own is not None:
            try:     # one launch on the matrix cores (K / V of a head staged in LDS): 900 queries x 8 heads x 32
                o = own(qkv.view(n, 3, HEADS, EMBED // HEADS)).view(n, 1, EMBED)
            except _lib.BevopsError as exc:
                if exc.status != _lib.NOT_SUPPORTED:
                    raise
        if o is None:
            q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))           # [1, heads, 900, 32] views
            o = F.scaled_dot_product_attention(q, k, v)                          # [1, heads, 900, 32]
            o = o.transpose(1, 2).reshape(n, 1, EMBED)
        return _dense(ops, mha.out_proj, o, query, False) if norm is None else _dense_norm(ops, mha.out_proj, o, query, norm)

    def forward(self, query, value, query_pos, reference_points, spatial_shapes):
        ops = self.cross_attn.ops
        query_n = self._self_attention(ops, query, query_pos, norm=self.norms[0])
        if query_n is None:
            qk = query + query_pos
            query_n = _layer_norm(ops, self.norms[0], query + self.self_attn(qk, qk, query, need_weights=False)[0])
        query = self.cross_attn(query_n, value, query_pos, reference_points, spatial_shapes, norm=self.norms[1])
        return self.ffn(query, ops, norm=self.norms[2])


class BEVFormer(nn.Module):
    """forward(image [1,6,3,H,W], prev_bev [nq,1,256], use_prev_bev (0/1 tensor), can_bus [18],
    lidar2img [1,6,4,4]) -> bev_embed [nq,1,256], outputs_classes [6,1,900,10], outputs_coords [6,1,900,10]."""

    def __init__(self, name="base", ops=None, seed=0, backbone_layout="auto"):
        super().__init__()
        torch.manual_seed(seed)
        cfg = CONFIGS[name]
        self.cfg, self.name = cfg, name
        self.ops = ops = ops if ops is not None else _hip_ops
        # "auto": channels-last backbone when the MI355X operators are in use and the frame is fp16
        self.backbone_layout = backbone_layout
        self._nhwc_ready = False
        self.bev_h, self.bev_w = cfg["bev"]
        nq = self.bev_h * self.bev_w
        self.backbone = ResNet(cfg["depth"], cfg["dcn"], cfg["out_indices"], ops, cfg["style"])
        self.neck = FPN(cfg["fpn_in"], EMBED, cfg["levels"])
        # head (bevformer_head.py): embeddings, learned positional encoding, branches
        self.bev_embedding = nn.Embedding(nq, EMBED)
        self.query_embedding = nn.Embedding(NUM_QUERY, EMBED * 2)
        self.row_embed, self.col_embed = nn.Embedding(self.bev_h, EMBED // 2), nn.Embedding(self.bev_w, EMBED // 2)
        self.cls_branches = nn.ModuleList(nn.Sequential(
            nn.Linear(EMBED, EMBED), nn.LayerNorm(EMBED), nn.ReLU(inplace=True),
            nn.Linear(EMBED, EMBED), nn.LayerNorm(EMBED), nn.ReLU(inplace=True), nn.Linear(EMBED, 10)) for _ in range(6))
        self.reg_branches = nn.ModuleList(nn.Sequential(
            nn.Linear(EMBED, EMBED), nn.ReLU(inplace=True), nn.Linear(EMBED, EMBED), nn.ReLU(inplace=True),
            nn.Linear(EMBED, 10)) for _ in range(6))
        # transformer (transformer.py)
        self.level_embeds = nn.Parameter(torch.randn(cfg["levels"], EMBED) * 0.02)
        self.cams_embeds = nn.Parameter(torch.randn(NUM_CAMS, EMBED) * 0.02)
        self.reference_points = nn.Linear(EMBED, 3)
        self.can_bus_mlp = nn.Sequential(nn.Linear(18, EMBED // 2), nn.ReLU(inplace=True),
                                         nn.Linear(EMBED // 2, EMBED), nn.ReLU(inplace=True), nn.LayerNorm(EMBED))
        self.encoder = nn.ModuleList(BEVFormerLayer(ops, cfg["levels"]) for _ in range(cfg["enc_layers"]))
        self.decoder = nn.ModuleList(DecoderLayer(ops) for _ in range(6))
        self.register_buffer("rotate_center", torch.tensor([100.0, 100.0]))   # transformer.py:26
        self._static = None
        self._pos_cache = self._query_cache = None
        self.eval()

    # ---- detector/bevformer.py:12-35
    def extract_feat(self, image, cams=None):
        B, N, C, H, W = image.shape
        img = image.view(B * N, C, H, W)
        if cams is not None:
            img = _take_cams(img, cams)
        nhwc = self.backbone_layout == "nhwc" or (self.backbone_layout == "auto" and self.ops is _hip_ops)
        chain = getattr(self, "int8_chain", None)     # quantization.Int8ChainBackbone, after its freeze()
        if chain is not None and chain.ready and img.dtype == torch.float16 and img.is_cuda:
            return chain(img)       # (img is already this rank's camera subset when `cams` is given)
        if nhwc and img.dtype == torch.float16 and img.is_cuda:
            if not self._nhwc_ready:   # MIOpen picks its NHWC kernels when the filters are channels-last too
                for m in list(self.backbone.modules()) + list(self.neck.modules()):
                    if isinstance(m, nn.Conv2d) and m.kernel_size != (1, 1):
                        m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
                self._nhwc_ready = True
            return self.neck.forward_nhwc(self.backbone.forward_nhwc(img, self.ops), self.ops)
        return self.neck(self.backbone(img))    # list of [cams, 256, h_l, w_l]

    def positional_encoding(self, dtype, device):   # mmdet LearnedPositionalEncoding
        x = self.col_embed(torch.arange(self.bev_w, device=device))
        y = self.row_embed(torch.arange(self.bev_h, device=device))
        pos = torch.cat((x.unsqueeze(0).repeat(self.bev_h, 1, 1), y.unsqueeze(1).repeat(1, self.bev_w, 1)), dim=-1)
        return pos.permute(2, 0, 1).unsqueeze(0).to(dtype)   # [1, 256, h, w]

    @torch.no_grad()
    def _geometry(self, dev):
        if self._static is None or self._static[0].device != dev:   # frame-independent geometry
            # evaluated on the HOST once (linspace / scalar divisions are not bit-stable across
            # devices) and uploaded: the anchors are then the reference's CPU values bit for bit
            ref_3d = G.reference_points_3d(self.bev_h, self.bev_w, PC_RANGE[5] - PC_RANGE[2], 4, device="cpu", dtype=torch.float)
            self._static = tuple(t.to(dev) for t in (ref_3d, G.reference_points_2d(ref_3d),
                                                     G.pillar_points(ref_3d, PC_RANGE)))
        return self._static

    @torch.no_grad()
    def project(self, lidar2img, image_shape, dtype, cams=None):
        """point_sampling_trt (encoder.py:197-259) of the BEV pillars for one rig: (reference_points_cam, bev_mask[,
        visibility plan]) in the model's dtype.  It depends on the calibration matrices only, and those change on EVERY
        nuScenes frame (ego motion between the camera and lidar timestamps; the reference evaluates this inside the
        engine per frame, lidar2img being an engine input: tools/bevformer/evaluate_trt.py:131-132, encoder.py:293):
        `forward` calls it on every frame, inside the frame's HIP graph -- one launch (ops.point_sampling, bit-identical
        to the torch op sequence of geometry.project_points) plus the two launches of the plan build.  `cams`: the
        cameras sampled on this rank (camera sharding); the plan then lists those only."""
        _, _, pillars = self._geometry(lidar2img.device)
        fused = getattr(self.ops, "point_sampling", None)
        if fused is not None and _R3["enabled"] and lidar2img.is_cuda and dtype in (torch.float16, torch.float32) \
                and pillars.shape[0] == 4:
            ref_cam, bev_mask = fused(pillars, lidar2img, image_shape, dtype)
        else:
            ref_cam, bev_mask = G.project_points(pillars, lidar2img.float(), image_shape, projection="fma")
            ref_cam, bev_mask = ref_cam.to(dtype), bev_mask.to(dtype)
        plan = self._sca_plan(bev_mask, cams)
        return (ref_cam, bev_mask) if plan is None else (ref_cam, bev_mask, plan)

    def _sca_plan(self, bev_mask, cams):
        """Visibility plan of the fused SCA sampling for the cameras sampled on this rank (None: no such operator, or
        outside its domain).  Depends on the calibration only, like bev_mask itself."""
        fn = getattr(self.ops, "spatial_cross_attention_plan", None)
        if fn is None or not _R3["enabled"] or not bev_mask.is_cuda or bev_mask.dtype != torch.float16 \
                or self.cfg["levels"] != 4:      # (the planned sampler's domain: the 4-level x 8-point pyramid of "base")
            return None
        mask_l = bev_mask if cams is None else _take_cams(bev_mask, cams)
        return fn(mask_l) if mask_l.shape[0] > 0 else None

    def forward(self, image, prev_bev, use_prev_bev, can_bus, lidar2img, cams=None, gather=None, shift=None, proj=None):
        """`shift` [1, 2]: optional precomputed G.bev_shift(can_bus) -- the frame loop evaluates it
        on the host (atan / sin / cos differ in the last ulp between host and device libraries;
        the host value is the reference's CPU path bit for bit).  `proj`: optional precomputed `project(lidar2img, ..., cams)`
        (callers that know the calibration to be constant; the frame loop does not: it changes every frame)."""
        mlvl = self.extract_feat(image, cams)
        if _OWN_ENCODER["enabled"] and image.is_cuda and image.dtype == torch.float16:
            # everything behind the backbone on the hand-written GEMMs (fixed summation order): see _OWN_ENCODER
            from .functions.linear import OWN_KERNELS
            was, OWN_KERNELS["enabled"] = OWN_KERNELS["enabled"], True
            try:
                return self._transformer(mlvl, image, prev_bev, use_prev_bev, can_bus, lidar2img, cams, gather, shift, proj)
            finally:
                OWN_KERNELS["enabled"] = was
        return self._transformer(mlvl, image, prev_bev, use_prev_bev, can_bus, lidar2img, cams, gather, shift, proj)

    def _transformer(self, mlvl, image, prev_bev, use_prev_bev, can_bus, lidar2img, cams, gather, shift, proj):
        """Everything behind the backbone + FPN: embeddings, encoder, decoder, heads."""
        dev, dtype = image.device, image.dtype
        image_shape = image.shape[-2:]
        bev_h, bev_w, nq = self.bev_h, self.bev_w, self.bev_h * self.bev_w
        bev_queries = self.bev_embedding.weight.to(dtype).unsqueeze(1)           # [nq, 1, 256]
        pkey = (self.col_embed.weight._version, self.row_embed.weight._version, dtype, dev)
        if _R3["enabled"] and self._pos_cache is not None and self._pos_cache[0] == pkey:
            bev_pos = self._pos_cache[1]     # frame-independent: the same TENSOR every frame (TSA keys its cached term on it)
        else:
            bev_pos = self.positional_encoding(dtype, dev).flatten(2).permute(2, 0, 1).contiguous()  # [nq, 1, 256]
            self._pos_cache = (pkey, bev_pos)

        # ---- transformer.get_bev_features_trt (:245-341); index/grid math in fp32 (a6)
        grid_length = ((PC_RANGE[4] - PC_RANGE[1]) / bev_h, (PC_RANGE[3] - PC_RANGE[0]) / bev_w)
        if shift is None:
            shift = G.bev_shift(can_bus.float(), bev_h, bev_w, grid_length)
        shift = shift.to(dtype)
        rot_hwc = getattr(self.ops, "rotate_hwc", None)
        stack = None     # TSA's keys [prev_bev | query] (encoder.py:586-588), assembled without a concatenation
        if rot_hwc is not None and prev_bev.is_cuda and prev_bev.shape[-1] % 8 == 0 and _R3["enabled"] \
                and prev_bev.dtype == dtype and prev_bev.shape[-1] == EMBED:
            # prev_bev [nq, 1, C] already is [H, W, C]: rotated straight into row 0 of the stack; the can_bus-shifted
            # queries are added straight into row 1 (round 6: the torch.cat of the two moved 41 MB per frame)
            stack = torch.empty((2, nq, EMBED), dtype=dtype, device=dev)
            rot_hwc(prev_bev.view(bev_h, bev_w, -1), can_bus[-1].float().reshape(1), self.rotate_center.float(),
                    out=stack[0].view(bev_h, bev_w, EMBED))
            prev_bev = stack[0].view(nq, 1, EMBED)
            # (inference only: `out=` does not record a graph, the operands are detached to say so)
            bev_queries = torch.add(bev_queries.detach(),
                                    self.can_bus_mlp(can_bus.view(1, -1).to(dtype)).view(1, 1, -1).detach(),
                                    out=stack[1].view(nq, 1, EMBED))
        elif rot_hwc is not None and prev_bev.is_cuda and prev_bev.shape[-1] % 8 == 0:
            prev_bev = rot_hwc(prev_bev.view(bev_h, bev_w, -1), can_bus[-1].float().reshape(1),
                               self.rotate_center.float()).view(nq, 1, -1)
        else:
            prev_bev = self.ops.rotate(prev_bev.view(bev_h, bev_w, -1).permute(2, 0, 1),
                                       can_bus[-1].float().reshape(1), self.rotate_center.float())
            prev_bev = prev_bev.permute(1, 2, 0).reshape(nq, 1, -1)
        if stack is None:
            bev_queries = bev_queries + self.can_bus_mlp(can_bus.view(1, -1).to(dtype)).view(1, 1, -1)
        feats, level_hw = [], []
        cam_embed = self.cams_embeds if cams is None else _take_cams(self.cams_embeds, cams)
        embed_fn = getattr(self.ops, "feat_embed_nhwc", None)
        fused_embed = embed_fn is not None and _FUSED_LINEAR["enabled"] and _R3["enabled"] and dtype == torch.float16 and image.is_cuda \
            and all(f.is_contiguous(memory_format=torch.channels_last) for f in mlvl) and mlvl[0].shape[0] > 0
        if fused_embed:   # one pass per level straight into the concatenated tensor (no adds, no cat)
            level_hw = [f.shape[-2:] for f in mlvl]
            feat_flatten = torch.empty((mlvl[0].shape[0], sum(h * w for h, w in level_hw), EMBED), dtype=dtype, device=dev)
            row = 0
            for lvl, feat in enumerate(mlvl):
                hw = feat.shape[2] * feat.shape[3]
                embed_fn(feat.permute(0, 2, 3, 1).reshape(feat.shape[0], hw, feat.shape[1]), cam_embed,
                         self.level_embeds[lvl], feat_flatten[:, row:row + hw, :])
                row += hw
        else:
            for lvl, feat in enumerate(mlvl):
                level_hw.append(feat.shape[-2:])
                if feat.is_contiguous(memory_format=torch.channels_last):            # already [cams, h, w, 256]
                    f = feat.permute(0, 2, 3, 1).reshape(feat.shape[0], feat.shape[2] * feat.shape[3], feat.shape[1])
                else:
                    f = feat.flatten(2).permute(0, 2, 1)                              # [cams, hw, 256]
                feats.append(f + cam_embed.to(dtype)[:, None, :] + self.level_embeds[lvl].to(dtype)[None, None, :])
            feat_flatten = torch.cat(feats, dim=1)                                   # [cams, sum hw, 256]
        # shape tensors live on the HOST: the operators cache a device copy per distinct value
        # and never have to synchronise to learn the pyramid geometry
        spatial_shapes, _ = G.level_layout(level_hw, "cpu")
        bev_shapes = torch.tensor([[bev_h, bev_w]])

        # ---- encoder.forward_trt (:261-334)
        ref_3d, ref_2d, pillars = self._geometry(dev)
        if cams is None and gather is not None and hasattr(gather, "cams"):
            cams = gather.cams     # the exchange object knows this rank's cameras: the plan must list exactly those
        # (a precomputed `proj` must have been made for the same `cams`: its plan lists the cameras sampled HERE)
        ref_cam, bev_mask, *rest = proj if proj is not None else self.project(lidar2img, image_shape, dtype, cams)
        plan = rest[0] if rest else None
        hybrid = G.hybrid_ref_2d(ref_2d, shift.float(), use_prev_bev).to(dtype)
        q = bev_queries.view(1, nq, EMBED)
        pos = bev_pos.view(1, nq, EMBED)
        prev = stack if stack is not None else torch.cat([prev_bev.view(1, nq, EMBED), q], dim=0)
        rows = None
        if getattr(gather, "mode", None) == "scatter":       # the encoder beyond the cameras sharded by query range
            lo, hi, _ = gather.query_range(nq)
            rows = (lo, hi)
            q = q[:, lo:hi].contiguous()
        for layer in self.encoder:
            q = layer(q, feat_flatten, pos, hybrid, ref_cam, bev_mask, spatial_shapes, bev_shapes, prev,
                      use_prev_bev, cams, gather, plan, rows)
        if rows is not None:
            q = gather.all_gather_queries(q, nq)              # the decoder (900 object queries) is replicated
        bev_embed = q.view(nq, 1, EMBED)

        # ---- decoder (transformer.forward_trt :375-398, decoder.py:52-112)
        qkey = _versions(self.query_embedding.weight, self.reference_points.weight, self.reference_points.bias) + (dtype, dev)
        if _R3["enabled"] and self._query_cache is not None and self._query_cache[0] == qkey:
            _, query_pos, query, reference_points = self._query_cache   # frame-independent: the same TENSORS every frame
        else:
            query_pos, query = (t.contiguous() for t in
                                torch.split(self.query_embedding.weight.to(dtype).unsqueeze(1), EMBED, dim=2))
            reference_points = self.reference_points(query_pos).sigmoid().view(1, NUM_QUERY, 3)
            self._query_cache = (qkey, query_pos, query, reference_points)
        init_reference = reference_points
        inter, inter_refs, regs = [], [], []
        out = query
        # The refinement is ONE launch (round 6, functions/refine.py): the framework's eight element-wise launches per
        # layer with its rounding after every step, log and sigmoid taken from tables the framework's own kernels
        # filled -- bit-exact on every binary16 input (round 4's one-launch build evaluated exp / log itself, missed
        # single ulps -- the refined points are sampling locations -- and was removed).
        fused_refine = getattr(self.ops, "refine_reference_points", None) if _FUSED_REFINE["enabled"] and _R3["enabled"] \
            and dtype == torch.float16 and dev.type == "cuda" else None
        ref_xy = reference_points[..., :2].unsqueeze(2).contiguous()
        for lid, layer in enumerate(self.decoder):
            out = layer(out, bev_embed, query_pos, ref_xy, bev_shapes)
            tmp = _mlp(self.ops, self.reg_branches[lid], out).view(1, -1, 10)
            if fused_refine is not None:
                reference_points, ref_xy = fused_refine(tmp, reference_points)
            else:
                reference_points = G.refine_reference_points(tmp, reference_points)      # decoder.py:93-103
                ref_xy = reference_points[..., :2].unsqueeze(2).contiguous()
            inter.append(out)
            inter_refs.append(reference_points)
            regs.append(tmp)

        # ---- head (bevformer_head.py:247-282).  The regression branch of level l on inter[l] is the tensor
        # the decoder loop already evaluated (same module, same input); the six levels' post-processing and
        # classification branches run as ONE batch over [6, 900, .] (element-wise ops: identical values;
        # branches: batched GEMMs with the stacked weights) instead of 6 x ~25 small launches
        if not _R3["enabled"]:   # A/B: the per-level loop of the reference head
            classes, coords = [], []
            for lvl in range(6):
                reference = inverse_sigmoid(init_reference if lvl == 0 else inter_refs[lvl - 1])
                hs = inter[lvl].view(1, NUM_QUERY, EMBED)
                crd = self.reg_branches[lvl](hs).clone()
                crd[..., 0:2] = (crd[..., 0:2] + reference[..., 0:2]).sigmoid()
                crd[..., 4:5] = (crd[..., 4:5] + reference[..., 2:3]).sigmoid()
                crd[..., 0:1] = crd[..., 0:1] * (PC_RANGE[3] - PC_RANGE[0]) + PC_RANGE[0]
                crd[..., 1:2] = crd[..., 1:2] * (PC_RANGE[4] - PC_RANGE[1]) + PC_RANGE[1]
                crd[..., 4:5] = crd[..., 4:5] * (PC_RANGE[5] - PC_RANGE[2]) + PC_RANGE[2]
                classes.append(self.cls_branches[lvl](hs))
                coords.append(crd)
            return bev_embed, torch.stack(classes), torch.stack(coords)
        crd = torch.stack(regs)                                                        # [6, 1, 900, 10]
        decode = getattr(self.ops, "decode_boxes", None) if fused_refine is not None else None
        hs = torch.stack(inter).view(6, NUM_QUERY, EMBED)
        if decode is not None:      # the ~20 element-wise launches below as ONE, bit-exact (functions/refine.py)
            crd = decode(crd, torch.stack([init_reference] + inter_refs[:-1]), PC_RANGE)
            return bev_embed, self._cls_batched(hs).view(6, 1, NUM_QUERY, -1), crd
        reference = inverse_sigmoid(torch.stack([init_reference] + inter_refs[:-1]))   # [6, 1, 900, 3]
        crd[..., 0:2] = (crd[..., 0:2] + reference[..., 0:2]).sigmoid()
        crd[..., 4:5] = (crd[..., 4:5] + reference[..., 2:3]).sigmoid()
        crd[..., 0:1] = crd[..., 0:1] * (PC_RANGE[3] - PC_RANGE[0]) + PC_RANGE[0]
        crd[..., 1:2] = crd[..., 1:2] * (PC_RANGE[4] - PC_RANGE[1]) + PC_RANGE[1]
        crd[..., 4:5] = crd[..., 4:5] * (PC_RANGE[5] - PC_RANGE[2]) + PC_RANGE[2]
        return bev_embed, self._cls_batched(hs).view(6, 1, NUM_QUERY, -1), crd

    def _cls_batched(self, hs):
        """cls_branches[l](hs[l]) for the six decoder levels as batched GEMMs: Linear -> LayerNorm -> ReLU ->
        Linear -> LayerNorm -> ReLU -> Linear with the per-level parameters stacked along the batch."""
        key = tuple(p._version for p in self.cls_branches.parameters()) + (hs.dtype, hs.device)
        if getattr(self, "_cls_stack", None) is None or self._cls_stack[0] != key:
            st = lambda k, attr: torch.stack([getattr(b[k], attr).detach() for b in self.cls_branches]).to(hs.dtype)
            self._cls_stack = (key, [(st(k, "weight").transpose(1, 2).contiguous(), st(k, "bias").unsqueeze(1)) for k in (0, 3, 6)],
                               [(st(k, "weight").unsqueeze(1), st(k, "bias").unsqueeze(1), self.cls_branches[0][k].eps) for k in (1, 4)])
        _, lin, norm = self._cls_stack
        x = hs
        for i in range(3):
            x = torch.baddbmm(lin[i][1], x, lin[i][0])
            if i < 2:
                g, b, eps = norm[i]
                x = F.relu(F.layer_norm(x, (x.shape[-1],), None, None, eps) * g + b, inplace=True)
        return x


def use_tuned_gemms(path=None):
    """Dense layers stay on hipBLASLt / rocBLAS; their per-shape solution choice
    was tuned once on an MI355X with PyTorch's TunableOp (36 s for the 35 GEMM shapes of
    BEVFormer-base, e.g. the 34800x256x1024 layer-3 1x1 convolution 61 -> 44 us) and is shipped as
    bevformer_tensorrt_amd/tunableop_gfx950.csv; this only reads it (no tuning at run time; the
    file is ignored by PyTorch if its library-version validators do not match)."""
    import os
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")
    try:
        if not os.path.exists(path) or os.environ.get("PYTORCH_TUNABLEOP_ENABLED") is not None:
            return False          # the user drives TunableOp through the environment
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(False)
        tunable.set_filename(path)
        tunable.read_file(path)
        return True
    except Exception:
        return False


class FrameRunner:
    """Stateful frame loop of tools/bevformer/evaluate_trt.py:76-154 with `prev_bev` kept on the
    device: can_bus position/angle deltas against the previous frame, `use_prev_bev = 0` on a
    scene change.  With `graph=True` the whole device-side frame (about 1 300 kernels at base)
    is captured into a HIP graph and replayed per frame from static input buffers -- shapes are
    static, exactly the property TensorRT exploits in the reference.  `use_prev_bev` is known on the
    host (a scene change), so it is a host value of the forward and there is one graph per value: the
    first frame of a scene replays the "no history" graph, every other frame the "history" graph, and
    neither carries the per-layer select between prev_bev and the repeated query."""

    def __init__(self, model, device, dtype, graph=False, cams=None, gather=None, clone_outputs=True):
        """clone_outputs=False: under graph replay `step` hands out the graph's own output buffers (valid until the
        next `step`, like the output bindings of a TensorRT execution context) instead of copies."""
        self.model, self.device, self.dtype, self.clone_outputs = model, device, dtype, clone_outputs
        self.cams, self.gather = cams, gather
        self.tuned_gemms = use_tuned_gemms() if device.type == "cuda" else False
        nq = model.bev_h * model.bev_w
        self.prev_bev = torch.zeros(nq, 1, EMBED, device=device, dtype=dtype)
        self.prev = {"scene": None, "pos": None, "angle": None}
        self.use_graph, self._graphs, self._use = graph, {}, 0.0
        H, W = model.cfg["image"]
        # the frame's small host-side inputs travel as ONE upload: [can_bus (18) | bev shift (2) | lidar2img (6 x 4 x 4)].
        # The calibration matrices are per-FRAME inputs, as in the reference (an engine input next to can_bus,
        # tools/bevformer/evaluate_trt.py:99,131-132): the camera projection of the BEV pillars and the SCA visibility
        # plan are evaluated from this buffer INSIDE the frame (and its graph), on every frame.
        n_small = 20 + NUM_CAMS * 16
        small = torch.zeros(n_small, device=device)
        self._host_small = torch.zeros(n_small)
        self._in = dict(image=torch.zeros(1, NUM_CAMS, 3, H, W, device=device, dtype=dtype),
                        small=small, can_bus=small[:18], shift=small[18:20].view(1, 2),
                        lidar2img=small[20:].view(1, NUM_CAMS, 4, 4),
                        use=torch.zeros((), device=device, dtype=dtype))

    @property
    def image_buffer(self):
        """The static [1, cams, 3, H, W] input buffer of the frame: a caller that writes its (normalised) camera
        images here -- as `step_raw` does -- and passes this very tensor to `step` saves the per-frame copy."""
        return self._in["image"]

    @property
    def _graph(self):   # the graph of the current use_prev_bev value (None: not captured yet)
        return self._graphs.get(self._use, (None, None))[0]

    def _forward(self):
        if self.gather is not None and self.device.type == "cuda" and not _SHARDED_TABLE_BACKBONE["enabled"]:
            # camera-sharded: every rank must evaluate the REPLICATED layers (TSA, FFN, decoder, heads) with the same
            # kernels.  With _OWN_ENCODER those layers -- everything behind the backbone -- already run on the
            # hand-written kernels, chosen by the shipped table (the same file on every rank) or, for a row count
            # the table has not seen, by a rule of the shape alone; the backbone is per-camera work no other rank
            # repeats and keeps the measured dispatch.  Without _OWN_ENCODER (or BEVOPS_SHARDED_RULE_DISPATCH=1, the
            # behaviour of rounds 3-5, 6-9 % slower) the rule-based dispatch is on for the whole forward.
            from .functions.linear import DETERMINISTIC
            prev, DETERMINISTIC["enabled"] = DETERMINISTIC["enabled"], True
            try:
                return self._forward_impl()
            finally:
                DETERMINISTIC["enabled"] = prev
        return self._forward_impl()

    def _forward_impl(self):
        i = self._in
        if not _R3["enabled"]:   # A/B: the device-side flag (one graph, per-layer select)
            return self.model(i["image"], self.prev_bev, i["use"], i["can_bus"], i["lidar2img"], self.cams, self.gather,
                              shift=i["shift"])
        return self.model(i["image"], self.prev_bev, self._use, i["can_bus"], i["lidar2img"], self.cams, self.gather,
                          shift=i["shift"])

    def _capture(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):       # warm-up on the capture stream (allocations, MIOpen find)
            for _ in range(2):
                self._forward()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        # camera-sharded frames capture their RCCL collectives too (the process group's stream joins the capture
        # through the events torch records around every collective); the group's watchdog thread polls events
        # meanwhile, which only the thread-local capture mode tolerates
        kw = {"capture_error_mode": "thread_local"} if self.gather is not None else {}
        if self.gather is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            # ... and even then NOT a query of the end event of a collective issued BEFORE the capture once the group's
            # stream has joined it ("operation not permitted when stream is capturing", raised on the watchdog thread =
            # process abort).  Collectives issued during a capture are not handed to the watchdog; the warm-up's are:
            # let them finish and give the watchdog (100 ms period) time to retire them before the capture begins.
            torch.cuda.synchronize()
            time.sleep(0.35)
        with torch.cuda.graph(graph, **kw):
            bev, cls, crd = self._forward()
            self.prev_bev.copy_(bev)     # state update is part of the graph
        self._graphs[self._use] = (graph, (cls, crd))

    def step_raw(self, raw_images, can_bus, lidar2img, scene_token):
        """Frame from RAW camera images [6, H0, W0, 3] (uint8 or fp32, BGR, on the device): the
        reference's NormalizeMultiviewImage + PadMultiViewImage(32) + format bundle
        (configs/bevformer/bevformer_base.py:11,228-231) run as one HIP pass straight into the frame's
        static input buffer, then `step`."""
        fn = getattr(self.model.ops, "image_normalize_pad", None)
        if fn is None:
            raise RuntimeError("the operator set has no image_normalize_pad")
        buf = self._in["image"][0]
        fn(raw_images, dtype=buf.dtype, out=buf)
        return self.step(buf[None], can_bus, lidar2img, scene_token)

    def step(self, image, can_bus, lidar2img, scene_token):
        can_bus = can_bus.clone().float()
        use_prev = 0.0 if scene_token != self.prev["scene"] else 1.0          # evaluate_trt.py:86-88
        pos, angle = can_bus[:3].clone(), can_bus[-1].clone()
        if use_prev:
            can_bus[:3] -= self.prev["pos"]                                     # :92-98
            can_bus[-1] -= self.prev["angle"]
        else:
            can_bus[:3] = 0
            can_bus[-1] = 0
        self.prev.update(scene=scene_token, pos=pos, angle=angle)
        i = self._in
        if image.data_ptr() != i["image"].data_ptr():      # (the caller may have filled the static buffer itself)
            i["image"].copy_(image, non_blocking=True)
        m = self.model
        grid_length = ((PC_RANGE[4] - PC_RANGE[1]) / m.bev_h, (PC_RANGE[3] - PC_RANGE[0]) / m.bev_w)
        can_host = can_bus.cpu()
        self._host_small[:18] = can_host
        self._host_small[18:20] = G.bev_shift(can_host, m.bev_h, m.bev_w, grid_length)[0]
        # this frame's calibration: no cache, no comparison -- nuScenes matrices differ on every frame (ego motion
        # between the camera and lidar timestamps; the reference's loop builds a fresh tensor per frame,
        # tools/bevformer/evaluate_pth.py:93).  A host tensor rides the one upload below; a device tensor is copied
        # device-to-device behind it (no host synchronisation either way).
        on_host = lidar2img.device.type == "cpu"
        if on_host:
            self._host_small[20:] = lidar2img.detach().reshape(-1).to(torch.float32)
        i["small"].copy_(self._host_small)                  # one upload (pageable source: staged before the call returns)
        if not on_host:
            i["lidar2img"].copy_(lidar2img.detach().reshape(i["lidar2img"].shape), non_blocking=True)
        if not _R3["enabled"]:
            i["use"].fill_(use_prev)
        self._use = use_prev
        if self.use_graph:
            if self._graph is None:
                saved = self.prev_bev.clone()
                self._capture()
                self.prev_bev.copy_(saved)   # capture/warm-up ran the model on scratch state
            graph, outs = self._graphs[self._use]
            graph.replay()
            # the capture's output buffers are overwritten by the next replay: hand out copies
            # (2 x 54 000 values), as the eager path hands out fresh tensors -- unless the caller asked for the
            # buffers themselves
            return tuple(t.clone() for t in outs) if self.clone_outputs else outs
        bev_embed, cls, crd = self._forward()
        self.prev_bev = bev_embed                                               # stays on device (:144)
        return cls, crd
