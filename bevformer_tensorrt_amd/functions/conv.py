"""Plain 3x3 convolutions of the channels-last backbone / neck (ResNet stages without DCN:
det2trt/models/backbones/resnet.py:106-260; FPN output convolutions: third_party/bev_mmdet3d/models/necks/
fpn.py:140-155) and the stride-2 1x1 convolutions at the head of a ResNet stage on the tiled MFMA GEMM skeleton as
an implicit GEMM (bevops_conv_tile_f16, csrc/tile_gemm.hip): no column buffer, no strided copy, shift + identity +
ReLU in the epilogue.  Not a reference plugin (TensorRT owns these layers
there).  `conv3x3_auto` measures it once per problem against the library convolution + epilogue pass and keeps
the faster one (BLOCKING on its first call per shape, outside stream capture)."""
import os

import torch
import torch.nn.functional as F

from ..utils import lib as _lib
from .modulated_deformable_conv2d import bias_act_nhwc_
from .multi_scale_deformable_attn import _TensorCache

_PACKED = _TensorCache()   # weight tensor -> taps-major copy (weakly keyed: dies with the model that owns the weight)
_CHOICE = {}          # problem -> "tile" | "library"
CONV_LOG = []         # (problem, {name: us})
CONV_MISSES = []   # problems conv3x3_auto met that dispatch_gfx950.json does not list
HALO = {"enabled": os.environ.get("BEVOPS_CONV_HALO", "1") == "1"}   # A/B: the LDS-resident 64-channel convolution


def pack_taps(weight):
    """[Cout, Cin, k, k] -> [Cout, k, k, Cin] contiguous (k = tap-major, channels innermost), cached per weight."""
    hit = _PACKED.get(weight)
    if hit is None:
        hit = _PACKED.put(weight, weight.detach().permute(0, 2, 3, 1).contiguous())
    return hit


def conv_nhwc(x, weight, bias=None, relu=False, residual=None, stride=1):
    """x [B, Cin, H, W] channels-last fp16, weight [Cout, Cin, k, k] with k in {1, 3} -> act(conv2d(x, weight,
    stride, pad k // 2) + bias + residual) [B, Cout, Hout, Wout] channels-last.  Cin % 32 == 0."""
    assert x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and weight.shape[2] == weight.shape[3]
    assert x.is_contiguous(memory_format=torch.channels_last)
    B, Cin, H, W = x.shape
    Cout, k = weight.shape[0], weight.shape[2]
    if weight.shape[1] != Cin:
        raise ValueError("weight does not match the input channels")
    wt = pack_taps(weight)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    out = torch.empty((B, Cout, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == x.dtype
        assert residual.is_contiguous(memory_format=torch.channels_last)
    if bias is not None:
        bias = bias.to(torch.float16).contiguous()
    if B == 0:
        return out
    handle = _lib.load_library()
    with torch.cuda.device(x.device):
        st = handle.bevops_conv_tile_f16(x.data_ptr(), wt.data_ptr(), bias.data_ptr() if bias is not None else None,
                                         residual.data_ptr() if residual is not None else None, out.data_ptr(),
                                         B, H, W, Cin, Cout, k, int(stride), int(bool(relu)),
                                         _lib.current_stream_ptr(x.device))
    _lib.check(st, "bevops_conv_tile_f16")
    return out


def conv3x3_c64(x, weight, bias=None, relu=False, residual=None, stride=1):
    """The 3x3 / stride 1 / pad 1 convolution of a 64 -> 64-channel layer with both operands in LDS
    (bevops_conv3x3_c64_f16, csrc/conv_halo.hip): x [B, 64, H, W] channels-last fp16, weight [64, 64, 3, 3] ->
    act(conv2d(x, weight, 1, 1) + bias), bit-identical to conv_nhwc.  Raises BevopsError(NOT_SUPPORTED) for any other
    layer (channel counts, a stride, identity rows)."""
    assert x.is_cuda and x.dtype == torch.float16 and x.dim() == 4
    assert x.is_contiguous(memory_format=torch.channels_last)
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    if residual is not None or stride != 1 or tuple(weight.shape[1:]) != (Cin, 3, 3):
        raise _lib.BevopsError("bevops_conv3x3_c64_f16: 3x3 / stride 1 layers without identity rows only", _lib.NOT_SUPPORTED)
    wt = pack_taps(weight)
    out = torch.empty((B, Cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if bias is not None:
        bias = bias.to(torch.float16).contiguous()
    if B == 0:
        return out
    handle = _lib.load_library()
    with torch.cuda.device(x.device):
        st = handle.bevops_conv3x3_c64_f16(x.data_ptr(), wt.data_ptr(), bias.data_ptr() if bias is not None else None,
                                           out.data_ptr(), B, H, W, Cin, Cout, int(bool(relu)),
                                           _lib.current_stream_ptr(x.device))
    _lib.check(st, "bevops_conv3x3_c64_f16")
    return out


def conv_int8_nhwc(x, scale_a, w_q_taps, scale_w, bias=None, relu=False, residual=None, stride=1):
    """The INT8 flavour (bevops_conv_tile_int8_fused): x [B, Cin, H, W] channels-last fp16, quantised with scale_a
    inside the kernel; w_q_taps [Cout, k, k, Cin] int8 (taps-major), scale_w a float or an fp32 [Cout] tensor; bias
    fp32; residual fp16 channels-last -> fp16 [B, Cout, Hout, Wout] channels-last.  Cin % 64 == 0."""
    assert x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and w_q_taps.dtype == torch.int8
    assert x.is_contiguous(memory_format=torch.channels_last) and w_q_taps.is_contiguous()
    B, Cin, H, W = x.shape
    Cout, k = w_q_taps.shape[0], w_q_taps.shape[1]
    assert w_q_taps.shape == (Cout, k, k, Cin)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    out = torch.empty((B, Cout, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == x.dtype
        assert residual.is_contiguous(memory_format=torch.channels_last)
    per_channel = torch.is_tensor(scale_w)
    ws = scale_w.float().contiguous() if per_channel else None
    b = bias.float().contiguous() if bias is not None else None
    if B == 0:
        return out
    handle = _lib.load_library()
    with torch.cuda.device(x.device):
        st = handle.bevops_conv_tile_int8_fused(
            x.data_ptr(), float(scale_a), w_q_taps.data_ptr(), ws.data_ptr() if per_channel else None,
            1.0 if per_channel else float(scale_w), b.data_ptr() if b is not None else None,
            residual.data_ptr() if residual is not None else None, out.data_ptr(), B, H, W, Cin, Cout, k, int(stride),
            int(bool(relu)), _lib.current_stream_ptr(x.device))
    _lib.check(st, "bevops_conv_tile_int8_fused")
    return out


def conv3x3_nhwc(x, weight, bias=None, relu=False, residual=None, stride=1):
    assert weight.shape[2:] == (3, 3)
    return conv_nhwc(x, weight, bias, relu, residual, stride)


def _library(x, weight, bias, relu, residual, stride=1):
    y = F.conv2d(x, weight, None, stride, weight.shape[2] // 2)
    if not y.is_contiguous(memory_format=torch.channels_last):
        y = y.contiguous(memory_format=torch.channels_last)
    return bias_act_nhwc_(y, bias, residual, relu)


def conv3x3_auto(x, weight, bias=None, relu=False, residual=None, stride=1):
    """conv_nhwc or the library convolution + one epilogue pass, whichever measured faster for the problem."""
    B, Cin, H, W = x.shape
    if B == 0 and Cin % 32 == 0:
        return conv_nhwc(x, weight, bias, relu, residual, stride)      # (empty result, nothing to measure)
    key = (str(x.device), B, H, W, Cin, weight.shape[0], weight.shape[2], stride, bool(relu), residual is not None)
    from .linear import DETERMINISTIC, _problem, _table
    # 64 -> 64 channels, stride 1, no identity rows (conv2 of the ResNet stage-1 bottlenecks): the LDS-resident kernel
    # (csrc/conv_halo.hip) -- bit-identical to the tiled implicit GEMM, so also the rule-based dispatch may take it
    halo_ok = HALO["enabled"] and Cin == 64 and weight.shape[0] == 64 and stride == 1 and residual is None \
        and weight.shape[2] == 3
    name = ("halo" if halo_ok else "tile") if (DETERMINISTIC["enabled"] and Cin % 32 == 0) else _CHOICE.get(key)
    if name is None:      # shipped choice (dispatch_gfx950.json): no measurement, the same kernel on every box
        name = _table()["conv"].get(_problem(key))
        if name == "tile" and halo_ok:
            # (the table was measured before this kernel existed: 47 against 88 us at the base shape, faster at every
            # shape of the four models, profiles/r06/conv_halo_time.jsonl)
            name = "halo"
        if name in ("tile", "library", "halo") and (name == "library" or Cin % 32 == 0) and (name != "halo" or halo_ok):
            _CHOICE[key] = name
        else:
            name = None
            if _problem(key) not in CONV_MISSES:
                CONV_MISSES.append(_problem(key))   # a problem the shipped table has never seen (bench.py reports them)
    if name is None:
        if Cin % 32 != 0:
            name = _CHOICE[key] = "library"
        elif torch.cuda.is_current_stream_capturing() or os.environ.get("BEVOPS_DENSE_TUNE", "1") == "0":
            name = "library"
        else:
            from .linear import graph_time_us
            times = {}
            for cand, fn in (("tile", conv_nhwc), ("library", _library)) + ((("halo", conv3x3_c64),) if halo_ok else ()):
                for _ in range(2):
                    fn(x, weight, bias, relu, residual, stride)
                torch.cuda.synchronize()
                # (under HIP-graph replay: an eager loop would time the host for the small late-stage convolutions)
                times[cand] = round(graph_time_us(lambda: fn(x, weight, bias, relu, residual, stride), 4, 3), 1)
            CONV_LOG.append((key, times))
            name = _CHOICE[key] = min(times, key=times.get)
    return {"tile": conv_nhwc, "halo": conv3x3_c64, "library": _library}[name](x, weight, bias, relu, residual, stride)


_STEM_PACKED = _TensorCache()   # stem weight -> (bias stamp, packed matrix-core operand image)
STEM_FUSED = {"enabled": os.environ.get("BEVOPS_STEM_FUSED", "1") != "0"}   # A/B: library convolution + pooling pass


def stem_conv_pool(x, weight, bias=None, scale_out=None):
    """The ResNet stem as ONE kernel (bevops_stem_conv_pool): max_pool2d(relu(conv2d(x, weight, bias, stride 2,
    pad 3)), 3, 2, 1) from the PLANAR images x [N, 3, H, W] (fp16, contiguous, W even) and weight [64, 3, 7, 7] to the
    pooled activation [N, 64, Hp, Wp] channels-last -- fp16, or int8 with `scale_out` (q = min(rne(v / scale_out),
    127): the first tensor of the INT8 engine's activation chain).  fp32 accumulation, one rounding.  The
    matrix-core operand image of (weight, bias) is built once per weight tensor (bevops_stem_pack)."""
    assert x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and x.is_contiguous()
    n, c, h, w = x.shape
    if tuple(weight.shape) != (64, 3, 7, 7) or c != 3 or w % 2:
        raise ValueError("stem_conv_pool: a 7x7 convolution from 3 to 64 channels on images of even width")
    handle = _lib.load_library()
    bstamp = None if bias is None else _TensorCache._stamp(bias)
    hit = _STEM_PACKED.get(weight)
    if hit is None or hit[0] != bstamp:
        wt = weight.detach().to(torch.float16).contiguous()
        bs = None if bias is None else bias.detach().to(torch.float16).contiguous()
        packed = torch.empty(handle.bevops_stem_packed_size(), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            st = handle.bevops_stem_pack(_lib.F16, wt.data_ptr(), bs.data_ptr() if bs is not None else None,
                                         packed.data_ptr(), _lib.current_stream_ptr(x.device))
        _lib.check(st, "bevops_stem_pack")
        hit = _STEM_PACKED.put(weight, (bstamp, packed))
    hc, wc = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    hp, wp = (hc - 1) // 2 + 1, (wc - 1) // 2 + 1
    out = torch.empty((n, 64, hp, wp), dtype=torch.float16 if scale_out is None else torch.int8, device=x.device,
                      memory_format=torch.channels_last)
    if n == 0:
        return out
    with torch.cuda.device(x.device):
        st = handle.bevops_stem_conv_pool(_lib.F16, _lib.F16 if scale_out is None else _lib.I8, x.data_ptr(),
                                          hit[1].data_ptr(), out.data_ptr(), n, h, w,
                                          0.0 if scale_out is None else float(scale_out),
                                          _lib.current_stream_ptr(x.device))
    _lib.check(st, "bevops_stem_conv_pool")
    return out

