"""Operators of the INT8 engine's int8 ACTIVATION CHAIN (not reference plugins: TensorRT builds these layers itself
from the reference's `Conv2dQ` / `LinearQ` modules, det2trt/models/utils/register.py:78-84, and from the INT8
flavour of its DCNv2 plugin, modulatedDeformableConv2dKernel.cu:463-607,897-978).  Every tensor between two layers
of a ResNet bottleneck is int8 with ONE per-tensor scale: a producer's epilogue requantises with its consumer's
calibrated input scale, the identity rows of a block are int8 with their own scale, so a layer moves one byte per
element in and one out.  Channels-last throughout: a 4-D tensor here is NCHW-shaped in torch.channels_last memory
format (= [B, H, W, C] in memory), int8 or fp16."""
import torch

from ..utils import lib as _lib
from ..utils import workspace as _ws
from .multi_scale_deformable_attn import _TensorCache


import os

# The persistent int8 GEMM of csrc/tsgemm.hip (bevops_tsgemm_s8; domain N % 256 == 0, K % 128 == 0).  Measured under
# graph replay against the tiled int8 GEMM (profiles/r04/tsgemm_s8_ab.jsonl): faster on the 256-column layers with a
# long K (ResNet stage-3 conv1, 34 800 x 256 x 1 024: 21.5 vs 24.6 us; small: 15.6 vs 17.8), slower wherever N > 256
# (every 256-column chunk re-reads the activation rows).  "enabled": None = that policy (BEVOPS_TSGEMM_S8 unset),
# True = wherever legal (tests, tools), False = never.
_TS_S8 = {"enabled": {"1": True, "0": False}.get(os.environ.get("BEVOPS_TSGEMM_S8", ""), None)}


def _ts_s8_pays(M, N, K):
    return N == 256 and K == 1024 and M >= 16384


def _code(dtype):
    return {torch.float16: _lib.F16, torch.int8: _lib.I8}[dtype]


def linear_int8_chain(a, scale_a, w_q, scale_w, bias=None, residual=None, scale_res=1.0, relu=False,
                      out_dtype=torch.float16, scale_out=1.0):
    """act((a . w_q^T) * scale_a * scale_w + bias + residual) with `a` [..., K] int8 (already quantised with
    scale_a) or fp16 (quantised inside the operand load), w_q [N, K] int8, scale_w a float or an fp32 [N] tensor,
    bias fp32 [N], residual [..., N] fp16 -- or int8 with scale_res (int8 `a` only) --, output fp16 or int8
    requantised with scale_out (bevops_linear_int8_chain)."""
    assert a.is_cuda and a.dtype in (torch.int8, torch.float16) and w_q.dtype == torch.int8
    K, N = a.shape[-1], w_q.shape[0]
    a2 = a.reshape(-1, K)
    if not a2.is_contiguous():
        a2 = a2.contiguous()
    w_q = w_q.contiguous()
    M = a2.shape[0]
    per_channel = torch.is_tensor(scale_w)
    ws = scale_w.float().contiguous() if per_channel else None
    b = bias.float().contiguous() if bias is not None else None
    r = None
    if residual is not None:
        assert residual.dtype in (torch.int8, torch.float16) and residual.numel() == M * N
        r = residual.reshape(M, N)
        if not r.is_contiguous():
            r = r.contiguous()
    out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    if M == 0:
        return out.view(*a.shape[:-1], N)
    handle = _lib.load_library()
    if a2.dtype == torch.int8 and N % 256 == 0 and K % 128 == 0 and \
            (_TS_S8["enabled"] or (_TS_S8["enabled"] is None and _ts_s8_pays(M, N, K))):
        with torch.cuda.device(a.device):
            st = handle.bevops_tsgemm_s8(
                a2.data_ptr(), float(scale_a), w_q.data_ptr(), ws.data_ptr() if per_channel else None,
                1.0 if per_channel else float(scale_w), b.data_ptr() if b is not None else None,
                r.data_ptr() if r is not None else None, _code(r.dtype) if r is not None else _lib.F16, float(scale_res),
                _code(out_dtype), out.data_ptr(), float(scale_out), M, N, K, int(bool(relu)),
                _lib.current_stream_ptr(a.device))
        _lib.check(st, "bevops_tsgemm_s8")
        return out.view(*a.shape[:-1], N)
    with torch.cuda.device(a.device):
        st = handle.bevops_linear_int8_chain(
            a2.data_ptr(), _code(a2.dtype), float(scale_a), w_q.data_ptr(), ws.data_ptr() if per_channel else None,
            1.0 if per_channel else float(scale_w), b.data_ptr() if b is not None else None,
            r.data_ptr() if r is not None else None, _code(r.dtype) if r is not None else _lib.F16, float(scale_res),
            _code(out_dtype), out.data_ptr(), float(scale_out), M, N, K, int(bool(relu)),
            _lib.current_stream_ptr(a.device))
    _lib.check(st, "bevops_linear_int8_chain")
    return out.view(*a.shape[:-1], N)


def conv_int8_chain_nhwc(x_q, scale_a, w_q_taps, scale_w, bias=None, relu=False, stride=1, out_dtype=torch.float16,
                         scale_out=1.0):
    """k x k (k in {1, 3}, pad k // 2) convolution of an int8 channels-last activation x_q [B, Cin, H, W] as an
    implicit int8 GEMM (bevops_conv_tile_int8): w_q_taps [Cout, k, k, Cin] int8 (taps-major), scale_w a float or an
    fp32 [Cout] tensor, bias fp32 -> act(conv + bias) as fp16 or as int8 requantised with scale_out, channels-last.
    Cin % 64 == 0."""
    assert x_q.is_cuda and x_q.dtype == torch.int8 and x_q.dim() == 4 and w_q_taps.dtype == torch.int8
    assert x_q.is_contiguous(memory_format=torch.channels_last) and w_q_taps.is_contiguous()
    B, Cin, H, W = x_q.shape
    Cout, k = w_q_taps.shape[0], w_q_taps.shape[1]
    assert w_q_taps.shape == (Cout, k, k, Cin)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    out = torch.empty((B, Cout, Ho, Wo), dtype=out_dtype, device=x_q.device, memory_format=torch.channels_last)
    per_channel = torch.is_tensor(scale_w)
    ws = scale_w.float().contiguous() if per_channel else None
    b = bias.float().contiguous() if bias is not None else None
    if B == 0:
        return out
    handle = _lib.load_library()
    with torch.cuda.device(x_q.device):
        st = handle.bevops_conv_tile_int8(
            x_q.data_ptr(), float(scale_a), w_q_taps.data_ptr(), ws.data_ptr() if per_channel else None,
            1.0 if per_channel else float(scale_w), b.data_ptr() if b is not None else None, None, _code(out_dtype),
            out.data_ptr(), float(scale_out), B, H, W, Cin, Cout, k, int(stride), int(bool(relu)),
            _lib.current_stream_ptr(x_q.device))
    _lib.check(st, "bevops_conv_tile_int8")
    return out


def bias_relu_maxpool_nhwc_int8(x, bias, scale_out):
    """max_pool2d(relu(x + bias), 3, 2, 1) of a channels-last fp16 activation, leaving as int8 quantised with
    scale_out (bevops_bias_relu_maxpool_nhwc_int8): the stem epilogue that starts the int8 chain."""
    assert x.is_cuda and x.dtype == torch.float16 and x.dim() == 4
    assert x.is_contiguous(memory_format=torch.channels_last)
    n, c, h, w = x.shape
    out = torch.empty((n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=torch.int8, device=x.device,
                      memory_format=torch.channels_last)
    if bias is not None:
        bias = bias.to(torch.float16).contiguous()
    if n == 0:
        return out
    handle = _lib.load_library()
    with torch.cuda.device(x.device):
        st = handle.bevops_bias_relu_maxpool_nhwc_int8(_lib.F16, x.data_ptr(),
                                                       bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                                       float(scale_out), n, h, w, c, _lib.current_stream_ptr(x.device))
    _lib.check(st, "bevops_bias_relu_maxpool_nhwc_int8")
    return out


_PACKED_S8 = _TensorCache()     # int8 DCNv2 weight [Cout, Cin / groups, Kh, Kw] -> its [Cout][tap][Cin / groups] image


def modulated_deformable_conv2d_int8_nhwc(x_q, scale_in, offset_mask_nhwc, scale_offset, scale_mask, weight_q,
                                          scale_weight, bias, scale_out, relu=False, stride=1, padding=1, dilation=1,
                                          groups=1, deform_groups=1, exact=False):
    """The DCNv2 block of the int8 chain (bevops_mdconv_forward_int8_nhwc): x_q int8 channels-last [B, Cin, H, W],
    offset_mask_nhwc the raw fp16 channels-last output [B, OC >= 3 K K, Ho, Wo] of the pack's offset convolution
    (2 K K offsets, then K K mask logits -- quantised with scale_offset / scale_mask inside the kernel, the
    sigmoid fused), weight_q int8 [Cout, Cin / groups, K, K] (packed once per tensor), bias fp32 -> int8
    channels-last [B, Cout, Ho, Wo] quantised with scale_out, ReLU folded in.  exact=True: the arithmetic is the INT8
    plugin's (modulatedDeformableConv2dKernel.cu:463-607) on those int8 operands, bit for bit; exact=False (default):
    the mask is folded into the quantised area weights, one requantisation per column element instead of two."""
    assert x_q.is_cuda and x_q.dtype == torch.int8 and x_q.dim() == 4 and weight_q.dtype == torch.int8
    assert x_q.is_contiguous(memory_format=torch.channels_last)
    assert offset_mask_nhwc.dtype == torch.float16 and offset_mask_nhwc.is_contiguous(memory_format=torch.channels_last)
    handle = _lib.load_library()
    B, Cin, H, W = x_q.shape
    Cout, cin_g, Kh, Kw = weight_q.shape
    Ho = (H + 2 * padding - (dilation * (Kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (Kw - 1) + 1)) // stride + 1
    assert tuple(offset_mask_nhwc.shape[0:1] + offset_mask_nhwc.shape[2:]) == (B, Ho, Wo)
    packed = _PACKED_S8.get(weight_q)
    if packed is None:
        nbytes = handle.bevops_mdconv_packed_weight_size(_lib.I8, Cout, cin_g, Kh, Kw)
        packed = torch.empty(nbytes, dtype=torch.uint8, device=weight_q.device)
        wc = weight_q.contiguous()
        with torch.cuda.device(weight_q.device):
            st = handle.bevops_mdconv_pack_weight(_lib.I8, wc.data_ptr(), packed.data_ptr(), Cout, cin_g, Kh, Kw,
                                                  _lib.current_stream_ptr(weight_q.device))
        _lib.check(st, "bevops_mdconv_pack_weight")
        packed = _PACKED_S8.put(weight_q, packed)
    out = torch.empty((B, Cout, Ho, Wo), dtype=torch.int8, device=x_q.device, memory_format=torch.channels_last)
    if B == 0:
        return out
    b = bias.float().contiguous() if bias is not None else None
    stream = _lib.current_stream_ptr(x_q.device)
    ws_bytes = handle.bevops_mdconv_int8_nhwc_workspace_size()
    ws = _ws.lend("mdconv_int8_nhwc", ws_bytes, x_q.device, stream)
    with torch.cuda.device(x_q.device):
        st = handle.bevops_mdconv_forward_int8_nhwc(
            x_q.data_ptr(), float(scale_in), offset_mask_nhwc.data_ptr(), int(offset_mask_nhwc.shape[1]),
            float(scale_offset), float(scale_mask), packed.data_ptr(), float(scale_weight),
            b.data_ptr() if b is not None else None, out.data_ptr(), float(scale_out), int(bool(relu)), int(bool(exact)),
            ws.data_ptr(),
            ws_bytes, B, Cin, H, W, Cout, Kh, Kw, stride, stride, padding, padding, dilation, dilation, groups,
            deform_groups, stream)
    _lib.check(st, "bevops_mdconv_forward_int8_nhwc")
    return out
