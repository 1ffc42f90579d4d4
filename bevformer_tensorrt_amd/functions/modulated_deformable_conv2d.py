"""modulated_deformable_conv2d / modulated_deformable_conv2d2 -- drop-in for
det2trt/models/functions/modulated_deformable_conv2d.py:205-291 (DCNv2 forward)."""
import torch
from torch.nn.modules.utils import _pair

from ..utils import lib as _lib
from .multi_scale_deformable_attn import _TensorCache

# weight tensor -> its [Cout][tap][Cin/groups] image (made once per weight version; inference
# calls the op with the same parameters frame after frame)
_PACKED = _TensorCache()


def _packed_weight(handle, weight, dt):
    hit = _PACKED.get(weight)
    if hit is not None:
        return hit
    Cout, cin_g, Kh, Kw = weight.shape
    nbytes = handle.bevops_mdconv_packed_weight_size(dt, Cout, cin_g, Kh, Kw)
    packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    with torch.cuda.device(weight.device):
        st = handle.bevops_mdconv_pack_weight(dt, weight.data_ptr(), packed.data_ptr(), Cout, cin_g, Kh, Kw,
                                              _lib.current_stream_ptr(weight.device))
    _lib.check(st, "bevops_mdconv_pack_weight")
    return _PACKED.put(weight, packed)


def _mdconv(input, offset, mask, weight, bias, stride, padding, dilation, groups, deform_groups):
    assert input.is_cuda, "modulated_deformable_conv2d: input must be on the GPU"
    if input.dim() != 4:
        raise ValueError(f"Expected 4D tensor as input, got {input.dim()}D tensor instead.")
    handle = _lib.load_library()
    # dtype follows `offset`, as the reference does (:73-75)
    input = input.type_as(offset).contiguous()
    w_src = weight
    weight = weight.type_as(input).contiguous()
    mask = mask.type_as(input).contiguous()
    offset = offset.contiguous()
    if bias is not None:
        bias = bias.type_as(input).contiguous()
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    B, Cin, H, W = input.shape
    Cout, _, Kh, Kw = weight.shape
    Ho = (H + 2 * ph - (dh * (Kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (Kw - 1) + 1)) // sw + 1
    if tuple(offset.shape) != (B, deform_groups * 2 * Kh * Kw, Ho, Wo):
        raise ValueError(f"offset shape {tuple(offset.shape)} != {(B, deform_groups * 2 * Kh * Kw, Ho, Wo)}")
    if tuple(mask.shape) != (B, deform_groups * Kh * Kw, Ho, Wo):
        raise ValueError(f"mask shape {tuple(mask.shape)} != {(B, deform_groups * Kh * Kw, Ho, Wo)}")
    dt = _lib.torch_dtype_code(input)
    if B == 0:      # empty batch: nothing to convolve
        return input.new_empty((0, Cout, max(Ho, 0), max(Wo, 0)))
    dims = (B, Cin, H, W, Cout, Kh, Kw, sh, sw, ph, pw, dh, dw, groups, deform_groups)
    ws_bytes = handle.bevops_mdconv_workspace_size(dt, *dims)
    if ws_bytes == 0:
        raise _lib.BevopsError("bevops_mdconv_workspace_size: unsupported arguments", _lib.NOT_SUPPORTED)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=input.device)
    out = torch.empty((B, Cout, Ho, Wo), dtype=input.dtype, device=input.device)
    # the cache is keyed on the caller's tensor object; a converted / re-laid-out temporary
    # (weight is not w_src) would never hit, so it takes the per-call path
    packed = None
    if weight is w_src:
        packed = _PACKED.get(weight)
        if packed is None and not torch.cuda.is_current_stream_capturing():
            packed = _packed_weight(handle, weight, dt)
    with torch.cuda.device(input.device):
        fn = handle.bevops_mdconv_forward_packed if packed is not None else handle.bevops_mdconv_forward
        st = fn(dt, input.data_ptr(), offset.data_ptr(), mask.data_ptr(),
                packed.data_ptr() if packed is not None else weight.data_ptr(),
                bias.data_ptr() if bias is not None else None, out.data_ptr(), ws.data_ptr(), ws_bytes,
                *dims, _lib.current_stream_ptr(input.device))
    _lib.check(st, "bevops_mdconv_forward")
    return out


def modulated_deformable_conv2d(input, offset, mask, weight, bias=None, stride=1, padding=0,
                                dilation=1, groups=1, deform_groups=1):
    """DCNv2 forward (plugin ModulatedDeformableConv2dTRT: fp32, fp16).  input [B,Cin,H,W],
    offset [B, dg*2*K*K, Ho, Wo], mask [B, dg*K*K, Ho, Wo], weight [Cout, Cin/groups, K, K]."""
    return _mdconv(input, offset, mask, weight, bias, stride, padding, dilation, groups, deform_groups)


def modulated_deformable_conv2d2(input, offset, mask, weight, bias=None, stride=1, padding=0,
                                 dilation=1, groups=1, deform_groups=1):
    """Same op under the half2 plugin name ModulatedDeformableConv2dTRT2."""
    return _mdconv(input, offset, mask, weight, bias, stride, padding, dilation, groups, deform_groups)


def modulated_deformable_conv2d_int8(input, offset, mask, weight, bias, scale_in, scale_offset, scale_mask,
                                     scale_weight, scale_out, stride=1, padding=0, dilation=1, groups=1,
                                     deform_groups=1):
    """INT8 flavour (modulatedDeformableConv2dKernel.cu:463-607): int8 input / offset / mask / weight
    with per-tensor scales, fp32 bias, int8 output."""
    assert input.is_cuda and input.dtype == torch.int8
    handle = _lib.load_library()
    input, offset, mask, weight = (t.contiguous() for t in (input, offset, mask, weight))
    for name, t in (("offset", offset), ("mask", mask), ("weight", weight)):
        if t.dtype != torch.int8:
            raise TypeError(f"{name} must be int8")
    if bias is not None:
        bias = bias.float().contiguous()
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    B, Cin, H, W = input.shape
    Cout, _, Kh, Kw = weight.shape
    Ho = (H + 2 * ph - (dh * (Kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (Kw - 1) + 1)) // sw + 1
    dims = (B, Cin, H, W, Cout, Kh, Kw, sh, sw, ph, pw, dh, dw, groups, deform_groups)
    ws_bytes = handle.bevops_mdconv_workspace_size(_lib.I8, *dims)
    if ws_bytes == 0:
        raise _lib.BevopsError("bevops_mdconv_workspace_size: unsupported arguments", _lib.NOT_SUPPORTED)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=input.device)
    out = torch.empty((B, Cout, Ho, Wo), dtype=torch.int8, device=input.device)
    # re-laid-out once per weight tensor (version); a first sighting under stream capture takes the per-call path
    packed = _PACKED.get(weight)
    if packed is None and not torch.cuda.is_current_stream_capturing():
        packed = _packed_weight(handle, weight, _lib.I8)
    with torch.cuda.device(input.device):
        fn = handle.bevops_mdconv_forward_int8_packed if packed is not None else handle.bevops_mdconv_forward_int8
        st = fn(input.data_ptr(), float(scale_in), offset.data_ptr(), float(scale_offset), mask.data_ptr(),
                float(scale_mask), packed.data_ptr() if packed is not None else weight.data_ptr(), float(scale_weight),
                bias.data_ptr() if bias is not None else None, out.data_ptr(), float(scale_out),
                ws.data_ptr(), ws_bytes, *dims, _lib.current_stream_ptr(input.device))
    _lib.check(st, "bevops_mdconv_forward_int8")
    return out


def modulated_deformable_conv2d_nhwc(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1,
                                     groups=1, deform_groups=1, relu=False, offset_mask_nhwc=None):
    """Channels-last DCNv2 for the re-hosted backbone (not a reference plugin): `input` is an
    NCHW-shaped tensor in torch.channels_last memory format (= [B, H, W, C] in memory), the
    result likewise; offset / mask keep the reference's planar layout; optional fused ReLU.
    fp16, fused-kernel domain only (raises BevopsError otherwise).
    offset_mask_nhwc: instead of (offset, mask), the raw channels-last output [B, OC, Ho, Wo] of
    the pack's offset convolution (2*KK offsets then KK mask logits); the sigmoid is fused."""
    assert input.is_cuda and input.dtype == torch.float16
    handle = _lib.load_library()
    if not input.is_contiguous(memory_format=torch.channels_last):
        input = input.contiguous(memory_format=torch.channels_last)
    om_ch = 0
    if offset_mask_nhwc is not None:
        assert offset_mask_nhwc.dtype == torch.float16
        if not offset_mask_nhwc.is_contiguous(memory_format=torch.channels_last):
            offset_mask_nhwc = offset_mask_nhwc.contiguous(memory_format=torch.channels_last)
        om_ch = offset_mask_nhwc.shape[1]
        offset, mask = offset_mask_nhwc, None
    else:
        offset, mask = offset.to(input.dtype).contiguous(), mask.to(input.dtype).contiguous()
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    B, Cin, H, W = input.shape
    Cout, _, Kh, Kw = weight.shape
    Ho = (H + 2 * ph - (dh * (Kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (Kw - 1) + 1)) // sw + 1
    dims = (B, Cin, H, W, Cout, Kh, Kw, sh, sw, ph, pw, dh, dw, groups, deform_groups)
    if weight.dtype != torch.float16 or not weight.is_contiguous():
        raise _lib.BevopsError("modulated_deformable_conv2d_nhwc: weight must be a contiguous fp16 tensor", _lib.BAD_PARAM)
    packed = _PACKED.get(weight)
    if packed is None:
        packed = _packed_weight(handle, weight, _lib.F16)
    ws_bytes = handle.bevops_mdconv_workspace_size(_lib.F16, *dims)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=input.device)
    out = torch.empty((B, Cout, Ho, Wo), dtype=input.dtype, device=input.device,
                      memory_format=torch.channels_last)
    if bias is not None:
        bias = bias.to(input.dtype).contiguous()
    if B == 0:          # a rank of the camera-sharded path that owns no camera
        return out
    with torch.cuda.device(input.device):
        st = handle.bevops_mdconv_forward_nhwc(
            _lib.F16, input.data_ptr(), offset.data_ptr(), mask.data_ptr() if mask is not None else None,
            packed.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), int(relu), om_ch,
            ws.data_ptr(), ws_bytes,
            *dims, _lib.current_stream_ptr(input.device))
    _lib.check(st, "bevops_mdconv_forward_nhwc")
    return out


def bias_act_nhwc_(x, bias=None, residual=None, relu=False):
    """In place on a channels-last activation (or any [rows, C] row-major tensor):
    x += bias (+ residual); optional ReLU -- one pass (bevops_bias_act_nhwc)."""
    assert x.is_cuda and x.dtype == torch.float16
    if x.dim() == 4:
        assert x.is_contiguous(memory_format=torch.channels_last)
        C = x.shape[1]
    else:
        assert x.is_contiguous()
        C = x.shape[-1]
    rows = x.numel() // C
    if rows == 0:
        return x
    if residual is not None:
        assert residual.shape == x.shape and residual.dtype == x.dtype
        assert residual.is_contiguous(memory_format=torch.channels_last) if x.dim() == 4 else residual.is_contiguous()
    handle = _lib.load_library()
    with torch.cuda.device(x.device):
        st = handle.bevops_bias_act_nhwc(_lib.F16, x.data_ptr(), bias.data_ptr() if bias is not None else None,
                                         residual.data_ptr() if residual is not None else None, rows, C,
                                         int(relu), _lib.current_stream_ptr(x.device))
    _lib.check(st, "bevops_bias_act_nhwc")
    return x


def bias_relu_maxpool_nhwc(x, bias=None):
    """max_pool2d(relu(x + bias), 3, 2, 1) on a channels-last fp16 activation in one pass
    (bevops_bias_relu_maxpool_nhwc): the stem epilogue of the re-hosted ResNet."""
    assert x.is_cuda and x.dtype == torch.float16 and x.dim() == 4
    assert x.is_contiguous(memory_format=torch.channels_last)
    n, c, h, w = x.shape
    out = torch.empty((n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=x.dtype, device=x.device,
                      memory_format=torch.channels_last)
    if bias is not None:
        bias = bias.to(torch.float16).contiguous()
    if n == 0:
        return out
    handle = _lib.load_library()
    with torch.cuda.device(x.device):
        st = handle.bevops_bias_relu_maxpool_nhwc(_lib.F16, x.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                  out.data_ptr(), n, h, w, c, _lib.current_stream_ptr(x.device))
    _lib.check(st, "bevops_bias_relu_maxpool_nhwc")
    return out


def upsample_add_nhwc_(a, b):
    """In place: a += nearest-up-sampled b (to a's spatial size), both channels-last fp16 [N, C, H, W] -- the
    FPN top-down step in one pass (bevops_upsample_add_nhwc), bit-equal to `a + F.interpolate(b, size=...)`."""
    assert a.is_cuda and a.dtype == torch.float16 and b.dtype == torch.float16 and a.dim() == 4 and b.dim() == 4
    assert a.is_contiguous(memory_format=torch.channels_last) and b.is_contiguous(memory_format=torch.channels_last)
    assert a.shape[0] == b.shape[0] and a.shape[1] == b.shape[1]
    if a.numel() == 0:
        return a
    handle = _lib.load_library()
    with torch.cuda.device(a.device):
        st = handle.bevops_upsample_add_nhwc(_lib.F16, a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[2], a.shape[3],
                                             b.shape[2], b.shape[3], a.shape[1], _lib.current_stream_ptr(a.device))
    _lib.check(st, "bevops_upsample_add_nhwc")
    return a


def feat_embed_nhwc(src, cam_embed, level_embed, dst):
    """dst[n, r, :] = (src[n, r, :] + cam_embed[n, :]) + level_embed (fp16, two roundings as the two adds of
    transformer.py:146-150).  src [N, rows, C] dense; dst = a [N, rows, C] slice (row range) of the concatenated
    [N, sum rows, C] feature tensor: written in place of the torch.cat copy."""
    assert src.is_cuda and src.dtype == torch.float16 and src.is_contiguous() and src.dim() == 3
    n, rows, c = src.shape
    assert dst.shape == src.shape and dst.dtype == src.dtype and dst.stride(2) == 1 and dst.stride(1) == c
    cam = cam_embed.to(src.dtype).contiguous()
    lvl = level_embed.to(src.dtype).contiguous()
    assert cam.shape == (n, c) and lvl.shape == (c,)
    handle = _lib.load_library()
    with torch.cuda.device(src.device):
        st = handle.bevops_feat_embed_nhwc(_lib.F16, src.data_ptr(), cam.data_ptr(), lvl.data_ptr(), dst.data_ptr(),
                                           n, rows, c, dst.stride(0), _lib.current_stream_ptr(src.device))
    _lib.check(st, "bevops_feat_embed_nhwc")
    return dst


_PACKED_C32 = _TensorCache()


def conv_offset_nhwc(input, weight, bias=None):
    """The DCNv2 pack's offset convolution (cnn/dcn.py:62-70) on a channels-last fp16 activation:
    3x3 / stride 1 / pad 1, weight [Cout <= 32, Cin, 3, 3], bias [Cout] -> [B, 32, H, W] in
    channels_last memory format (channels >= Cout are zero), ready to be the `offset_mask_nhwc`
    operand of modulated_deformable_conv2d_nhwc.  One implicit-GEMM launch with the bias in its
    epilogue (bevops_conv3x3_c32_forward_nhwc).  Raises BevopsError (status 3) for unsupported Cin."""
    assert input.is_cuda and input.dtype == torch.float16 and weight.dtype == torch.float16
    if weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) or weight.shape[0] > 32:
        raise ValueError(f"conv_offset_nhwc: weight {tuple(weight.shape)} is not [<=32, Cin, 3, 3]")
    handle = _lib.load_library()
    if not input.is_contiguous(memory_format=torch.channels_last):
        input = input.contiguous(memory_format=torch.channels_last)
    B, Cin, H, W = input.shape
    Cout = weight.shape[0]
    if weight.shape[1] != Cin:
        raise ValueError("conv_offset_nhwc: weight / input channel mismatch")
    hit = _PACKED_C32.get(weight)
    if hit is None:
        nbytes = handle.bevops_conv3x3_c32_packed_weight_size(_lib.F16, Cin)
        if nbytes == 0:
            raise _lib.BevopsError("bevops_conv3x3_c32_pack_weight: dtype/shape combination not supported (status 3)", _lib.NOT_SUPPORTED)
        packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
        wc = weight.detach().contiguous()
        b32 = torch.zeros(32, dtype=torch.float16, device=weight.device)
        with torch.cuda.device(weight.device):
            st = handle.bevops_conv3x3_c32_pack_weight(_lib.F16, wc.data_ptr(), packed.data_ptr(), Cout, Cin,
                                                       _lib.current_stream_ptr(weight.device))
        _lib.check(st, "bevops_conv3x3_c32_pack_weight")
        hit = _PACKED_C32.put(weight, [packed, b32, None])
    packed, b32, bias_version = hit
    if bias is not None and bias_version != (id(bias), bias._version):   # refresh the padded bias only when it changed
        b32[:Cout].copy_(bias.detach())
        hit[2] = (id(bias), bias._version)
    out = torch.empty((B, 32, H, W), dtype=input.dtype, device=input.device, memory_format=torch.channels_last)
    if B == 0:
        return out
    with torch.cuda.device(input.device):
        st = handle.bevops_conv3x3_c32_forward_nhwc(_lib.F16, input.data_ptr(), packed.data_ptr(),
                                                    b32.data_ptr() if bias is not None else None,
                                                    out.data_ptr(), B, H, W, Cin,
                                                    _lib.current_stream_ptr(input.device))
    _lib.check(st, "bevops_conv3x3_c32_forward_nhwc")
    return out
