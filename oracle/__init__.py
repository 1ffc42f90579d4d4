"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

May be imported from `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg, and nowhere else.  The product package
(`bevformer_tensorrt_amd`) never imports this module; it fails loudly when its
HIP library is missing instead of falling back to anything here.

`liboracle.so` is plain C (gcc, OpenMP) built by `make -C oracle`
(`__graft_entry__.build()` does that).  Numpy in, numpy out.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith("_ref.c")]
    stale = (not os.path.exists(so)) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _msda_dims(value, shapes, ref, off, logit):
    bs, nk, heads, C = value.shape
    L = shapes.shape[0]
    nq = off.shape[1]
    ppg = ref.shape[-1] // 2
    P = logit.shape[-1] // L
    assert off.shape[-1] == L * P * 2 and int((shapes[:, 0] * shapes[:, 1]).sum()) == nk
    return [ctypes.c_int(int(x)) for x in (bs, nk, heads, C, L, nq, P, ppg)]


def msda_f32(value, shapes, ref, off, logit):
    """fp32 MSDA (msda_ref.c:oracle_msda_f32).  Accepts any float dtype; upcasts
    to fp32 exactly like the eager path does for fp16 inputs
    (det2trt/models/functions/multi_scale_deformable_attn.py:96-101)."""
    value, ref, off, logit = (_c(x, np.float32) for x in (value, ref, off, logit))
    shapes = _c(shapes, np.int32)
    out = np.empty(off.shape[:3] + (value.shape[-1],), np.float32)
    lib().oracle_msda_f32(_p(value), _p(shapes), _p(ref), _p(off), _p(logit), _p(out),
                          *_msda_dims(value, shapes, ref, off, logit))
    return out


def msda_s8(value, s_v, shapes, ref, off, s_o, logit, s_w, s_out, u8_weights=False):
    """int8 MSDA; `u8_weights=False`: <float> flavour (kernel.cu:848-955),
    True: <__half2> flavour (kernel.cu:957-1104)."""
    value, off, logit = (_c(x, np.int8) for x in (value, off, logit))
    ref = _c(ref, np.float32)
    shapes = _c(shapes, np.int32)
    out = np.empty(off.shape[:3] + (value.shape[-1],), np.int8)
    fn = lib().oracle_msda_s8_u8w if u8_weights else lib().oracle_msda_s8
    f = ctypes.c_float
    fn(_p(value), f(s_v), _p(shapes), _p(ref), _p(off), f(s_o), _p(logit), f(s_w),
       _p(out), f(s_out), *_msda_dims(value, shapes, ref, off, logit))
    return out


def grid_sampler(inp, grid, interp, pad, align):
    """sampler_ref.c: oracle_grid_sampler_2d / _3d (fp32).  grid is channel-first in
    [-10, 10] units like the reference op (functions/grid_sampler.py:140-236)."""
    inp, grid = _c(inp, np.float32), _c(grid, np.float32)
    i = ctypes.c_int
    if grid.ndim == 4:
        N, C, H, W = inp.shape
        Ho, Wo = grid.shape[2:]
        out = np.empty((N, C, Ho, Wo), np.float32)
        lib().oracle_grid_sampler_2d(_p(inp), _p(grid), _p(out), i(N), i(C), i(H), i(W), i(Ho),
                                     i(Wo), i(interp), i(pad), i(int(align)))
    else:
        N, C, D, H, W = inp.shape
        Do, Ho, Wo = grid.shape[2:]
        out = np.empty((N, C, Do, Ho, Wo), np.float32)
        lib().oracle_grid_sampler_3d(_p(inp), _p(grid), _p(out), i(N), i(C), i(D), i(H), i(W),
                                     i(Do), i(Ho), i(Wo), i(interp), i(pad), i(int(align)))
    return out


def rotate(img, angle, center, interp):
    """sampler_ref.c: oracle_rotate (functions/rotate.py:12-80).  interp 0=bilinear 1=nearest."""
    img = _c(img, np.float32)
    C, H, W = img.shape
    out = np.empty_like(img)
    f, i = ctypes.c_float, ctypes.c_int
    lib().oracle_rotate(_p(img), f(float(angle)), f(float(center[0])), f(float(center[1])),
                        _p(out), i(C), i(H), i(W), i(interp))
    return out


def grid_sampler_s8(inp, grid, interp, pad, align, s_in, s_grid, s_out):
    inp, grid = _c(inp, np.int8), _c(grid, np.int8)
    N, C, H, W = inp.shape
    Ho, Wo = grid.shape[2:]
    out = np.empty((N, C, Ho, Wo), np.int8)
    i, f = ctypes.c_int, ctypes.c_float
    lib().oracle_grid_sampler_2d_s8(_p(inp), _p(grid), _p(out), i(N), i(C), i(H), i(W), i(Ho), i(Wo),
                                    i(interp), i(pad), i(int(align)), f(s_in), f(s_grid), f(s_out))
    return out


def rotate_s8(img, angle, center, interp, s_in, s_out):
    img = _c(img, np.int8)
    C, H, W = img.shape
    out = np.empty_like(img)
    f, i = ctypes.c_float, ctypes.c_int
    lib().oracle_rotate_s8(_p(img), f(float(angle)), f(float(center[0])), f(float(center[1])),
                           _p(out), i(C), i(H), i(W), i(interp), f(s_in), f(s_out))
    return out


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                interval_lengths, out_height, out_width, scale_io=None):
    """bev_pool_ref.c.  depth [N,D,H,W], feat [N,H,W,C] -> [1,out_h,out_w,C].
    int8 when scale_io (= s_depth*s_feat/s_out) is given."""
    r = [_c(x, np.int32) for x in (ranks_depth, ranks_feat, ranks_bev, interval_starts,
                                   interval_lengths)]
    c = feat.shape[-1]
    n_out = out_height * out_width * c
    i = ctypes.c_int
    if scale_io is None:
        depth, feat = _c(depth, np.float32), _c(feat, np.float32)
        out = np.empty((1, out_height, out_width, c), np.float32)
        lib().oracle_bev_pool_v2_f32(_p(depth), _p(feat), *[_p(x) for x in r], _p(out), i(c),
                                     i(len(r[3])), ctypes.c_long(n_out))
    else:
        depth, feat = _c(depth, np.int8), _c(feat, np.int8)
        out = np.empty((1, out_height, out_width, c), np.int8)
        lib().oracle_bev_pool_v2_s8(_p(depth), _p(feat), *[_p(x) for x in r], _p(out), i(c),
                                    i(len(r[3])), ctypes.c_long(n_out), ctypes.c_float(scale_io))
    return out


def mdconv(x, offset, mask, weight, bias, stride, padding, dilation, groups, deform_groups):
    """mdconv_ref.c: DCNv2 forward (fp32).  stride/padding/dilation are (h, w) pairs."""
    x, offset, mask, weight = (_c(a, np.float32) for a in (x, offset, mask, weight))
    B, Cin, H, W = x.shape
    Cout, _, Kh, Kw = weight.shape
    Ho = (H + 2 * padding[0] - (dilation[0] * (Kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (Kw - 1) + 1)) // stride[1] + 1
    assert offset.shape == (B, deform_groups * 2 * Kh * Kw, Ho, Wo), offset.shape
    assert mask.shape == (B, deform_groups * Kh * Kw, Ho, Wo), mask.shape
    out = np.empty((B, Cout, Ho, Wo), np.float32)
    b = _c(bias, np.float32) if bias is not None else None
    i = ctypes.c_int
    lib().oracle_mdconv_f32(_p(x), _p(offset), _p(mask), _p(weight), _p(b) if b is not None else None,
                            _p(out), i(B), i(Cin), i(H), i(W), i(Cout), i(Kh), i(Kw), i(stride[0]),
                            i(stride[1]), i(padding[0]), i(padding[1]), i(dilation[0]), i(dilation[1]),
                            i(groups), i(deform_groups))
    return out


def mdconv_s8(x, s_in, offset, s_off, mask, s_mask, weight, s_w, bias, s_out, stride, padding, dilation,
              groups, deform_groups):
    """mdconv_ref.c: INT8 DCNv2 forward (modulatedDeformableConv2dKernel.cu:190-257,463-607)."""
    x, offset, mask, weight = (_c(a, np.int8) for a in (x, offset, mask, weight))
    B, Cin, H, W = x.shape
    Cout, _, Kh, Kw = weight.shape
    Ho = (H + 2 * padding[0] - (dilation[0] * (Kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (Kw - 1) + 1)) // stride[1] + 1
    out = np.empty((B, Cout, Ho, Wo), np.int8)
    b = _c(bias, np.float32) if bias is not None else None
    i, f = ctypes.c_int, ctypes.c_float
    lib().oracle_mdconv_s8(_p(x), f(s_in), _p(offset), f(s_off), _p(mask), f(s_mask), _p(weight), f(s_w),
                           _p(b) if b is not None else None, _p(out), f(s_out), i(B), i(Cin), i(H), i(W),
                           i(Cout), i(Kh), i(Kw), i(stride[0]), i(stride[1]), i(padding[0]), i(padding[1]),
                           i(dilation[0]), i(dilation[1]), i(groups), i(deform_groups))
    return out
