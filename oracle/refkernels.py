"""oracle/refkernels.py -- TEST INFRASTRUCTURE ONLY.

numpy doors onto `oracle/_ref/libbevref.so`: the REFERENCE's own plugin kernels
(`/root/reference/TensorRT/plugin/*/*Kernel.cu`), compiled for the host by
`make -C oracle` (see `oracle/Makefile`, `oracle/cuda_on_cpu/`) and executed thread by
thread on the CPU.  Used to pin the C restatements in `oracle/*_ref.c` and to generate the
golden fixtures under `tests/golden/ref_kernels_*.npz` (`tests/golden/make_ref_kernel_golden.py`).

Signatures mirror `oracle/__init__.py`; every function takes dense NCHW / reference-layout
numpy arrays and packs them into the TensorRT formats the plugins negotiate
(kLINEAR, kCHW2 for the __half2 kernels, kCHW4 for int8; `supportsFormatCombination` of each
plugin) before calling the reference's host function.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libbevref.so")
_LIB = None

F32, F16, H2 = 0, 1, 2


def available():
    return os.path.exists(_SO)


def lib():
    global _LIB
    if _LIB is None:
        if not available():
            raise RuntimeError("oracle/_ref/libbevref.so is not built (needs /root/reference; make -C oracle)")
        _LIB = ctypes.CDLL(_SO)
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _np_dt(dtype):
    return np.float32 if dtype == F32 else np.float16


def chw_pack(a, k):
    """[..., C, H, W] -> kCHW<k>: [..., ceil(C/k), H, W, k], zero-padded channels."""
    a = np.asarray(a)
    C, H, W = a.shape[-3:]
    Cp = -(-C // k) * k
    if Cp != C:
        pad = np.zeros(a.shape[:-3] + (Cp - C, H, W), a.dtype)
        a = np.concatenate([a, pad], axis=-3)
    a = a.reshape(a.shape[:-3] + (Cp // k, k, H, W))
    return np.ascontiguousarray(np.moveaxis(a, -3, -1))


def chw_unpack(a, C):
    """inverse of chw_pack: [..., C/k, H, W, k] -> [..., C, H, W]"""
    a = np.moveaxis(a, -1, -3)
    a = a.reshape(a.shape[:-4] + (a.shape[-4] * a.shape[-3],) + a.shape[-2:])
    return np.ascontiguousarray(a[..., :C, :, :])


def _msda_dims(value, shapes, ref, off, logit):
    bs, nk, heads, C = value.shape
    L = shapes.shape[0]
    nq = off.shape[1]
    ppg = ref.shape[-1] // 2
    P = logit.shape[-1] // L
    return [ctypes.c_int(int(x)) for x in (bs, nk, heads, C, L, nq, P, ppg)]


def msda(value, shapes, ref, off, logit, dtype=F32):
    """ms_deformable_im2col_cuda<float|__half> / _h2 (multiScaleDeformableAttnKernel.cu:1106-1168)."""
    dt = _np_dt(dtype)
    value, ref, off, logit = (_c(x, dt) for x in (value, ref, off, logit))
    shapes = _c(shapes, np.int32)
    out = np.empty(off.shape[:3] + (value.shape[-1],), dt)
    rc = lib().bevref_msda(ctypes.c_int(dtype), _p(value), _p(shapes), _p(ref), _p(off), _p(logit), _p(out),
                           *_msda_dims(value, shapes, ref, off, logit))
    assert rc == 0
    return out


def msda_s8(value, s_v, shapes, ref, off, s_o, logit, s_w, s_out, ref_half=False):
    """ms_deformable_im2col_cuda_int8<float> (ref_half=False, kernel.cu:848-955) or
    <__half2> (ref_half=True: reference points as fp16 pairs, kernel.cu:957-1104)."""
    value, off, logit = (_c(x, np.int8) for x in (value, off, logit))
    ref = _c(ref, np.float16 if ref_half else np.float32)
    shapes = _c(shapes, np.int32)
    out = np.empty(off.shape[:3] + (value.shape[-1],), np.int8)
    f = ctypes.c_float
    lib().bevref_msda_int8(ctypes.c_int(int(ref_half)), _p(value), f(s_v), _p(shapes), _p(ref), _p(off), f(s_o),
                           _p(logit), f(s_w), _p(out), f(s_out), *_msda_dims(value, shapes, ref, off, logit))
    return out


def rotate(img, angle, center, interp, dtype=F32):
    """rotate<float|__half> / rotate_h2 (rotateKernel.cu:708-733).  img [C,H,W]."""
    dt = _np_dt(dtype)
    C, H, W = img.shape
    ang = np.array([angle], dt)
    cen = _c(center, dt)
    i = ctypes.c_int
    if dtype == H2:
        x = chw_pack(_c(img, dt), 2)
        out = np.empty_like(x)
        lib().bevref_rotate(i(dtype), _p(out), _p(x), _p(ang), _p(cen), i(C), i(H), i(W), i(interp))
        return chw_unpack(out, C)
    x = _c(img, dt)
    out = np.empty_like(x)
    lib().bevref_rotate(i(dtype), _p(out), _p(x), _p(ang), _p(cen), i(C), i(H), i(W), i(interp))
    return out


def rotate_s8(img, angle, center, interp, s_in, s_out, angle_half=False):
    """rotate_int8<float|__half> (rotateKernel.cu:735-748): kCHW4 in/out."""
    C, H, W = img.shape
    dt = np.float16 if angle_half else np.float32
    ang, cen = np.array([angle], dt), _c(center, dt)
    x = chw_pack(_c(img, np.int8), 4)
    out = np.zeros_like(x)
    i, f = ctypes.c_int, ctypes.c_float
    lib().bevref_rotate_int8(i(int(angle_half)), _p(out), f(s_out), _p(x), f(s_in), _p(ang), _p(cen), i(C), i(H),
                             i(W), i(interp))
    return chw_unpack(out, C)


def _dims(*shape):
    return (ctypes.c_int * len(shape))(*[int(s) for s in shape])


def grid_sampler(inp, grid, interp, pad, align, dtype=F32):
    """grid_sample<float|__half|__half2> (gridSamplerKernel.cu:1933-2006).  grid channel-first
    in [-10, 10]; 4-D or 5-D (the __half2 flavour is 4-D only, kCHW2)."""
    dt = _np_dt(dtype)
    inp, grid = _c(inp, dt), _c(grid, dt)
    nb = inp.ndim
    out_shape = inp.shape[:2] + grid.shape[2:]
    i = ctypes.c_int
    if dtype == H2:
        assert nb == 4
        C = inp.shape[1]
        x, g = chw_pack(inp, 2), chw_pack(grid, 2)
        out = np.empty(x.shape[:2] + grid.shape[2:] + (2,), dt)
        lib().bevref_grid_sample(i(dtype), _p(out), _p(x), _p(g), _dims(*out_shape), _dims(*inp.shape),
                                 _dims(*grid.shape), i(nb), i(interp), i(pad), i(int(align)))
        return chw_unpack(out, C)
    out = np.empty(out_shape, dt)
    lib().bevref_grid_sample(i(dtype), _p(out), _p(inp), _p(grid), _dims(*out_shape), _dims(*inp.shape),
                             _dims(*grid.shape), i(nb), i(interp), i(pad), i(int(align)))
    return out


def grid_sampler_s8(inp, grid, interp, pad, align, s_in, s_grid, s_out):
    """grid_sample_int8 (gridSamplerKernel.cu:2008-2043): input, grid and output in kCHW4."""
    inp, grid = _c(inp, np.int8), _c(grid, np.int8)
    N, C, H, W = inp.shape
    out_shape = (N, C) + grid.shape[2:]
    x, g = chw_pack(inp, 4), chw_pack(grid, 4)
    out = np.zeros(x.shape[:2] + grid.shape[2:] + (4,), np.int8)
    i, f = ctypes.c_int, ctypes.c_float
    lib().bevref_grid_sample_int8(_p(out), f(s_out), _p(x), f(s_in), _p(g), f(s_grid), _dims(*out_shape),
                                  _dims(*inp.shape), _dims(*grid.shape), i(4), i(interp), i(pad), i(int(align)))
    return chw_unpack(out, C)


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, out_height,
                out_width, dtype=F32, scales=None):
    """bev_pool_v2<float|__half> / _h2 / _int8 (bevPoolKernel.cu:151-190).  feat [N,H,W,C] ->
    [1,out_h,out_w,C]; int8 when scales=(s_depth, s_feat, s_out)."""
    r = [_c(x, np.int32) for x in (ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths)]
    c = feat.shape[-1]
    n_out = out_height * out_width * c
    i, f = ctypes.c_int, ctypes.c_float
    if scales is None:
        dt = _np_dt(dtype)
        depth, feat = _c(depth, dt), _c(feat, dt)
        out = np.empty((1, out_height, out_width, c), dt)
        lib().bevref_bev_pool_v2(i(dtype), i(c), i(len(r[3])), i(n_out), _p(depth), _p(feat), *[_p(x) for x in r],
                                 _p(out))
    else:
        depth, feat = _c(depth, np.int8), _c(feat, np.int8)
        out = np.empty((1, out_height, out_width, c), np.int8)
        lib().bevref_bev_pool_v2_int8(i(c), i(len(r[3])), i(n_out), _p(depth), f(scales[0]), _p(feat), f(scales[1]),
                                      *[_p(x) for x in r], _p(out), f(scales[2]))
    return out


def mdconv(x, offset, mask, weight, bias, stride, padding, dilation, groups, deform_groups, dtype=F32):
    """ModulatedDeformConvForwardCUDAKernel<float|__half|__half2>
    (modulatedDeformableConv2dKernel.cu:695-894); square stride/padding/dilation."""
    dt = _np_dt(dtype)
    x, offset, mask, weight = (_c(a, dt) for a in (x, offset, mask, weight))
    B, Cin, H, W = x.shape
    Cout, _, Kh, Kw = weight.shape
    Ho = (H + 2 * padding - (dilation * (Kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (Kw - 1) + 1)) // stride + 1
    b = _c(bias, dt) if bias is not None else None
    i = ctypes.c_int
    if dtype == H2:
        raise NotImplementedError("the kCHW2 DCN flavour is not exposed (packed weights / offsets)")
    out = np.empty((B, Cout, Ho, Wo), dt)
    rc = lib().bevref_mdconv(i(dtype), _p(x), _p(weight), _p(b), _p(offset), _p(mask), _p(out), i(B), i(Cin), i(H),
                             i(W), i(Cout), i(Kh), i(Kw), i(stride), i(padding), i(dilation), i(groups),
                             i(deform_groups))
    assert rc == 0
    return out


def mdconv_s8(x, s_in, offset, s_off, mask, s_mask, weight, s_w, bias, s_out, stride, padding, dilation, groups,
              deform_groups):
    """ModulatedDeformConvForwardCUDAKernel_int8<float> (kernel.cu:896-978): x and weight kCHW4."""
    x, offset, mask, weight = (_c(a, np.int8) for a in (x, offset, mask, weight))
    B, Cin, H, W = x.shape
    Cout, _, Kh, Kw = weight.shape
    Ho = (H + 2 * padding - (dilation * (Kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (Kw - 1) + 1)) // stride + 1
    xp, wp = chw_pack(x, 4), chw_pack(weight, 4)
    out = np.empty((B, Cout, Ho, Wo), np.int8)
    b = _c(bias, np.float32) if bias is not None else None
    i, f = ctypes.c_int, ctypes.c_float
    lib().bevref_mdconv_int8(_p(xp), f(s_in), _p(wp), f(s_w), _p(b), _p(offset), f(s_off), _p(mask), f(s_mask),
                             _p(out), f(s_out), i(B), i(Cin), i(H), i(W), i(Cout), i(Kh), i(Kw), i(stride),
                             i(padding), i(dilation), i(groups), i(deform_groups))
    return out
