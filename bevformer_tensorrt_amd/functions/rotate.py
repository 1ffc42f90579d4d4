"""rotate / rotate2 -- drop-in for det2trt/models/functions/rotate.py:99-134.

The reference's eager path builds an affine grid with torch ops and calls
torch.grid_sampler (:12-80); its engine runs the fused RotateTRT plugin.  Here one HIP
kernel (csrc/rotate.hip) does both, reached through `bevops_rotate_forward`.
"""
import torch

from ..utils import lib as _lib

_MODE = {"bilinear": 0, "nearest": 1}


def _scalar_dev(x, device, n):
    if not torch.is_tensor(x):
        x = torch.tensor(x, dtype=torch.float32)
    x = x.to(device).reshape(-1)
    if x.numel() != n:
        raise ValueError(f"expected {n} element(s), got {x.numel()}")
    if x.dtype not in (torch.float32, torch.float16):
        x = x.float()
    return x.contiguous()


def _rotate(img, angle, center, interpolation, scales=(1.0, 1.0)):
    assert img.is_cuda, "rotate: img must be on the GPU"
    assert img.ndim == 3  # functions/rotate.py:13
    if interpolation not in _MODE:
        raise KeyError(interpolation)
    handle = _lib.load_library()
    img = img.contiguous()
    angle = _scalar_dev(angle, img.device, 1)
    center = _scalar_dev(center, img.device, 2)
    if center.dtype != angle.dtype:
        center = center.to(angle.dtype)
    if img.dtype == torch.float32 and angle.dtype != torch.float32:
        angle, center = angle.float(), center.float()
    out = torch.empty_like(img)
    C, H, W = img.shape
    if out.numel() == 0:
        return out
    with torch.cuda.device(img.device):
        st = handle.bevops_rotate_forward(
            _lib.torch_dtype_code(img), img.data_ptr(), angle.data_ptr(), center.data_ptr(),
            _lib.torch_dtype_code(angle), out.data_ptr(), C, H, W, _MODE[interpolation],
            float(scales[0]), float(scales[1]), _lib.current_stream_ptr(img.device))
    _lib.check(st, "bevops_rotate_forward")
    return out


def rotate(img, angle, center, interpolation="nearest"):
    """Rotate `img` [C,H,W] by `angle` degrees (counter-clockwise) about `center` (x, y).
    Plugin RotateTRT (fp32, fp16).  Signature as functions/rotate.py:99."""
    return _rotate(img, angle, center, interpolation)


def rotate2(img, angle, center, interpolation="nearest"):
    """Same op under the reference's half2 plugin name RotateTRT2 (functions/rotate.py:118)."""
    return _rotate(img, angle, center, interpolation)


def rotate_int8(img, angle, center, scale_in, scale_out, interpolation="nearest"):
    """INT8 flavour (rotateKernel.cu:415-706): int8 image + per-tensor scales."""
    return _rotate(img, angle, center, interpolation, (scale_in, scale_out))


def rotate_hwc(img, angle, center, interpolation="nearest", out=None):
    """`rotate` on channels-last data: img [H, W, C] -> [H, W, C] (fp32 / fp16).  Element for element
    the same result as rotate(img.permute(2, 0, 1), ...).permute(1, 2, 0), without the two layout
    copies (not a reference plugin: the layout prev_bev already has between frames).  `out`: optional contiguous
    destination of the same shape and dtype (the model rotates prev_bev straight into its [prev_bev | query] stack)."""
    assert img.is_cuda and img.ndim == 3
    if interpolation not in _MODE:
        raise KeyError(interpolation)
    handle = _lib.load_library()
    img = img.contiguous()
    angle = _scalar_dev(angle, img.device, 1)
    center = _scalar_dev(center, img.device, 2)
    if center.dtype != angle.dtype:
        center = center.to(angle.dtype)
    if img.dtype == torch.float32 and angle.dtype != torch.float32:
        angle, center = angle.float(), center.float()
    if out is None:
        out = torch.empty_like(img)
    elif out.shape != img.shape or out.dtype != img.dtype or not out.is_contiguous() or out.device != img.device \
            or out.data_ptr() == img.data_ptr():
        raise ValueError("rotate_hwc: `out` must be another contiguous tensor of the input's shape, dtype and device")
    H, W, C = img.shape
    with torch.cuda.device(img.device):
        st = handle.bevops_rotate_forward_hwc(
            _lib.torch_dtype_code(img), img.data_ptr(), angle.data_ptr(), center.data_ptr(),
            _lib.torch_dtype_code(angle), out.data_ptr(), C, H, W, _MODE[interpolation],
            _lib.current_stream_ptr(img.device))
    _lib.check(st, "bevops_rotate_forward_hwc")
    return out
