"""image_normalize_pad -- the camera-image front end of the frame loop: the
reference's NormalizeMultiviewImage + PadMultiViewImage(size_divisor=32) + DefaultFormatBundle3D
(configs/bevformer/bevformer_base.py:11,228-231) as one HIP pass over the raw images."""
import ctypes

import torch

from ..utils import lib as _lib

IMG_NORM_CFG = dict(mean=[103.530, 116.280, 123.675], std=[1.0, 1.0, 1.0], to_rgb=False)   # bevformer_base.py:11


def padded_size(h, w, divisor=32):
    """mmcv.impad_to_multiple: bottom / right padding to the next multiple of `divisor`."""
    return -(-h // divisor) * divisor, -(-w // divisor) * divisor


def image_normalize_pad(images, mean=None, std=None, to_rgb=False, size_divisor=32, dtype=torch.float16,
                        channels_last=False, out=None):
    """images [N, H0, W0, 3] uint8 or float32 on the GPU (BGR, as cv2 loads them) ->
    [N, 3, Hp, Wp] `dtype` (memory format channels_last if asked), normalised and zero padded."""
    assert images.is_cuda and images.dim() == 4 and images.shape[-1] == 3
    if images.dtype not in (torch.uint8, torch.float32):
        raise TypeError("images must be uint8 or float32")
    mean = IMG_NORM_CFG["mean"] if mean is None else mean
    std = IMG_NORM_CFG["std"] if std is None else std
    N, H0, W0, _ = images.shape
    Hp, Wp = padded_size(H0, W0, size_divisor)
    images = images.contiguous()
    if out is not None and (tuple(out.shape) != (N, 3, Hp, Wp) or not out.is_cuda or not (
            out.is_contiguous(memory_format=torch.channels_last) if channels_last else out.is_contiguous())):
        raise ValueError(f"out must be a dense [N, 3, {Hp}, {Wp}] tensor on the GPU in the requested layout")
    if out is None:
        out = torch.empty((N, 3, Hp, Wp), dtype=dtype, device=images.device,
                          memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    m = (ctypes.c_double * 3)(*[float(v) for v in mean])
    s = (ctypes.c_double * 3)(*[float(v) for v in std])
    handle = _lib.load_library()
    with torch.cuda.device(images.device):
        st = handle.bevops_image_normalize_pad(
            _lib.U8 if images.dtype == torch.uint8 else _lib.F32, images.data_ptr(), _lib.torch_dtype_code(out),
            out.data_ptr(), N, H0, W0, Hp, Wp, m, s, int(bool(to_rgb)), int(bool(channels_last)),
            _lib.current_stream_ptr(images.device))
    _lib.check(st, "bevops_image_normalize_pad")
    return out
