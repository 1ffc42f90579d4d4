"""point_sampling -- the camera projection of the BEV pillars (BEVFormerEncoderTRTP.point_sampling_trt,
det2trt/models/modules/encoder.py:197-259) as one launch.  Not one of the reference's plugin functions: there it is
~30 torch ops inside the exported engine, evaluated on every frame because `lidar2img` is an engine input
(tools/bevformer/evaluate_trt.py:131-132).  `geometry.project_points` is the same arithmetic spelled in torch ops
(the host / other-device path); tests/test_geometry_gpu.py holds the two bit-identical."""
import torch

from ..utils import lib as _lib


def point_sampling(pillars, lidar2img, image_shape, dtype=torch.float16):
    """
    Args:
        pillars: (D, 1, 1, num_query, 4, 1) or (D, num_query, 4) fp32 -- geometry.pillar_points(ref_3d, pc_range)
        lidar2img: (*, num_cams, 4, 4) fp32 on the same device
        image_shape: (h, w) of the padded camera images
        dtype: torch.float16 | torch.float32 of the two results
    Returns:
        reference_points_cam (num_cams, 1, num_query, D, 2), bev_mask (num_cams, num_query, 1)
    Raises BevopsError (NOT_SUPPORTED) unless D == 4.
    """
    assert pillars.is_cuda and lidar2img.is_cuda, "point_sampling: operands must be on the GPU"
    handle = _lib.load_library()
    D = pillars.shape[0]
    pts = pillars.reshape(D, -1, 4).to(torch.float32).contiguous()
    nq = pts.shape[1]
    l2i = lidar2img.reshape(-1, 4, 4).to(torch.float32).contiguous()
    ncam = l2i.shape[0]
    code = {torch.float16: _lib.F16, torch.float32: _lib.F32}.get(dtype)
    if code is None:
        raise TypeError("point_sampling: dtype must be torch.float16 or torch.float32")
    ref = torch.empty((ncam, 1, nq, D, 2), dtype=dtype, device=pts.device)
    mask = torch.empty((ncam, nq, 1), dtype=dtype, device=pts.device)
    with torch.cuda.device(pts.device):
        st = handle.bevops_point_sampling(code, pts.data_ptr(), l2i.data_ptr(), ref.data_ptr(), mask.data_ptr(), ncam, nq, D,
                                          float(image_shape[0]), float(image_shape[1]), _lib.current_stream_ptr(pts.device))
    _lib.check(st, "bevops_point_sampling")
    return ref, mask
