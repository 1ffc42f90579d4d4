"""grid_sampler / grid_sampler2 -- drop-in for
det2trt/models/functions/grid_sampler.py:140-305 (4-D and 5-D inputs, channel-first grid
scaled to [-10, 10])."""
import torch

from ..utils import lib as _lib
from ..utils import workspace as _ws

_MODE = {"bilinear": 0, "nearest": 1, "bicubic": 2}
_PAD = {"zeros": 0, "border": 1, "reflection": 2}


def _workspace(device, nbytes, stream):
    """Scratch per (device, stream); superseded buffers stay alive (utils/workspace.py)."""
    return _ws.lend("grid_sampler", nbytes, device, stream)


def _grid_sampler(input, grid, interpolation_mode, padding_mode, align_corners,
                  scales=(1.0, 1.0, 1.0)):
    assert input.is_cuda, "grid_sampler: input must be on the GPU"
    handle = _lib.load_library()
    mode, pad = _MODE[interpolation_mode], _PAD[padding_mode]
    if grid.dtype != input.dtype:
        raise TypeError(f"grid dtype {grid.dtype} != input dtype {input.dtype}")
    input, grid = input.contiguous(), grid.contiguous()
    dt = _lib.torch_dtype_code(input)
    stream = _lib.current_stream_ptr(input.device)
    if grid.dim() == 4:
        N, C, H, W = input.shape
        if grid.shape[0] != N or grid.shape[1] != 2:
            raise ValueError(f"grid must be [N,2,H_out,W_out], got {tuple(grid.shape)}")
        Ho, Wo = grid.shape[2:]
        out = torch.empty((N, C, Ho, Wo), dtype=input.dtype, device=input.device)
        if out.numel() == 0:
            return out
        if input.numel() == 0:      # no source pixels: every mode reads padding zeros
            return out.zero_()
        # lend a scratch buffer: up-sampling calls stage the input channels-last (same results)
        nws = handle.bevops_grid_sampler_2d_workspace_size(dt, N, C, H, W) if dt != _lib.I8 else 0
        ws = _workspace(input.device, nws, stream) if nws and Ho * Wo >= 2 * H * W else None
        with torch.cuda.device(input.device):
            st = handle.bevops_grid_sampler_2d_forward_ws(
                dt, input.data_ptr(), grid.data_ptr(), out.data_ptr(), N, C, H, W, Ho, Wo, mode,
                pad, int(bool(align_corners)), float(scales[0]), float(scales[1]),
                float(scales[2]), ws.data_ptr() if ws is not None else None, nws if ws is not None else 0, stream)
        _lib.check(st, "bevops_grid_sampler_2d_forward_ws")
        return out
    if grid.dim() == 5:
        N, C, D, H, W = input.shape
        if grid.shape[0] != N or grid.shape[1] != 3:
            raise ValueError(f"grid must be [N,3,D_out,H_out,W_out], got {tuple(grid.shape)}")
        Do, Ho, Wo = grid.shape[2:]
        out = torch.empty((N, C, Do, Ho, Wo), dtype=input.dtype, device=input.device)
        if out.numel() == 0:
            return out
        with torch.cuda.device(input.device):
            st = handle.bevops_grid_sampler_3d_forward(
                dt, input.data_ptr(), grid.data_ptr(), out.data_ptr(), N, C, D, H, W, Do, Ho, Wo,
                mode, pad, int(bool(align_corners)), stream)
        _lib.check(st, "bevops_grid_sampler_3d_forward")
        return out
    raise RuntimeError  # grid_sampler.py:236


def grid_sampler(input, grid, interpolation_mode, padding_mode, align_corners):
    """Plugin GridSampler2DTRT / GridSampler3DTRT (fp32, fp16).  input [N,C,(D,)H,W],
    grid [N,2,Ho,Wo] or [N,3,Do,Ho,Wo] with values in [-10, 10]; modes
    bilinear|nearest|bicubic(4-D only), paddings zeros|border|reflection."""
    return _grid_sampler(input, grid, interpolation_mode, padding_mode, align_corners)


def grid_sampler2(input, grid, interpolation_mode, padding_mode, align_corners):
    """Same op under the half2 plugin names GridSampler2DTRT2 / GridSampler3DTRT2."""
    return _grid_sampler(input, grid, interpolation_mode, padding_mode, align_corners)


def grid_sampler_int8(input, grid, interpolation_mode, padding_mode, align_corners, scale_in,
                      scale_grid, scale_out):
    """INT8 2-D flavour (gridSamplerKernel.cu:1082-1268), bilinear / nearest."""
    return _grid_sampler(input, grid, interpolation_mode, padding_mode, align_corners,
                         (scale_in, scale_grid, scale_out))
