"""bev_pool_v2 / bev_pool_v2_2 -- drop-in for det2trt/models/functions/bev_pool_v2.py:105-151."""
import torch

from ..utils import lib as _lib


def _i32(t, device):
    return t.to(device=device, dtype=torch.int32).contiguous()


def _bev_pool(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths,
              out_height, out_width, scales=(1.0, 1.0, 1.0)):
    assert depth.is_cuda and feat.is_cuda, "bev_pool_v2: depth/feat must be on the GPU"
    if depth.dtype != feat.dtype:
        raise TypeError(f"depth dtype {depth.dtype} != feat dtype {feat.dtype}")
    handle = _lib.load_library()
    dev = feat.device
    depth, feat = depth.contiguous(), feat.contiguous()
    # ranks arrive as float tensors in the reference pipeline and are cast with .int() (:116-120)
    rd, rf, rb, ist, il = (_i32(t, dev) for t in (ranks_depth, ranks_feat, ranks_bev,
                                                  interval_starts, interval_lengths))
    c = feat.shape[-1]
    out = torch.empty((1, out_height, out_width, c), dtype=feat.dtype, device=dev)
    with torch.cuda.device(dev):
        st = handle.bevops_bev_pool_v2_forward(
            _lib.torch_dtype_code(feat), depth.data_ptr(), feat.data_ptr(), rd.data_ptr(),
            rf.data_ptr(), rb.data_ptr(), ist.data_ptr(), il.data_ptr(), out.data_ptr(), c,
            ist.numel(), out_height, out_width, float(scales[0]), float(scales[1]),
            float(scales[2]), _lib.current_stream_ptr(dev))
    _lib.check(st, "bevops_bev_pool_v2_forward")
    return out


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                interval_lengths, out_height=128, out_width=128):
    """depth [N,D,H,W], feat [N,H,W,C] -> [1, out_height, out_width, C] (plugin BEVPoolV2TRT)."""
    return _bev_pool(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                     interval_lengths, out_height, out_width)


def bev_pool_v2_2(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                  interval_lengths, out_height=128, out_width=128):
    """Same op under the half2 plugin name BEVPoolV2TRT2."""
    return _bev_pool(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                     interval_lengths, out_height, out_width)


def bev_pool_v2_int8(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                     interval_lengths, scale_depth, scale_feat, scale_out, out_height=128,
                     out_width=128):
    """INT8 flavour (bevPoolKernel.cu:115-149): int32 accumulate, requantise."""
    return _bev_pool(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                     interval_lengths, out_height, out_width, (scale_depth, scale_feat, scale_out))
