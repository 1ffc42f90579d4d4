"""spatial_cross_attention_sample -- fused sampling step of BEVFormer's spatial cross-attention
(SURVEY.md 8f-3).  NOT one of the reference's plugin functions: it stands for the sequence

    queries = multi_scale_deformable_attn(value, shapes, ref_cam, offsets.repeat(num_cams), weights.repeat(num_cams))
    slots   = (queries.flatten(2) * bev_mask).sum(0, keepdim=True)

of det2trt/models/modules/spatial_cross_attention.py:254-270 (offsets / weights are identical for
every camera because the query is repeated, :254), skipping the (camera, query) pairs whose
bev_mask weight is zero like the original PyTorch implementation's rebatching
(third_party/bev_mmdet3d/models/modules/spatial_cross_attention.py:143-191).
"""
import torch

from ..utils import lib as _lib
from .multi_scale_deformable_attn import _host_shapes, _shapes_i32, _workspace


def spatial_cross_attention_sample(value, value_spatial_shapes, reference_points_cam, sampling_offsets,
                                   attention_weights, bev_mask):
    """
    Args:
        value: (num_cams, num_keys, num_heads, 32) fp16
        value_spatial_shapes: (num_levels, 2)
        reference_points_cam: (num_cams, num_query, 1, 2 * points_per_group), normalised
        sampling_offsets: (1, num_query, num_heads, num_levels * num_points * 2), shared by the cameras
        attention_weights: (1, num_query, num_heads, num_levels * num_points), pre-softmax
        bev_mask: (num_cams, num_query[, 1]) visibility weight of each (camera, query); 0 = skip
    Returns: (1, num_query, num_heads * 32) = sum over cameras of bev_mask * sampled
    """
    assert value.is_cuda, "spatial_cross_attention_sample: value must be on the GPU"
    if value.dtype != torch.float16:
        raise TypeError("spatial_cross_attention_sample is fp16-only; compose "
                        "multi_scale_deformable_attn + the masked sum for other dtypes")
    handle = _lib.load_library()
    ncam, nk, heads, ch = value.shape
    nq = sampling_offsets.shape[1]
    L = value_spatial_shapes.shape[0]
    ppg = reference_points_cam.shape[-1] // 2
    if sampling_offsets.shape[0] != 1 or attention_weights.shape[0] != 1:
        raise ValueError("sampling_offsets / attention_weights must be the camera-shared [1, nq, heads, .] tensors")
    P = attention_weights.numel() // (nq * heads * L)
    if sampling_offsets.numel() != nq * heads * L * P * 2:
        raise ValueError("sampling_offsets / attention_weights shapes disagree")
    if reference_points_cam.numel() != ncam * nq * ppg * 2:
        raise ValueError("reference_points_cam must be [num_cams, num_query, 1, 2*points_per_group]")
    mask = bev_mask.reshape(ncam, nq)
    value, ref, off, w, mask = (t.to(torch.float16).contiguous()
                                for t in (value, reference_points_cam, sampling_offsets, attention_weights, mask))
    shapes_dev, shapes_host = _shapes_i32(value_spatial_shapes, value.device)
    if shapes_host is None:
        shapes_host = _host_shapes(shapes_dev)
    out = torch.empty((1, nq, heads * ch), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device):
        stream = _lib.current_stream_ptr(value.device)
        ws_bytes = handle.bevops_sca_workspace_size(_lib.F16, shapes_host.data_ptr(), ncam, nk, heads, ch, L, nq, P)
        if ws_bytes == 0:
            raise _lib.BevopsError("bevops_sca_workspace_size: unsupported arguments (needs 32 channels per head)", _lib.NOT_SUPPORTED)
        ws = _workspace(ws_bytes, value.device, stream)
        st = handle.bevops_sca_forward(_lib.F16, value.data_ptr(), shapes_host.data_ptr(), ref.data_ptr(),
                                       off.data_ptr(), w.data_ptr(), mask.data_ptr(), out.data_ptr(), ncam, nk,
                                       heads, ch, L, nq, P, ppg, ws.data_ptr(), ws.numel(), stream)
    _lib.check(st, "bevops_sca_forward")
    return out


PLANNED = {"enabled": __import__("os").environ.get("BEVOPS_SCA_PLAN", "1") != "0"}   # A/B: balanced slices of a visibility plan


def spatial_cross_attention_plan(bev_mask):
    """Visibility plan of the fused SCA sampling for one rig (bevops_sca_plan_build): per camera the ascending list of
    the queries whose bev_mask weight is non-zero, as opaque device bytes.  bev_mask [num_cams, num_query(, 1)] depends
    on lidar2img only (modules/encoder.py:255-258), so a frame loop builds this once per calibration and hands it to
    every spatial_cross_attention_projected call of every layer and frame.  None when the mask is outside the plan's
    domain (more than 16 cameras or 65 535 queries) -- the caller then runs without a plan."""
    assert bev_mask.is_cuda
    handle = _lib.load_library()
    ncam = bev_mask.shape[0]
    mask = bev_mask.reshape(ncam, -1).to(torch.float16).contiguous()
    nq = mask.shape[1]
    size = handle.bevops_sca_plan_size(ncam, nq) if ncam > 0 else 0
    if size == 0:
        return None
    plan = torch.empty(size, dtype=torch.uint8, device=mask.device)
    with torch.cuda.device(mask.device):
        st = handle.bevops_sca_plan_build(_lib.F16, mask.data_ptr(), ncam, nq, plan.data_ptr(), size,
                                          _lib.current_stream_ptr(mask.device))
    _lib.check(st, "bevops_sca_plan_build")
    return plan


def spatial_cross_attention_projected(features, weight, bias, value_spatial_shapes, reference_points_cam,
                                      sampling_offsets, attention_weights, bev_mask, num_heads=8, plan=None):
    """value_proj + fused SCA sampling in two launches without the [cams, keys, heads, 32] tensor in between
    (spatial_cross_attention.py:754 followed by :254-270): the value projection runs on the tall-skinny MFMA
    GEMM whose epilogue stores straight into the sampler's padded head-major planes
    (bevops_value_proj_packed), then the fused sampling reads them (bevops_sca_forward_prepacked).

        features: (num_cams, num_keys, embed) fp16 -- the encoder input (FPN levels + camera / level embeddings)
        weight, bias: value_proj parameters [embed, embed], [embed]
        the other arguments as spatial_cross_attention_sample
        plan: spatial_cross_attention_plan(bev_mask) of THIS bev_mask, or None (the sampling kernel then compacts
              its own chunk of queries per block: same results, unbalanced blocks)
    Returns (1, num_query, embed).  Raises BevopsError (NOT_SUPPORTED) outside the 4-level x 8-point domain."""
    assert features.is_cuda and features.dtype == torch.float16
    handle = _lib.load_library()
    ncam, nk, embed = features.shape
    heads, ch = num_heads, embed // num_heads
    nq = sampling_offsets.shape[1]
    L = value_spatial_shapes.shape[0]
    ppg = reference_points_cam.shape[-1] // 2
    P = attention_weights.numel() // (nq * heads * L)
    if sampling_offsets.shape[0] != 1 or attention_weights.shape[0] != 1:
        raise ValueError("sampling_offsets / attention_weights must be the camera-shared [1, nq, heads, .] tensors")
    mask = bev_mask.reshape(ncam, nq)
    feats, ref, off, w, mask = (t.to(torch.float16).contiguous()
                                for t in (features, reference_points_cam, sampling_offsets, attention_weights, mask))
    weight = weight.to(torch.float16).contiguous()
    bias = None if bias is None else bias.to(torch.float16).contiguous()
    shapes_dev, shapes_host = _shapes_i32(value_spatial_shapes, features.device)
    if shapes_host is None:
        shapes_host = _host_shapes(shapes_dev)
    geom = (shapes_host, ncam, nk, heads, ch, L, nq, P, ppg)
    planes = _project_planes(handle, feats, weight, bias, geom)
    return _sample_planes(handle, planes, geom, ref, off, w, mask, plan)


def _project_planes(handle, feats, weight, bias, geom):
    """bevops_value_proj_packed into a lent workspace: (workspace tensor, plane bytes, offset of the sampler's own
    workspace, its size)."""
    shapes_host, ncam, nk, heads, ch, L, nq, P, _ = geom
    with torch.cuda.device(feats.device):
        stream = _lib.current_stream_ptr(feats.device)
        pk_bytes = handle.bevops_value_proj_packed_size(shapes_host.data_ptr(), ncam, nk, heads, ch, L, nq, P)
        if pk_bytes == 0:
            raise _lib.BevopsError("bevops_value_proj_packed_size: shape outside the packed-projection domain",
                                   _lib.NOT_SUPPORTED)
        ws_bytes = handle.bevops_sca_prepacked_workspace_size(ncam, heads, ch, nq)
        pk_room = (pk_bytes + 255) & ~255
        ws = _workspace(pk_room + ws_bytes, feats.device, stream)
        st = handle.bevops_value_proj_packed(feats.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                             shapes_host.data_ptr(), ws.data_ptr(), pk_bytes, ncam, nk, heads, ch, L, nq,
                                             P, stream)
    _lib.check(st, "bevops_value_proj_packed")
    return ws, pk_bytes, pk_room, ws_bytes


def _sample_planes(handle, planes, geom, ref, off, w, mask, plan=None):
    """The fused sampling on planes `_project_planes` left in the workspace (tools time this half alone)."""
    shapes_host, ncam, nk, heads, ch, L, nq, P, ppg = geom
    ws, pk_bytes, pk_room, ws_bytes = planes
    out = torch.empty((1, nq, heads * ch), dtype=ref.dtype, device=ref.device)
    with torch.cuda.device(ref.device):
        stream = _lib.current_stream_ptr(ref.device)
        if plan is not None and PLANNED["enabled"]:
            assert plan.is_cuda and plan.dtype == torch.uint8 and plan.is_contiguous()
            st = handle.bevops_sca_forward_planned(_lib.F16, ws.data_ptr(), pk_bytes, shapes_host.data_ptr(),
                                                   ref.data_ptr(), off.data_ptr(), w.data_ptr(), mask.data_ptr(),
                                                   plan.data_ptr(), plan.numel(), out.data_ptr(), ncam, nk, heads, ch, L,
                                                   nq, P, ppg, ws.data_ptr() + pk_room, ws_bytes, stream)
            _lib.check(st, "bevops_sca_forward_planned")
            return out
        st = handle.bevops_sca_forward_prepacked(_lib.F16, ws.data_ptr(), pk_bytes, shapes_host.data_ptr(), ref.data_ptr(),
                                                 off.data_ptr(), w.data_ptr(), mask.data_ptr(), out.data_ptr(), ncam, nk,
                                                 heads, ch, L, nq, P, ppg, ws.data_ptr() + pk_room, ws_bytes, stream)
    _lib.check(st, "bevops_sca_forward_prepacked")
    return out
