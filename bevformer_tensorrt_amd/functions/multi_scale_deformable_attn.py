"""multi_scale_deformable_attn / multi_scale_deformable_attn2 -- drop-in for
det2trt/models/functions/multi_scale_deformable_attn.py:150-217.

The reference's eager `forward()` materialises sampling locations and a softmax
and then calls mmcv's CUDA extension (:58-115); its TensorRT engine runs the fused
plugin instead.  Here both are one fused HIP kernel (csrc/msda.hip) reached
through `bevops_msda_forward`.  fp16 inputs are upcast inside the kernel (fp32
location/softmax/accumulate), output returned in the input dtype like :123.
"""
import weakref

import torch

from ..utils import lib as _lib
from ..utils import workspace as _ws

# Caches are keyed by the tensor OBJECT (weakly) and validated by everything that can change
# under a live object: the in-place version counter, and -- because nn.Module.half() / .to() /
# `param.data = ...` swap the storage of the SAME Parameter object without bumping the version --
# dtype, device, data pointer and shape.
class _TensorCache:
    """id(tensor) -> value (WeakKeyDictionary cannot be used: Tensor.__eq__ is element-wise)."""

    def __init__(self):
        self._d = {}

    @staticmethod
    def _stamp(t):
        return (t._version, t.dtype, t.device, t.data_ptr(), tuple(t.shape))

    def get(self, t):
        hit = self._d.get(id(t))
        if hit is not None and hit[0]() is t and hit[1] == self._stamp(t):
            return hit[2]
        return None

    def put(self, t, value):
        if len(self._d) > 256:
            self._d = {k: v for k, v in self._d.items() if v[0]() is not None}
        self._d[id(t)] = (weakref.ref(t), self._stamp(t), value)
        return value


_CPU_SHAPES = {}
_DEV_I32 = _TensorCache()
_HOST_SHAPES = _TensorCache()


def _shapes_i32(shapes, device):
    """int32 device copy (+ host copy when free) of value_spatial_shapes."""
    if shapes.device.type == "cpu":
        key = (tuple(shapes.flatten().tolist()), str(device))
        hit = _CPU_SHAPES.get(key)
        if hit is None:
            host = shapes.to(torch.int32).contiguous()
            hit = (host.to(device), host)
            _CPU_SHAPES[key] = hit
        return hit
    if shapes.dtype == torch.int32 and shapes.is_contiguous():
        return shapes, None
    hit = _DEV_I32.get(shapes)
    if hit is None:
        hit = _DEV_I32.put(shapes, shapes.to(torch.int32).contiguous())
    return hit, None


def _host_shapes(shapes_dev):
    """Host copy of a device-resident value_spatial_shapes (lets the library stage the small
    pyramid levels in LDS).  One blocking .cpu() per distinct tensor object, then cached."""
    hit = _HOST_SHAPES.get(shapes_dev)
    if hit is None:
        hit = _HOST_SHAPES.put(shapes_dev, shapes_dev.to("cpu", torch.int32).contiguous())
    return hit


def _workspace(nbytes, device, stream_ptr):
    """Scratch for the head-major re-layouts: see utils/workspace.py (per device AND stream,
    superseded buffers kept alive for captured graphs)."""
    return _ws.lend("msda", nbytes, device, stream_ptr)


release_workspaces = _ws.release


def _msda(value, value_spatial_shapes, reference_points, sampling_offsets, attention_weights,
          scales=(1.0, 1.0, 1.0, 1.0), out=None):
    assert value.is_cuda, "multi_scale_deformable_attn: value must be on the GPU"
    if value.dim() != 4:
        raise ValueError(f"value must be [bs, num_keys, heads, channels], got {tuple(value.shape)}")
    handle = _lib.load_library()
    bs, nk, heads, ch = value.shape
    L = value_spatial_shapes.shape[0]
    nq = sampling_offsets.shape[1]
    ppg = reference_points.shape[-1] // 2
    if bs == 0 or nq == 0 or heads == 0 or ch == 0:
        # empty batch / no queries: nothing to sample (the eager reference path returns the empty tensor too;
        # the C ABI itself treats a zero dimension as a bad parameter, like a zero-sized launch)
        return value.new_empty((bs, nq, heads, ch)) if out is None else out
    P = attention_weights.numel() // (bs * nq * heads * L)
    if sampling_offsets.numel() != bs * nq * heads * L * P * 2:
        raise ValueError("sampling_offsets / attention_weights shapes disagree")
    if reference_points.numel() != bs * nq * ppg * 2:
        raise ValueError("reference_points must be [bs, num_query, 1, 2*points_per_group]")
    dt = _lib.torch_dtype_code(value)
    for name, t in (("sampling_offsets", sampling_offsets), ("attention_weights", attention_weights)):
        if t.dtype != value.dtype:
            raise TypeError(f"{name} dtype {t.dtype} != value dtype {value.dtype}")
    rdt = _lib.torch_dtype_code(reference_points)
    # offsets / logits expanded over the batch (stride 0 views, e.g. the SCA query repeated for
    # every camera): hand the single copy to the kernel instead of materialising bs copies
    shared = (bs > 1 and sampling_offsets.dim() == 4 and attention_weights.dim() == 4
              and sampling_offsets.stride(0) == 0 and attention_weights.stride(0) == 0)
    if shared:
        sampling_offsets, attention_weights = sampling_offsets[:1], attention_weights[:1]
    value, reference_points, sampling_offsets, attention_weights = (
        t.contiguous() for t in (value, reference_points, sampling_offsets, attention_weights))
    shapes_dev, shapes_host = _shapes_i32(value_spatial_shapes, value.device)
    if out is None:
        out = torch.empty((bs, nq, heads, ch), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device):
        stream = _lib.current_stream_ptr(value.device)
        # the library validates sum(H*W) == num_keys (and plans the head-major layouts) from the
        # HOST copy of the shapes: always hand it over (one blocking copy per distinct shapes
        # tensor, then cached; skipped only for a first sighting under stream capture)
        if shapes_host is None and (_HOST_SHAPES.get(shapes_dev) is not None
                                    or not torch.cuda.is_current_stream_capturing()):
            shapes_host = _host_shapes(shapes_dev)
        ws_bytes = handle.bevops_msda_workspace_size(dt, bs, nk, heads, ch, L, nq, P)
        ws = None
        if ws_bytes and shapes_host is not None:
            ws_bytes = handle.bevops_msda_workspace_size_shapes(
                dt, shapes_host.data_ptr(), bs, nk, heads, ch, L, nq, P)
            ws = _workspace(ws_bytes, value.device, stream)
        st = handle.bevops_msda_forward_ws(
            dt, value.data_ptr(), shapes_dev.data_ptr(),
            shapes_host.data_ptr() if shapes_host is not None else None,
            reference_points.data_ptr(), rdt, sampling_offsets.data_ptr(),
            attention_weights.data_ptr(), out.data_ptr(), bs, nk, heads, ch, L, nq, P, ppg,
            float(scales[0]), float(scales[1]), float(scales[2]), float(scales[3]), int(shared),
            ws.data_ptr() if ws is not None else None, ws_bytes if ws is not None else 0, stream)
    _lib.check(st, "bevops_msda_forward_ws")
    return out


def multi_scale_deformable_attn(value, value_spatial_shapes, reference_points, sampling_offsets,
                                attention_weights):
    """Multi-scale deformable attention (plugin MultiScaleDeformableAttnTRT: fp32, fp16).

    Args (as det2trt/models/functions/multi_scale_deformable_attn.py:153-174):
        value: (bs, num_keys, num_heads, embed_dims // num_heads)
        value_spatial_shapes: (num_levels, 2), last dim (h, w)
        reference_points: (bs, num_queries, 1, 2 * points_per_group), normalised
        sampling_offsets: (bs, num_queries, num_heads, num_levels * num_points * 2), (x, y) pixels
        attention_weights: (bs, num_queries, num_heads, num_levels * num_points), pre-softmax
    Returns: (bs, num_queries, num_heads, embed_dims // num_heads)
    """
    return _msda(value, value_spatial_shapes, reference_points, sampling_offsets, attention_weights)


def multi_scale_deformable_attn_local(value, value_spatial_shapes, reference_points, sampling_offsets,
                                      attention_weights):
    """The same operator for callers that KNOW their reference points have locality -- neighbouring queries sample
    neighbouring pixels, as the BEV grid of temporal self-attention does (encoder.py:170-195): it runs the
    layout-preserving quad kernel directly on `value` (no head-major re-layout pass), whatever the size of the maps.
    The default dispatch must assume the op test's uniform-random points, for which maps beyond an XCD's L2 want the
    head-major form.  Same values (tests); not a reference name."""
    handle = _lib.load_library()
    prev = handle.bevops_msda_set_variant(10)          # thread-local: "never the head-major kernels"
    try:
        return _msda(value, value_spatial_shapes, reference_points, sampling_offsets, attention_weights)
    finally:
        handle.bevops_msda_set_variant(prev)


def multi_scale_deformable_attn2(value, value_spatial_shapes, reference_points, sampling_offsets,
                                 attention_weights):
    """Same op under the reference's `half2` plugin name (MultiScaleDeformableAttnTRT2,
    :184-217).  On MI355X both names run the same 16-byte-vectorised kernel."""
    return _msda(value, value_spatial_shapes, reference_points, sampling_offsets, attention_weights)


def multi_scale_deformable_attn_int8(value, value_spatial_shapes, reference_points,
                                     sampling_offsets, attention_weights, scale_value,
                                     scale_offset, scale_weight, scale_out):
    """INT8 flavour.  In the reference the int8 tensors and their per-tensor scales come
    from TensorRT (PluginTensorDesc::scale, multiScaleDeformableAttnPlugin.cpp:75-77);
    here the caller passes int8 tensors + scales explicitly.  reference_points stay
    fp32 (signed x127 weights) or fp16 (unsigned x255 weights)."""
    return _msda(value, value_spatial_shapes, reference_points, sampling_offsets,
                 attention_weights, (scale_value, scale_offset, scale_weight, scale_out))


class PackedValue:
    """`value` in the padded head-major form of the library (bevops_msda_pack_value): opaque bytes
    plus the call geometry they were packed for."""

    def __init__(self, data, dtype, ref_dtype, shapes_host, shapes_dev, dims):
        self.data, self.dtype, self.ref_dtype = data, dtype, ref_dtype
        self.shapes_host, self.shapes_dev, self.dims = shapes_host, shapes_dev, dims


def msda_pack_value(value, value_spatial_shapes, num_query, num_point, reference_dtype=None, out=None):
    """Re-lay `value` [bs, nk, heads, 32] (fp16 or int8) once for several
    `multi_scale_deformable_attn_prepacked` calls with the same (num_query, num_point).  For int8 the
    packed form depends on the flavour: pass the dtype of the reference points (float32 -> x127
    weights, float16 -> x255 weights)."""
    assert value.is_cuda and value.dim() == 4
    handle = _lib.load_library()
    bs, nk, heads, ch = value.shape
    L = value_spatial_shapes.shape[0]
    dt = _lib.torch_dtype_code(value)
    rdt = _lib.F16 if reference_dtype in (None, torch.float16) else _lib.F32
    if dt == _lib.F16:
        rdt = _lib.F16
    shapes_dev, shapes_host = _shapes_i32(value_spatial_shapes, value.device)
    if shapes_host is None:
        shapes_host = _host_shapes(shapes_dev)
    nbytes = handle.bevops_msda_packed_size(dt, shapes_host.data_ptr(), bs, nk, heads, ch, L, num_query, num_point)
    if nbytes == 0:
        raise _lib.BevopsError("bevops_msda_packed_size: shape outside the head-major domain", _lib.NOT_SUPPORTED)
    if out is None or out.numel() < nbytes:
        out = torch.empty(nbytes, dtype=torch.uint8, device=value.device)
    value = value.contiguous()
    with torch.cuda.device(value.device):
        st = handle.bevops_msda_pack_value(dt, rdt, value.data_ptr(), shapes_host.data_ptr(), out.data_ptr(),
                                           out.numel(), bs, nk, heads, ch, L, num_query, num_point,
                                           _lib.current_stream_ptr(value.device))
    _lib.check(st, "bevops_msda_pack_value")
    return PackedValue(out, value.dtype, rdt, shapes_host, shapes_dev, (bs, nk, heads, ch, L, num_query, num_point))


def multi_scale_deformable_attn_prepacked(packed, reference_points, sampling_offsets, attention_weights,
                                          scales=(1.0, 1.0, 1.0, 1.0), out=None):
    """The sampling half of the op on a `PackedValue` (bevops_msda_forward_prepacked); same arguments
    and result as `multi_scale_deformable_attn[_int8]` otherwise."""
    handle = _lib.load_library()
    bs, nk, heads, ch, L, nq, P = packed.dims
    ppg = reference_points.shape[-1] // 2
    if sampling_offsets.shape[1] != nq or attention_weights.numel() // (attention_weights.shape[0] * nq * heads * L) != P:
        raise ValueError("offsets / weights do not match the geometry the value was packed for")
    if _lib.torch_dtype_code(reference_points) != packed.ref_dtype:
        raise TypeError("reference_points dtype differs from the flavour the value was packed for")
    shared = (bs > 1 and sampling_offsets.stride(0) == 0 and attention_weights.stride(0) == 0)
    if shared:
        sampling_offsets, attention_weights = sampling_offsets[:1], attention_weights[:1]
    reference_points, sampling_offsets, attention_weights = (
        t.contiguous() for t in (reference_points, sampling_offsets, attention_weights))
    dev = packed.data.device
    if out is None:
        out = torch.empty((bs, nq, heads, ch), dtype=packed.dtype, device=dev)
    with torch.cuda.device(dev):
        st = handle.bevops_msda_forward_prepacked(
            _lib.torch_dtype_code(out), packed.data.data_ptr(), packed.data.numel(), packed.shapes_host.data_ptr(),
            reference_points.data_ptr(), packed.ref_dtype, sampling_offsets.data_ptr(), attention_weights.data_ptr(),
            out.data_ptr(), bs, nk, heads, ch, L, nq, P, ppg, float(scales[0]), float(scales[1]), float(scales[2]),
            float(scales[3]), int(shared), _lib.current_stream_ptr(dev))
    _lib.check(st, "bevops_msda_forward_prepacked")
    return out
