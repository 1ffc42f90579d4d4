"""Decoder reference-point refinement (det2trt/models/modules/decoder.py:93-103) as one bit-exact launch
(bevops_refine_reference_points, csrc/refine.hip).  Not a reference plugin: the reference's engine evaluates these
element-wise layers itself."""
import torch

from ..utils import lib as _lib

_TABLES = {}      # device -> (log table, sigmoid table): fp16 [65536], entry i = f(binary16 value with bit pattern i)


def _tables(device):
    """log and sigmoid of EVERY binary16 value, evaluated once per process by the framework's own element-wise kernels:
    inside the fused launch the two transcendental steps are then the framework's results bit for bit (this library's
    compiler does not produce the framework binary's logf: csrc/refine.hip)."""
    key = str(device)
    hit = _TABLES.get(key)
    if hit is None:
        assert not torch.cuda.is_current_stream_capturing(), "refine tables must be built before stream capture"
        v = torch.arange(0, 1 << 16, dtype=torch.int32, device=device).to(torch.int16).view(torch.float16)
        hit = _TABLES[key] = (torch.log(v).contiguous(), torch.sigmoid(v).contiguous())
    return hit


def refine_reference_points(tmp, reference_points):
    """tmp [1, n, >= 5] (regression branch output), reference_points [1, n, 3], both fp16 on the device ->
    (new_reference_points [1, n, 3], reference_xy [1, n, 1, 2] contiguous): sigmoid(tmp[..., (0, 1, 4)] +
    inverse_sigmoid(reference_points)) with the framework's rounding after every step."""
    assert tmp.is_cuda and tmp.dtype == torch.float16 and reference_points.dtype == torch.float16
    n = reference_points.shape[-2]
    assert reference_points.shape[-1] == 3 and tmp.shape[-2] == n and tmp.shape[-1] >= 5
    t2 = tmp.reshape(n, tmp.shape[-1])
    if not t2.is_contiguous():
        t2 = t2.contiguous()
    r2 = reference_points.reshape(n, 3)
    if not r2.is_contiguous():
        r2 = r2.contiguous()
    new = torch.empty((1, n, 3), dtype=torch.float16, device=tmp.device)
    xy = torch.empty((1, n, 1, 2), dtype=torch.float16, device=tmp.device)
    log_t, sig_t = _tables(tmp.device)
    handle = _lib.load_library()
    with torch.cuda.device(tmp.device):
        st = handle.bevops_refine_reference_points(_lib.F16, t2.data_ptr(), r2.data_ptr(), new.data_ptr(), xy.data_ptr(), n,
                                                   t2.shape[-1], log_t.data_ptr(), sig_t.data_ptr(),
                                                   _lib.current_stream_ptr(tmp.device))
    _lib.check(st, "bevops_refine_reference_points")
    return new, xy


def decode_boxes(regs, refs, pc_range):
    """The head's box decoding (bevformer_head.py:247-282) on stacked levels: regs [..., 10], refs [..., 3] fp16 on the
    device -> boxes [..., 10]: columns 0 / 1 / 4 = sigmoid(reg + inverse_sigmoid(ref)) * (pc_range[3 + a] - pc_range[a]) +
    pc_range[a] with the framework's rounding after every step (mmdet's inverse_sigmoid), the other columns copied."""
    assert regs.is_cuda and regs.dtype == torch.float16 and refs.dtype == torch.float16
    assert regs.shape[-1] == 10 and refs.shape[-1] == 3 and regs.shape[:-1] == refs.shape[:-1]
    r2 = regs.reshape(-1, 10)
    f2 = refs.reshape(-1, 3)
    r2 = r2 if r2.is_contiguous() else r2.contiguous()
    f2 = f2 if f2.is_contiguous() else f2.contiguous()
    out = torch.empty_like(r2)
    log_t, sig_t = _tables(regs.device)
    s = [float(pc_range[3 + a] - pc_range[a]) for a in range(3)]
    o = [float(pc_range[a]) for a in range(3)]
    handle = _lib.load_library()
    with torch.cuda.device(regs.device):
        st = handle.bevops_decode_boxes(_lib.F16, r2.data_ptr(), f2.data_ptr(), out.data_ptr(), r2.shape[0], s[0], o[0], s[1], o[1],
                                        s[2], o[2], log_t.data_ptr(), sig_t.data_ptr(), _lib.current_stream_ptr(regs.device))
    _lib.check(st, "bevops_decode_boxes")
    return out.view(regs.shape)
