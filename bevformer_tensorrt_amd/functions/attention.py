"""self_attention_qkv -- the decoder's self-attention over its object queries (mmcv MultiheadAttention inside the
decoder layer, det2trt/models/modules/decoder.py:52-112) as one launch on the matrix cores.  Not one of the reference's
plugin functions (TensorRT fuses the attention itself)."""
import torch

from ..utils import lib as _lib


def self_attention_qkv(qkv, scale=None):
    """
    Args:
        qkv: (num_query, 3, num_heads, 32) fp16 -- q, k, v of every head as the in-projection leaves them
        scale: softmax scale (default 1 / sqrt(32))
    Returns: (num_query, num_heads * 32) = concat over heads of softmax(scale q k^T) v
    Raises BevopsError (NOT_SUPPORTED) for another head width or more queries than the kernel stages in LDS."""
    assert qkv.is_cuda and qkv.dtype == torch.float16 and qkv.ndim == 4 and qkv.shape[1] == 3
    n, _, heads, hd = qkv.shape
    qkv = qkv.contiguous()
    out = torch.empty((n, heads * hd), dtype=qkv.dtype, device=qkv.device)
    if n == 0:
        return out
    handle = _lib.load_library()
    with torch.cuda.device(qkv.device):
        st = handle.bevops_mha_selfattn_f16(qkv.data_ptr(), out.data_ptr(), n, heads, hd,
                                            float(scale if scale is not None else hd ** -0.5),
                                            _lib.current_stream_ptr(qkv.device))
    _lib.check(st, "bevops_mha_selfattn_f16")
    return out
