"""Scratch buffers the operators LEND to the library (the C ABI allocates nothing; the reference's
plugins get theirs from TensorRT through getWorkspaceSize, e.g.
modulatedDeformableConv2dPlugin.cpp:73-115).

One buffer per (purpose, device, stream): launches on one stream are ordered, so re-use is safe;
another stream gets its own buffer and never races.  A buffer that has to grow is REPLACED, not
freed: a HIP graph captured earlier may have its address baked in, so the superseded buffer stays
alive in `_RETIRED` until `release()`.  Growth is geometric, so callers of different sizes that
alternate on one stream settle on one buffer instead of thrashing."""
import torch

_LIVE = {}
_RETIRED = []


def lend(tag, nbytes, device, stream_ptr):
    key = (tag, str(device), int(stream_ptr))
    buf = _LIVE.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _RETIRED.append(buf)
            nbytes = max(int(nbytes), 2 * buf.numel())
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _LIVE[key] = buf
    return buf


def release():
    """Drop every cached buffer (only when no captured graph that used them is still alive)."""
    _LIVE.clear()
    del _RETIRED[:]
