"""utils/workspace.py -- the scratch buffers the Python operators lend to the library: one per (purpose,
device, stream); a buffer that must grow is replaced, never freed (a captured graph may hold its address);
growth is geometric; release() drops everything."""
import torch

from bevformer_tensorrt_amd.utils import workspace as W


def setup_function(_):
    W.release()


def test_one_buffer_per_purpose_device_and_stream():
    a = W.lend("msda", 1000, "cpu", 0)
    assert a.numel() >= 1000 and a.dtype == torch.uint8
    assert W.lend("msda", 500, "cpu", 0) is a            # smaller request: same buffer
    assert W.lend("msda", 1000, "cpu", 1) is not a       # another stream: its own buffer
    assert W.lend("linear", 1000, "cpu", 0) is not a     # another purpose: its own buffer


def test_growth_is_geometric_and_keeps_the_superseded_buffer_alive():
    a = W.lend("msda", 1000, "cpu", 0)
    ptr = a.data_ptr()
    b = W.lend("msda", 1001, "cpu", 0)
    assert b is not a and b.numel() >= 2000              # grows by at least 2x: alternating callers settle
    assert any(t.data_ptr() == ptr for t in W._RETIRED)  # the old address stays valid (captured graphs)
    assert W.lend("msda", 1800, "cpu", 0) is b
    W.release()
    assert not W._LIVE and not W._RETIRED
