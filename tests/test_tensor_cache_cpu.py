"""_TensorCache (functions/multi_scale_deformable_attn.py): a hit needs the same tensor OBJECT and
an unchanged stamp (version, dtype, device, data pointer, shape) -- `param.data = ...`,
nn.Module.half() / .to() keep the object and the version but change the storage."""
import torch

from bevformer_tensorrt_amd.functions.multi_scale_deformable_attn import _TensorCache


def test_hit_and_invalidation():
    c = _TensorCache()
    lin = torch.nn.Linear(4, 4)
    w = lin.weight
    assert c.get(w) is None
    c.put(w, "packed-fp32")
    assert c.get(w) == "packed-fp32"
    with torch.no_grad():
        w.add_(1.0)                      # in-place edit: version bump
    assert c.get(w) is None
    c.put(w, "again")
    lin.half()                           # same Parameter object, new storage, same version
    assert lin.weight is w and c.get(w) is None
    c.put(w, "packed-fp16")
    w.data = torch.zeros(4, 4, dtype=torch.float16)   # storage swap, dtype unchanged
    assert c.get(w) is None
    other = torch.nn.Parameter(torch.zeros(4, 4))
    assert c.get(other) is None
