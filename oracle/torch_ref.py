"""Torch (CPU) port of the reference's PyTorch path for MSDA -- TEST/BASELINE
INFRASTRUCTURE ONLY (see oracle/__init__.py for who may import this).

This is the code path the reference takes when `value` is not on a GPU
(det2trt/models/modules/spatial_cross_attention.py:539-552,
temporal_self_attention.py:282-294, decoder.py:323-335), i.e. "the reference's own
PyTorch CPU path" that BASELINE.md section 3 names as the CPU baseline:

  pre-processing : det2trt/models/functions/multi_scale_deformable_attn.py:58-92
                   (offsets -> normalised locations, softmax over L*P)
  sampling       : det2trt/models/utils/trt_ops.py:4-85
                   (per level F.grid_sample(bilinear, zeros, align_corners=False),
                    weighted sum)

`bench.py` times it as `cpu_baseline` (kind "port": /root/reference does not exist
on the GPU box, so the reference file itself cannot be imported there);
tests/test_oracle_golden.py pins it to the golden vectors the reference produced.
"""
import torch
import torch.nn.functional as F


def msda_locations_and_weights(shapes, ref, off, logit):
    """functions/multi_scale_deformable_attn.py:58-92."""
    bs, nq, heads = off.shape[:3]
    L = shapes.shape[0]
    ppg = ref.shape[-1] // 2
    off = off.view(bs, nq, heads, L, -1, ppg, 2)
    P = off.shape[4] * ppg
    normalizer = torch.stack([shapes[..., 1], shapes[..., 0]], -1).to(off.dtype)
    loc = ref.view(bs, nq, 1, 1, 1, ppg, 2) + off / normalizer.view(1, 1, 1, L, 1, 1, 2)
    loc = loc.view(bs, nq, heads, L, P, 2)
    w = logit.reshape(-1, L * P).softmax(-1).view(bs, nq, heads, L, P)
    return loc, w


def msda_sample(value, shapes, loc, w):
    """utils/trt_ops.py:4-85 (value [bs, nk, heads, C] split per level)."""
    bs, _, heads, C = value.shape
    _, nq, _, L, P, _ = loc.shape
    sizes = [int(h) * int(wd) for h, wd in shapes.tolist()]
    w = w.transpose(1, 2).reshape(bs * heads, 1, nq, L * P)
    grids = 2 * loc - 1
    out = 0
    for lvl, v in enumerate(value.split(sizes, dim=1)):
        H, W = (int(x) for x in shapes[lvl])
        v = v.flatten(2).transpose(1, 2).reshape(bs * heads, C, H, W)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        s = F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False)
        s = s * w[..., lvl * P:(lvl + 1) * P]
        out = out + s.sum(-1)
    return out.view(bs, heads * C, nq).transpose(1, 2).contiguous().view(bs, nq, heads, C)


def msda(value, shapes, ref, off, logit):
    loc, w = msda_locations_and_weights(shapes, ref, off, logit)
    return msda_sample(value.float(), shapes, loc.float(), w.float())
