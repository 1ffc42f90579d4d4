#!/usr/bin/env python3
"""Base SCA with the model's own geometry (6-camera rig): reference op sequence vs fused op."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bevformer_tensorrt_amd as bev
from bevformer_tensorrt_amd import geometry as G
from msda_sweep import time_call
levels = [[116, 200], [58, 100], [29, 50], [15, 25]]
nq, heads, C, P = 40000, 8, 32, 8
g = torch.Generator().manual_seed(0)
nk = sum(h * w for h, w in levels)
img = (928, 1600)
ref3d = G.reference_points_3d(200, 200, 8, 4, device="cpu")
cam, mask = G.point_sampling(ref3d, [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], G.synthetic_lidar2img(img), img)
ref = cam.reshape(6, nq, 1, 8).half().cuda()
vis = mask.reshape(6, nq, -1).any(-1)
w = (vis.float() / vis.sum(0).clamp(min=1)).half().cuda()
print(json.dumps(dict(visible_fraction=round(vis.float().mean().item(), 3))))
value = torch.randn(6, nk, heads, C, generator=g).half().cuda()
off = torch.randn(1, nq, heads, 4 * P * 2, generator=g).half().cuda()
logit = torch.randn(1, nq, heads, 4 * P, generator=g).half().cuda()
sh = torch.tensor(levels, dtype=torch.int32).cuda()
def unfused():
    q = bev.multi_scale_deformable_attn(value, sh, ref, off.expand(6, -1, -1, -1), logit.expand(6, -1, -1, -1))
    return (q.flatten(2) * w.unsqueeze(-1)).sum(0, keepdim=True)
def fused():
    return bev.spatial_cross_attention_sample(value, sh, ref, off, logit, w)
a, b = unfused(), fused()
print(json.dumps(dict(max_abs_diff=round((a.float() - b.float()).abs().max().item(), 5))))
for name, fn in (("reference sequence (shared offsets)", unfused), ("fused", fused)):
    med, mn = time_call(fn)
    print(json.dumps(dict(op=name, us_med=round(med, 1), us_min=round(mn, 1))), flush=True)
