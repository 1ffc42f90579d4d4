#!/usr/bin/env python3
"""Diagnostic: the SCA calls of one eager base frame, each replayed on the visibility plan (direct stores / scratch) and
without a plan; reports the first call whose results differ and what the differing rows have in common."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402
from bevformer_tensorrt_amd.functions import spatial_cross_attention as S  # noqa: E402
from bevformer_tensorrt_amd.utils import lib as L  # noqa: E402

dev, dtype = torch.device("cuda"), torch.float16
model = B.BEVFormer("base", seed=0).to(dev, dtype)
H, W = B.CONFIGS["base"]["image"]
l2i = G.synthetic_lidar2img((H, W)).to(dev)
calls = []
orig = S.spatial_cross_attention_projected


def rec(*a, **k):
    calls.append(([t.clone() if torch.is_tensor(t) else t for t in a], dict(k)))
    return orig(*a, **k)


import bevformer_tensorrt_amd.functions as _ops  # noqa: E402
_ops.spatial_cross_attention_projected = rec      # (the model's operator namespace is this module)
g = torch.Generator().manual_seed(1)
img = torch.randn(1, 6, 3, H, W, generator=g).to(dev, dtype)
nq = model.bev_h * model.bev_w
prev = torch.zeros(nq, 1, B.EMBED, device=dev, dtype=dtype)
with torch.no_grad():
    model(img, prev, torch.tensor(0.0, device=dev), torch.zeros(18, device=dev), l2i)
print("recorded SCA calls:", len(calls))
h = L.load_library()
for ci, (a, k) in enumerate(calls):
    bm = a[7]
    plan = k.get("plan")
    if plan is None:
        plan = S.spatial_cross_attention_plan(bm)
    want = orig(*a)
    res = {}
    for name, v in (("direct", 3012), ("scratch", 3013)):
        h.bevops_msda_set_variant(v)
        res[name] = orig(*a, plan=plan)
    h.bevops_msda_set_variant(3012)
    torch.cuda.synchronize()
    for name, got in res.items():
        bad = (got != want).any(-1).flatten()
        print(f"call {ci} {name}: rows differing {int(bad.sum())} of {bad.numel()}, max |d| {float((got.float() - want.float()).abs().max()):.3e}")
        if bad.any():
            rows = torch.nonzero(bad).flatten()[:8]
            seen = (bm.reshape(bm.shape[0], -1) != 0)
            print("   rows", rows.tolist(), "cameras seeing them", seen[:, rows].sum(0).tolist(),
                  "weights", bm.reshape(bm.shape[0], -1)[:, rows].max(0).values.tolist())
            print("   want finite", bool(torch.isfinite(want.float()).all()), "got finite", bool(torch.isfinite(got.float()).all()))
            r0 = int(rows[0])
            print("   want", want[0, r0, :6].tolist(), "got", got[0, r0, :6].tolist())
    if ci >= 1:
        break
