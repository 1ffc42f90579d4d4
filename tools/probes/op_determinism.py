#!/usr/bin/env python3
"""Diagnostic: one eager base frame twice; every call of the operator namespace is recorded (inputs and outputs) and the
first calls whose INPUTS agree between the two evaluations while their OUTPUTS do not are printed."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bevformer_tensorrt_amd.functions as ops  # noqa: E402
from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402

log = []


def wrap(name, fn):
    def rec(*a, **k):
        ins = [t.detach().clone() for t in list(a) + list(k.values()) if torch.is_tensor(t)]
        out = fn(*a, **k)
        outs = [out] if torch.is_tensor(out) else [t for t in (out if isinstance(out, (list, tuple)) else []) if torch.is_tensor(t)]
        log.append((name, ins, [t.detach().clone() for t in outs]))
        return out
    return rec


for name in ("dense_auto", "spatial_cross_attention_projected", "spatial_cross_attention_plan", "layer_norm",
             "multi_scale_deformable_attn", "multi_scale_deformable_attn_local", "tsa_split", "queue_mean2", "rotate_hwc",
             "linear_bias_act"):
    if hasattr(ops, name):
        setattr(ops, name, wrap(name, getattr(ops, name)))

dev, dtype = torch.device("cuda"), torch.float16
model = B.BEVFormer("base", seed=0).to(dev, dtype)
H, W = B.CONFIGS["base"]["image"]
l2i = G.synthetic_lidar2img((H, W)).to(dev)
g = torch.Generator().manual_seed(1)
img = torch.randn(1, 6, 3, H, W, generator=g).to(dev, dtype)
nq = model.bev_h * model.bev_w
prev = torch.zeros(nq, 1, B.EMBED, device=dev, dtype=dtype)
runs = []
with torch.no_grad():
    model(img, prev, torch.tensor(0.0, device=dev), torch.zeros(18, device=dev), l2i)
    for _ in range(2):
        log.clear()
        model(img, prev, torch.tensor(0.0, device=dev), torch.zeros(18, device=dev), l2i)
        torch.cuda.synchronize()
        runs.append(list(log))
        if len(runs[-1]) > 60:      # (memory: the encoder's first two layers are enough)
            runs[-1] = runs[-1][:60]
a, b = runs
print("recorded calls:", len(a), len(b))
shown = 0
for i, ((na, ia, oa), (nb, ib, ob)) in enumerate(zip(a, b)):
    same_in = len(ia) == len(ib) and all(p.shape == q.shape and torch.equal(p, q) for p, q in zip(ia, ib))
    same_out = all(torch.equal(p, q) for p, q in zip(oa, ob))
    if not same_out:
        d = max(float((p.float() - q.float()).abs().max()) for p, q in zip(oa, ob))
        print(i, na, "inputs equal:", same_in, "max |d| %.2e" % d, [tuple(t.shape) for t in ia][:4], "->", [tuple(t.shape) for t in oa])
        shown += 1
        if shown >= 6:
            break
