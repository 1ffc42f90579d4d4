#!/usr/bin/env python3
"""Stress diagnostic: the inputs of encoder.0.sca of one real frame are recorded, then the module -- and, separately,
its fused sampling operator on the recorded projections -- are evaluated `--iters` times on exactly those inputs; the
number of evaluations that differ from the first.  usage: sca_module_stress.py [--model small] [--iters 2000]"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=2000)
ap.add_argument("--model", default="small")
a = ap.parse_args()
dev = torch.device("cuda")
model = B.BEVFormer(a.model, seed=0).to(dev, torch.float16)
H, W = B.CONFIGS[a.model]["image"]
l2i = G.synthetic_lidar2img((H, W)).to(dev)
img = torch.randn(1, 6, 3, H, W, generator=torch.Generator().manual_seed(1)).to(dev, torch.float16)
nq = model.bev_h * model.bev_w
prev = torch.zeros(nq, 1, B.EMBED, device=dev, dtype=torch.float16)
rec = {}
sca = model.encoder[0].sca
h = sca.register_forward_pre_hook(lambda m, args, kwargs: rec.setdefault("in", ([t.clone() if torch.is_tensor(t) else t for t in args], dict(kwargs))), with_kwargs=True)
ops_rec = {}
orig = {}
for name in ("spatial_cross_attention_sample", "spatial_cross_attention_projected", "multi_scale_deformable_attn"):
    fn = getattr(model.ops, name, None)
    if fn is None:
        continue
    orig[name] = fn

    def wrap(*x, _n=name, _f=fn, **k):
        if _n not in ops_rec and "in" in rec and "done" not in rec:
            ops_rec[_n] = ([t.clone() if torch.is_tensor(t) else t for t in x], dict(k))
        return _f(*x, **k)
    setattr(model.ops, name, wrap)
with torch.no_grad():
    model(img, prev, torch.tensor(0.0, device=dev), torch.zeros(18, device=dev), l2i)
    rec["done"] = True
    h.remove()
    args, kwargs = rec["in"]
    first = sca(*args, **kwargs).clone()
    bad = sum(int(not torch.equal(sca(*args, **kwargs), first)) for _ in range(a.iters))
    print(json.dumps({"model": a.model, "what": "encoder.0.sca module", "iters": a.iters, "differing": bad}), flush=True)
    for name, (x, k) in ops_rec.items():
        f = orig[name]
        first = f(*x, **k).clone()
        bad = sum(int(not torch.equal(f(*x, **k), first)) for _ in range(a.iters))
        print(json.dumps({"model": a.model, "what": name, "iters": a.iters, "differing": bad}), flush=True)
