#!/usr/bin/env python3
"""Diagnostic: one eager base frame evaluated twice on the same inputs; the outputs of every module are compared in
execution order and the first modules whose outputs differ between the two evaluations are printed (a kernel that is
not run-to-run deterministic: an atomic split-K of a library GEMM, or a race)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "base"
dev, dtype = torch.device("cuda"), torch.float16
model = B.BEVFormer(name, seed=0).to(dev, dtype)
H, W = B.CONFIGS[name]["image"]
l2i = G.synthetic_lidar2img((H, W)).to(dev)
g = torch.Generator().manual_seed(1)
img = torch.randn(1, 6, 3, H, W, generator=g).to(dev, dtype)
nq = model.bev_h * model.bev_w
prev = torch.zeros(nq, 1, B.EMBED, device=dev, dtype=dtype)
log = []


def hook(mod_name):
    def fn(mod, inp, out):
        ts = [out] if torch.is_tensor(out) else [t for t in (out if isinstance(out, (list, tuple)) else []) if torch.is_tensor(t)]
        log.append((mod_name, type(mod).__name__, [t.detach().clone() for t in ts]))
    return fn


for mod_name, mod in model.named_modules():
    if mod_name and type(mod).__module__.startswith("bevformer_tensorrt_amd"):
        mod.register_forward_hook(hook(mod_name))
runs = []
with torch.no_grad():
    model(img, prev, torch.tensor(0.0, device=dev), torch.zeros(18, device=dev), l2i)      # warm-up: measured dispatch
    for _ in range(2):
        log.clear()
        out = model(img, prev, torch.tensor(0.0, device=dev), torch.zeros(18, device=dev), l2i)
        torch.cuda.synchronize()
        runs.append((list(log), [t.clone() for t in out]))
a, b = runs
print("hooked module calls:", len(a[0]), len(b[0]))
shown = 0
for (na, ta, xa), (nb, tb, xb) in zip(a[0], b[0]):
    bad = [float((p.float() - q.float()).abs().max()) for p, q in zip(xa, xb) if not torch.equal(p, q)]
    if bad:
        print("differs:", na, ta, ["%.2e" % v for v in bad], [tuple(p.shape) for p in xa][:2])
        shown += 1
        if shown >= 12:
            break
print("final outputs equal:", [bool(torch.equal(p, q)) for p, q in zip(a[1], b[1])])
