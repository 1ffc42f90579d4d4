#!/usr/bin/env python3
"""Stress diagnostic for a RARE run-to-run difference (one of ~10 small frames under BEVOPS_OWN_ENCODER=1 differed in
its last encoder FFN): every hand-written dense kernel on the encoder's shapes of small / base, `--iters` evaluations of
the same operands each, the number of evaluations that differ from the first; then `--frames` pairs of whole frames.
usage: own_kernel_stress.py [--iters 300] [--frames 12] [--model small]"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402
from bevformer_tensorrt_amd.functions import linear as L  # noqa: E402
from bevformer_tensorrt_amd.utils import lib as _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--model", default="small")
a = ap.parse_args()
g = torch.Generator().manual_seed(0)
dev = torch.device("cuda")
SHAPES = [(M, N, K, res, relu) for M in (22500, 40000)
          for N, K, res, relu in ((512, 256, False, True), (256, 256, False, False), (192, 256, False, False),
                                  (256, 256, True, False), (256, 512, True, False))]
SHAPES += [(5520, 256, 256, False, False), (22500, 128, 256, False, False), (22500, 64, 256, False, False),   # small's SCA
           (45000, 256, 256, False, False), (900, 256, 256, True, False), (900, 768, 256, True, False)]
for M, N, K, res, relu in SHAPES:
    if True:
        x = torch.randn(M, K, generator=g).half().to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).half().to(dev)
        b = torch.randn(N, generator=g).half().to(dev)
        r = torch.randn(M, N, generator=g).half().to(dev) if res else None
        lw, lb = torch.randn(N, generator=g).half().to(dev), torch.randn(N, generator=g).half().to(dev)
        cands = {k: (lambda x, w, b, r, relu, f=f: f(x, w, b, r, relu)) for k, f in L._DENSE.items() if k in ("tile", "tsgemm", "small")}
        if N == 256 and res:
            cands["tsgemm_ln"] = lambda x, w, b, r, relu: bev.tsgemm_ln(x, w, b, r, lw, lb, 1e-5)
        row = {}
        for name, fn in cands.items():
            try:
                first = fn(x, w, b, r, relu).clone()
            except _lib.BevopsError:
                continue
            bad = 0
            for i in range(a.iters):
                y = fn(x, w, b, r, relu)
                if i % 3 == 0:      # (vary what runs between two evaluations)
                    x.add_(0)
                bad += int(not torch.equal(y, first))
            row[name] = bad
        print(json.dumps({"M": M, "N": N, "K": K, "residual": res, "relu": relu, "iters": a.iters, "differing_evaluations": row}), flush=True)

model = B.BEVFormer(a.model, seed=0).to(dev, torch.float16)
H, W = B.CONFIGS[a.model]["image"]
l2i = G.synthetic_lidar2img((H, W)).to(dev)
img = torch.randn(1, 6, 3, H, W, generator=torch.Generator().manual_seed(1)).to(dev, torch.float16)
nq = model.bev_h * model.bev_w
prev = torch.zeros(nq, 1, B.EMBED, device=dev, dtype=torch.float16)
log = []
for mod_name, mod in model.named_modules():
    if mod_name and type(mod).__module__.startswith("bevformer_tensorrt_amd"):
        mod.register_forward_hook(lambda m, i, o, n=mod_name: log.append((n, o.detach().clone())) if torch.is_tensor(o) else None)
# (the backbone's blocks run through forward_nhwc, which hooks do not see: record what encoder.0.sca is handed)
def _pre(m, args):
    for i, t in enumerate(args[:4]):
        if torch.is_tensor(t):
            log.append(("encoder.0.sca.input%d" % i, t.detach().clone()))


model.encoder[0].sca.register_forward_pre_hook(_pre)
def _stages(x, ops, _bb=model.backbone):
    outs = []
    for i, st in enumerate(_bb.stages):
        for j, blk in enumerate(st):
            DETAIL["on"] = STAGE_DETAIL == i
            DETAIL["tag"] = "stage%d.block%d" % (i, j)
            x = blk.forward_nhwc(x, ops)
            DETAIL["on"] = False
        log.append(("stage%d" % i, x.detach().clone()))
        if i in _bb.out_indices:
            outs.append(x)
    return outs


STAGE_DETAIL = int(os.environ.get("STAGE_DETAIL", "-1"))
DETAIL = {"on": False, "tag": "", "n": 0}


def _wrap(fn, kind):
    def inner(*x, **k):
        y = fn(*x, **k)
        if DETAIL["on"]:
            log.append(("%s.%s%s" % (DETAIL["tag"], kind, tuple(y.shape[1:])), y.detach().clone()))
        return y
    return inner


B._conv1x1_nhwc, B._conv_nhwc = _wrap(B._conv1x1_nhwc, "conv1x1"), _wrap(B._conv_nhwc, "conv")
model.backbone._stages_nhwc = _stages
_feat = model.extract_feat
model.extract_feat = lambda *x, **k: [log.append(("extract_feat.level%d" % i, t.detach().clone())) or t for i, t in enumerate(_feat(*x, **k))]
with torch.no_grad():
    ref = None
    for f in range(a.frames):
        log.clear()
        model(img, prev, torch.tensor(0.0, device=dev), torch.zeros(18, device=dev), l2i)
        torch.cuda.synchronize()
        if ref is None:
            ref = list(log)
            continue
        diff = [(n, float((p.float() - q.float()).abs().max()), int((p != q).sum())) for (n, p), (_, q) in zip(ref, log) if not torch.equal(p, q)]
        print(json.dumps({"frame": f, "own_encoder": B._OWN_ENCODER["enabled"], "first_differing_modules": diff[:4]}), flush=True)
