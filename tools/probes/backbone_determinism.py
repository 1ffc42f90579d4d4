#!/usr/bin/env python3
"""Diagnostic: where do two evaluations of the same base frame first differ INSIDE the channels-last backbone?  (Its
blocks are called through forward_nhwc, which module hooks do not see: tools/probes/frame_determinism.py reports the
first consumer of the image features instead.)  Every bottleneck's conv1 / conv2 / conv3 output and every FPN output of
two eager runs are compared in execution order."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bevformer_tensorrt_amd import bevformer as B  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "base"
dev, dtype = torch.device("cuda"), torch.float16
model = B.BEVFormer(name, seed=0).to(dev, dtype)
H, W = B.CONFIGS[name]["image"]
img = torch.randn(1, 6, 3, H, W, generator=torch.Generator().manual_seed(1)).to(dev, dtype)
log = []
conv1x1, conv, dcn_fwd = B._conv1x1_nhwc, B._conv_nhwc, B.DCNv2Pack.forward_nhwc


def wrap(fn, tag):
    def inner(*a, **k):
        y = fn(*a, **k)
        shape = tuple(y.shape)
        log.append((tag, shape, y.detach().clone()))
        return y
    return inner


B._conv1x1_nhwc, B._conv_nhwc = wrap(conv1x1, "conv1x1"), wrap(conv, "conv")
B.DCNv2Pack.forward_nhwc = wrap(dcn_fwd, "dcn")
runs = []
with torch.no_grad():
    model.extract_feat(img)          # warm-up: measured dispatch
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    for i in range(N):
        log.clear()
        outs = model.extract_feat(img)
        torch.cuda.synchronize()
        if i == 0 or any(not torch.equal(x, y[2]) for (_, _, x), y in zip(log, runs[0][0])):
            runs.append((list(log), [t.clone() for t in outs]))     # (kept: the first run and every run that differs from it)
print("runs:", N, "of which differ from run 0:", len(runs) - 1)
for other in range(1, min(len(runs), 6)):
    a, b = runs[0], runs[other]
    print("run 0 vs differing run %d: recorded calls %d" % (other, len(a[0])))
    shown = 0
    for i, ((ta, sa, xa), (tb, sb, xb)) in enumerate(zip(a[0], b[0])):
        if not torch.equal(xa, xb):
            d = (xa.float() - xb.float()).abs()
            print("  differs: call %d %s %s max %.2e, %d elements" % (i, ta, sa, float(d.max()), int((d > 0).sum())))
            shown += 1
            if shown >= 6:
                break
    print("  pyramid equal:", [bool(torch.equal(p, q)) for p, q in zip(a[1], b[1])])
