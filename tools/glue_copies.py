#!/usr/bin/env python3
"""Where do the frame's framework copies come from?  Wraps the tensor methods that can materialise a copy
(contiguous / clone / copy_ / to / cat / repeat / mean / permute+contiguous) during ONE eager base frame and prints per
call site (file:line inside bevformer_tensorrt_amd) the number of real copies and their megabytes."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402

name = "base"
dev, dtype = torch.device("cuda"), torch.float16
model = B.BEVFormer(name, seed=0).to(dev, dtype)
H, W = B.CONFIGS[name]["image"]
l2i = G.synthetic_lidar2img((H, W)).to(dev)
g = torch.Generator().manual_seed(0)
img = torch.randn(1, 6, 3, H, W, generator=g).to(dev, dtype)
r = B.FrameRunner(model, dev, dtype)
r.image_buffer.copy_(img)
can = torch.zeros(18)
for i in range(3):
    can[0], can[-1] = 0.5 * i, 0.8 * i
    r.step(r.image_buffer, can, l2i, "s")
torch.cuda.synchronize()
log = collections.defaultdict(lambda: [0, 0.0])


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "bevformer_tensorrt_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"


def note(kind, nbytes):
    k = (kind, site())
    log[k][0] += 1
    log[k][1] += nbytes / 1e6


_contig, _clone, _copy, _to, _cat, _mean, _repeat = (torch.Tensor.contiguous, torch.Tensor.clone, torch.Tensor.copy_,
                                                     torch.Tensor.to, torch.cat, torch.mean, torch.Tensor.repeat)


def contiguous(self, *a, **k):
    mf = k.get("memory_format", a[0] if a else torch.contiguous_format)
    if self.is_cuda and not self.is_contiguous(memory_format=mf):
        note("contiguous", self.numel() * self.element_size())
    return _contig(self, *a, **k)


def clone(self, *a, **k):
    if self.is_cuda:
        note("clone", self.numel() * self.element_size())
    return _clone(self, *a, **k)


def copy_(self, src, *a, **k):
    if self.is_cuda:
        note("copy_", self.numel() * self.element_size())
    return _copy(self, src, *a, **k)


def to(self, *a, **k):
    out = _to(self, *a, **k)
    if out.is_cuda and out.data_ptr() != self.data_ptr():
        note("to", out.numel() * out.element_size())
    return out


def cat(ts, *a, **k):
    out = _cat(ts, *a, **k)
    if out.is_cuda:
        note("cat", out.numel() * out.element_size())
    return out


def mean(x, *a, **k):
    if x.is_cuda:
        note("mean", x.numel() * x.element_size())
    return _mean(x, *a, **k)


def repeat(self, *a, **k):
    out = _repeat(self, *a, **k)
    if out.is_cuda:
        note("repeat", out.numel() * out.element_size())
    return out


torch.Tensor.contiguous, torch.Tensor.clone, torch.Tensor.copy_, torch.Tensor.to = contiguous, clone, copy_, to
torch.cat, torch.mean, torch.Tensor.repeat = cat, mean, repeat
can[0], can[-1] = 2.0, 3.2
r.step(r.image_buffer, can, l2i, "s")
torch.cuda.synchronize()
print("kind        site                              copies      MB")
for (kind, st), (n, mb) in sorted(log.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{kind:11s} {st:32s} {n:6d} {mb:9.2f}")
