#!/usr/bin/env python3
"""The ResNet stem at the base frame's shape (6 x 3 x 928 x 1600 fp16 -> 6 x 232 x 400 x 64): the one-kernel form
(bevops_stem_conv_pool; pooling neighbours through DPP wave shifts / ds_bpermute; fp16 and int8 output) against the form
it replaces (images to channels-last -> library convolution -> bevops_bias_relu_maxpool_nhwc[_int8]), under HIP-graph
replay, interleaved.  One JSON line.  --once K: K plain launches of each (for rocprofv3)."""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.functions import int8_chain as C  # noqa: E402
from bevformer_tensorrt_amd.functions import linear as L  # noqa: E402
from bevformer_tensorrt_amd.utils import lib as _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="6,928,1600")
ap.add_argument("--once", type=int, default=0)
args = ap.parse_args()
n, h, w = (int(v) for v in args.shape.split(","))
g = torch.Generator().manual_seed(0)
x = torch.randn(n, 3, h, w, generator=g).half().cuda()
wt = (torch.randn(64, 3, 7, 7, generator=g) / 12).half().cuda()
wt_cl = wt.contiguous(memory_format=torch.channels_last)
b = torch.randn(64, generator=g).half().cuda()
handle = _lib.load_library()
s8 = 0.05


def two_pass(int8=False):
    def fn():
        y = F.conv2d(x.contiguous(memory_format=torch.channels_last), wt_cl, None, 2, 3)
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        return C.bias_relu_maxpool_nhwc_int8(y, b, s8) if int8 else bev.bias_relu_maxpool_nhwc(y, b)
    return fn


def fused(variant, int8=False):
    def fn():
        handle.bevops_stem_set_variant(variant)
        return bev.stem_conv_pool(x, wt, b, s8 if int8 else None)
    return fn


fns = {"two_pass": two_pass(), "fused_dpp": fused(0), "fused_bpermute": fused(1), "two_pass_int8": two_pass(True),
       "fused_dpp_int8": fused(0, True)}
ref = fns["two_pass"]().float()
for name in ("fused_dpp", "fused_bpermute"):
    err = (fns[name]().float() - ref).abs()
    assert bool((err <= 4e-3 * ref.abs() + 4e-3).all()), (name, float(err.max()))
d8 = (fns["fused_dpp_int8"]().float() - fns["two_pass_int8"]().float()).abs()
assert float(d8.max()) <= 1, float(d8.max())
if args.once:
    for fn in fns.values():
        for _ in range(args.once):
            fn()
    torch.cuda.synchronize()
    sys.exit(0)
res = {k: [] for k in fns}
for _ in range(3):
    for name, fn in fns.items():
        res[name].append(round(L.graph_time_us(fn, 4, 3), 2))
med = {k: sorted(v)[1] for k, v in res.items()}
hc, wc = (h - 1) // 2 + 1, (w - 1) // 2 + 1
hp, wp = (hc - 1) // 2 + 1, (wc - 1) // 2 + 1
flop = 2.0 * n * hc * wc * 64 * 147
byt = n * 3 * h * w * 2 + n * hp * wp * 64 * 2
print(json.dumps({"what": "ResNet stem (conv 7x7/2 + shift + ReLU + max-pool 3/2), planar fp16 images -> pooled channels-last",
                  "shape": [n, 3, h, w], "us": med, "int8_max_lsb_diff": float(d8.max()),
                  "useful_gflop": round(flop / 1e9, 2), "algorithmic_MB": round(byt / 1e6, 1),
                  "fused_dpp": {"TFLOPs": round(flop / med["fused_dpp"] / 1e6, 1),
                                "GBs": round(byt / med["fused_dpp"] / 1e3, 1)}}), flush=True)
handle.bevops_stem_set_variant(0)
