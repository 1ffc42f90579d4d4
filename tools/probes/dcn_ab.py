#!/usr/bin/env python3
"""Interleaved A/B of bevops_mdconv_set_variant values at the two ResNet-101 DCN shapes (fp16 and int8):
usage: dcn_ab.py VARIANT_A VARIANT_B [rounds] -- the two variants alternate, so clock / cache warm-up drifts
hit both alike."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tools/ (msda_sweep)
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402
from dcn_int8_time import med  # noqa: E402


def main():
    va, vb = int(sys.argv[1]), int(sys.argv[2])
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    lib = load_library()
    g = torch.Generator().manual_seed(0)
    for (B, C, H, W) in ((6, 256, 58, 100), (6, 512, 29, 50)):
        xq = torch.randint(-127, 128, (B, C, H, W), generator=g, dtype=torch.int8).cuda()
        oq = torch.randint(-127, 128, (B, 18, H, W), generator=g, dtype=torch.int8).cuda()
        mq = torch.randint(0, 128, (B, 9, H, W), generator=g, dtype=torch.int8).cuda()
        wq = torch.randint(-127, 128, (C, C, 3, 3), generator=g, dtype=torch.int8).cuda()
        bq = torch.zeros(C).cuda()
        xh = torch.randn(B, C, H, W, generator=g).half().cuda()
        oh = torch.randn(B, 18, H, W, generator=g).half().cuda()
        mh = torch.rand(B, 9, H, W, generator=g).half().cuda()
        wh = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).half().cuda()
        bh = torch.zeros(C).half().cuda()
        calls = {"int8": lambda: bev.modulated_deformable_conv2d_int8(xq, oq, mq, wq, bq, 0.02, 0.03, 1 / 127, 0.01, 0.05, 1, 1, 1, 1, 1),
                 "f16": lambda: bev.modulated_deformable_conv2d(xh, oh, mh, wh, bh, 1, 1, 1, 1, 1)}
        for name, fn in calls.items():
            res = {va: [], vb: []}
            for _ in range(rounds):
                for v in (va, vb):
                    lib.bevops_mdconv_set_variant(v)
                    try:
                        res[v].append(med(fn, iters=20, warm=5)[0])
                    finally:
                        lib.bevops_mdconv_set_variant(0)
            print(json.dumps({"op": "dcn_" + name, "shape": [B, C, H, W], "variants": [va, vb],
                              "us_a": res[va], "us_b": res[vb]}), flush=True)


if __name__ == "__main__":
    main()
