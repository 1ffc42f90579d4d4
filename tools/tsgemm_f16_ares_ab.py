#!/usr/bin/env python3
"""STAGED kernel's A/B (run it once bevops_tsgemm_f16_ares has passed its gated test on the device): the fp16 conv3 /
short-K layers of ResNet on every dense implementation -- hipBLASLt entry, tiled GEMM, persistent GEMM, and the two
plans of the A-resident persistent GEMM -- under HIP-graph replay, interleaved.  One JSON line per layer."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_tensorrt_amd.functions import linear as L  # noqa: E402

SHAPES = [("s2.conv3", 139200, 512, 128), ("s3.conv3", 34800, 1024, 256), ("small.s3.conv3", 22080, 1024, 256),
          ("small.s2.conv3", 88320, 512, 128),
          # the encoder's K = 256 layers fit the kernel's domain as well (one 160-row tile per CU at 40 000 rows)
          ("enc.output_proj", 40000, 256, 256), ("sca.offsets / ffn.fc1", 40000, 512, 256), ("tsa.value_proj", 80000, 256, 256)]
for name, M, N, K in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    r = torch.randn(M, N, generator=g).half().cuda()
    fns = {"blaslt": lambda: L.linear_bias_act(x, w, b, r, True), "tile": lambda: L.tile_gemm(x, w, b, r, True),
           "tsgemm": lambda: L.tsgemm(x, w, b, r, True), "ares_plan0": lambda: L.tsgemm_ares(x, w, b, r, True, 0),
           "ares_plan1": lambda: L.tsgemm_ares(x, w, b, r, True, 1)}
    res = {k: [] for k in fns}
    for _ in range(3):
        for k, fn in fns.items():
            res[k].append(round(L.graph_time_us(fn), 2))
    byt = (M * K + N * K + 2 * M * N) * 2
    print(json.dumps({"layer": name, "M": M, "N": N, "K": K, "us": {k: sorted(v)[1] for k, v in res.items()},
                      "byte_floor_us_at_8TBs": round(byt / 8e6, 1)}), flush=True)
