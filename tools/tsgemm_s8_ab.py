#!/usr/bin/env python3
"""The persistent int8 GEMM (bevops_tsgemm_s8) against the tiled int8 GEMM on the int8 chain's 1x1 convolutions of
ResNet stages 2-4 (base and small), under HIP-graph replay, interleaved.  One JSON line per layer."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_tensorrt_amd.functions import int8_chain as C  # noqa: E402
from bevformer_tensorrt_amd.functions import linear as L  # noqa: E402

SHAPES = [("s2.conv3", 139200, 512, 128, True), ("s3.conv1", 34800, 256, 1024, False),
          ("s3.conv3", 34800, 1024, 256, True), ("s3.down", 34800, 1024, 512, False),
          ("s4.conv1", 8700, 512, 2048, False), ("s4.conv3", 8700, 2048, 512, True),
          ("small.s3.conv1", 22080, 256, 1024, False), ("small.s3.conv3", 22080, 1024, 256, True)]

for name, M, N, K, has_res in SHAPES:
    g = torch.Generator().manual_seed(0)
    a8 = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).cuda()
    w8 = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).cuda()
    r8 = torch.randint(-127, 128, (M, N), generator=g, dtype=torch.int8).cuda() if has_res else None
    bf = torch.randn(N, generator=g).cuda()
    fn = lambda: C.linear_int8_chain(a8, 0.02, w8, 0.001, bf, r8, 0.03, True, torch.int8, 0.05)  # noqa: E731
    res = {False: [], True: []}
    for _ in range(3):
        for v in (False, True):
            C._TS_S8["enabled"] = v
            try:
                res[v].append(round(L.graph_time_us(fn), 2))
            finally:
                C._TS_S8["enabled"] = None
    byt = M * K + N * K + M * N * (2 if has_res else 1)
    t, p = sorted(res[False])[1], sorted(res[True])[1]
    print(json.dumps({"layer": name, "M": M, "N": N, "K": K, "us_tile": t, "us_tsgemm_s8": p,
                      "TBs_tile": round(byt / t / 1e6, 2), "TBs_tsgemm_s8": round(byt / p / 1e6, 2)}), flush=True)
