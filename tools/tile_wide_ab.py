#!/usr/bin/env python3
"""128-byte against 64-byte k-steps of the int8 chain's plain GEMMs (bevops_tile_gemm_set_variant 256 / 255) on the
ResNet layer shapes of BEVFormer-base / small, timed under HIP-graph replay, interleaved.  One JSON line per layer."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_tensorrt_amd.functions import int8_chain as C  # noqa: E402
from bevformer_tensorrt_amd.functions import linear as L  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402

# name, M, N, K, int8 identity rows
SHAPES = [("s1.conv3", 556800, 256, 64, True), ("s2.conv1", 139200, 128, 512, False),
          ("s2.conv3", 139200, 512, 128, True), ("s3.conv1", 34800, 256, 1024, False),
          ("s3.conv3", 34800, 1024, 256, True), ("s3.down", 34800, 1024, 512, False),
          ("s4.conv1", 8700, 512, 2048, False), ("s4.conv1_first", 8700, 512, 1024, False),
          ("s4.conv3", 8700, 2048, 512, True),
          ("small.s3.conv1", 22080, 256, 1024, False), ("small.s3.conv3", 22080, 1024, 256, True),
          ("small.s4.conv1", 5520, 512, 2048, False), ("small.s4.conv3", 5520, 2048, 512, True)]

lib = load_library()
for name, M, N, K, has_res in SHAPES:
    g = torch.Generator().manual_seed(0)
    a8 = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).cuda()
    w8 = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).cuda()
    r8 = torch.randint(-127, 128, (M, N), generator=g, dtype=torch.int8).cuda() if has_res else None
    bf = torch.randn(N, generator=g).cuda()
    fn = lambda: C.linear_int8_chain(a8, 0.02, w8, 0.001, bf, r8, 0.03, True, torch.int8, 0.05)  # noqa: E731
    res = {255: [], 256: []}
    for _ in range(3):
        for v in (255, 256):
            prev = lib.bevops_tile_gemm_set_variant(v)
            try:
                res[v].append(round(L.graph_time_us(fn), 2))
            finally:
                lib.bevops_tile_gemm_set_variant(prev)
    byt = M * K + N * K + M * N * (2 if has_res else 1)
    n, w = sorted(res[255])[1], sorted(res[256])[1]
    print(json.dumps({"layer": name, "M": M, "N": N, "K": K, "us_step64": n, "us_step128": w,
                      "TBs_step64": round(byt / n / 1e6, 2), "TBs_step128": round(byt / w / 1e6, 2)}), flush=True)
