#!/usr/bin/env python3
"""The decoder's few-row dense layers (900 object queries) on each implementation, timed under HIP-graph replay
(functions/linear.py: graph_time_us -- an eager loop measures the ~12 us Python wrapper, not the kernel): the
no-pipeline GEMM (bevops_small_gemm_f16) against the tiled / tall-skinny / library GEMMs.  One JSON line per layer."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.functions import linear as L  # noqa: E402
from bevformer_tensorrt_amd.utils import lib as _lib  # noqa: E402

SHAPES = [("dec.in_proj", 900, 768, 256, True, False), ("dec.out_proj", 900, 256, 256, True, False),
          ("dec.offsets", 900, 64, 256, True, False), ("dec.weights", 900, 32, 256, True, False),
          ("dec.ffn.fc1", 900, 512, 256, False, True), ("dec.ffn.fc2", 900, 256, 512, True, False),
          ("head.reg0", 900, 256, 256, False, True), ("head.reg2", 900, 10, 256, False, False),
          ("can_bus.fc", 1, 256, 128, False, True)]
for name, M, N, K, has_res, relu in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    r = torch.randn(M, N, generator=g).half().cuda() if has_res else None
    us = {}
    for cand, fn in L._DENSE.items():
        try:
            fn(x, w, b, r, relu)
            us[cand] = round(L.graph_time_us(lambda: fn(x, w, b, r, relu), 16, 3), 2)
        except _lib.BevopsError:
            pass
    print(json.dumps({"layer": name, "M": M, "N": N, "K": K, "us_graph_replay": us, "best": min(us, key=us.get)}), flush=True)
