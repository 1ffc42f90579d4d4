#!/usr/bin/env python3
"""tsgemm vs the library GEMM paths on the dense-layer shapes of BEVFormer-base (interleaved, HIP events)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd import bevformer as B  # noqa: E402
from msda_sweep import time_call  # noqa: E402

SHAPES = [("s3.conv1", 34800, 256, 1024, False, True), ("s3.conv3", 34800, 1024, 256, True, True),
          ("s2.conv3", 139200, 512, 128, True, True), ("s1.conv3", 556800, 256, 64, True, True),
          ("s4.conv1", 8700, 512, 2048, False, True), ("s4.conv3", 8700, 2048, 512, True, True),
          ("s3.down", 34800, 1024, 512, False, False), ("fpn.lat2", 34800, 256, 1024, False, False),
          ("sca.value_proj", 184950, 256, 256, False, False), ("tsa.value_proj", 80000, 256, 256, False, False),
          ("enc.output_proj", 40000, 256, 256, True, False), ("ffn.fc1", 40000, 512, 256, False, True),
          ("ffn.fc2", 40000, 256, 512, True, False)]

B.use_tuned_gemms()
for name, M, N, K, has_res, relu in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    r = torch.randn(M, N, generator=g).half().cuda() if has_res else None
    wt = w.t()
    if has_res:
        lib = lambda: bev.linear_bias_act(x, w, b, r, relu)
    elif relu:
        lib = lambda: torch._addmm_activation(b, x, wt)
    else:
        lib = lambda: torch.addmm(b, x, wt)
    ours = lambda: bev.tsgemm(x, w, b, r, relu)
    res = {"lib": [], "ts": []}
    for _ in range(3):
        res["lib"].append(round(time_call(lib, iters=20, warm=5)[0], 1))
        res["ts"].append(round(time_call(ours, iters=20, warm=5)[0], 1))
    byt = (M * K + N * K + M * N * (2 if has_res else 1)) * 2
    ts = sorted(res["ts"])[1]
    print(json.dumps({"layer": name, "M": M, "N": N, "K": K, "us_lib": sorted(res["lib"])[1], "us_tsgemm": ts,
                      "GBs_tsgemm": round(byt / ts / 1e3, 1), "TFLOPs_tsgemm": round(2.0 * M * N * K / ts / 1e6, 1)}), flush=True)
