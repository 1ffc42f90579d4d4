#!/usr/bin/env python3
"""The fp16 dense-layer implementations on the layer shapes of BEVFormer-base: what functions/linear.py's
dense_auto measures (tsgemm / tile_gemm / hipBLASLt entry / framework addmm) and which one it picks."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd import bevformer as B  # noqa: E402
from bevformer_tensorrt_amd.functions import linear as L  # noqa: E402

SHAPES = [("s1.conv1", 556800, 64, 256, False, True), ("s1.conv3", 556800, 256, 64, True, True),
          ("s2.conv1", 139200, 128, 512, False, True), ("s2.conv3", 139200, 512, 128, True, True),
          ("s3.conv1", 34800, 256, 1024, False, True), ("s3.conv3", 34800, 1024, 256, True, True),
          ("s4.conv1", 8700, 512, 2048, False, True), ("s4.conv3", 8700, 2048, 512, True, True),
          ("fpn.lat2", 34800, 256, 1024, False, False), ("tsa.value_proj", 80000, 256, 256, False, False),
          ("tsa.split", 40000, 192, 256, True, False), ("enc.output_proj", 40000, 256, 256, True, False),
          ("sca.offsets", 40000, 512, 256, False, False), ("sca.weights", 40000, 256, 256, False, False),
          ("ffn.fc1", 40000, 512, 256, False, True), ("ffn.fc2", 40000, 256, 512, True, False),
          ("dec.value_proj", 40000, 256, 256, False, False), ("dec.in_proj", 900, 768, 256, True, False),
          ("dec.out_proj", 900, 256, 256, True, False), ("dec.offsets", 900, 64, 256, True, False),
          ("dec.ffn.fc1", 900, 512, 256, False, True), ("dec.ffn.fc2", 900, 256, 512, True, False),
          ("head.reg", 900, 256, 256, False, True)]

B.use_tuned_gemms()
for name, M, N, K, has_res, relu in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    r = torch.randn(M, N, generator=g).half().cuda() if has_res else None
    bev.dense_auto(x, w, b, r, relu)
    key, times = L.DENSE_LOG[-1]
    byt = (M * K + N * K + M * N * (2 if has_res else 1)) * 2
    best = L._DENSE_CHOICE[key]
    print(json.dumps({"layer": name, "M": M, "N": N, "K": K, "us": times, "pick": best,
                      "GBs_pick": round(byt / times[best] / 1e3, 1)}), flush=True)
