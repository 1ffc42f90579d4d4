#!/usr/bin/env python3
"""INT8 dense layers on the shapes of BEVFormer-base: quantise pass + int8 GEMM (bevops_quantize_rows +
bevops_linear_int8) vs the GEMM that quantises its fp16 operand itself (bevops_linear_int8_fused) vs the fp16
library GEMM of the same layer (interleaved, HIP events)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd import bevformer as B  # noqa: E402
from msda_sweep import time_call  # noqa: E402

SHAPES = [("s1.conv1", 556800, 64, 256, False, True), ("s1.conv3", 556800, 256, 64, True, True),
          ("s2.conv1", 139200, 128, 512, False, True), ("s2.conv3", 139200, 512, 128, True, True),
          ("s3.conv1", 34800, 256, 1024, False, True), ("s3.conv3", 34800, 1024, 256, True, True),
          ("s4.conv1", 8700, 512, 2048, False, True), ("s4.conv3", 8700, 2048, 512, True, True),
          ("sca.value_proj", 184950, 256, 256, False, False), ("tsa.value_proj", 80000, 256, 256, False, False),
          ("tsa.offsets", 40000, 128, 512, False, False), ("enc.output_proj", 40000, 256, 256, True, False),
          ("sca.offsets", 40000, 512, 256, False, False), ("ffn.fc1", 40000, 512, 256, False, True),
          ("ffn.fc2", 40000, 256, 512, True, False), ("dec.output_proj", 900, 256, 256, True, False)]

B.use_tuned_gemms()
for name, M, N, K, has_res, relu in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).half().cuda()
    w = torch.randn(N, K, generator=g) / K ** 0.5
    wh = w.half().cuda()
    bh = torch.randn(N, generator=g).half().cuda()
    bf = bh.float()
    r = torch.randn(M, N, generator=g).half().cuda() if has_res else None
    s_x, s_w = float(x.abs().max()) / 127, float(w.abs().max()) / 127
    wq = torch.clamp(torch.round(w / s_w), -127, 127).to(torch.int8).cuda()
    qbuf = torch.empty(M, K, dtype=torch.int8, device="cuda")
    q = bev.quantize_rows(x, s_x, out=qbuf)
    lib = lambda: bev.linear_bias_act(x, wh, bh, r, relu)
    two = lambda: bev.linear_int8(bev.quantize_rows(x, s_x, out=qbuf), s_x, wq, s_w, bf, r, relu)
    gemm = lambda: bev.linear_int8(q, s_x, wq, s_w, bf, r, relu)
    fused = lambda: bev.linear_int8(x, s_x, wq, s_w, bf, r, relu)
    res = {"lib": [], "two": [], "gemm": [], "fused": []}
    for _ in range(3):
        for key, fn in (("lib", lib), ("two", two), ("gemm", gemm), ("fused", fused)):
            try:
                res[key].append(round(time_call(fn, iters=20, warm=5)[0], 1))
            except Exception:     # no library algorithm for the shape
                res[key].append(float("nan"))
    med = {k: sorted(v)[1] for k, v in res.items()}
    byt = (M * K + M * N * (2 if has_res else 1)) * 2 + N * K
    print(json.dumps({"layer": name, "M": M, "N": N, "K": K, "us_fp16_lib": med["lib"], "us_quantize_plus_int8": med["two"],
                      "us_int8_gemm_alone": med["gemm"], "us_int8_fused": med["fused"],
                      "GBs_fused": round(byt / med["fused"] / 1e3, 1)}), flush=True)
