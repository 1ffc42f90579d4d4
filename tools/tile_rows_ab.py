#!/usr/bin/env python3
"""64-row vs 128-row tiles of the tiled GEMM family (bevops_tile_gemm_set_variant) on the layer shapes of
BEVFormer-base, interleaved on one box: fp16 (bevops_tile_gemm_f16), the int8 chain's flavours (int8 in / int8 out
with int8 identity rows: ResNet conv3; int8 in / int8 out: conv1) and the encoder's fused-quantise flavour.  One
JSON line per layer: microseconds per call for each tile height, and what the launcher's own policy (0) picks."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.functions import int8_chain as C  # noqa: E402
from bevformer_tensorrt_amd.utils import load_library  # noqa: E402
from msda_sweep import time_call  # noqa: E402

# name, M, N, K, identity rows, flavours to time
SHAPES = [("s1.conv1", 556800, 64, 256, False, "f16 s8"), ("s1.conv3", 556800, 256, 64, True, "f16 s8"),
          ("s2.conv1", 139200, 128, 512, False, "f16 s8"), ("s2.conv3", 139200, 512, 128, True, "f16 s8"),
          ("s3.conv1", 34800, 256, 1024, False, "f16 s8"), ("s3.conv3", 34800, 1024, 256, True, "f16 s8"),
          ("s4.conv1", 8700, 512, 2048, False, "f16 s8"), ("s4.conv3", 8700, 2048, 512, True, "f16 s8"),
          ("tsa.value_proj", 80000, 256, 256, False, "f16 f16q"), ("enc.output_proj", 40000, 256, 256, True, "f16 f16q"),
          ("sca.offsets", 40000, 512, 256, False, "f16 f16q"), ("ffn.fc1", 40000, 512, 256, False, "f16 f16q"),
          ("ffn.fc2", 40000, 256, 512, True, "f16 f16q"), ("dec.in_proj", 900, 768, 256, True, "f16")]

lib = load_library()
for name, M, N, K, has_res, flav in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    r = torch.randn(M, N, generator=g).half().cuda() if has_res else None
    a8 = torch.randint(-127, 128, (M, K), generator=g, dtype=torch.int8).cuda()
    w8 = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8).cuda()
    r8 = torch.randint(-127, 128, (M, N), generator=g, dtype=torch.int8).cuda() if has_res else None
    bf = b.float()
    fns = {}
    if "f16" in flav.split():
        fns["f16"] = lambda: bev.tile_gemm(x, w, b, r, True)
    if "s8" in flav.split():
        fns["s8"] = lambda: C.linear_int8_chain(a8, 0.02, w8, 0.001, bf, r8, 0.03, True, torch.int8, 0.05)
    if "f16q" in flav.split():
        fns["f16q"] = lambda: C.linear_int8_chain(x, 0.02, w8, 0.001, bf, r, 1.0, False, torch.float16)
    out = {"layer": name, "M": M, "N": N, "K": K}
    for key, fn in fns.items():
        res = {64: [], 128: [], 0: []}
        for _ in range(3):
            for rows in (64, 128, 0):
                prev = lib.bevops_tile_gemm_set_variant(rows)
                try:
                    res[rows].append(round(time_call(fn, iters=20, warm=5)[0], 1))
                finally:
                    lib.bevops_tile_gemm_set_variant(prev)
        out[key] = {"us_64": sorted(res[64])[1], "us_128": sorted(res[128])[1], "us_policy": sorted(res[0])[1]}
    print(json.dumps(out), flush=True)
