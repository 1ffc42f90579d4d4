#!/usr/bin/env python3
"""The six encoder (and decoder) layers project the SAME rows with six weight sets (TSA: the [prev_bev | query] stack,
SCA: the camera features, decoder: bev_embed).  Six 256-column GEMMs against one 1 536-column GEMM, HIP-graph replay."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.functions import linear as L  # noqa: E402

g = torch.Generator().manual_seed(0)
for M, what in ((80000, "TSA value_proj"), (40000, "decoder value_proj"), (184950, "SCA value_proj (row-major output)")):
    x = torch.randn(M, 256, generator=g).half().cuda()
    w = (torch.randn(1536, 256, generator=g) / 16).half().cuda()
    b = torch.randn(1536, generator=g).half().cuda()
    ws, bs = [w[i * 256:(i + 1) * 256].contiguous() for i in range(6)], [b[i * 256:(i + 1) * 256].contiguous() for i in range(6)]
    row = {"rows": M, "what": what}
    for name in ("tsgemm", "tile", "blaslt"):
        fn = L._DENSE[name]
        row[name] = {"six_calls_us": round(L.graph_time_us(lambda: [fn(x, ws[i], bs[i], None, False) for i in range(6)], 2, 3), 1),
                     "one_call_1536_us": round(L.graph_time_us(lambda: fn(x, w, b, None, False), 2, 3), 1)}
    print(json.dumps(row), flush=True)
