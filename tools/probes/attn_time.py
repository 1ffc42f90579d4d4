#!/usr/bin/env python3
"""The decoder's self-attention (900 x 8 x 32) under HIP-graph replay: torch's fused kernel against csrc/attention.hip."""
import json, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bevformer_tensorrt_amd as bev  # noqa: E402
from bevformer_tensorrt_amd.functions.linear import graph_time_us  # noqa: E402
g = torch.Generator().manual_seed(0)
qkv = torch.randn(900, 3, 8, 32, generator=g).half().cuda()
q, k, v = (qkv[:, i].transpose(0, 1)[None] for i in range(3))
res = {"torch_sdpa": [], "own": []}
for _ in range(3):
    res["torch_sdpa"].append(round(graph_time_us(lambda: F.scaled_dot_product_attention(q, k, v)), 2))
    res["own"].append(round(graph_time_us(lambda: bev.self_attention_qkv(qkv)), 2))
print(json.dumps(res))
