#!/usr/bin/env python3
"""The decoder's self-attention (900 object queries, 8 heads x 32; modules/decoder.py:52-112) under HIP-graph replay:
torch's fused kernel (flash / aotriton) against the math form (two batched GEMMs + softmax) and the efficient backend."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevformer_tensorrt_amd.functions.linear import graph_time_us  # noqa: E402

g = torch.Generator().manual_seed(0)
qkv = torch.randn(1, 900, 3, 8, 32, generator=g).half().cuda()
q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
res = {}
res["default"] = graph_time_us(lambda: F.scaled_dot_product_attention(q, k, v))
from torch.nn.attention import SDPBackend, sdpa_kernel
for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
    try:
        def f():
            with sdpa_kernel(be):
                return F.scaled_dot_product_attention(q, k, v)
        res[name] = graph_time_us(f)
    except Exception as exc:
        res[name] = repr(exc)[:80]


def manual():
    s = torch.matmul(q, k.transpose(-1, -2)) * (32 ** -0.5)
    return torch.matmul(torch.softmax(s, -1), v)


res["manual_bmm_softmax_bmm"] = graph_time_us(manual)
want = F.scaled_dot_product_attention(q.float(), k.float(), v.float())
res["manual_max_err"] = (manual().float() - want).abs().max().item()
print(json.dumps({k_: (round(v_, 2) if isinstance(v_, float) else v_) for k_, v_ in res.items()}))
