#!/usr/bin/env python3
"""What HBM gives a streaming kernel on this box, next to the 8 TB/s the rooflines are priced against: the framework's
elementwise kernels (16 bytes per lane, grid-stride) under HIP-graph replay on tensors of the backbone's sizes --
fill (write only), copy (1 read : 1 write), add (2 reads : 1 write; the mix of a 1x1 convolution with identity rows),
sum (read only).  usage: hbm_rates.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevformer_tensorrt_amd.functions.linear import graph_time_us  # noqa: E402

dev = torch.device("cuda")
for rows, cols, what in ((34800, 1024, "stage-3 activation, 71 MB"), (139200, 512, "stage-2 activation, 143 MB"),
                         (556800, 256, "stage-1 activation, 285 MB"), (34800, 256, "stage-3 bottleneck, 17.8 MB")):
    a = torch.randn(rows, cols, device=dev).half()
    b = torch.randn(rows, cols, device=dev).half()
    c = torch.empty_like(a)
    mb = a.numel() * 2 / 1e6
    row = {"tensor": what, "MB": round(mb, 1)}
    for name, fn, traffic in (("fill", lambda: c.zero_(), mb), ("copy", lambda: c.copy_(a), 2 * mb),
                              ("add", lambda: torch.add(a, b, out=c), 3 * mb), ("relu_inplace", lambda: torch.relu_(c), 2 * mb),
                              ("sum", lambda: a.sum(dtype=torch.float32), mb)):
        us = min(graph_time_us(fn, iters=4) for _ in range(2))
        row[name] = {"us": round(us, 1), "TB/s": round(traffic / us, 2)}
    print(json.dumps(row), flush=True)
