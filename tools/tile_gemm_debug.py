#!/usr/bin/env python3
"""tile_gemm vs fp32 linear: error statistics per shape (debug aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev

for M, N, K in [(4096, 256, 1024), (4096, 256, 64), (4096, 256, 32), (4096, 128, 128), (512, 128, 256), (34800, 256, 1024)]:
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    out = bev.tile_gemm(x, w).float()
    want = x.float() @ w.float().t()
    want64 = (x[:512].double() @ w.double().t())
    ts = bev.tsgemm(x, w).float() if N % 256 == 0 and K % 64 == 0 else None
    err = (out - want).abs()
    bad = (err > 1e-3 * want.abs().clamp_min(1) + 2e-3)
    rows_bad = bad.any(1).nonzero().flatten()[:8].tolist()
    cols_bad = bad.any(0).nonzero().flatten()[:8].tolist()
    print(M, N, K, "max", float(err.max()), "bad", int(bad.sum()), "rows", rows_bad, "cols", cols_bad,
          "ref32-vs-64", float((want[:512].double() - want64).abs().max()),
          "tile-vs-64", float((out[:512].double() - want64).abs().max()),
          "tsgemm-vs-64", None if ts is None else float((ts[:512].double() - want64).abs().max()), flush=True)
