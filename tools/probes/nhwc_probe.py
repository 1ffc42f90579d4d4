#!/usr/bin/env python3
"""Is an NHWC (channels_last) backbone faster than NCHW on this stack?  conv / linear micro-timings."""
import sys, os, json, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msda_sweep import time_call
dev = "cuda"
def t(fn): return round(time_call(fn, iters=20, warm=5)[0], 1)
for (Cin, Cout, H, W, k) in ((256, 64, 232, 400, 1), (64, 64, 232, 400, 3), (64, 256, 232, 400, 1),
                             (512, 128, 116, 200, 1), (128, 128, 116, 200, 3), (1024, 256, 58, 100, 1),
                             (256, 1024, 58, 100, 1), (2048, 512, 29, 50, 1), (3, 64, 928, 1600, 7)):
    x = torch.randn(6, Cin, H, W, device=dev, dtype=torch.half)
    w = torch.randn(Cout, Cin, k, k, device=dev, dtype=torch.half) * 0.05
    b = torch.randn(Cout, device=dev, dtype=torch.half)
    s, p = (2, 3) if k == 7 else (1, k // 2)
    r = dict(shape=[Cin, Cout, H, W, k])
    r["nchw_conv_bias_relu"] = t(lambda: F.relu(F.conv2d(x, w, b, s, p), inplace=True))
    r["nchw_conv_nobias"] = t(lambda: F.conv2d(x, w, None, s, p))
    xl = x.contiguous(memory_format=torch.channels_last)
    wl = w.contiguous(memory_format=torch.channels_last)
    r["nhwc_conv_bias_relu"] = t(lambda: F.relu(F.conv2d(xl, wl, b, s, p), inplace=True))
    r["nhwc_conv_nobias"] = t(lambda: F.conv2d(xl, wl, None, s, p))
    if k == 1:
        x2 = xl.permute(0, 2, 3, 1).reshape(-1, Cin)
        w2 = w.view(Cout, Cin)
        r["nhwc_linear_bias_relu"] = t(lambda: F.relu(F.linear(x2, w2, b), inplace=True))
        w2t = w2.t().contiguous()
        r["nhwc_addmm_act"] = t(lambda: torch._addmm_activation(b, x2, w2t))
    print(json.dumps(r), flush=True)
