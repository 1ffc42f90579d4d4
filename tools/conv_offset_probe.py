import sys, os, json, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from msda_sweep import time_call
for (C, H, W) in ((256, 58, 100), (512, 29, 50)):
    x = torch.randn(6, C, H, W, device="cuda", dtype=torch.half).contiguous(memory_format=torch.channels_last)
    xn = x.contiguous()
    for co in (27, 32, 64, 128):
        w = (torch.randn(co, C, 3, 3, device="cuda", dtype=torch.half) * 0.01)
        wl = w.contiguous(memory_format=torch.channels_last)
        b = torch.zeros(co, device="cuda", dtype=torch.half)
        r = dict(C=C, cout=co)
        r["nhwc"] = round(time_call(lambda: F.conv2d(x, wl, b, 1, 1))[0], 1)
        r["nchw"] = round(time_call(lambda: F.conv2d(xn, w, b, 1, 1))[0], 1)
        print(json.dumps(r), flush=True)
