#!/usr/bin/env python3
"""How the base SCA call's time splits over pyramid levels (kernel-design probe)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tools/ (msda_sweep)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bevformer_tensorrt_amd as bev
from bevformer_tensorrt_amd.utils import load_library
load_library().bevops_msda_set_variant(int(os.environ.get('VARIANT', '0')))
from msda_sweep import gen, time_call
LV = [[116, 200], [58, 100], [29, 50], [15, 25]]
for name, levels in (("all", LV), ("l01", LV[:2]), ("l23", LV[2:]), ("l0", LV[:1]), ("l1", LV[1:2]),
                     ("l2", LV[2:3]), ("l3", LV[3:])):
    for dist in ("uniform", "rig"):
        args, byt = gen((6, levels, 40000, 8, 4), torch.float16, dist)
        med, mn = time_call(lambda: bev.multi_scale_deformable_attn(*args))
        print(json.dumps(dict(levels=name, dist=dist, us_med=round(med, 1), us_min=round(mn, 1))), flush=True)
