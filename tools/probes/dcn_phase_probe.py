#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the fused DCN kernel (needs a library built with
-DDCN_PROFILE: make -C bevformer_tensorrt_amd/csrc EXTRA=-DDCN_PROFILE)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tools/ (msda_sweep)
import bevformer_tensorrt_amd as bev
from bevformer_tensorrt_amd.functions import multi_scale_deformable_attn as M
from bevformer_tensorrt_amd.utils import load_library
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
g = torch.Generator().manual_seed(0)
C, H, W = 256, 58, 100
x = torch.randn(6, C, H, W, generator=g).half().cuda()
off = torch.randn(6, 18, H, W, generator=g).half().cuda()
mask = torch.rand(6, 9, H, W, generator=g).half().cuda()
w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).half().cuda()
b = torch.randn(C, generator=g).half().cuda()
lib = load_library()
lib.bevops_mdconv_set_variant(variant)
from bevformer_tensorrt_amd.utils import lib as L
dims = (6, C, H, W, C, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1)
nbytes = lib.bevops_mdconv_workspace_size(L.F16, *dims)
buf = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
out = torch.empty(6, C, H, W, dtype=torch.half, device="cuda")
for _ in range(3):
    st = lib.bevops_mdconv_forward(L.F16, x.data_ptr(), off.data_ptr(), mask.data_ptr(), w.data_ptr(), b.data_ptr(),
                                   out.data_ptr(), buf.data_ptr(), nbytes, *dims, L.current_stream_ptr(x.device))
    assert st == 0
torch.cuda.synchronize()
xt_bytes = (6 * C * H * W * 2 + 255) // 256 * 256
base = xt_bytes + 8 * 1000000
nw = 8 if variant != 2 else 4
raw = buf[base: base + 64 * nw * 64].view(torch.int64).view(64 * nw, 8).cpu()
names = ["prologue", "blend(wait loads)", "bar+LDS write+bar", "prefetch issue", "MFMA loop"]
tot = raw[:, :5].double().mean(0)
for n, v in zip(names, tot.tolist()):
    print(f"{n:22s} {v:12.0f} ticks  {100 * v / tot.sum().item():5.1f}%")
print("sum", tot.sum().item(), "per step", tot[1:].sum().item() / 36)
