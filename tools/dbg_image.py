import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevformer_tensorrt_amd as bev
from oracle.image_ref import image_normalize_pad as ref
g = torch.Generator().manual_seed(0)
img = torch.randint(0, 256, (2, 45, 70, 3), generator=g, dtype=torch.uint8).float() + torch.rand((2, 45, 70, 3), generator=g) * 0.5
for to_rgb, std in ((True, (58.395, 57.12, 57.375)), (False, (58.395, 57.12, 57.375)), (True, (1.0, 1.0, 1.0))):
    want = ref(img.numpy(), std=std, to_rgb=to_rgb)
    got = bev.image_normalize_pad(img.cuda(), std=std, to_rgb=to_rgb, dtype=torch.float32).cpu().numpy()
    bad = np.argwhere(got != want)
    print("to_rgb", to_rgb, "std", std, "mismatches", len(bad), "of", got.size)
    for i in bad[:4]:
        i = tuple(i); n, c, y, x = i
        src = img.numpy()[n, y, x] if y < 45 and x < 70 else None
        print("   at", i, "got", repr(got[i]), "want", repr(want[i]), "src", src)
# frame runner
from bevformer_tensorrt_amd import bevformer as B, geometry as G
dev = torch.device("cuda")
model = B.BEVFormer("tiny", seed=0).to(dev, torch.float16)
H, W = B.CONFIGS["tiny"]["image"]
raw = torch.randint(0, 256, (6, H - 30, W, 3), generator=g, dtype=torch.uint8)
l2i = G.synthetic_lidar2img((H, W))
a = B.FrameRunner(model, dev, torch.float16); b = B.FrameRunner(model, dev, torch.float16)
ca, ra = a.step_raw(raw.to(dev), torch.zeros(18), l2i, "s")
pre = bev.image_normalize_pad(raw.to(dev), dtype=torch.float16)[None]
print("buffer == pre:", torch.equal(a._in["image"], pre), float(pre.abs().max()))
cb, rb = b.step(pre, torch.zeros(18), l2i, "s")
cb2, rb2 = b.step(pre, torch.zeros(18), l2i, "s2")
print("cls diff", float((ca.float() - cb.float()).abs().max()), "crd diff", float((ra.float() - rb.float()).abs().max()),
      "finite", bool(torch.isfinite(ca).all()), "repeat diff", float((cb.float() - cb2.float()).abs().max()))
