#!/usr/bin/env python3
"""SHA-256 of the three outputs of one fixed frame (seeded weights, seeded inputs): compare between PROCESSES (the
library algorithms are selected by timing once per process) and boxes.  usage: frame_hash.py [tiny|small|base ...]"""
import hashlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402

dev, dtype = torch.device("cuda"), torch.float16
for name in (sys.argv[1:] or ["tiny", "small", "base"]):
    model = B.BEVFormer(name, seed=0).to(dev, dtype)
    H, W = B.CONFIGS[name]["image"]
    l2i = G.synthetic_lidar2img((H, W)).to(dev)
    img = torch.randn(1, 6, 3, H, W, generator=torch.Generator().manual_seed(1)).to(dev, dtype)
    nq = model.bev_h * model.bev_w
    prev = (torch.randn(nq, 1, B.EMBED, generator=torch.Generator().manual_seed(2)) * 0.5).to(dev, dtype)
    can = torch.zeros(18, device=dev)
    can[0], can[-1] = 0.5, 0.8
    with torch.no_grad():
        model(img, prev, torch.tensor(1.0, device=dev), can, l2i)
        out = model(img, prev, torch.tensor(1.0, device=dev), can, l2i)
        feats = model.extract_feat(img)
    h = lambda t: hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]
    print(json.dumps({"model": name, "pyramid": [h(f) for f in feats], "bev_embed": h(out[0]), "cls": h(out[1]), "boxes": h(out[2])}), flush=True)
    del model
