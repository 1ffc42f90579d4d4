#!/usr/bin/env python3
"""Does the ResNet's front (stem + stages 1-2, whose activations of six cameras are 140-570 MB per tensor) run faster in
camera chunks whose working set stays inside the 256 MB memory-side cache?  Times, under HIP-graph replay, the front of
the base backbone on all six cameras at once against chunks of 3 / 2 / 1 cameras (results concatenated), and the rest
of the backbone (stages 3-4, DCNv2) for scale.  usage: backbone_chunks.py [--rounds R]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevformer_tensorrt_amd import bevformer as B  # noqa: E402
from bevformer_tensorrt_amd.functions.linear import graph_time_us  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--model", default="base")
a = ap.parse_args()
dev = torch.device("cuda")
model = B.BEVFormer(a.model).to(dev, torch.float16)
ops, bb = model.ops, model.backbone
H, W = B.CONFIGS[a.model]["image"]
img = torch.randn(6, 3, H, W, generator=torch.Generator().manual_seed(0)).to(dev, torch.float16)
model.extract_feat(img[None])      # channels-last filters, dispatch measured for the six-camera shapes


def front(x, upto=2):
    y = bb._stem_fused(x, ops)
    for st in bb.stages[:upto]:
        for blk in st:
            y = blk.forward_nhwc(y, ops)
    return y


def chunked(x, n, upto=2):
    return torch.cat([front(x[i:i + n], upto) for i in range(0, 6, n)])


def rest(y):
    for st in bb.stages[2:]:
        for blk in st:
            y = blk.forward_nhwc(y, ops)
    return y


def two_streams(fn, x, parts=2):
    """fn on `parts` camera groups, each on its own stream (forked from / joined to the current one: inside a graph
    capture these are parallel branches) -- do a load-path-bound kernel of one group (DCNv2) and an HBM-bound one of
    the other (1 x 1 convolutions) overlap?"""
    cur = torch.cuda.current_stream()
    n, outs = 6 // parts, []
    for i, s in enumerate(SIDE[:parts]):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(fn(x[i * n:(i + 1) * n]))
    for s in SIDE[:parts]:
        cur.wait_stream(s)
    return torch.cat(outs)


SIDE = [torch.cuda.Stream() for _ in range(3)]
with torch.no_grad():
    ref = front(img)
    for n in (3, 2, 1):
        assert torch.equal(chunked(img, n), ref) or (chunked(img, n) - ref).abs().max() < 0.05   # (measures the new shapes)
    mid = ref
    for parts in (2, 3):
        two_streams(rest, mid, parts)      # (measures the new shapes)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        row = {"front_6": round(graph_time_us(lambda: front(img), iters=2), 1)}
        for n in (3, 2, 1):
            row["front_%dx%d" % (6 // n, n)] = round(graph_time_us(lambda: chunked(img, n), iters=2), 1)
        row["stage1_6"] = round(graph_time_us(lambda: front(img, 1), iters=2), 1)
        row["stage1_6x1"] = round(graph_time_us(lambda: chunked(img, 1, 1), iters=2), 1)
        row["stages34_6"] = round(graph_time_us(lambda: rest(mid), iters=2), 1)
        row["stages34_2streams"] = round(graph_time_us(lambda: two_streams(rest, mid), iters=2), 1)
        row["stages34_3streams"] = round(graph_time_us(lambda: two_streams(rest, mid, 3), iters=2), 1)
        row["front_2streams"] = round(graph_time_us(lambda: two_streams(front, img), iters=2), 1)
        print(json.dumps(row), flush=True)
