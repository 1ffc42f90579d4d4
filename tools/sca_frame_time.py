#!/usr/bin/env python3
"""The SCA sampling call the model frame replays (value-projection planes -> fused sampling -> camera reduce) on the
reference points of the 6-camera rig, WITHOUT the projection GEMM: one block per 1 280-query chunk
(bevops_sca_forward_prepacked) against the balanced slices of a visibility plan (bevops_sca_forward_planned; k = slices
per CU; "_scratch": every pair through the per-camera scratch instead of single-camera pairs stored straight into the
output; "_rolled": the rolled camera reduce = the round-5 first-half build), under HIP-graph replay, interleaved.  One JSON line; --offsets S scales the N(0, 1) sampling offsets (pixels).
--once K: K plain launches of every flavour (for rocprofv3 --kernel-trace / --pmc runs).  (The ablation builds whose
timings are in profiles/r05/sca_plan_ablation.jsonl were removed from the library after that measurement.)"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_tensorrt_amd import geometry as G  # noqa: E402
from bevformer_tensorrt_amd.functions import linear as L  # noqa: E402
from bevformer_tensorrt_amd.functions import spatial_cross_attention as S  # noqa: E402
from bevformer_tensorrt_amd.functions.multi_scale_deformable_attn import _host_shapes, _shapes_i32  # noqa: E402
from bevformer_tensorrt_amd.utils import lib as _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--offsets", type=float, default=1.0)
ap.add_argument("--once", type=int, default=0)
ap.add_argument("--ks", default="1,2,3")
ap.add_argument("--only", default="", help="comma-separated flavour names: run just these (for counter passes)")
args = ap.parse_args()

g = torch.Generator().manual_seed(0)
levels = [[116, 200], [58, 100], [29, 50], [15, 25]]
nk = sum(h * w for h, w in levels)
nq, heads, embed = 40000, 8, 256
feats = (torch.randn(6, nk, embed, generator=g) * 0.5).half().cuda()
wgt = (torch.randn(embed, embed, generator=g) / 16).half().cuda()
bias = (torch.randn(embed, generator=g) * 0.1).half().cuda()
ref3d = G.reference_points_3d(200, 200, 8, 4, device="cpu")
cam, mask = G.point_sampling(ref3d, [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], G.synthetic_lidar2img((928, 1600)), (928, 1600))
ref = cam.reshape(6, nq, 1, 8).half().cuda()
vis = mask.reshape(6, nq, -1).any(-1)
bm = (vis.float() / vis.sum(0).clamp(min=1)).half().cuda()
off = (torch.randn(1, nq, heads, 64, generator=g) * args.offsets).half().cuda()
w = torch.randn(1, nq, heads, 32, generator=g).half().cuda()
sh = torch.tensor(levels, dtype=torch.int32)
handle = _lib.load_library()
shapes_dev, shapes_host = _shapes_i32(sh, feats.device)
if shapes_host is None:
    shapes_host = _host_shapes(shapes_dev)
geom = (shapes_host, 6, nk, heads, 32, 4, nq, 8, 4)
planes = S._project_planes(handle, feats, wgt, bias, geom)
plan = S.spatial_cross_attention_plan(bm)
torch.cuda.synchronize()
ks = [int(k) for k in args.ks.split(",")]


def chunked():
    return S._sample_planes(handle, planes, geom, ref, off, w, bm, None)


def planned(k, direct=True, unrolled=True, fold=3014):
    """k slices per CU; direct: single-camera pairs stored by the sampler into the output rows (3012) or every pair
    through the per-camera scratch (3013); unrolled: the camera reduce with its loop unrolled (3010) or rolled (3011)."""
    def fn():
        handle.bevops_msda_set_variant(3000 + k)
        handle.bevops_msda_set_variant(3012 if direct else 3013)
        handle.bevops_msda_set_variant(3010 if unrolled else 3011)
        handle.bevops_msda_set_variant(fold)     # 3014: record broadcasts folded into their consumers, LDS row taps fused (default); 3015: the round-5 build
        return S._sample_planes(handle, planes, geom, ref, off, w, bm, plan)
    return fn


fns = {"chunked": chunked, **{f"planned_k{k}": planned(k) for k in ks},
       "planned_k2_nofold": planned(2, fold=3015),
       "planned_k2_scratch": planned(2, direct=False), "planned_k2_scratch_rolled": planned(2, False, False)}

want = chunked()
if args.only:
    fns = {k: v for k, v in fns.items() if k in args.only.split(",")}
for name, fn in fns.items():
    assert torch.equal(fn(), want), name
if args.once:
    for name, fn in fns.items():
        for _ in range(args.once):
            fn()
    torch.cuda.synchronize()
    sys.exit(0)
res = {k: [] for k in fns}
for _ in range(3):
    for name, fn in fns.items():
        res[name].append(round(L.graph_time_us(fn), 2))
pairs = int(vis.sum())
fused_bytes = (6 * nk * heads * 32 + nq * heads * 32 * 3 + 6 * nq * 8 + 6 * nq + nq * heads * 32) * 2 + 32
med = {k: sorted(v)[1] for k, v in res.items()}
print(json.dumps({"what": "fused SCA sampling call on prepacked planes (sampler + camera reduce), rig geometry",
                  "offsets_sigma_px": args.offsets, "visible_pairs": pairs, "visible_frac": round(pairs / (6 * nq), 4),
                  "us": med, "algorithmic_bytes": fused_bytes,
                  "frac_of_8TBs": {k: round(fused_bytes / v / 8e6, 4) for k, v in med.items()}}), flush=True)
for v in (3002, 3012, 3010, 3014, 0):
    handle.bevops_msda_set_variant(v)
