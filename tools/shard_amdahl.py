#!/usr/bin/env python3
"""Amdahl table of the camera-sharded frame from ONE GPU (no multi-GPU box is lent to this build): the per-camera
part of a BEVFormer-base frame -- backbone + FPN + encoder-input assembly, and per encoder layer the value
projection + fused SCA sampling of the LOCAL cameras -- timed with 6 / 3 / 2 / 1 cameras (what a rank owns at
G = 1 / 2 / 4 (max shard) / 8), next to the whole frame (HIP-graph replay).  Replicated time = whole frame - the
6-camera per-camera part; the all-reduce of the "reduce" exchange is priced from the message size (20.48 MB fp16
per layer) at the xGMI link rate.  Prints one JSON line per row and the predicted frame time per G.
Round 5 adds the "scatter" exchange (cameras sharded AND the rest of the encoder sharded by query range): rank 0's
compute of a G-rank frame is MEASURED as a HIP-graph replay of the real sharded forward with a wire-less stand-in for
the exchange object (own rows = the first ceil(nq / G), the all-gather a local copy of the right size, the
reduce-scatter a slice) -- every kernel rank 0 would launch, at the size it would launch it.
usage: shard_amdahl.py [base] [--iters N]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevformer_tensorrt_amd import bevformer as B, geometry as G  # noqa: E402


def time_ms(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters


class _LocalOnly:
    """the "reduce" exchange without a wire: this rank's masked camera sum is the result"""
    mode = "reduce"

    def __init__(self, cams):
        self.cams = cams

    def reduce(self, t):
        return t


class _LocalScatter:
    """rank 0 of a G-rank "scatter" exchange without a wire (timing stand-in: the gathered rows are copies of rank
    0's rows, so the VALUES of the frame are not the model's -- shapes, kernels and bytes moved locally are)"""
    mode = "scatter"

    def __init__(self, world, cams):
        self.world, self.rank, self.cams = world, 0, cams

    def query_range(self, nq):
        per = -(-nq // self.world)
        return 0, per, per

    def all_gather_queries(self, local, nq):
        return local.repeat(1, self.world, 1)[:, :nq].contiguous()

    def reduce_scatter_queries(self, partial, nq):
        return partial[:, : -(-nq // self.world)].contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default="base")
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--link-gbs", type=float, default=153.0, help="one xGMI link, GB/s per direction (MI355X_MICROARCH)")
    a = ap.parse_args()
    dev, dtype = torch.device("cuda"), torch.float16
    name = a.model
    model = B.BEVFormer(name, seed=0).to(dev, dtype)
    H, W = B.CONFIGS[name]["image"]
    nq = model.bev_h * model.bev_w
    l2i = G.synthetic_lidar2img((H, W)).to(dev)
    g = torch.Generator().manual_seed(0)
    img = torch.randn(1, 6, 3, H, W, generator=g).to(dev, dtype)
    # whole frame, graph replay, back to back
    runner = B.FrameRunner(model, dev, dtype, graph=True, clone_outputs=False)
    runner.image_buffer.copy_(img)
    can = torch.zeros(18)
    for i in range(3):
        can[0], can[-1] = 0.5 * i, 0.8 * i
        runner.step(runner.image_buffer, can, l2i, "s")
    whole = time_ms(lambda: runner.step(runner.image_buffer, can, l2i, "s"), a.iters)
    print(json.dumps({"row": "whole frame, 1 GPU, HIP-graph replay", "ms": round(whole, 3)}), flush=True)
    # geometry of one frame (for the sampler's inputs)
    ref_3d = G.reference_points_3d(model.bev_h, model.bev_w, B.PC_RANGE[5] - B.PC_RANGE[2], 4, device="cpu", dtype=torch.float)
    pillars = G.pillar_points(ref_3d, B.PC_RANGE).to(dev)
    ref_cam, bev_mask = G.project_points(pillars, l2i.float(), (H, W), projection="fma")
    ref_cam, bev_mask = ref_cam.to(dtype), bev_mask.to(dtype)
    query = torch.randn(1, nq, B.EMBED, generator=g).to(dev, dtype)
    per_cam = {}
    for ncam in (6, 3, 2, 1):
        cams = list(range(ncam))
        t_feat = time_ms(lambda: model.extract_feat(img, cams), a.iters)
        mlvl = model.extract_feat(img, cams)
        level_hw = [f.shape[-2:] for f in mlvl]
        feat = torch.cat([f.permute(0, 2, 3, 1).reshape(ncam, -1, B.EMBED) for f in mlvl], 1).contiguous()
        shapes, _ = G.level_layout(level_hw, "cpu")
        sca = model.encoder[0].sca
        ex = _LocalOnly(cams)
        full = time_ms(lambda: sca(query, feat, ref_cam, bev_mask, shapes, None, ex), a.iters)
        # the replicated share of that call: offsets / weights / output projections of the one BEV query set
        repl = time_ms(lambda: (B._dense(model.ops, sca.sampling_offsets, query), B._dense(model.ops, sca.attention_weights, query),
                                B._dense(model.ops, sca.output_proj, query, query, False)), a.iters)
        per_cam[ncam] = (t_feat, max(full - repl, 0.0))
        print(json.dumps({"row": f"{ncam} local cameras", "backbone_fpn_ms": round(t_feat, 3),
                          "sca_local_ms_per_layer": round(full - repl, 3), "sca_replicated_ms_per_layer": round(repl, 3)}),
              flush=True)
    layers = len(model.encoder)
    shard6 = per_cam[6][0] + layers * per_cam[6][1]
    replicated = whole - shard6
    print(json.dumps({"row": "split of the 1-GPU frame", "per_camera_ms": round(shard6, 3), "replicated_ms": round(replicated, 3),
                      "per_camera_share": round(shard6 / whole, 3)}), flush=True)
    msg = nq * B.EMBED * 2
    # ---- "scatter": rank 0's whole sharded frame, measured (graph replay, wire-less exchange)
    for Gn, ncam in ((1, 6), (2, 3), (4, 2), (8, 1)):
        cams = list(range(ncam))
        r2 = B.FrameRunner(model, dev, dtype, graph=True, cams=cams, gather=_LocalScatter(Gn, cams), clone_outputs=False)
        r2.image_buffer.copy_(img)
        for i in range(3):
            r2.step(r2.image_buffer, can, l2i, "s")
        t = time_ms(lambda: r2.step(r2.image_buffer, can, l2i, "s"), a.iters)
        # per layer one all-gather + one reduce-scatter of the query rows = the bytes of one all-reduce; + the final
        # all-gather.  "ring": every byte over ONE link pair; "direct": the (G - 1) peers' shards over their own links
        ring = 0.0 if Gn == 1 else (layers * 2 + 1) * ((Gn - 1) / Gn * msg / (a.link_gbs * 1e9)) * 1e3
        direct = 0.0 if Gn == 1 else (layers * 2 + 1) * (msg / Gn / (a.link_gbs * 1e9)) * 1e3
        print(json.dumps({"row": f"scatter G={Gn} (rank 0 measured, no wire)", "max_local_cameras": ncam,
                          "compute_ms": round(t, 3), "wire_ms_ring_one_link": round(ring, 3),
                          "wire_ms_direct_links": round(direct, 3),
                          "speedup_compute_only": round(whole / t, 2), "speedup_ring_exposed": round(whole / (t + ring), 2),
                          "speedup_direct_exposed": round(whole / (t + direct), 2)}), flush=True)
        del r2
        torch.cuda.empty_cache()
    for Gn, ncam in ((1, 6), (2, 3), (4, 2), (8, 1)):
        # ring all-reduce on point-to-point links: 2 (G - 1) / G of the message crosses each link, both directions busy
        comm = 0.0 if Gn == 1 else layers * (2 * (Gn - 1) / Gn * msg / (a.link_gbs * 1e9)) * 1e3
        t = per_cam[ncam][0] + layers * per_cam[ncam][1] + replicated
        print(json.dumps({"row": f"predicted G={Gn}", "max_local_cameras": ncam, "compute_ms": round(t, 3),
                          "all_reduce_ms_if_exposed": round(comm, 3), "frames_per_s_compute_only": round(1000 / t, 1),
                          "frames_per_s_comm_exposed": round(1000 / (t + comm), 1),
                          "speedup_vs_1": round(whole / (t + comm), 2)}), flush=True)


if __name__ == "__main__":
    main()
