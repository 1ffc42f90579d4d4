"""Synthetic bev_pool_v2 index generator with the structure of
LSSViewTransformer.voxel_pooling_prepare_v2 (third_party/bev_mmdet3d/models/necks/
view_transformer.py:239-312): every kept frustum point (n,d,h,w) has a depth index, the
feature index of its (n,h,w) pixel and a BEV cell; points are sorted by cell and
run-length encoded into intervals."""
import numpy as np


def make_indices(N, D, H, W, out_h, out_w, keep=0.27, seed=0):
    rng = np.random.default_rng(seed)
    n_pts = N * D * H * W
    kept = np.flatnonzero(rng.random(n_pts) < keep).astype(np.int64)
    ranks_depth = kept
    n, rem = np.divmod(kept, D * H * W)
    hw = rem % (H * W)
    ranks_feat = n * (H * W) + hw
    # cells: clustered like a real rig (few cells get very long intervals)
    cells = (rng.beta(0.8, 2.5, kept.size) * (out_h * out_w * 0.42)).astype(np.int64)
    cells = np.minimum(cells, out_h * out_w - 1)
    order = np.argsort(cells, kind="stable")
    ranks_bev, ranks_depth, ranks_feat = cells[order], ranks_depth[order], ranks_feat[order]
    first = np.ones(kept.size, bool)
    first[1:] = ranks_bev[1:] != ranks_bev[:-1]
    starts = np.flatnonzero(first)
    lengths = np.diff(np.append(starts, kept.size))
    i32 = lambda a: a.astype(np.int32)
    return i32(ranks_depth), i32(ranks_feat), i32(ranks_bev), i32(starts), i32(lengths)


def index_add_reference(depth, feat, ranks_depth, ranks_feat, ranks_bev, out_h, out_w):
    """Independent statement of the op with torch.index_add_ (fp64 accumulate)."""
    import torch
    d = torch.from_numpy(np.asarray(depth, np.float64)).flatten()
    f = torch.from_numpy(np.asarray(feat, np.float64)).reshape(-1, feat.shape[-1])
    contrib = d[torch.from_numpy(ranks_depth).long()].unsqueeze(1) * f[torch.from_numpy(ranks_feat).long()]
    out = torch.zeros(out_h * out_w, feat.shape[-1], dtype=torch.float64)
    out.index_add_(0, torch.from_numpy(ranks_bev).long(), contrib)
    return out.view(1, out_h, out_w, -1).numpy()
