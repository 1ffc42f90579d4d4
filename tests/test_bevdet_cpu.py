"""BEVDet view-transformer geometry (BASELINE config 5): bevformer_tensorrt_amd/bevdet.py restates
LSSViewTransformer.create_frustum / get_lidar_coor / voxel_pooling_prepare_v2; the golden holds
SHA-256 digests of what the reference's OWN methods produce (lifted from third_party/bev_mmdet3d/
models/necks/view_transformer.py by tests/golden/make_wrapper_golden.py) at the BEVDet-R50 config with
the calibration of the reference's bev_pool test.  Index generation: bit-exact."""
import hashlib

import numpy as np
import torch

from conftest import golden
from bevformer_tensorrt_amd.bevdet import BEVDET_R50, LSSViewTransformer


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_ranks_and_intervals_bit_exact():
    g = golden("bevdet_geometry")
    vt = LSSViewTransformer(**BEVDET_R50, ops=object())
    assert vt.D == int(g["D"]) == 59
    assert _digest(vt.frustum.contiguous().numpy()) == str(g["frustum_sha256"])
    t = lambda k: torch.from_numpy(g[k])
    coor = vt.get_lidar_coor(t("sensor2ego"), None, t("cam2imgs"), t("post_rots"), t("post_trans"), t("bda"))
    assert np.array_equal(coor[0, :, ::7, ::3, ::5].numpy(), g["coor_sample"])
    assert _digest(coor.contiguous().numpy()) == str(g["coor_sha256"])
    ranks = vt.voxel_pooling_prepare_v2(coor)
    for name, r in zip(["ranks_bev", "ranks_depth", "ranks_feat", "interval_starts", "interval_lengths"], ranks):
        a = r.numpy()
        assert a.dtype == np.int32 and a.size == int(g[name + "_len"])
        assert np.array_equal(a[:64], g[name + "_head"])
        assert _digest(a) == str(g[name + "_sha256"]), name
    # structure: sorted cells, intervals partition the points
    rb, st, ln = ranks[0].numpy(), ranks[3].numpy(), ranks[4].numpy()
    assert (np.diff(rb) >= 0).all() and st[0] == 0 and (st[1:] == np.cumsum(ln)[:-1]).all() and ln.sum() == rb.size
    assert rb.max() < 128 * 128
