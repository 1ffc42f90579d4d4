"""Index / grid generation must be BIT-EXACT against the reference's
own functions (golden produced by executing encoder.py:170-259 and transformer.py:262-294
on CPU, tests/golden/make_golden.py:make_geometry)."""
import numpy as np
import pytest
import torch

from conftest import golden
from bevformer_tensorrt_amd import geometry as G

PC_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]


@pytest.mark.parametrize("tag", ["tiny", "small40"])
def test_reference_points_and_point_sampling_bit_exact(tag):
    g = golden("geometry")
    bh, bw, ih, iw = (int(x) for x in g[f"{tag}_meta"])
    ref_3d = G.reference_points_3d(bh, bw, PC_RANGE[5] - PC_RANGE[2], 4, device="cpu")
    assert np.array_equal(ref_3d.numpy(), g[f"{tag}_ref3d"])
    l2i = G.synthetic_lidar2img((ih, iw))
    assert np.array_equal(l2i.numpy(), g[f"{tag}_lidar2img"])
    cam, mask = G.point_sampling(ref_3d, PC_RANGE, l2i, (ih, iw))
    assert np.array_equal(cam.numpy(), g[f"{tag}_cam"])
    assert np.array_equal(mask.numpy(), g[f"{tag}_mask"])
    # sanity of the synthetic rig: every pillar is seen by 0..3 cameras, most by >= 1
    seen = (mask > 0).sum(0).flatten()
    assert seen.max() <= 3 and (seen >= 1).float().mean() > 0.6


def _digest(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("projection", ["matmul", "fma"])
def test_base_size_point_sampling_bit_exact(projection):
    """BEVFormer-base: 200x200 pillars x 4 anchors x 6 cameras of 928x1600 (the addresses the
    base SCA call samples).  The golden holds SHA-256 digests of the reference's arrays (they are
    11 MB) plus every 37th query; both projection spellings must reproduce them on the CPU."""
    g = golden("geometry_base")
    bh, bw, ih, iw, step = (int(x) for x in g["meta"])
    ref_3d = G.reference_points_3d(bh, bw, PC_RANGE[5] - PC_RANGE[2], 4, device="cpu")
    l2i = G.synthetic_lidar2img((ih, iw))
    assert np.array_equal(l2i.numpy(), g["lidar2img"])
    cam, mask = G.point_sampling(ref_3d, PC_RANGE, l2i, (ih, iw), projection=projection)
    assert np.array_equal(ref_3d.numpy()[:, :, ::step], g["ref3d_sample"])
    assert np.array_equal(cam.numpy()[:, :, ::step], g["cam_sample"])
    assert np.array_equal(mask.numpy()[:, ::step], g["mask_sample"])
    assert [_digest(ref_3d.numpy()), _digest(cam.numpy()), _digest(mask.numpy())] == list(g["sha256"])


def test_bev_shift_bit_exact():
    g = golden("geometry")
    for can, want in zip(g["can_bus"], g["shift"]):
        got = G.bev_shift(torch.from_numpy(can), 200, 200)
        assert np.array_equal(got.numpy(), want)


def test_ref_2d_and_hybrid():
    ref_3d = G.reference_points_3d(5, 7, 8, 4, device="cpu")
    ref_2d = G.reference_points_2d(ref_3d)
    assert ref_2d.shape == (1, 35, 1, 2)
    assert torch.equal(ref_2d[0, :, 0], ref_3d[0, 0, :, :2])
    shift = torch.tensor([[0.01, -0.02]])
    hy = G.hybrid_ref_2d(ref_2d, shift, 1)
    assert hy.shape == (2, 35, 1, 2) and torch.equal(hy[1], ref_2d[0])
    assert torch.equal(G.hybrid_ref_2d(ref_2d, shift, 0)[0], ref_2d[0])


def test_level_layout():
    shapes, start = G.level_layout([(116, 200), (58, 100), (29, 50), (15, 25)])
    assert shapes.tolist() == [[116, 200], [58, 100], [29, 50], [15, 25]]
    assert start.tolist() == [0, 23200, 29000, 30450]
    assert int((shapes[:, 0] * shapes[:, 1]).sum()) == 30825
