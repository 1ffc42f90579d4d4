"""CPU: pin the bev_pool oracle on the reference test's own index tensors and on
synthetic ones against an independent torch.index_add_ statement."""
import numpy as np
import pytest

from conftest import golden
from util_bevpool import index_add_reference, make_indices


def test_oracle_on_reference_test_indices(oracle_mod):
    g = golden("bev_pool_ref_ranks")   # test_bev_pool_v2.py:6-13 -> 699899 points, 29351 intervals
    assert g["ranks_bev"].shape == (699899,) and g["interval_starts"].shape == (29351,)
    rng = np.random.default_rng(0)
    depth = rng.random((6, 160, 32, 88), dtype=np.float32)
    feat = rng.standard_normal((6, 32, 88, 16), dtype=np.float32)   # C=16 keeps the CPU test quick
    out = oracle_mod.bev_pool_v2(depth, feat, g["ranks_depth"], g["ranks_feat"], g["ranks_bev"],
                                 g["interval_starts"], g["interval_lengths"], 200, 200)
    want = index_add_reference(depth, feat, g["ranks_depth"], g["ranks_feat"], g["ranks_bev"], 200, 200)
    np.testing.assert_allclose(out, want, rtol=1e-4, atol=1e-3)   # fp32 sums of up to 1930 terms


def test_oracle_synthetic_and_empty_cells(oracle_mod):
    rd, rf, rb, ist, il = make_indices(2, 5, 4, 6, 8, 8, keep=0.5, seed=3)
    rng = np.random.default_rng(1)
    depth = rng.random((2, 5, 4, 6), dtype=np.float32)
    feat = rng.standard_normal((2, 4, 6, 7), dtype=np.float32)
    out = oracle_mod.bev_pool_v2(depth, feat, rd, rf, rb, ist, il, 8, 8)
    want = index_add_reference(depth, feat, rd, rf, rb, 8, 8)
    np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-5)
    untouched = np.setdiff1d(np.arange(64), rb)
    assert untouched.size and not out.reshape(64, -1)[untouched].any()


def test_oracle_int8_exact(oracle_mod):
    rd, rf, rb, ist, il = make_indices(2, 5, 4, 6, 8, 8, keep=0.5, seed=4)
    rng = np.random.default_rng(2)
    depth = rng.integers(0, 128, (2, 5, 4, 6), dtype=np.int8)
    feat = rng.integers(-127, 128, (2, 4, 6, 8), dtype=np.int8)
    sio = 0.004
    out = oracle_mod.bev_pool_v2(depth, feat, rd, rf, rb, ist, il, 8, 8, scale_io=sio)
    acc = index_add_reference(depth, feat, rd, rf, rb, 8, 8) * np.float32(sio)
    want = np.clip(acc, -128, 127)
    want = np.trunc(want + np.where(want > 0, 0.5, -0.5))
    assert np.abs(out.astype(np.int32) - want.astype(np.int32)).max() <= 1
