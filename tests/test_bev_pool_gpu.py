"""GPU parity: bev_pool_v2 HIP kernel vs oracle at the reference test shape and the
BEVDet-R50 model shape."""
import numpy as np
import pytest
import torch

from conftest import golden
from util_bevpool import make_indices

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bev():
    import bevformer_tensorrt_amd as b
    return b


def t(a):
    return torch.from_numpy(a).cuda()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_reference_test_shape(bev, oracle_mod, dtype):
    """depth [6,160,32,88], feat [6,32,88,128] -> [1,200,200,128] with the reference
    test's own 699 899 points / 29 351 intervals (test_bev_pool_v2.py:6-13)."""
    g = golden("bev_pool_ref_ranks")
    gen = torch.Generator().manual_seed(0)
    depth = torch.rand(6, 160, 32, 88, generator=gen).softmax(1).to(dtype)
    feat = torch.randn(6, 32, 88, 128, generator=gen).to(dtype)
    # the reference pipeline hands the ranks over as float tensors (:242-246)
    ranks = [torch.from_numpy(g[k]).float().cuda() for k in
             ("ranks_depth", "ranks_feat", "ranks_bev", "interval_starts", "interval_lengths")]
    out = bev.bev_pool_v2(depth.cuda(), feat.cuda(), *ranks, 200, 200)
    assert out.shape == (1, 200, 200, 128)
    want = oracle_mod.bev_pool_v2(depth.float().numpy(), feat.float().numpy(), g["ranks_depth"],
                                  g["ranks_feat"], g["ranks_bev"], g["interval_starts"],
                                  g["interval_lengths"], 200, 200)
    got = out.float().cpu().numpy()
    if dtype == torch.float32:
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    else:
        assert np.abs(got - want).max() <= 1e-2 * max(1.0, np.abs(want).max())
    assert torch.equal(out, bev.bev_pool_v2_2(depth.cuda(), feat.cuda(), *ranks, 200, 200))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("C", [64, 7])
def test_bevdet_r50_shape(bev, oracle_mod, dtype, C):
    """BEVDet-R50: depth [6,59,16,44], feat [6,16,44,64] -> [1,128,128,64]."""
    rd, rf, rb, ist, il = make_indices(6, 59, 16, 44, 128, 128, keep=0.72, seed=0)
    gen = torch.Generator().manual_seed(0)
    depth = torch.rand(6, 59, 16, 44, generator=gen).to(dtype)
    feat = torch.randn(6, 16, 44, C, generator=gen).to(dtype)
    out = bev.bev_pool_v2(depth.cuda(), feat.cuda(), t(rd), t(rf), t(rb), t(ist), t(il), 128, 128)
    want = oracle_mod.bev_pool_v2(depth.float().numpy(), feat.float().numpy(), rd, rf, rb, ist, il,
                                  128, 128)
    got = out.float().cpu().numpy()
    if dtype == torch.float32:
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-5)
    else:
        assert np.abs(got - want).max() <= 1e-2 * max(1.0, np.abs(want).max())
    untouched = np.setdiff1d(np.arange(128 * 128), rb)
    assert not got.reshape(128 * 128, -1)[untouched].any()


@pytest.mark.parametrize("C", [64, 20, 3])
def test_int8_bit_exact(bev, oracle_mod, C):
    rd, rf, rb, ist, il = make_indices(6, 59, 16, 44, 128, 128, keep=0.72, seed=1)
    gen = torch.Generator().manual_seed(0)
    depth = torch.randint(0, 128, (6, 59, 16, 44), generator=gen, dtype=torch.int8)
    feat = torch.randint(-127, 128, (6, 16, 44, C), generator=gen, dtype=torch.int8)
    s_d, s_f, s_o = 1 / 127, 0.03, 0.21
    out = bev.bev_pool_v2_int8(depth.cuda(), feat.cuda(), t(rd), t(rf), t(rb), t(ist), t(il), s_d, s_f,
                               s_o, 128, 128)
    want = oracle_mod.bev_pool_v2(depth.numpy(), feat.numpy(), rd, rf, rb, ist, il, 128, 128,
                                  scale_io=np.float32(s_d) * np.float32(s_f) / np.float32(s_o))
    assert np.array_equal(out.cpu().numpy(), want)   # integer arithmetic: bit-exact


def test_no_intervals_gives_zeros(bev):
    z = torch.zeros(0, dtype=torch.int32).cuda()
    out = bev.bev_pool_v2(torch.rand(1, 2, 2, 2).cuda(), torch.rand(1, 2, 2, 8).cuda(), z, z, z, z, z, 4, 4)
    assert out.shape == (1, 4, 4, 8) and not out.any()
