"""Edge cases of the drop-in operators: empty inputs (the eager reference path returns empty tensors; the
plugins would be asked for a zero-sized launch), degenerate maps (1-pixel levels, single rows / columns),
reference points exactly on and beyond the borders, and ragged counts that do not fill a wave / tile."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bev():
    import bevformer_tensorrt_amd as b
    return b


@pytest.fixture(scope="module")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


def _msda_args(bs, levels, nq, heads, P, dtype=torch.float32, ppg=1, seed=0):
    g = torch.Generator().manual_seed(seed)
    nk = sum(h * w for h, w in levels)
    L = len(levels)
    value = torch.randn(bs, nk, heads, 32, generator=g)
    shapes = torch.tensor(levels, dtype=torch.int64).reshape(L, 2)
    ref = torch.rand(bs, nq, 1, 2 * ppg, generator=g)
    off = torch.randn(bs, nq, heads, L * P * 2, generator=g) * 2
    w = torch.randn(bs, nq, heads, L * P, generator=g)
    return [value.to(dtype).cuda(), shapes.cuda(), ref.to(dtype).cuda(), off.to(dtype).cuda(), w.to(dtype).cuda()]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_msda_empty_queries_and_empty_batch(bev, dtype):
    a = _msda_args(2, [[5, 7]], 0, 8, 4, dtype)
    out = bev.multi_scale_deformable_attn(*a)
    assert out.shape == (2, 0, 8, 32) and out.dtype == dtype
    a = _msda_args(0, [[5, 7]], 11, 8, 4, dtype)
    assert bev.multi_scale_deformable_attn(*a).shape == (0, 11, 8, 32)


@pytest.mark.parametrize("levels", [[[1, 1]], [[1, 9]], [[9, 1]], [[1, 1], [2, 3], [1, 5]]])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 1e-2)])
def test_msda_degenerate_maps(bev, oracle_mod, levels, dtype, tol):
    a = _msda_args(2, levels, 37, 8, 4, dtype, seed=3)
    a[2] = (a[2] * 1.6 - 0.3)                       # reference points from -0.3 to 1.3: on and beyond every border
    out = bev.multi_scale_deformable_attn(*a).float().cpu().numpy()
    want = oracle_mod.msda_f32(*[t.float().cpu().numpy() if t.is_floating_point() else t.cpu().numpy() for t in a])
    assert np.isfinite(out).all()
    assert np.abs(out - want).max() <= tol * max(1.0, np.abs(want).max())


def test_msda_reference_points_exactly_on_the_borders(bev, oracle_mod):
    a = _msda_args(1, [[4, 6], [2, 3]], 64, 8, 4, torch.float32, seed=5)
    edge = torch.tensor([0.0, 1.0, 0.5, 1.0 - 2 ** -24])
    a[2] = edge[torch.randint(0, 4, a[2].shape)].cuda()
    a[3] = torch.zeros_like(a[3])                   # samples land exactly on x = -0.5, W - 0.5, ...
    out = bev.multi_scale_deformable_attn(*a).cpu().numpy()
    want = oracle_mod.msda_f32(*[t.cpu().numpy() for t in a])
    assert np.abs(out - want).max() <= 2e-5


def test_rotate_and_grid_sampler_empty(bev):
    img = torch.randn(0, 8, 8).cuda()
    assert bev.rotate(img, torch.tensor(10.0), torch.tensor([4.0, 4.0])).shape == (0, 8, 8)
    inp = torch.randn(2, 3, 4, 5).cuda()
    grid = torch.zeros(2, 2, 0, 7).cuda()
    assert bev.grid_sampler(inp, grid, "bilinear", "zeros", False).shape == (2, 3, 0, 7)
    inp0 = torch.randn(0, 3, 4, 5).cuda()
    assert bev.grid_sampler(inp0, torch.zeros(0, 2, 6, 7).cuda(), "bilinear", "zeros", False).shape == (0, 3, 6, 7)


def test_mdconv_empty_batch_and_single_pixel(bev, oracle_mod):
    w = torch.randn(8, 8, 3, 3).cuda()
    b = torch.randn(8).cuda()
    x0 = torch.randn(0, 8, 5, 6).cuda()
    y0 = bev.modulated_deformable_conv2d(x0, torch.zeros(0, 18, 5, 6).cuda(), torch.zeros(0, 9, 5, 6).cuda(), w, b, 1, 1, 1, 1, 1)
    assert y0.shape == (0, 8, 5, 6)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 8, 1, 1, generator=g).cuda()               # a 1 x 1 image: every tap but the centre is padding
    off = (torch.randn(1, 18, 1, 1, generator=g) * 0.7).cuda()
    m = torch.rand(1, 9, 1, 1, generator=g).cuda()
    y = bev.modulated_deformable_conv2d(x, off, m, w, b, 1, 1, 1, 1, 1).cpu().numpy()
    want = oracle_mod.mdconv(x.cpu().numpy(), off.cpu().numpy(), m.cpu().numpy(), w.cpu().numpy(), b.cpu().numpy(),
                             (1, 1), (1, 1), (1, 1), 1, 1)
    assert np.abs(y - want).max() <= 1e-5


@pytest.mark.parametrize("nq", [1, 7, 63, 65])
def test_msda_int8_ragged_query_counts(bev, oracle_mod, nq):
    a = _msda_args(2, [[6, 10], [3, 5]], nq, 8, 4, torch.float32, seed=nq)

    def q(t):
        s = float(t.abs().max()) / 127.0
        return torch.clamp(torch.round(t / s), -127, 127).to(torch.int8), s
    qv, sv = q(a[0]); qo, so = q(a[3]); qw, sw = q(a[4])
    out = bev.multi_scale_deformable_attn_int8(qv, a[1], a[2], qo, qw, sv, so, sw, 0.05).cpu().numpy()
    want = oracle_mod.msda_s8(qv.cpu().numpy(), sv, a[1].cpu().numpy(), a[2].cpu().numpy(), qo.cpu().numpy(), so,
                              qw.cpu().numpy(), sw, 0.05)
    assert np.array_equal(out, want)
