"""GPU parity against the reference's own kernels, LIVE, at model-sized shapes.

`oracle/_ref/libbevref.so` (the reference's .cu kernels compiled for the host, oracle/Makefile) travels
to the GPU box with the snapshot; here it is executed on the host cores next to the HIP operators on
inputs far larger than the committed fixtures of tests/test_ref_kernels_gpu.py: BEVFormer tiny /
base call shapes (a query slice for the base SCA call), a ResNet-101 stage-3 DCNv2 image, the
reference test's own bev_pool index set, a slice of the reference's grid_sampler test, a 200x200
BEV rotate.  Skipped when the library was not built (no /root/reference at build time).
"""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    from oracle import refkernels
    # under `-m gpu` the reference-built checker must be there (it travels with the snapshot, built by
    # __graft_entry__.build()): a missing library is a FAILURE, not a silent coverage drop
    assert refkernels.available(), "oracle/_ref/libbevref.so not built -- run `make -C oracle` where /root/reference exists"
    refkernels.lib()
    return refkernels


@pytest.fixture(scope="module")
def bev():
    import bevformer_tensorrt_amd as b
    return b


def cu(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dt is None else t.to(dt)


def lsb(a, b):
    return np.abs(a.astype(np.int32) - b.astype(np.int32))


MSDA_SHAPES = {
    "tiny_sca": (6, [[15, 25]], 2500, 8, 4),
    "tiny_tsa": (2, [[50, 50]], 2500, 4, 1),
    "base_sca_q3k": (6, [[116, 200], [58, 100], [29, 50], [15, 25]], 3000, 8, 4),
    "base_tsa_q5k": (2, [[200, 200]], 5000, 4, 1),
    "base_dec": (1, [[200, 200]], 900, 4, 1),
}


def msda_inputs(shape, heads=8, C=32, seed=0):
    bs, levels, nq, P, ppg = shape
    rng = np.random.default_rng(seed)
    L = len(levels)
    nk = sum(h * w for h, w in levels)
    f = np.float32
    return (rng.standard_normal((bs, nk, heads, C)).astype(f), np.array(levels, np.int32),
            rng.uniform(0, 1, (bs, nq, 1, 2 * ppg)).astype(f),
            rng.standard_normal((bs, nq, heads, L * P * 2)).astype(f),
            rng.standard_normal((bs, nq, heads, L * P)).astype(f))


@pytest.mark.parametrize("name", list(MSDA_SHAPES))
def test_msda_fp32_fp16_vs_live_reference_kernels(bev, R, name):
    v, sh, r, o, w = msda_inputs(MSDA_SHAPES[name])
    want = R.msda(v, sh, r, o, w, R.F32)
    got = bev.multi_scale_deformable_attn(cu(v), cu(sh), cu(r), cu(o), cu(w)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-5)
    h = torch.float16
    got16 = bev.multi_scale_deformable_attn(cu(v, h), cu(sh), cu(r, h), cu(o, h), cu(w, h)).float().cpu().numpy()
    ref16 = R.msda(v, sh, r, o, w, R.H2).astype(np.float32)     # the reference's all-binary16 <__half2> kernel
    want16 = R.msda(v.astype(np.float16), sh, r.astype(np.float16), o.astype(np.float16), w.astype(np.float16), R.F32)
    # the half2 kernel's binary16 coordinates cost it 0.002 (SCA pyramids) ... 0.02 (200x200 maps) mean
    # abs error against fp32 math; ours must be within the reference's fp16 criterion (0.01) of it or
    # as close to it as fp32 math itself is, element-wise within 1e-2 of fp32 math, and closer than it
    assert np.abs(got16 - ref16).mean() <= max(0.01, 1.1 * np.abs(ref16 - want16).mean())
    assert np.abs(got16 - want16).max() <= 1e-2
    assert np.abs(got16 - want16).mean() <= np.abs(ref16 - want16).mean() + 1e-4


@pytest.mark.parametrize("name", ["tiny_sca", "base_sca_q3k", "base_dec"])
def test_msda_int8_vs_live_reference_kernels(bev, R, name):
    v, sh, r, o, w = msda_inputs(MSDA_SHAPES[name])
    q = lambda x: (np.clip(np.rint(x / (np.abs(x).max() / 127)), -127, 127).astype(np.int8), float(np.abs(x).max() / 127))
    (vq, sv), (oq, so), (wq, sw) = q(v), q(o), q(w)
    s_out = float(np.abs(R.msda(v, sh, r, o, w, R.F32)).max() / 127)
    want = R.msda_s8(vq, sv, sh, r, oq, so, wq, sw, s_out, ref_half=False)
    got = bev.multi_scale_deformable_attn_int8(cu(vq), cu(sh), cu(r), cu(oq), cu(wq), sv, so, sw, s_out).cpu().numpy()
    # <float> flavour: bit-exact against the reference's own kernel
    assert np.array_equal(got, want), (lsb(got, want).max(), (got != want).mean())
    want = R.msda_s8(vq, sv, sh, r.astype(np.float16), oq, so, wq, sw, s_out, ref_half=True)
    got = bev.multi_scale_deformable_attn_int8(cu(vq), cu(sh), cu(r, torch.float16), cu(oq), cu(wq), sv, so, sw,
                                               s_out).cpu().numpy()
    # <__half2> flavour: same integer pipeline, but the kernel evaluates sampling locations, the softmax
    # sum and the requantisation in binary16 (0.1 px on a 200-wide map; overflow beyond 65 504).  Measured
    # agreement of the fp32-math restatement with it: 87 % / 73 % / 40 % identical at tiny-SCA / base-SCA /
    # base-decoder shapes, >= 79 % within 1 LSB, >= 97.9 % within 3 -- and ours is the one closer to fp32
    d = lsb(got, want)
    assert (d <= 1).mean() >= 0.75 and (d <= 3).mean() >= 0.97, ((d <= 1).mean(), (d <= 3).mean())
    truth = R.msda(v, sh, r, o, w, R.F32) / s_out
    assert np.abs(got - truth).mean() <= np.abs(want - truth).mean() + 0.02


def test_dcn_r101_stage3_image_vs_live_reference_launcher(bev, R):
    """One camera of the stage-3 call (256 -> 256 channels, 58x100, 3x3): the reference's im2col kernel
    + GEMM + bias kernel on the host vs the fp32 and fp16 (fused MFMA) operators."""
    rng = np.random.default_rng(0)
    B, C, H, W = 1, 256, 58, 100
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    off = rng.standard_normal((B, 18, H, W)).astype(np.float32)
    mask = rng.uniform(0, 1, (B, 9, H, W)).astype(np.float32)
    w = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    want = R.mdconv(x, off, mask, w, b, 1, 1, 1, 1, 1, R.F32)
    scale = max(1.0, float(np.abs(want).max()))
    got = bev.modulated_deformable_conv2d(cu(x), cu(off), cu(mask), cu(w), cu(b), 1, 1, 1, 1, 1).cpu().numpy()
    assert np.abs(got - want).max() <= 1e-4 * scale
    h = torch.float16
    got16 = bev.modulated_deformable_conv2d(cu(x, h), cu(off, h), cu(mask, h), cu(w, h), cu(b, h), 1, 1, 1, 1,
                                            1).float().cpu().numpy()
    assert np.abs(got16 - want).max() <= 1e-2 * scale and np.abs(got16 - want).mean() <= 0.05


def test_bev_pool_reference_index_set_vs_live_reference_kernels(bev, R):
    g = golden("bev_pool_ref_ranks")
    rng = np.random.default_rng(0)
    depth = rng.uniform(0, 1, (6, 160, 32, 88)).astype(np.float32)
    feat = rng.standard_normal((6, 32, 88, 128)).astype(np.float32)
    idx = [g[k] for k in ("ranks_depth", "ranks_feat", "ranks_bev", "interval_starts", "interval_lengths")]
    want = R.bev_pool_v2(depth, feat, *idx, 200, 200, R.F32)
    got = bev.bev_pool_v2(cu(depth), cu(feat), *[cu(a) for a in idx], 200, 200).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-5)
    sd, sf = 1.0 / 127, float(np.abs(feat).max() / 127)
    so = float(np.abs(want).max() / 127)
    dq = np.clip(np.rint(depth / sd), -127, 127).astype(np.int8)
    fq = np.clip(np.rint(feat / sf), -127, 127).astype(np.int8)
    want8 = R.bev_pool_v2(dq, fq, *idx, 200, 200, scales=(sd, sf, so))
    got8 = bev.bev_pool_v2_int8(cu(dq), cu(fq), *[cu(a) for a in idx], sd, sf, so, 200, 200).cpu().numpy()
    assert np.array_equal(got8, want8)


def test_grid_sampler_reference_test_slice_vs_live_reference_kernel(bev, R):
    """Two images of the reference's own test (test_grid_sampler.py: input [8,32,100,100], grid =
    linspace(-15, 15) meshgrid, 1001x1001 cut to 301x301 around the image border) -- exercises the
    channels-last staged path of the operator."""
    rng = np.random.default_rng(0)
    inp = rng.standard_normal((2, 32, 100, 100)).astype(np.float32)
    # + 7e-4: the regular grid otherwise puts 4 % of the samples exactly on .5 ties, where the plugin
    # (::round, half away from zero) and the PyTorch path we follow (nearbyint, half to even) differ
    lin = np.linspace(-15, 15, 1001, dtype=np.float32)[250:551] + np.float32(7e-4)
    gy, gx = np.meshgrid(lin, lin, indexing="ij")
    grid = np.repeat(np.stack([gx, gy], 0)[None], 2, 0).astype(np.float32)
    for mode, mi in (("bilinear", 0), ("nearest", 1), ("bicubic", 2)):
        for pad, pi in (("zeros", 0), ("reflection", 2)):
            want = R.grid_sampler(inp, grid, mi, pi, 0, R.F32)
            got = bev.grid_sampler(cu(inp), cu(grid), mode, pad, False).cpu().numpy()
            if mi == 1:
                assert (got != want).mean() <= 2e-3     # .5 ties: the plugin rounds half away, aten to even
            else:
                np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)


def test_rotate_bev_sized_vs_live_reference_kernel(bev, R):
    rng = np.random.default_rng(0)
    img = rng.standard_normal((64, 200, 200)).astype(np.float32)
    for ang in (1.5, -37.0):
        for nm, mi in (("bilinear", 0), ("nearest", 1)):
            want = R.rotate(img, ang, (100.0, 100.0), mi, R.F32)
            got = bev.rotate(cu(img), torch.tensor(ang).cuda(), torch.tensor([100.0, 100.0]).cuda(), nm).cpu().numpy()
            if mi == 1:
                assert (got != want).mean() <= 2e-3
            else:
                np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)
            hwc = bev.rotate_hwc(cu(img).permute(1, 2, 0).contiguous(), torch.tensor(ang).cuda(),
                                 torch.tensor([100.0, 100.0]).cuda(), nm).permute(2, 0, 1).cpu().numpy()
            assert np.array_equal(hwc, got)
