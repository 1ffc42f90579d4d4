"""GPU parity tests: rotate and grid_sampler HIP kernels vs the CPU oracle and the
golden vectors the reference's Python functions produced.
Tolerances (reference tests in brackets, mean-abs):
  fp32 bilinear/bicubic: element-wise 1e-5 (+1e-5 rel)       [grid_sampler 1e-5, rotate 1e-4]
  fp32 nearest: identical except <=0.1% of pixels (.5 ties)    [0.1 mean]
  fp16: element-wise 1e-2 vs fp32 evaluation of the fp16-rounded inputs [0.05 .. 0.6 mean]
  int8: |err| <= 1 LSB, <=1% of elements off by one            [0.1 .. 0.5 mean]
"""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

MODES = ("bilinear", "nearest", "bicubic")
PADS = ("zeros", "border", "reflection")


@pytest.fixture(scope="module")
def bev():
    import bevformer_tensorrt_amd as b
    from bevformer_tensorrt_amd.utils import load_library
    load_library()
    return b


def ref_grid(N, Ho, Wo, lo=-15.0, hi=15.0, seed=0, jitter=0.0):
    """test_grid_sampler.py:22-37: linspace(-15, 15) meshgrid (50% out of range)."""
    gy, gx = torch.meshgrid(torch.linspace(lo, hi, Ho), torch.linspace(lo, hi, Wo), indexing="ij")
    grid = torch.stack([gx, gy], 0)[None].repeat(N, 1, 1, 1)
    if jitter:
        grid = grid + torch.randn(grid.shape, generator=torch.Generator().manual_seed(seed)) * jitter
    return grid


def close_nearest(out, want, frac=1e-3):
    return float((out != want).mean()) <= frac


# ------------------------------------------------------------------ grid_sampler 2-D
@pytest.mark.parametrize("mode", range(3))
@pytest.mark.parametrize("pad", range(3))
@pytest.mark.parametrize("align", [False, True])
def test_grid_sampler_2d_golden(bev, mode, pad, align):
    g = golden("grid_sampler_2d")
    out = bev.grid_sampler(torch.from_numpy(g["input"]).cuda(), torch.from_numpy(g["grid"]).cuda(),
                           MODES[mode], PADS[pad], align).cpu().numpy()
    want = g[f"{MODES[mode]}_{PADS[pad]}_{int(align)}"]
    if mode == 1:
        assert close_nearest(out, want)
    else:
        np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("mode", range(3))
@pytest.mark.parametrize("pad", range(3))
@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_grid_sampler_2d_reference_shape(bev, oracle_mod, mode, pad, align, dtype):
    """reference test shape [8,32,100,100] (test_grid_sampler.py:5-7), output 301x301."""
    g = torch.Generator().manual_seed(0)
    inp = torch.randn(8, 32, 100, 100, generator=g).to(dtype)
    grid = ref_grid(8, 301, 301, jitter=0.3).to(dtype)
    out = bev.grid_sampler(inp.cuda(), grid.cuda(), MODES[mode], PADS[pad], align).float().cpu().numpy()
    want = oracle_mod.grid_sampler(inp.float().numpy(), grid.float().numpy(), mode, pad, align)
    if dtype == torch.float32:
        if mode == 1:
            assert close_nearest(out, want)
        else:
            np.testing.assert_allclose(out, want, rtol=1e-5, atol=2e-5)
    else:
        if mode == 1:
            assert close_nearest(out, want)
        else:
            assert np.abs(out - want).max() <= 1e-2 * max(1.0, np.abs(want).max() / 4)


def test_grid_sampler_2d_full_reference_size(bev, oracle_mod):
    """The full reference test size: grid [8,2,1001,1001] (test_grid_sampler.py:22-37)."""
    g = torch.Generator().manual_seed(0)
    inp = torch.randn(8, 32, 100, 100, generator=g)
    grid = ref_grid(8, 1001, 1001)
    out = bev.grid_sampler(inp.cuda(), grid.cuda(), "bilinear", "zeros", False)
    torch.cuda.synchronize()
    want = oracle_mod.grid_sampler(inp.numpy(), grid.numpy(), 0, 0, False)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=2e-5)
    # identity grid property at full size: align_corners=True, grid = pixel centres -> copy
    gy, gx = torch.meshgrid(torch.linspace(-10, 10, 100), torch.linspace(-10, 10, 100), indexing="ij")
    ident = torch.stack([gx, gy], 0)[None].repeat(8, 1, 1, 1)
    out = bev.grid_sampler(inp.cuda(), ident.cuda(), "bilinear", "zeros", True).cpu()
    assert (out - inp).abs().max().item() <= 2e-4  # linspace rounding ~1e-5 px
    out = bev.grid_sampler(inp.cuda(), ident.cuda(), "nearest", "border", True).cpu()
    assert torch.equal(out, inp)


@pytest.mark.parametrize("mode", range(2))
@pytest.mark.parametrize("pad", range(3))
@pytest.mark.parametrize("align", [False, True])
def test_grid_sampler_3d_golden(bev, mode, pad, align):
    g = golden("grid_sampler_3d")
    out = bev.grid_sampler(torch.from_numpy(g["input"]).cuda(), torch.from_numpy(g["grid"]).cuda(),
                           MODES[mode], PADS[pad], align).cpu().numpy()
    want = g[f"{MODES[mode]}_{PADS[pad]}_{int(align)}"]
    if mode == 1:
        assert close_nearest(out, want)
    else:
        np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("mode", range(2))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_grid_sampler_3d_reference_shape(bev, oracle_mod, mode, dtype):
    """[8,32,10,10,10] -> grid [8,3,31,31,31] (test_grid_sampler.py:9-11 uses 101^3)."""
    g = torch.Generator().manual_seed(0)
    inp = torch.randn(8, 32, 10, 10, 10, generator=g).to(dtype)
    lin = torch.linspace(-14, 14, 31)
    gz, gy, gx = torch.meshgrid(lin, lin, lin, indexing="ij")
    grid = (torch.stack([gx, gy, gz], 0)[None].repeat(8, 1, 1, 1, 1)
            + torch.randn(8, 3, 31, 31, 31, generator=g) * 0.3).to(dtype)
    for pad in range(3):
        out = bev.grid_sampler(inp.cuda(), grid.cuda(), MODES[mode], PADS[pad], False).float().cpu().numpy()
        want = oracle_mod.grid_sampler(inp.float().numpy(), grid.float().numpy(), mode, pad, False)
        if mode == 1:
            assert close_nearest(out, want)
        elif dtype == torch.float32:
            np.testing.assert_allclose(out, want, rtol=1e-5, atol=2e-5)
        else:
            assert np.abs(out - want).max() <= 1e-2


def test_grid_sampler_bicubic_3d_not_supported(bev):
    from bevformer_tensorrt_amd.utils.lib import BevopsError
    with pytest.raises(BevopsError):
        bev.grid_sampler(torch.zeros(1, 1, 2, 2, 2).cuda(), torch.zeros(1, 3, 2, 2, 2).cuda(),
                         "bicubic", "zeros", False)


@pytest.mark.parametrize("mode", range(3))
@pytest.mark.parametrize("pad", range(3))
def test_grid_sampler_int8(bev, oracle_mod, mode, pad):
    g = torch.Generator().manual_seed(0)
    inp = torch.randint(-127, 128, (4, 16, 40, 50), generator=g, dtype=torch.int8)
    s_grid = 15.0 / 127
    grid = torch.round(ref_grid(4, 77, 91, jitter=0.5) / s_grid).clamp(-127, 127).to(torch.int8)
    s_in, s_out = 0.031, 0.027
    out = bev.grid_sampler_int8(inp.cuda(), grid.cuda(), MODES[mode], PADS[pad], False, s_in, s_grid,
                                s_out).cpu().numpy().astype(np.int32)
    want = oracle_mod.grid_sampler_s8(inp.numpy(), grid.numpy(), mode, pad, False, s_in, s_grid,
                                      s_out).astype(np.int32)
    d = np.abs(out - want)
    assert d.max() <= 1 and (d > 0).mean() <= 0.01
    # and the int8 result tracks the fp32 op within the quantisation error
    f = oracle_mod.grid_sampler(inp.numpy().astype(np.float32) * s_in,
                                grid.numpy().astype(np.float32) * s_grid, mode, pad, False)
    assert np.abs(np.clip(f / s_out, -128, 127) - out).mean() <= (1.0 if mode < 2 else 12.0)  # int8 bicubic: truncated x127 coefficients, two
    # truncating divisions, white-noise overshoot (reference tolerance 0.4 real units ~ 13 LSB)


# ------------------------------------------------------------------ rotate
@pytest.mark.parametrize("case", ["sq_small", "rect", "offcenter", "bev_like"])
def test_rotate_golden(bev, case):
    g = golden("rotate_" + case)
    img = torch.from_numpy(g["img"]).cuda()
    ang = torch.tensor(float(g["angle"])).cuda()
    ctr = torch.from_numpy(g["center"]).cuda()
    ob = bev.rotate(img, ang, ctr, "bilinear").cpu().numpy()
    np.testing.assert_allclose(ob, g["bilinear"], rtol=1e-4, atol=1e-4)
    on = bev.rotate(img, ang, ctr, "nearest").cpu().numpy()
    assert close_nearest(on, g["nearest"], 2e-3)


@pytest.mark.parametrize("shape", [(256, 50, 50), (256, 150, 150), (256, 200, 200)],
                         ids=["tiny", "small", "base"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("interp", ["nearest", "bilinear"])
def test_rotate_model_shapes(bev, oracle_mod, shape, dtype, interp):
    """prev_bev [256, bev_h, bev_w], centre = rotate_center [100,100]
    (det2trt/models/modules/transformer.py:26,298-302), small ego-motion angle."""
    g = torch.Generator().manual_seed(0)
    img = torch.randn(*shape, generator=g).to(dtype)
    for angle in (1.7, -38.25):
        center = (100.0, 100.0)
        out = bev.rotate(img.cuda(), torch.tensor(angle), torch.tensor(center), interp).float().cpu().numpy()
        want = oracle_mod.rotate(img.float().numpy(), angle, center, 0 if interp == "bilinear" else 1)
        if interp == "nearest":
            assert close_nearest(out, want, 2e-3)   # pure copies: bit-exact off the .5 ties
        elif dtype == torch.float32:
            np.testing.assert_allclose(out, want, rtol=1e-4, atol=1e-4)
        else:
            assert np.abs(out - want).max() <= 1e-2


def test_rotate_reference_test_shape(bev, oracle_mod):
    """test_rotate.py:6-9,21-25: img randn[256,512,512], angle randn*360, centre [500,500]."""
    g = torch.Generator().manual_seed(0)
    img = torch.randn(256, 512, 512, generator=g)
    angle = float(torch.randn(1, generator=g)) * 360
    out = bev.rotate(img.cuda(), torch.tensor(angle).cuda(), torch.tensor([500.0, 500.0]).cuda(),
                     "bilinear").cpu().numpy()
    want = oracle_mod.rotate(img.numpy(), angle, (500.0, 500.0), 0)
    # source coordinates reach ~1e3 px here, so fp32 sin/cos ulps show up at ~1e-4 px
    assert np.abs(out - want).mean() <= 1e-4           # the reference's own criterion
    assert np.abs(out - want).max() <= 5e-3


def test_rotate_properties_full_base_size(bev):
    g = torch.Generator().manual_seed(1)
    img = torch.randn(256, 200, 200, generator=g).half().cuda()
    ctr = torch.tensor([100.0, 100.0]).cuda()
    # angle 0 about the image centre is the identity for both modes
    assert torch.equal(bev.rotate(img, torch.tensor(0.0).cuda(), ctr, "nearest"), img)
    # (bilinear: the affine grid of functions/rotate.py is only integer to ~1e-5 px)
    ident = bev.rotate(img, torch.tensor(0.0).cuda(), ctr, "bilinear")
    assert (ident.float() - img.float()).abs().max().item() <= 4e-3
    # 90 + 270 degrees (nearest, centre) returns the original image
    r = bev.rotate(bev.rotate(img, torch.tensor(90.0).cuda(), ctr, "nearest"),
                   torch.tensor(270.0).cuda(), ctr, "nearest")
    assert (r != img).float().mean().item() <= 2e-2  # border row/column may fall outside
    # rotate2 is the same op
    a = torch.tensor(12.5).cuda()
    assert torch.equal(bev.rotate(img, a, ctr), bev.rotate2(img, a, ctr))


@pytest.mark.parametrize("interp", ["nearest", "bilinear"])
def test_rotate_int8(bev, oracle_mod, interp):
    g = torch.Generator().manual_seed(0)
    img = torch.randint(-127, 128, (64, 50, 50), generator=g, dtype=torch.int8)
    s_in, s_out = 0.02, 0.025
    out = bev.rotate_int8(img.cuda(), torch.tensor(17.0), torch.tensor([25.0, 25.0]), s_in, s_out,
                          interp).cpu().numpy().astype(np.int32)
    want = oracle_mod.rotate_s8(img.numpy(), 17.0, (25.0, 25.0), 0 if interp == "bilinear" else 1,
                                s_in, s_out).astype(np.int32)
    d = np.abs(out - want)
    assert d.max() <= 1 and (d > 0).mean() <= 0.01


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("interp", ["nearest", "bilinear"])
def test_rotate_hwc_equals_rotate_on_permuted_data(dtype, interp):
    """The channels-last entry (prev_bev's own [H, W, C] layout) gives, element for element, what
    the plugin-layout op gives on the permuted tensor."""
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(0)
    for C, H, W, ang, ctr in [(256, 50, 50, 7.3, (25.0, 25.0)), (64, 31, 45, -101.0, (20.0, 11.0)), (8, 5, 7, 45.0, (3.0, 2.0))]:
        img = torch.randn(C, H, W, generator=g).to(dtype).cuda()
        a, c = torch.tensor(ang).cuda(), torch.tensor(ctr).cuda()
        want = bev.rotate(img, a, c, interp)
        got = bev.rotate_hwc(img.permute(1, 2, 0).contiguous(), a, c, interp).permute(2, 0, 1)
        assert torch.equal(got, want), (C, H, W, (got.float() - want.float()).abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_grid_sampler_staged_path_is_bit_identical(dtype):
    """Up-sampling bilinear / nearest calls stage the input channels-last in a lent workspace
    (bevops_grid_sampler_2d_forward_ws); the planar kernel must give the same bits."""
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.utils import lib as L
    h = L.load_library()
    g = torch.Generator().manual_seed(0)
    for (N, C, H, W, Ho, Wo) in [(2, 16, 9, 11, 40, 37), (1, 32, 20, 20, 101, 99), (3, 8, 5, 7, 13, 6)]:
        x = torch.randn(N, C, H, W, generator=g).to(dtype).cuda()
        grid = ((torch.rand(N, 2, Ho, Wo, generator=g) * 2 - 1) * 12).to(dtype).cuda()
        assert h.bevops_grid_sampler_2d_workspace_size(L.torch_dtype_code(x), N, C, H, W) > 0
        for mode, mi in (("bilinear", 0), ("nearest", 1)):
            for pad, pi in (("zeros", 0), ("border", 1), ("reflection", 2)):
                for align in (False, True):
                    staged = bev.grid_sampler(x, grid, mode, pad, align)          # workspace lent by the wrapper
                    planar = torch.empty_like(staged)
                    st = h.bevops_grid_sampler_2d_forward(L.torch_dtype_code(x), x.data_ptr(), grid.data_ptr(),
                                                          planar.data_ptr(), N, C, H, W, Ho, Wo, mi, pi, int(align),
                                                          1.0, 1.0, 1.0, torch.cuda.current_stream().cuda_stream)
                    assert st == 0
                    torch.cuda.synchronize()
                    assert torch.equal(staged, planar), (N, C, H, W, mode, pad, align)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32, torch.int8])
@pytest.mark.parametrize("interp", ["nearest", "bilinear"])
@pytest.mark.parametrize("shape", [(256, 200, 200), (20, 50, 52), (9, 31, 40), (5, 7, 9)])
def test_rotate_wide_stores_equal_per_lane_stores(dtype, interp, shape):
    """bevops_rotate_forward stores through an LDS transpose of 8-pixel runs (16-byte stores) where every plane
    starts 16-byte aligned; `bevops_rotate_set_variant(1)` keeps the per-lane stores of rounds 1-3.  Same values
    bit for bit -- full and partial pixel blocks, partial channel chunks, plane sizes that fall back."""
    import bevformer_tensorrt_amd as bevm
    from bevformer_tensorrt_amd.utils import load_library
    lib = load_library()
    g = torch.Generator().manual_seed(sum(shape))
    C, H, W = shape
    ang, ctr = torch.tensor(17.5).cuda(), torch.tensor([W * 0.45, H * 0.55]).cuda()
    if dtype == torch.int8:
        img = torch.randint(-127, 128, shape, generator=g, dtype=torch.int8).cuda()
        run = lambda: bevm.rotate_int8(img, ang, ctr, 0.03, 0.04, interp)
    else:
        img = torch.randn(shape, generator=g).to(dtype).cuda()
        run = lambda: bevm.rotate(img, ang, ctr, interp)
    outs = []
    for variant in (0, 1):
        prev = lib.bevops_rotate_set_variant(variant)
        try:
            outs.append(run())
        finally:
            lib.bevops_rotate_set_variant(prev)
    assert torch.equal(outs[0], outs[1])
