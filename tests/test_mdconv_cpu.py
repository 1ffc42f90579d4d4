"""CPU: pin the DCNv2 oracle by construction (no runnable reference for this op here):
zero offsets + unit mask == conv2d; integer offsets == conv over a shifted image; mask
scaling is linear; groups / deform_groups / stride / dilation handled like conv2d."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def rnd(*s, seed=0):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("cfg", [
    dict(B=2, Cin=8, Cout=6, H=9, W=11, K=3, stride=1, pad=1, dil=1, g=1, dg=1),
    dict(B=1, Cin=8, Cout=8, H=10, W=7, K=3, stride=2, pad=1, dil=1, g=2, dg=2),
    dict(B=2, Cin=4, Cout=4, H=8, W=8, K=3, stride=1, pad=2, dil=2, g=1, dg=4),
    dict(B=1, Cin=6, Cout=3, H=5, W=6, K=1, stride=1, pad=0, dil=1, g=3, dg=1),
])
def test_zero_offset_is_conv2d(oracle_mod, cfg):
    c = cfg
    x, w, b = rnd(c["B"], c["Cin"], c["H"], c["W"]), rnd(c["Cout"], c["Cin"] // c["g"], c["K"], c["K"], seed=1), rnd(c["Cout"], seed=2)
    want = F.conv2d(x, w, b, c["stride"], c["pad"], c["dil"], c["g"])
    Ho, Wo = want.shape[2:]
    off = torch.zeros(c["B"], c["dg"] * 2 * c["K"] ** 2, Ho, Wo)
    mask = torch.ones(c["B"], c["dg"] * c["K"] ** 2, Ho, Wo)
    out = oracle_mod.mdconv(x.numpy(), off.numpy(), mask.numpy(), w.numpy(), b.numpy(),
                            (c["stride"],) * 2, (c["pad"],) * 2, (c["dil"],) * 2, c["g"], c["dg"])
    np.testing.assert_allclose(out, want.numpy(), rtol=1e-4, atol=1e-4)
    # mask linearity
    out_half = oracle_mod.mdconv(x.numpy(), off.numpy(), 0.5 * mask.numpy(), w.numpy(), None,
                                 (c["stride"],) * 2, (c["pad"],) * 2, (c["dil"],) * 2, c["g"], c["dg"])
    np.testing.assert_allclose(out_half, 0.5 * (want - b.view(1, -1, 1, 1)).numpy(), rtol=1e-4, atol=1e-4)


def test_integer_offsets_shift_the_image(oracle_mod):
    x, w = rnd(1, 4, 12, 12), rnd(5, 4, 3, 3, seed=1)
    dy, dx = 2, -1
    off = torch.zeros(1, 18, 12, 12)
    off[:, 0::2] = dy   # h offsets first, then w (kernel.cu:288-296)
    off[:, 1::2] = dx
    mask = torch.ones(1, 9, 12, 12)
    out = oracle_mod.mdconv(x.numpy(), off.numpy(), mask.numpy(), w.numpy(), None, (1, 1), (1, 1), (1, 1), 1, 1)
    # sampling x at (h+dy, w+dx) == conv over the image shifted by (-dy, -dx) with zero fill
    xs = torch.zeros(1, 4, 12 + 8, 12 + 8)
    xs[:, :, 4:16, 4:16] = x
    shifted = xs[:, :, 4 + dy:16 + dy, 4 + dx:16 + dx]
    # zero-padding semantics differ only where the 3x3 window leaves the ORIGINAL image
    want = F.conv2d(F.pad(x, (4, 4, 4, 4)), w)[:, :, 3 + dy:15 + dy, 3 + dx:15 + dx]
    np.testing.assert_allclose(out, want.numpy(), rtol=1e-4, atol=1e-4)
    assert shifted.shape == x.shape


def test_fractional_offset_is_bilinear_blend(oracle_mod):
    """A uniform (0, +0.25) offset equals 0.75*conv(x) + 0.25*conv(x shifted by one column)."""
    x, w = rnd(1, 2, 8, 10), rnd(3, 2, 3, 3, seed=1)
    off = torch.zeros(1, 18, 8, 10)
    off[:, 1::2] = 0.25
    mask = torch.ones(1, 9, 8, 10)
    out = oracle_mod.mdconv(x.numpy(), off.numpy(), mask.numpy(), w.numpy(), None, (1, 1), (1, 1), (1, 1), 1, 1)
    xp = F.pad(x, (2, 2, 2, 2))
    a = F.conv2d(xp, w)[:, :, 1:9, 1:11]
    b = F.conv2d(xp, w)[:, :, 1:9, 2:12]
    np.testing.assert_allclose(out, (0.75 * a + 0.25 * b).numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("cfg", [dict(B=2, Cin=8, Cout=6, H=9, W=11, stride=1, pad=1, dil=1),
                                 dict(B=1, Cin=4, Cout=5, H=12, W=10, stride=2, pad=1, dil=1),
                                 dict(B=1, Cin=4, Cout=4, H=8, W=9, stride=1, pad=2, dil=2)])
def test_torch_statement_of_dcn_matches_the_c_oracle(oracle_mod, cfg):
    """oracle/ref_ops.py: mdconv_torch (the DCNv2 formulation the model-level parity tests of the big configs evaluate
    on the device) against oracle.mdconv (pinned bit-exact against the reference's own kernels compiled for the host,
    tests/test_ref_kernels_cpu.py): offsets of several pixels, taps that leave the image on every side, masks in [0, 1]."""
    from oracle.ref_ops import mdconv_torch
    c = cfg
    x, w, b = rnd(c["B"], c["Cin"], c["H"], c["W"]), rnd(c["Cout"], c["Cin"], 3, 3, seed=1), rnd(c["Cout"], seed=2)
    Ho = (c["H"] + 2 * c["pad"] - c["dil"] * 2 - 1) // c["stride"] + 1
    Wo = (c["W"] + 2 * c["pad"] - c["dil"] * 2 - 1) // c["stride"] + 1
    off = rnd(c["B"], 18, Ho, Wo, seed=3) * 2.5
    mask = torch.sigmoid(rnd(c["B"], 9, Ho, Wo, seed=4))
    want = oracle_mod.mdconv(x.numpy(), off.numpy(), mask.numpy(), w.numpy(), b.numpy(), (c["stride"],) * 2,
                             (c["pad"],) * 2, (c["dil"],) * 2, 1, 1)
    got = mdconv_torch(x, off, mask, w, b, c["stride"], c["pad"], c["dil"]).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
