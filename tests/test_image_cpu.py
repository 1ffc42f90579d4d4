"""oracle/image_ref.py (the camera-image front end's checker) against first principles: mmcv.imnormalize_ is
`cv2.subtract(img, mean)` then `cv2.multiply(img, 1 / std)` on a float32 image (third_party/bev_mmdet3d/datasets/
pipelines/transform_3d.py:57-88 -> mmcv/image/photometric.py), and OpenCV converts a scalar operand to the array's depth:
per pixel  r1 = fl32(fl32(x) - fl32(mean)),  r2 = fl32(r1 * fl32(1 / std))  -- two float32 roundings, NOT the float64
value of (x - mean) / std rounded once.  The expected values below are built with exact rational arithmetic
(fractions.Fraction) and an explicit round-to-nearest-even to binary32, independent of numpy's float32 operators."""
from fractions import Fraction

import numpy as np

from oracle import image_ref


def _fl32(q):
    """exact Fraction -> nearest binary32 (ties to even), returned as a Fraction."""
    if q == 0:
        return Fraction(0)
    s = -1 if q < 0 else 1
    q = abs(q)
    e = 0
    while q >= 2:
        q /= 2; e += 1
    while q < 1:
        q *= 2; e -= 1
    m = q * (1 << 23)                      # 1.xxx * 2^23, round to an integer
    lo = m.numerator // m.denominator
    rem = m - lo
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (lo & 1)):
        lo += 1
    return s * Fraction(lo, 1 << 23) * (Fraction(2) ** e)


def test_normalize_is_two_float32_roundings():
    mean, std = (103.530, 116.280, 123.675), (58.395, 57.12, 57.375)     # the reference's two config families
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(1, 5, 7, 3)).astype(np.uint8)
    for m, s in ((mean, (1.0, 1.0, 1.0)), (mean, std)):
        got = image_ref.image_normalize_pad(img, m, s, to_rgb=False, size_divisor=1)
        assert got.shape == (1, 3, 5, 7) and got.dtype == np.float32
        differs_from_float64 = False
        for y in range(5):
            for x in range(7):
                for c in range(3):
                    m32 = _fl32(Fraction(m[c]))
                    sinv = _fl32(Fraction(1.0 / np.float64(s[c])))       # 1 / std is formed in float64 by mmcv
                    r1 = _fl32(Fraction(int(img[0, y, x, c])) - m32)
                    want = _fl32(r1 * sinv)
                    assert Fraction(float(got[0, c, y, x])) == want, (y, x, c)
                    f64 = np.float32((np.float64(img[0, y, x, c]) - m[c]) / s[c])
                    differs_from_float64 |= Fraction(float(f64)) != want
        if s != (1.0, 1.0, 1.0):
            assert differs_from_float64      # the float64 formula is a DIFFERENT function: this test can tell them apart


def test_bgr_to_rgb_and_padding():
    img = np.arange(2 * 3 * 5 * 3, dtype=np.float32).reshape(2, 3, 5, 3)
    out = image_ref.image_normalize_pad(img, (0, 0, 0), (1, 1, 1), to_rgb=True, size_divisor=4)
    assert out.shape == (2, 3, 4, 8)
    assert np.array_equal(out[:, :, :3, :5], img[..., ::-1].transpose(0, 3, 1, 2))
    assert not out[:, :, 3:, :].any() and not out[:, :, :, 5:].any()       # bottom / right zero padding (impad_to_multiple)
