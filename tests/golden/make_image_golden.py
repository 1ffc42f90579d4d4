#!/usr/bin/env python3
"""Golden vectors for the camera-image front end (NormalizeMultiviewImage + PadMultiViewImage,
third_party/bev_mmdet3d/datasets/pipelines/transform_3d.py:99-150,171-190).  mmcv / cv2 are not installed
here and the reference ships no fixture of theirs, so the pin is the PUBLISHED definition of
mmcv.imnormalize evaluated exactly: (float64(pixel) - mean) * (1 / std) in float64 (pixels are integers
0..255, so the float64 evaluation is exact up to the one rounding of 1 / std), bottom / right zero padding
to a multiple of 32.  The float32 pipeline of mmcv (two roundings per pixel) must agree with it within
2 float32 ulps of the result -- tests/test_oracle_golden.py::test_image_ref_against_float64_definition.
Writes tests/golden/image_norm.npz (small: 2 images of 37 x 45)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, size=(2, 37, 45, 3), dtype=np.uint8)
    img[0, 0, 0] = (0, 0, 0)
    img[0, 0, 1] = (255, 255, 255)
    cases = {}
    for name, mean, std, to_rgb in (("caffe", (103.530, 116.280, 123.675), (1.0, 1.0, 1.0), False),
                                    ("torch", (123.675, 116.28, 103.53), (58.395, 57.12, 57.375), True)):
        x = img.astype(np.float64)
        if to_rgb:
            x = x[..., ::-1]
        y = (x - np.asarray(mean, np.float64)) * (1.0 / np.asarray(std, np.float64))
        out = np.zeros((2, 64, 64, 3), np.float64)
        out[:, :37, :45] = y
        cases[name + "_out64"] = np.ascontiguousarray(out.transpose(0, 3, 1, 2))
        cases[name + "_mean"] = np.asarray(mean, np.float64)
        cases[name + "_std"] = np.asarray(std, np.float64)
        cases[name + "_to_rgb"] = np.asarray(to_rgb)
    np.savez_compressed(os.path.join(HERE, "image_norm.npz"), img=img, **cases)


if __name__ == "__main__":
    main()
