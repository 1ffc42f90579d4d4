"""CPU restatement of the reference's camera-image test pipeline (test infrastructure only):
NormalizeMultiviewImage -> PadMultiViewImage(size_divisor=32) -> DefaultFormatBundle3D
(configs/bevformer/bevformer_base.py:11,228-231; third_party/bev_mmdet3d/datasets/pipelines/
transform_3d.py:57-150, formating.py).  The arithmetic lives in mmcv.imnormalize_ (mmcv-full 1.5.0,
absent here): `cv2.cvtColor(BGR2RGB)` if to_rgb, `cv2.subtract(img, float64(mean))`,
`cv2.multiply(img, 1 / float64(std))` on a float32 image -- OpenCV converts the scalar operand to
the array's depth, i.e. two float32 roundings per pixel; mmcv.impad_to_multiple pads bottom / right
with zeros.  PARITY UNPINNED against mmcv / cv2 themselves (neither is installed): this is the
published algorithm, and the HIP op is held bit-exact against it."""
import numpy as np


def image_normalize_pad(images, mean=(103.530, 116.280, 123.675), std=(1.0, 1.0, 1.0), to_rgb=False, size_divisor=32):
    """images [N, H0, W0, 3] uint8 / float32 (BGR) -> float32 [N, 3, Hp, Wp]."""
    img = np.asarray(images).astype(np.float32)
    if to_rgb:
        img = img[..., ::-1]
    mean32 = np.asarray(mean, np.float64).astype(np.float32)
    stdinv = (1.0 / np.asarray(std, np.float64)).astype(np.float32)
    img = ((img - mean32).astype(np.float32) * stdinv).astype(np.float32)
    n, h, w, _ = img.shape
    hp, wp = -(-h // size_divisor) * size_divisor, -(-w // size_divisor) * size_divisor
    out = np.zeros((n, hp, wp, 3), np.float32)
    out[:, :h, :w] = img
    return np.ascontiguousarray(out.transpose(0, 3, 1, 2))
