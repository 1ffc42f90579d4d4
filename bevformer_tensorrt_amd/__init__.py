"""bevformer_tensorrt_amd -- MI355X-native sampling hot path for BEVFormer/BEVDet.

Host-side mirror of the reference's operator API (det2trt/models/functions/*.py,
det2trt/models/utils/register.py) over the C ABI of libbevops_hip.so
(include/bevops.h).  PyTorch is used for device memory, streams and
torch.distributed only; every op below runs a hand-written HIP kernel and fails
loudly if the library is missing -- there is no CPU or eager fallback.
"""
from .utils.register import TRT_FUNCTIONS  # noqa: F401
from . import functions  # noqa: F401  (registers the ops)
from .functions import *  # noqa: F401,F403

__version__ = "0.1.0"
