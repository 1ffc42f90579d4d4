"""ctypes loader for libbevops_hip.so (the C ABI in include/bevops.h).

Mirrors the reference's load hook -- `ctypes.CDLL(os.path.realpath(
"TensorRT/lib/libtensorrt_ops.so"))` at import (det2trt/models/utils/register.py:72-75)
-- except that the path is package-relative and a missing library is a hard error
with the build command, never a silent fallback.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); kept in step with include/bevops.h (tests check this)
SIGNATURES = {
    "bevops_version": (ctypes.c_char_p, []),
    "bevops_status_string": (ctypes.c_char_p, [c_int]),
    "bevops_query": (c_void_p, [ctypes.c_char_p]),
    "bevops_msda_set_variant": (c_int, [c_int]),
    "bevops_msda_forward": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_void_p, c_void_p] + [c_int] * 8 +
                            [c_float] * 4 + [c_void_p]),
    "bevops_msda_workspace_size": (c_size_t, [c_int] * 8),
    "bevops_msda_workspace_size_shapes": (c_size_t, [c_int, c_void_p] + [c_int] * 7),
    "bevops_msda_forward_ws": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p, c_void_p, c_void_p] + [c_int] * 8 +
                               [c_float] * 4 + [c_int, c_void_p, c_size_t, c_void_p]),
    "bevops_msda_packed_size": (c_size_t, [c_int, c_void_p] + [c_int] * 7),
    "bevops_msda_pack_value": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t] + [c_int] * 7 + [c_void_p]),
    "bevops_msda_forward_prepacked": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                              c_void_p] + [c_int] * 8 + [c_float] * 4 + [c_int, c_void_p]),
    "bevops_sca_workspace_size": (c_size_t, [c_int, c_void_p] + [c_int] * 7),
    "bevops_sca_forward": (c_int, [c_int] + [c_void_p] * 7 + [c_int] * 8 + [c_void_p, c_size_t, c_void_p]),
    "bevops_rotate_forward": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                      c_int, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "bevops_grid_sampler_2d_forward": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 9 +
                                       [c_float] * 3 + [c_void_p]),
    "bevops_grid_sampler_2d_workspace_size": (c_size_t, [c_int] * 5),
    "bevops_grid_sampler_2d_forward_ws": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 9 +
                                          [c_float] * 3 + [c_void_p, c_size_t, c_void_p]),
    "bevops_bev_pool_v2_forward": (c_int, [c_int] + [c_void_p] * 8 + [c_int] * 4 + [c_float] * 3 +
                                   [c_void_p]),
    "bevops_mdconv_set_variant": (c_int, [c_int]),
    "bevops_mdconv_workspace_size": (c_size_t, [c_int] * 16),
    "bevops_mdconv_forward_int8": (c_int, [c_void_p, c_float, c_void_p, c_float, c_void_p, c_float,
                                           c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p,
                                           c_size_t] + [c_int] * 15 + [c_void_p]),
    "bevops_mdconv_forward_int8_packed": (c_int, [c_void_p, c_float, c_void_p, c_float, c_void_p, c_float,
                                           c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p,
                                           c_size_t] + [c_int] * 15 + [c_void_p]),
    "bevops_mdconv_forward": (c_int, [c_int] + [c_void_p] * 7 + [c_size_t] + [c_int] * 15 +
                              [c_void_p]),
    "bevops_mdconv_forward_packed": (c_int, [c_int] + [c_void_p] * 7 + [c_size_t] + [c_int] * 15 +
                                     [c_void_p]),
    "bevops_mdconv_forward_nhwc": (c_int, [c_int] + [c_void_p] * 6 + [c_int, c_int, c_void_p, c_size_t] + [c_int] * 15 +
                                   [c_void_p]),
    "bevops_bias_act_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "bevops_value_proj_packed_size": (c_size_t, [c_void_p] + [c_int] * 7),
    "bevops_value_proj_packed": (c_int, [c_void_p] * 5 + [c_size_t] + [c_int] * 7 + [c_void_p]),
    "bevops_value_pack_planes": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t] + [c_int] * 7 + [c_void_p]),
    "bevops_sca_prepacked_workspace_size": (c_size_t, [c_int] * 4),
    "bevops_sca_forward_prepacked": (c_int, [c_int, c_void_p, c_size_t] + [c_void_p] * 6 + [c_int] * 8 +
                                     [c_void_p, c_size_t, c_void_p]),
    "bevops_sca_plan_size": (c_size_t, [c_int, c_int]),
    "bevops_sca_plan_build": (c_int, [c_int, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "bevops_sca_forward_planned": (c_int, [c_int, c_void_p, c_size_t] + [c_void_p] * 6 + [c_size_t, c_void_p] + [c_int] * 8 +
                                   [c_void_p, c_size_t, c_void_p]),
    "bevops_point_sampling": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float,
                                      c_void_p]),
    "bevops_tsgemm_f16": (c_int, [c_void_p] * 5 + [ctypes.c_longlong, c_int, c_int, c_int, c_void_p]),
    "bevops_tsgemm_f16_ln": (c_int, [c_void_p] * 6 + [c_float, c_void_p, ctypes.c_longlong, c_int, c_int, c_void_p]),
    "bevops_mha_selfattn_max_queries": (c_size_t, []),
    "bevops_mha_selfattn_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "bevops_tsgemm_s8": (c_int, [c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_float, c_int,
                                 c_void_p, c_float, ctypes.c_longlong, c_int, c_int, c_int, c_void_p]),
    "bevops_tsa_split": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "bevops_queue_mean2": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "bevops_upsample_add_nhwc": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "bevops_feat_embed_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_int, c_size_t,
                                       c_void_p]),
    "bevops_conv3x3_c32_set_variant": (c_int, [c_int]),
    "bevops_conv3x3_c32_packed_weight_size": (c_size_t, [c_int, c_int]),
    "bevops_conv3x3_c32_pack_weight": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "bevops_conv3x3_c32_forward_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                c_int, c_void_p]),
    "bevops_rotate_set_variant": (c_int, [c_int]),
    "bevops_rotate_forward_hwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                          c_int, c_int, c_int, c_int, c_void_p]),
    "bevops_layer_norm": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_float, c_void_p]),
    "bevops_linear_workspace_size": (c_size_t, []),
    "bevops_linear_bias_act": (c_int, [c_int] + [c_void_p] * 5 + [ctypes.c_longlong, c_int, c_int, c_int,
                                                                  c_void_p, c_size_t, c_void_p]),
    "bevops_image_normalize_pad": (c_int, [c_int, c_void_p, c_int, c_void_p] + [c_int] * 5 +
                                   [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), c_int, c_int, c_void_p]),
    "bevops_quantize_rows": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_float, c_void_p]),
    "bevops_dequantize_rows": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_float, c_void_p]),
    "bevops_linear_int8": (c_int, [c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p,
                                   c_float, ctypes.c_longlong, c_int, c_int, c_int, c_void_p]),
    "bevops_linear_int8_fused": (c_int, [c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p,
                                         c_float, ctypes.c_longlong, c_int, c_int, c_int, c_void_p]),
    "bevops_small_gemm_f16": (c_int, [c_void_p] * 5 + [ctypes.c_longlong, c_int, c_int, c_int, c_void_p]),
    "bevops_tile_gemm_f16": (c_int, [c_void_p] * 5 + [ctypes.c_longlong, c_int, c_int, c_int, c_void_p]),
    "bevops_conv_tile_f16": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    "bevops_conv3x3_c64_f16": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "bevops_refine_reference_points": (c_int, [c_int] + [c_void_p] * 4 + [c_int] * 2 + [c_void_p] * 3),
    "bevops_decode_boxes": (c_int, [c_int] + [c_void_p] * 3 + [c_int] + [c_float] * 6 + [c_void_p] * 3),
    "bevops_bias_relu_maxpool_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "bevops_stem_packed_size": (c_size_t, []),
    "bevops_stem_pack": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bevops_stem_conv_pool": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "bevops_stem_set_variant": (c_int, [c_int]),
    "bevops_conv_tile_int8_fused": (c_int, [c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]
                                    + [c_int] * 8 + [c_void_p]),
    "bevops_linear_int8_chain": (c_int, [c_void_p, c_int, c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int,
                                         c_float, c_int, c_void_p, c_float, ctypes.c_longlong, c_int, c_int, c_int, c_void_p]),
    "bevops_conv_tile_int8": (c_int, [c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p,
                                      c_float] + [c_int] * 8 + [c_void_p]),
    "bevops_bias_relu_maxpool_nhwc_int8": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int,
                                                   c_void_p]),
    "bevops_mdconv_int8_nhwc_workspace_size": (c_size_t, []),
    "bevops_mdconv_forward_int8_nhwc": (c_int, [c_void_p, c_float, c_void_p, c_int, c_float, c_float, c_void_p, c_float,
                                                c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_size_t] + [c_int] * 15 +
                                        [c_void_p]),
    "bevops_linear_tune": (c_int, [c_int] + [c_void_p] * 5 + [ctypes.c_longlong, c_int, c_int, c_int,
                                                              c_void_p, c_size_t, c_void_p]),
    "bevops_mdconv_packed_weight_size": (c_size_t, [c_int] * 5),
    "bevops_mdconv_pack_weight": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "bevops_grid_sampler_3d_forward": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 11 +
                                       [c_void_p]),
}

F32, F16, I8, U8 = 0, 1, 2, 3


def lib_path():
    return os.environ.get("BEVOPS_LIB", os.path.join(_PKG, "libbevops_hip.so"))


def load_library():
    """Load (once) and return the ctypes handle.  `import torch` first so that the
    HIP runtime torch ships (same SONAME libamdhip64.so.7) is the one both sides use."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build the HIP kernels first "
            "(`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C bevformer_tensorrt_amd/csrc`). There is no fallback path.")
    import torch  # noqa: F401
    handle = ctypes.CDLL(os.path.realpath(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the .so is stale
        fn.restype, fn.argtypes = res, args
    _LIB = handle
    return handle


SUCCESS, FAILURE, BAD_PARAM, NOT_SUPPORTED, NOT_INITIALIZED = range(5)   # include/bevops.h


class BevopsError(RuntimeError):
    """A non-zero pluginStatus_t-style code came back through the C ABI; `.status` holds it."""

    def __init__(self, message, status):
        super().__init__(message)
        self.status = status


def check(status, what):
    if status != 0:
        msg = load_library().bevops_status_string(status).decode()
        raise BevopsError(f"{what}: {msg} (status {status})", status)


def torch_dtype_code(t):
    import torch
    code = {torch.float32: F32, torch.float16: F16, torch.int8: I8}.get(t.dtype)
    if code is None:
        raise TypeError(f"unsupported dtype {t.dtype}; expected float32, float16 or int8")
    return code


def current_stream_ptr(device):
    import torch
    return torch.cuda.current_stream(device).cuda_stream
