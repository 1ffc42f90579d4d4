"""Operator API: same names and positional signatures as the reference's
det2trt/models/functions/__init__.py:1-35, registered in TRT_FUNCTIONS."""
from .multi_scale_deformable_attn import (
    multi_scale_deformable_attn,
    multi_scale_deformable_attn2,
    multi_scale_deformable_attn_int8,
    multi_scale_deformable_attn_local,
    msda_pack_value,
    multi_scale_deformable_attn_prepacked,
)
from .rotate import rotate, rotate2, rotate_int8, rotate_hwc
from .grid_sampler import grid_sampler, grid_sampler2, grid_sampler_int8
from .bev_pool_v2 import bev_pool_v2, bev_pool_v2_2, bev_pool_v2_int8
from .modulated_deformable_conv2d import (modulated_deformable_conv2d, modulated_deformable_conv2d2,
                                          modulated_deformable_conv2d_int8, modulated_deformable_conv2d_nhwc,
                                          bias_act_nhwc_, bias_relu_maxpool_nhwc, conv_offset_nhwc, upsample_add_nhwc_,
                                          feat_embed_nhwc)
from .spatial_cross_attention import (spatial_cross_attention_sample, spatial_cross_attention_projected,
                                      spatial_cross_attention_plan)
from .linear import (linear_bias_act, layer_norm, quantize_rows, dequantize_rows, linear_int8, tsgemm, tsgemm_ln, tile_gemm, small_gemm,
                     dense_auto, tsa_split, queue_mean2)
from .conv import conv_nhwc, conv3x3_nhwc, conv3x3_auto, conv3x3_c64, conv_int8_nhwc, stem_conv_pool
from .image import image_normalize_pad, padded_size
from .point_sampling import point_sampling
from .attention import self_attention_qkv
from .refine import refine_reference_points, decode_boxes
from ..utils.register import TRT_FUNCTIONS

TRT_FUNCTIONS.register_module(module=multi_scale_deformable_attn)
TRT_FUNCTIONS.register_module(module=multi_scale_deformable_attn2)
TRT_FUNCTIONS.register_module(module=multi_scale_deformable_attn_int8)
for _f in (rotate, rotate2, rotate_int8, grid_sampler, grid_sampler2, grid_sampler_int8,
           bev_pool_v2, bev_pool_v2_2, bev_pool_v2_int8, modulated_deformable_conv2d,
           modulated_deformable_conv2d2, modulated_deformable_conv2d_int8,
           spatial_cross_attention_sample):
    TRT_FUNCTIONS.register_module(module=_f)

__all__ = [
    "multi_scale_deformable_attn",
    "multi_scale_deformable_attn2",
    "multi_scale_deformable_attn_int8",
    "rotate", "rotate2", "rotate_int8",
    "grid_sampler", "grid_sampler2", "grid_sampler_int8",
    "bev_pool_v2", "bev_pool_v2_2", "bev_pool_v2_int8",
    "modulated_deformable_conv2d", "modulated_deformable_conv2d2", "modulated_deformable_conv2d_int8",
    "spatial_cross_attention_sample", "spatial_cross_attention_projected", "spatial_cross_attention_plan", "modulated_deformable_conv2d_nhwc", "bias_act_nhwc_", "linear_bias_act", "layer_norm", "rotate_hwc", "conv_offset_nhwc", "upsample_add_nhwc_", "feat_embed_nhwc",
    "msda_pack_value", "multi_scale_deformable_attn_prepacked", "multi_scale_deformable_attn_local", "image_normalize_pad", "padded_size", "quantize_rows", "dequantize_rows", "linear_int8", "tsgemm", "tsgemm_ln", "tile_gemm", "small_gemm", "dense_auto", "tsa_split", "queue_mean2", "conv_nhwc", "conv3x3_nhwc", "conv3x3_auto", "conv3x3_c64", "conv_int8_nhwc", "bias_relu_maxpool_nhwc", "stem_conv_pool", "point_sampling", "self_attention_qkv", "refine_reference_points", "decode_boxes",
]
