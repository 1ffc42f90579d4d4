"""Operator API: same names and positional signatures as the reference's
det2trt/models/functions/__init__.py:1-35, registered in TRT_FUNCTIONS."""
from .multi_scale_deformable_attn import (
    multi_scale_deformable_attn,
    multi_scale_deformable_attn2,
    multi_scale_deformable_attn_int8,
)
from ..utils.register import TRT_FUNCTIONS

TRT_FUNCTIONS.register_module(module=multi_scale_deformable_attn)
TRT_FUNCTIONS.register_module(module=multi_scale_deformable_attn2)
TRT_FUNCTIONS.register_module(module=multi_scale_deformable_attn_int8)

__all__ = [
    "multi_scale_deformable_attn",
    "multi_scale_deformable_attn2",
    "multi_scale_deformable_attn_int8",
]
