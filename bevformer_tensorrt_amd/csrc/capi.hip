// Library identification + name -> entry-point table (stands in for TensorRT's
// plugin-creator registry: REGISTER_TENSORRT_PLUGIN,
// TensorRT/plugin/multi_scale_deformable_attn/multiScaleDeformableAttnPlugin.cpp:345-346).
#include <string.h>

#include "common.h"

extern "C" const char *bevops_version(void) { return "bevops-hip 0.1 (gfx950)"; }

extern "C" const char *bevops_status_string(int status) {
  switch (status) {
    case BEVOPS_SUCCESS: return "success";
    case BEVOPS_FAILURE: return "failure (kernel launch error)";
    case BEVOPS_BAD_PARAM: return "bad parameter";
    case BEVOPS_NOT_SUPPORTED: return "dtype/shape combination not supported";
    case BEVOPS_NOT_INITIALIZED: return "not initialized";
    default: return "unknown status";
  }
}

namespace {
struct Entry {
  const char *name;
  void *fn;
};
const Entry kTable[] = {
    {"bevops_msda_forward", (void *)&bevops_msda_forward},
    {"bevops_msda_forward_ws", (void *)&bevops_msda_forward_ws},
    {"bevops_msda_workspace_size", (void *)&bevops_msda_workspace_size},
    {"MultiScaleDeformableAttnTRT", (void *)&bevops_msda_forward},
    {"MultiScaleDeformableAttnTRT2", (void *)&bevops_msda_forward},
    {"bevops_rotate_forward", (void *)&bevops_rotate_forward},
    {"RotateTRT", (void *)&bevops_rotate_forward},
    {"RotateTRT2", (void *)&bevops_rotate_forward},
    {"bevops_grid_sampler_2d_forward", (void *)&bevops_grid_sampler_2d_forward},
    {"GridSampler2DTRT", (void *)&bevops_grid_sampler_2d_forward},
    {"GridSampler2DTRT2", (void *)&bevops_grid_sampler_2d_forward},
    {"bevops_grid_sampler_3d_forward", (void *)&bevops_grid_sampler_3d_forward},
    {"GridSampler3DTRT", (void *)&bevops_grid_sampler_3d_forward},
    {"GridSampler3DTRT2", (void *)&bevops_grid_sampler_3d_forward},
    {"bevops_bev_pool_v2_forward", (void *)&bevops_bev_pool_v2_forward},
    {"BEVPoolV2TRT", (void *)&bevops_bev_pool_v2_forward},
    {"BEVPoolV2TRT2", (void *)&bevops_bev_pool_v2_forward},
    {"bevops_mdconv_forward", (void *)&bevops_mdconv_forward},
    {"bevops_mdconv_workspace_size", (void *)&bevops_mdconv_workspace_size},
    {"bevops_mdconv_forward_int8", (void *)&bevops_mdconv_forward_int8},
    {"bevops_mdconv_forward_int8_packed", (void *)&bevops_mdconv_forward_int8_packed},
    {"ModulatedDeformableConv2dTRT", (void *)&bevops_mdconv_forward},
    {"ModulatedDeformableConv2dTRT2", (void *)&bevops_mdconv_forward},
    // entries that are not reference plugins: workspace-lending / channels-last / fused forms
    {"bevops_grid_sampler_2d_forward_ws", (void *)&bevops_grid_sampler_2d_forward_ws},
    {"bevops_grid_sampler_2d_workspace_size", (void *)&bevops_grid_sampler_2d_workspace_size},
    {"bevops_rotate_forward_hwc", (void *)&bevops_rotate_forward_hwc},
    {"bevops_sca_forward", (void *)&bevops_sca_forward},
    {"bevops_mdconv_forward_nhwc", (void *)&bevops_mdconv_forward_nhwc},
    {"bevops_conv3x3_c32_forward_nhwc", (void *)&bevops_conv3x3_c32_forward_nhwc},
    {"bevops_bias_act_nhwc", (void *)&bevops_bias_act_nhwc},
    {"bevops_bias_relu_maxpool_nhwc", (void *)&bevops_bias_relu_maxpool_nhwc},
    {"bevops_stem_packed_size", (void *)&bevops_stem_packed_size},
    {"bevops_stem_pack", (void *)&bevops_stem_pack},
    {"bevops_stem_conv_pool", (void *)&bevops_stem_conv_pool},
    {"bevops_stem_set_variant", (void *)&bevops_stem_set_variant},
    {"bevops_upsample_add_nhwc", (void *)&bevops_upsample_add_nhwc},
    {"bevops_tsa_split", (void *)&bevops_tsa_split},
    {"bevops_queue_mean2", (void *)&bevops_queue_mean2},
    {"bevops_tsgemm_f16", (void *)&bevops_tsgemm_f16},
    {"bevops_tsgemm_s8", (void *)&bevops_tsgemm_s8},
    {"bevops_tsgemm_f16_ln", (void *)&bevops_tsgemm_f16_ln},
    {"bevops_mha_selfattn_f16", (void *)&bevops_mha_selfattn_f16},
    {"bevops_mha_selfattn_max_queries", (void *)&bevops_mha_selfattn_max_queries},
    {"bevops_value_proj_packed_size", (void *)&bevops_value_proj_packed_size},
    {"bevops_value_proj_packed", (void *)&bevops_value_proj_packed},
    {"bevops_value_pack_planes", (void *)&bevops_value_pack_planes},
    {"bevops_sca_prepacked_workspace_size", (void *)&bevops_sca_prepacked_workspace_size},
    {"bevops_sca_forward_prepacked", (void *)&bevops_sca_forward_prepacked},
    {"bevops_sca_plan_size", (void *)&bevops_sca_plan_size},
    {"bevops_sca_plan_build", (void *)&bevops_sca_plan_build},
    {"bevops_sca_forward_planned", (void *)&bevops_sca_forward_planned},
    {"bevops_point_sampling", (void *)&bevops_point_sampling},
    {"bevops_feat_embed_nhwc", (void *)&bevops_feat_embed_nhwc},
    {"bevops_linear_bias_act", (void *)&bevops_linear_bias_act},
    {"bevops_linear_tune", (void *)&bevops_linear_tune},
    {"bevops_quantize_rows", (void *)&bevops_quantize_rows},
    {"bevops_dequantize_rows", (void *)&bevops_dequantize_rows},
    {"bevops_linear_int8", (void *)&bevops_linear_int8},
    {"bevops_linear_int8_fused", (void *)&bevops_linear_int8_fused},
    {"bevops_tile_gemm_f16", (void *)&bevops_tile_gemm_f16},
    {"bevops_small_gemm_f16", (void *)&bevops_small_gemm_f16},
    {"bevops_conv_tile_f16", (void *)&bevops_conv_tile_f16},
    {"bevops_conv3x3_c64_f16", (void *)&bevops_conv3x3_c64_f16},
    {"bevops_refine_reference_points", (void *)&bevops_refine_reference_points},
    {"bevops_decode_boxes", (void *)&bevops_decode_boxes},
    {"bevops_conv_tile_int8_fused", (void *)&bevops_conv_tile_int8_fused},
    {"bevops_linear_int8_chain", (void *)&bevops_linear_int8_chain},
    {"bevops_conv_tile_int8", (void *)&bevops_conv_tile_int8},
    {"bevops_bias_relu_maxpool_nhwc_int8", (void *)&bevops_bias_relu_maxpool_nhwc_int8},
    {"bevops_mdconv_int8_nhwc_workspace_size", (void *)&bevops_mdconv_int8_nhwc_workspace_size},
    {"bevops_mdconv_forward_int8_nhwc", (void *)&bevops_mdconv_forward_int8_nhwc},
    {"bevops_image_normalize_pad", (void *)&bevops_image_normalize_pad},
    {"bevops_msda_packed_size", (void *)&bevops_msda_packed_size},
    {"bevops_msda_pack_value", (void *)&bevops_msda_pack_value},
    {"bevops_msda_forward_prepacked", (void *)&bevops_msda_forward_prepacked},
    {"bevops_linear_workspace_size", (void *)&bevops_linear_workspace_size},
    {"bevops_msda_workspace_size_shapes", (void *)&bevops_msda_workspace_size_shapes},
    {"bevops_sca_workspace_size", (void *)&bevops_sca_workspace_size},
    {"bevops_mdconv_forward_packed", (void *)&bevops_mdconv_forward_packed},
    {"bevops_mdconv_pack_weight", (void *)&bevops_mdconv_pack_weight},
    {"bevops_mdconv_packed_weight_size", (void *)&bevops_mdconv_packed_weight_size},
    {"bevops_conv3x3_c32_pack_weight", (void *)&bevops_conv3x3_c32_pack_weight},
    {"bevops_conv3x3_c32_packed_weight_size", (void *)&bevops_conv3x3_c32_packed_weight_size},
    {"bevops_layer_norm", (void *)&bevops_layer_norm},
};
}  // namespace

extern "C" void *bevops_query(const char *name) {
  if (!name) return nullptr;
  for (const Entry &e : kTable)
    if (strcmp(e.name, name) == 0) return e.fn;
  return nullptr;
}
