/*
 * bevops.h -- C ABI of libbevops_hip.so: MI355X (gfx950) HIP kernels for the
 * BEVFormer / BEVDet sampling hot path.
 *
 * This is the drop-in boundary.  Every entry point replaces the device side of
 * one TensorRT plugin of the reference (DerryHub/BEVFormer_tensorrt), i.e. what
 * `IPluginV2DynamicExt::enqueue(inputDesc, outputDesc, inputs, outputs,
 * workspace, stream)` did there; the tensor descriptors are flattened into
 * plain ints/floats.  The reference interface each function replaces is cited
 * as file:line relative to the reference tree.
 *
 * Conventions (restating the reference's, TensorRT/common/helper.h:19-25 and
 * SURVEY.md section 8b):
 *   - return value: bevops_status_t (0 = success).  Never aborts, exits or
 *     throws across the ABI; launch errors are returned, not printed.
 *   - all pointers are DEVICE pointers unless the name ends in `_host`.
 *   - the caller owns every buffer (inputs, outputs, workspace); the library
 *     allocates no device memory.  Process state it DOES keep, all of it
 *     invisible in results: (1) per kernel instance, the dynamic-LDS limit it
 *     has already raised on a device; (2) bevops_linear_*: one hipBLASLt handle
 *     per device and the algorithm chosen per problem, under a mutex (as the
 *     reference keeps the cuBLAS handle TensorRT attaches,
 *     modulatedDeformableConv2dPlugin.cpp:286-290); (3) the `*_set_variant`
 *     A/B hooks (thread-local, default 0 = automatic choice; tests and probes
 *     only).  Every operator entry is re-entrant from several host threads.
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as
 *     void*; NULL = the default stream).  No operator synchronises the host or
 *     is illegal under stream capture; the ONE blocking entry is the optional
 *     tuning call bevops_linear_tune, which says so.
 *   - tensors are dense, row-major ("kLINEAR"), 16-byte aligned.
 *   - dtype enums: BEVOPS_F32 / BEVOPS_F16 / BEVOPS_I8.  INT8 tensors carry a
 *     per-tensor scale (real = int8 * scale), as TensorRT's
 *     PluginTensorDesc::scale did.
 */
#ifndef BEVOPS_H_
#define BEVOPS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  BEVOPS_SUCCESS = 0,         /* STATUS_SUCCESS          helper.h:20 */
  BEVOPS_FAILURE = 1,         /* STATUS_FAILURE          helper.h:21 */
  BEVOPS_BAD_PARAM = 2,       /* STATUS_BAD_PARAM        helper.h:22 */
  BEVOPS_NOT_SUPPORTED = 3,   /* STATUS_NOT_SUPPORTED    helper.h:23 */
  BEVOPS_NOT_INITIALIZED = 4  /* STATUS_NOT_INITIALIZED  helper.h:24 */
} bevops_status_t;

typedef enum { BEVOPS_F32 = 0, BEVOPS_F16 = 1, BEVOPS_I8 = 2, BEVOPS_U8 = 3 /* raw camera images only */ } bevops_dtype_t;

/* interpolation / padding enums: functions/grid_sampler.py:134-136,
 * gridSamplerKernel.h:9-12 */
typedef enum { BEVOPS_BILINEAR = 0, BEVOPS_NEAREST = 1, BEVOPS_BICUBIC = 2 } bevops_interp_t;
typedef enum { BEVOPS_PAD_ZEROS = 0, BEVOPS_PAD_BORDER = 1, BEVOPS_PAD_REFLECTION = 2 } bevops_pad_t;

/* Library identification. */
const char *bevops_version(void);
/* Human-readable text for a status code (static storage). */
const char *bevops_status_string(int status);
/* Address of an entry point by plugin/op name, or NULL.  Accepts the ABI symbol
 * ("bevops_msda_forward") and the reference's plugin type names
 * ("MultiScaleDeformableAttnTRT", "MultiScaleDeformableAttnTRT2", "RotateTRT",
 * ...; multiScaleDeformableAttnPlugin.cpp:17-18 etc.), standing in for
 * TensorRT's plugin-creator registry (REGISTER_TENSORRT_PLUGIN,
 * multiScaleDeformableAttnPlugin.cpp:345-346). */
void *bevops_query(const char *name);

/* ------------------------------------------------------------------------
 * Multi-scale deformable attention, fused softmax + sampling + weighted sum.
 * Replaces MultiScaleDeformableAttnPlugin::enqueue
 *   (TensorRT/plugin/multi_scale_deformable_attn/multiScaleDeformableAttnPlugin.cpp:71-140)
 * and the launchers ms_deformable_im2col_cuda{,_h2,_int8}
 *   (multiScaleDeformableAttnKernel.h:12-38, multiScaleDeformableAttnKernel.cu:1106-1218).
 *
 *   value            [bs, nk, heads, channels]       dtype
 *   spatial_shapes   [num_levels, 2] int32 (h, w)    device
 *   spatial_shapes_host  same values on the host, or NULL (enables the
 *                    host-side shape check nk == sum h*w; never required)
 *   reference_points [bs, num_query, 1, 2*points_per_group]  ref_dtype (F32|F16);
 *                    must equal dtype for F32/F16 values, either for I8
 *                    (supportsFormatCombination, ...Plugin.cpp:148-189)
 *   sampling_offsets [bs, num_query, heads, num_levels*num_point*2]  dtype
 *   attention_weights[bs, num_query, heads, num_levels*num_point]    dtype, PRE-softmax
 *   output           [bs, num_query, heads, channels] dtype
 *   scale_*          INT8 per-tensor scales (ignored otherwise)
 * INT8 requires channels % 4 == 0 and num_point % 4 == 0 (...Plugin.cpp:151-156),
 * else BEVOPS_NOT_SUPPORTED.  With I8 values, ref_dtype F32 selects the signed
 * x127 weight flavour (kernel.cu:848-955), F16 the unsigned x255 flavour
 * (kernel.cu:957-1104).
 * ------------------------------------------------------------------------ */
int bevops_msda_forward(int dtype, const void *value, const int32_t *spatial_shapes,
                        const int32_t *spatial_shapes_host, const void *reference_points,
                        int ref_dtype, const void *sampling_offsets,
                        const void *attention_weights, void *output, int bs, int nk,
                        int heads, int channels, int num_levels, int num_query,
                        int num_point, int points_per_group, float scale_value,
                        float scale_offset, float scale_weight, float scale_out,
                        void *stream);

/* Same op with a caller-owned scratch buffer (TensorRT's `workspace` argument of enqueue):
 * lets the library re-lay `value` out head-major ([bs][heads][keys][32]) so that the two
 * x-corners of a sample share one cache line and each XCD's L2 holds the (camera, head)
 * plane it gathers from.  bevops_msda_workspace_size returns the bytes required (0 = the
 * path does not apply: non-fp16, channels != 32, ...); a NULL / too small workspace simply
 * selects the layout-preserving kernels, results are identical within fp32 rounding.
 * shared_offsets = 1: sampling_offsets / attention_weights are [1, num_query, heads, .] and
 * apply to every one of the bs value batches (BEVFormer's SCA repeats the same query for
 * all cameras, det2trt/models/modules/spatial_cross_attention.py:254) -- identical result
 * to passing them repeated bs times, without the 6x redundant HBM reads. */
size_t bevops_msda_workspace_size(int dtype, int bs, int nk, int heads, int channels,
                                  int num_levels, int num_query, int num_point);
/* Same, for a caller that knows the level shapes on the host ([num_levels][2] = (H, W)):
 * also covers the zero-padded re-layout with LDS-resident small levels (msda_hm3.hip), whose
 * size depends on the level shapes.  >= bevops_msda_workspace_size. */
size_t bevops_msda_workspace_size_shapes(int dtype, const int32_t *spatial_shapes_host, int bs,
                                         int nk, int heads, int channels, int num_levels,
                                         int num_query, int num_point);
int bevops_msda_forward_ws(int dtype, const void *value, const int32_t *spatial_shapes,
                           const int32_t *spatial_shapes_host, const void *reference_points,
                           int ref_dtype, const void *sampling_offsets,
                           const void *attention_weights, void *output, int bs, int nk,
                           int heads, int channels, int num_levels, int num_query,
                           int num_point, int points_per_group, float scale_value,
                           float scale_offset, float scale_weight, float scale_out,
                           int shared_offsets, void *workspace, size_t workspace_bytes,
                           void *stream);

/* Tuning hook: selects an internal MSDA kernel variant for subsequent calls from
 * this thread (0 = automatic).  Results are identical across variants; exists so
 * tests / tuning scripts can A/B a default against its partner in one process.  The complete list (decoded in
 * csrc/msda.hip: bevops_msda_set_variant, bevops_msda_forward_ws):
 *    1, 2     other point-splits of the layout-preserving quad kernel;   99  the one-thread-per-output generic kernel
 *    10       never a head-major kernel (what multi_scale_deformable_attn_local uses)
 *    11 / 15  head-major generations hm / hm2 forced;  16  hm3;  17  hm4 wherever it is instantiated
 *    19       int8 hm4 on the one-block-per-CU plan (partner of the default two-blocks plan)
 *    1000 / 1001  the fp16 SCA sampler hm5 with / without its visibility pre-pass
 *    3001 .. 3008 slices per CU of the planned fused SCA sampling (default 2; sticky until set again)
 *    3010 / 3011  camera reduce of the fused SCA op with its camera loop unrolled (default) / rolled (sticky)
 *    3012 / 3013  planned fused SCA sampling stores the pairs only one camera sees straight into the output and the
 *                 reduce skips those rows (default) / every pair through the per-camera scratch (sticky)
 *    3014 / 3015  planned fused SCA sampling with the DPP broadcasts of the sample records folded into the instructions
 *                 that consume them and the LDS row taps fused (default, round 6) / the round-5 build (sticky); same bits
 * (The measured-and-rejected builds of rounds 1-4 -- LDS-staged hm, two-copy hm, hm4 chunk sizes / schedule ablations,
 * int8 pixel-pair entries, hm5 with 768 threads / mailbox / persistent blocks / level-class split -- are no longer in
 * the library; their measurements are under profiles/.)  A packed value (bevops_msda_pack_value) must be sampled under
 * the variant it was packed under.  Returns the previously REQUESTED kernel-family value; the 30xx knobs are independent
 * of the family selection: they are neither recorded as, nor returned as, the requested value, so
 * `prev = set_variant(10); ...; set_variant(prev)` restores the family whatever 30xx calls came before or in between. */
int bevops_msda_set_variant(int variant);


/* ------------------------------------------------------------------------
 * rotate: img [channels, height, width] rotated by *angle degrees (counter-clockwise)
 * about *center (x, y in pixels); zeros padding, align_corners = false.
 * Replaces RotatePlugin::enqueue (TensorRT/plugin/rotate/rotatePlugin.cpp:75-116) and
 * rotate<T> / rotate_h2 / rotate_int8 (rotateKernel.h:14-26, rotateKernel.cu:708-748).
 *   angle  : 1 element, center : 2 elements, DEVICE memory, angle_dtype F32|F16
 *            (F32 images need F32 angle/center, rotatePlugin.cpp:125-156)
 *   interpolation : BEVOPS_BILINEAR | BEVOPS_NEAREST
 *   I8 : dense [C,H,W] int8 (the reference used kCHW4), real = int8 * scale_in;
 *        output requantised with scale_out.
 * ------------------------------------------------------------------------ */
int bevops_rotate_forward(int dtype, const void *img, const void *angle, const void *center,
                          int angle_dtype, void *output, int channels, int height, int width,
                          int interpolation, float scale_in, float scale_out, void *stream);

/* ------------------------------------------------------------------------
 * grid_sampler, 2-D: input [N,C,H_in,W_in], grid [N,2,H_out,W_out] channel-first
 * (x, y) in [-10, 10] units, output [N,C,H_out,W_out]; all of dtype.
 * Replaces GridSamplerPlugin::enqueue (TensorRT/plugin/grid_sampler/gridSamplerPlugin.cpp:110-156),
 * grid_sample<T> and grid_sample_int8 (gridSamplerKernel.h:14-26, gridSamplerKernel.cu:1933-2043).
 *   interpolation : bilinear | nearest | bicubic;  padding : zeros | border | reflection
 *   I8 : all three interpolation modes; grid is int8 with scale_grid.
 * 3-D: input [N,C,D,H,W], grid [N,3,D_out,H_out,W_out] (x, y, z); F32 / F16;
 *      bilinear (trilinear) | nearest.
 * ------------------------------------------------------------------------ */
int bevops_grid_sampler_2d_forward(int dtype, const void *input, const void *grid, void *output,
                                   int N, int C, int H_in, int W_in, int H_out, int W_out,
                                   int interpolation, int padding, int align_corners,
                                   float scale_in, float scale_grid, float scale_out,
                                   void *stream);
/* bevops_grid_sampler_2d_forward with a caller-lent scratch buffer (mirrors getWorkspaceSize): fp32 /
 * fp16 bilinear / nearest calls with C % 8 == 0 whose output is >= 2x the input stage the input
 * channels-last in the workspace (one 16-byte load per tap and 8-channel chunk instead of 8 two-byte
 * gathers); every other call runs the planar kernel.  Results are bit-identical either way.
 * bevops_grid_sampler_2d_workspace_size: bytes to lend (0 = the staged path does not apply). */
size_t bevops_grid_sampler_2d_workspace_size(int dtype, int N, int C, int H_in, int W_in);
int bevops_grid_sampler_2d_forward_ws(int dtype, const void *input, const void *grid, void *output,
                                      int N, int C, int H_in, int W_in, int H_out, int W_out,
                                      int interpolation, int padding, int align_corners,
                                      float scale_in, float scale_grid, float scale_out,
                                      void *workspace, size_t workspace_bytes, void *stream);
int bevops_grid_sampler_3d_forward(int dtype, const void *input, const void *grid, void *output,
                                   int N, int C, int D_in, int H_in, int W_in, int D_out,
                                   int H_out, int W_out, int interpolation, int padding,
                                   int align_corners, void *stream);

/* ------------------------------------------------------------------------
 * bev_pool_v2 (BEVDet pillar pooling):
 *   out[ranks_bev[s_k], :] = sum_{i < len_k} depth.flat[ranks_depth[s_k+i]] * feat.flat[ranks_feat[s_k+i], :]
 * all other output cells are zero.  Replaces BEVPoolPlugin::enqueue
 * (TensorRT/plugin/bev_pool_v2/bevPoolPlugin.cpp:68-109) and bev_pool_v2 / _h2 / _int8
 * (bevPoolKernel.h:13-31, bevPoolKernel.cu:151-190).
 *   depth  [N,D,H,W] dtype, feat [N,H,W,channels] dtype, ranks_* [n_points] int32,
 *   interval_starts / interval_lengths [n_intervals] int32,
 *   output [1, out_height, out_width, channels] dtype (fully written: cleared on `stream`).
 *   I8: out = T2int8(acc * scale_depth * scale_feat / scale_out), int32 accumulation.
 * ------------------------------------------------------------------------ */
int bevops_bev_pool_v2_forward(int dtype, const void *depth, const void *feat,
                               const int32_t *ranks_depth, const int32_t *ranks_feat,
                               const int32_t *ranks_bev, const int32_t *interval_starts,
                               const int32_t *interval_lengths, void *output, int channels,
                               int n_intervals, int out_height, int out_width,
                               float scale_depth, float scale_feat, float scale_out,
                               void *stream);

/* ---------------------------------------------------------------------------
 * Fused spatial cross-attention sampling.  NOT a reference plugin: it replaces the
 * sequence of det2trt/models/modules/spatial_cross_attention.py:254-270 --
 *   query.repeat(num_cams) -> MultiScaleDeformableAttnTRT on [num_cams, nq, ...] ->
 *   slots = (queries * bev_mask).sum(0)
 * -- with one call that (a) takes the camera-independent sampling_offsets / attention_weights
 * once ([1, nq, heads, L*P*2] / [1, nq, heads, L*P]), (b) skips every (camera, query) pair whose
 * bev_mask weight is 0 (the rebatching of the original PyTorch SCA,
 * third_party/bev_mmdet3d/models/modules/spatial_cross_attention.py:143-191) and (c) returns the
 * masked camera sum  output[q, heads, C] = sum_cam bev_mask[cam, q] * sample(cam, q).
 *   value [num_cams, nk, heads, C] fp16, reference_points_cam [num_cams, nq, 1, 2*ppg] fp16,
 *   bev_mask [num_cams, nq] fp16 (0 = not visible, else the weight), output [nq, heads, C] fp16.
 * F16, C == 32 only (NOT_SUPPORTED otherwise: compose the reference sequence instead).
 * ------------------------------------------------------------------------ */
size_t bevops_sca_workspace_size(int dtype, const int32_t *spatial_shapes_host, int num_cams, int nk,
                                 int heads, int channels, int num_levels, int num_query,
                                 int num_point);
int bevops_sca_forward(int dtype, const void *value, const int32_t *spatial_shapes_host,
                       const void *reference_points_cam, const void *sampling_offsets,
                       const void *attention_weights, const void *bev_mask, void *output,
                       int num_cams, int nk, int heads, int channels, int num_levels, int num_query,
                       int num_point, int points_per_group, void *workspace, size_t workspace_bytes,
                       void *stream);

/* ------------------------------------------------------------------------
 * Modulated deformable convolution (DCNv2) forward.
 * Replaces ModulatedDeformableConv2dPlugin::enqueue / getWorkspaceSize
 * (TensorRT/plugin/modulated_deformable_conv2d/modulatedDeformableConv2dPlugin.cpp:73-197)
 * and ModulatedDeformConvForwardCUDAKernel<T> (modulatedDeformableConv2dKernel.h:11-30,
 * modulatedDeformableConv2dKernel.cu:695-978).
 *   input  [B,Cin,H,W], offset [B, deform_groups*2*Kh*Kw, Ho, Wo] (per tap: h then w),
 *   mask   [B, deform_groups*Kh*Kw, Ho, Wo], weight [Cout, Cin/groups, Kh, Kw],
 *   bias   [Cout] or NULL, output [B,Cout,Ho,Wo];  Ho/Wo from the usual conv formula.
 *   workspace: caller-owned scratch of at least bevops_mdconv_workspace_size(...) bytes
 *   (mirrors getWorkspaceSize; 0 = unsupported arguments), 16-byte aligned.
 * F32 and F16 (F16 needs (Cin/groups*Kh*Kw) % 8 == 0) through bevops_mdconv_forward; INT8
 * through bevops_mdconv_forward_int8 (int8 input / offset / mask / weight with per-tensor
 * scales, fp32 bias, int8 output: out = T2int8((acc*scale_in*scale_weight + bias)/scale_out),
 * modulatedDeformableConv2dKernel.cu:463-607,897-978; needs Cin % 4 == 0,
 * (Cout/groups) % 4 == 0 like ...Plugin.cpp:217-219).
 * bevops_mdconv_workspace_size(BEVOPS_I8, ...) sizes its workspace.
 * ------------------------------------------------------------------------ */
/* Tuning hook like bevops_msda_set_variant: 0 = automatic (fused implicit GEMM when the
 * channel counts allow), 1 = force the im2col + GEMM pipeline, 2 / 3 = the register-staged
 * fused kernels (256 / 512 threads), 4 = LDS-DMA kernel with 64-pixel tiles and no split-K
 * tail, 5 = LDS-DMA kernel with 128-pixel tiles, 7 / 13 = LDS-DMA kernel with the round-2 wave rotation / with one
 * segment order for all waves (rounds 2-5; the default since round 6 lets the lower half of a block's waves issue all
 * the weight DMA and the upper half run its matrix segment first: same bits); INT8: 6 = im2col + GEMM pair, 8 = register-staged
 * fused kernel, 9 = LDS-DMA kernel whatever the tile count, 20 + mask = timing experiments (wrong
 * results).  Returns the previous value. */
int bevops_mdconv_set_variant(int variant);
size_t bevops_mdconv_workspace_size(int dtype, int B, int Cin, int H, int W, int Cout, int Kh,
                                    int Kw, int stride_h, int stride_w, int pad_h, int pad_w,
                                    int dil_h, int dil_w, int groups, int deform_groups);
int bevops_mdconv_forward_int8(const void *input, float scale_in, const void *offset,
                               float scale_offset, const void *mask, float scale_mask,
                               const void *weight, float scale_weight, const float *bias,
                               void *output, float scale_out, void *workspace,
                               size_t workspace_bytes, int B, int Cin, int H, int W, int Cout, int Kh,
                               int Kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h,
                               int dil_w, int groups, int deform_groups, void *stream);
/* the same with the weights already re-laid-out by bevops_mdconv_pack_weight(BEVOPS_I8, ...) */
int bevops_mdconv_forward_int8_packed(const void *input, float scale_in, const void *offset,
                                      float scale_offset, const void *mask, float scale_mask,
                                      const void *packed_weight, float scale_weight, const float *bias,
                                      void *output, float scale_out, void *workspace,
                                      size_t workspace_bytes, int B, int Cin, int H, int W, int Cout, int Kh,
                                      int Kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h,
                                      int dil_w, int groups, int deform_groups, void *stream);
int bevops_mdconv_forward(int dtype, const void *input, const void *offset, const void *mask,
                          const void *weight, const void *bias, void *output, void *workspace,
                          size_t workspace_bytes, int B, int Cin, int H, int W, int Cout, int Kh,
                          int Kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h,
                          int dil_w, int groups, int deform_groups, void *stream);
/* Inference keeps the same weights call after call: the [Cout][tap][Cin/groups] re-layout that
 * bevops_mdconv_forward makes inside its workspace on every call can be made once
 * (the reference builds its weight tensors once in the plugin constructor,
 * modulatedDeformableConv2dPlugin.cpp:33-71).  F32 / F16 / I8 (I8: rows zero-padded to 16 bytes, for
 * bevops_mdconv_forward_int8_packed); `packed` holds bevops_mdconv_packed_weight_size bytes, 16-byte aligned. */
size_t bevops_mdconv_packed_weight_size(int dtype, int Cout, int Cin_per_group, int Kh, int Kw);
int bevops_mdconv_pack_weight(int dtype, const void *weight, void *packed, int Cout,
                              int Cin_per_group, int Kh, int Kw, void *stream);
/* Channels-last variant for a caller whose activations are [B, H, W, C] (the re-hosted backbone):
 * input and output NHWC, packed weights, optional fused ReLU; fp16 fused-kernel domain only
 * (NOT_SUPPORTED otherwise).  offset_mask_channels == 0: offset / mask keep the reference's
 * planar [B, ., Ho, Wo] layout.  offset_mask_channels == OC > 0: `offset` is the raw channels-last
 * output [B, Ho, Wo, OC] of the pack's offset convolution (cnn/dcn.py:62-70: 2*KK offset channels
 * then KK mask logits per deform group, OC >= deform_groups*3*KK, even), `mask` is ignored and
 * the sigmoid is applied in the kernel. */
int bevops_mdconv_forward_nhwc(int dtype, const void *input_nhwc, const void *offset,
                               const void *mask, const void *packed_weight, const void *bias,
                               void *output_nhwc, int relu, int offset_mask_channels,
                               void *workspace, size_t workspace_bytes,
                               int B, int Cin, int H, int W, int Cout, int Kh, int Kw, int stride_h,
                               int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int groups,
                               int deform_groups, void *stream);
/* x[rows, channels] += bias[channels] (+ residual[rows, channels]), optional ReLU, in place: the
 * folded-BN convolution epilogue of the re-hosted backbone as one pass (fp16, channels % 8 == 0). */
int bevops_bias_act_nhwc(int dtype, void *x, const void *bias, const void *residual, size_t rows,
                         int channels, int relu, void *stream);
/* Stem epilogue of the channels-last backbone in one pass: out [N, Ho, Wo, C] = max_pool2d(relu(x + bias), 3,
 * stride 2, pad 1) over the convolution's raw output x [N, H, W, C] (fp16, C % 8 == 0; Ho = (H - 1) / 2 + 1).
 * Bit-equal to bevops_bias_act_nhwc followed by the framework's max_pool2d (rounding is monotonic). */
int bevops_bias_relu_maxpool_nhwc(int dtype, const void *x, const void *bias, void *out, int n, int h, int w,
                                  int channels, void *stream);
/* The whole ResNet stem of the re-hosted backbone as ONE kernel (round 5; not a reference plugin): conv 7x7 / stride 2 /
 * pad 3 from 3 to 64 channels (folded-BN shift `bias`) -> ReLU -> max_pool2d 3 / stride 2 / pad 1
 * (third_party/bev_mmdet3d/models/backbones/resnet.py:619-624: conv1, norm1, relu, maxpool), from the PLANAR camera
 * images x [n, 3, h, w] (fp16, w even) to the pooled channels-last activation out [n, hp, wp, 64]
 * (hc = (h - 1) / 2 + 1, hp = (hc - 1) / 2 + 1, likewise w), fp16 or -- out_dtype BEVOPS_I8, scale_out -- int8 with
 * q = min(rne(v / scale_out), 127): the first tensor of the INT8 engine's activation chain.  fp32 accumulation, one
 * rounding at the end (the two-pass form rounds the convolution's output to fp16 first: results agree to that
 * rounding).  bevops_stem_pack turns weight [64, 3, 7, 7] + bias [64] (fp16; bias may be NULL) into the kernel's
 * matrix-core operand image (bevops_stem_packed_size bytes, 16-byte aligned, device memory): once per model.
 * bevops_stem_set_variant (thread-local, A/B switch of the tests): 0 = pooling neighbours through wave-wide DPP
 * shifts, 1 = through ds_bpermute; identical results.  Returns the previous value. */
size_t bevops_stem_packed_size(void);
int bevops_stem_pack(int dtype, const void *weight, const void *bias, void *packed, void *stream);
int bevops_stem_conv_pool(int dtype, int out_dtype, const void *x, const void *packed, void *out, int n, int h, int w,
                          float scale_out, void *stream);
int bevops_stem_set_variant(int variant);

/* FPN top-down step of the re-hosted neck on channels-last fp16 activations (not a reference plugin):
 * a[n, y, x, :] += b[n, sy(y), sx(x), :], source indices as aten's nearest up-sampling -- the reference's
 * `laterals[i-1] += F.interpolate(laterals[i], size=..., mode="nearest")`
 * (third_party/bev_mmdet3d/models/necks/fpn.py:170-176) in one pass, bit-equal to it.  In place on `a`. */
int bevops_upsample_add_nhwc(int dtype, void *a, const void *b, int n, int h, int w, int hb, int wb,
                             int channels, void *stream);

/* Encoder input assembly (det2trt/models/modules/transformer.py:138-152): dst[n, r, :] = (src[n, r, :] +
 * cam_embed[n, :]) + level_embed[:], both sums rounded to fp16 like the two tensor adds they replace; `dst` is
 * the level's first row inside the concatenated [cams, sum hw, C] feature tensor, `dst_batch_stride` its
 * element stride between cameras (so no torch.cat copy is needed).  src: [n, rows, C] dense. */
int bevops_feat_embed_nhwc(int dtype, const void *src, const void *cam_embed, const void *level_embed, void *dst,
                           int n, size_t rows, int channels, size_t dst_batch_stride, void *stream);
/* The DCNv2 pack's offset convolution (cnn/dcn.py:62-70: 3x3, stride 1, pad 1, Cout <= 32) on
 * channels-last fp16 activations with its bias in the epilogue:
 *   output_nhwc[B, H, W, 32] = conv3x3(input_nhwc[B, H, W, Cin], weight[Cout, Cin, 3, 3]) + bias32
 * (channels >= Cout are the bias32 values, normally 0).  Cin in {64, 128, 256} or a multiple of
 * 256.  The weight is packed once (bevops_conv3x3_c32_pack_weight into a caller buffer of
 * bevops_conv3x3_c32_packed_weight_size bytes; 0 = unsupported Cin); bias32 has 32 entries or is
 * NULL.  The result is the `offset` operand of bevops_mdconv_forward_nhwc with
 * offset_mask_channels = 32. */
size_t bevops_conv3x3_c32_packed_weight_size(int dtype, int Cin);
int bevops_conv3x3_c32_pack_weight(int dtype, const void *weight, void *packed, int Cout, int Cin, void *stream);
/* 0 = default: at Cin == 256 the build with the weights in registers and 8 x 8-pixel image tiles in LDS (round 6),
 * the tile kernel with three waves per 32-pixel tile otherwise; 3 = that tile kernel everywhere (round 5's default;
 * bit-identical to 0); 2 = the tile kernel with one wave per tile (rounds 1-4); 1 = the variant that stages the
 * image rows in LDS (A/B, tests).  Returns the previous value. */
int bevops_conv3x3_c32_set_variant(int variant);
int bevops_conv3x3_c32_forward_nhwc(int dtype, const void *input_nhwc, const void *packed_weight,
                                    const void *bias32, void *output_nhwc, int B, int H, int W, int Cin,
                                    void *stream);
/* A/B switch of bevops_rotate_forward's store path (thread-local): 0 = 16-byte stores through an LDS transpose of
 * 8-pixel runs where every plane starts 16-byte aligned, 1 = one store per lane and plane (rounds 1-3).  Same values
 * either way.  Returns the previous setting. */
int bevops_rotate_set_variant(int variant);
/* Temporal self-attention glue (round 5; not reference plugins; modules/temporal_self_attention.py:350-457).
 * bevops_tsa_split: one row of the stacked sampling_offsets | attention_weights projection, laid out as the reference
 * views it -- [heads][bev_queue 2][points][xy] then [heads][bev_queue 2][points] -- split into the queue-major operands
 * of the MSDA call: offsets [2, num_query, heads, points * 2], weights [2, num_query, heads, points] (what
 * .view(...).permute(0, 3, 1, 2, 4, 5[, 6]).contiguous() produces twice, as one pass).  fp16, points == 4.
 * bevops_queue_mean2: out[i] = (x[i] + x[count + i]) / 2 in fp32 with one rounding (= torch.mean over the two queue
 * entries), fp16, count % 8 == 0. */
int bevops_tsa_split(int dtype, const void *both, void *offsets, void *weights, int num_query, int heads, int points,
                     void *stream);
int bevops_queue_mean2(int dtype, const void *x, void *out, size_t count, void *stream);
/* bevops_rotate_forward on channels-last data: img / output are [height, width, channels] (fp32 /
 * fp16, channels a multiple of 4 / 8) -- the layout prev_bev [H*W, 1, C] already has in the model,
 * so the permute-copy to [C, H, W] and back around the plugin (transformer.py:296-303) disappears.
 * Same arithmetic, element for element. */
int bevops_rotate_forward_hwc(int dtype, const void *img, const void *angle, const void *center,
                              int angle_dtype, void *output, int channels, int height, int width,
                              int interpolation, void *stream);
/* out[rows, channels] = (x - mean) * rsqrt(var + eps) * gamma + beta over the last dimension
 * (biased variance, fp32 statistics; torch.nn.LayerNorm semantics); fp16, channels in
 * {64, 128, 256, 512}; gamma / beta optional; out may alias x.  The norms between the attention
 * blocks of the re-hosted encoder / decoder (encoder.py:510-636) as one streaming pass. */
int bevops_layer_norm(int dtype, const void *x, const void *gamma, const void *beta, void *out,
                      size_t rows, int channels, float eps, void *stream);
/* INT8 dense layers: what TensorRT builds from the reference's `LinearQ` /
 * `Conv2dQ` (= pytorch_quantization QuantLinear / QuantConv2d, det2trt/models/utils/register.py:78-84):
 * per-tensor symmetric quantisation of the layer input, int8 x int8 -> int32 GEMM on the matrix cores,
 * de-quantising epilogue.  Not plugins.
 *   bevops_quantize_rows: q = clamp(rne(x / scale), -127, 127), x fp16, count % 8 == 0.
 *   bevops_linear_int8:   out[M, N] = act((a_q[M, K] . w_q[N, K]^T) * scale_a * w_scale[n] + bias[n]
 *                         + residual[M, N]); w_scales (device, fp32 [N]) per output channel, or NULL
 *                         for the per-tensor scale_w; bias fp32 (device) optional; residual fp16
 *                         optional; out fp16, or int8 requantised with scale_out.  K % 16 == 0.
 *   bevops_linear_int8_fused: the same layer taking the fp16 activation x[M, K] itself: it is quantised
 *                         inside the GEMM's operand load, q = clamp(rne(x * (1 / scale_a)), -127, 127) with
 *                         the product and the rounding in one fused multiply-add -- no quantise pass, no int8
 *                         copy of the activation.  (x * fl(1 / s) against fl(x / s): the two quantisers can
 *                         differ by one step only where x / s is within 1e-5 of a rounding tie.)
 *   bevops_dequantize_rows: out (fp16) = q (int8) * scale, product in fp32, one rounding; count % 8 == 0. */
int bevops_quantize_rows(int dtype, const void *x, void *q, size_t count, float scale, void *stream);
int bevops_dequantize_rows(int dtype, const void *q, void *out, size_t count, float scale, void *stream);
int bevops_linear_int8(const void *a_q, float scale_a, const void *w_q, const float *w_scales,
                       float scale_w, const float *bias, const void *residual, int out_dtype,
                       void *out, float scale_out, long long M, int N, int K, int relu, void *stream);
int bevops_linear_int8_fused(const void *x_f16, float scale_a, const void *w_q, const float *w_scales,
                             float scale_w, const float *bias, const void *residual, int out_dtype,
                             void *out, float scale_out, long long M, int N, int K, int relu, void *stream);
/* The same tiled GEMM skeleton (csrc/tile_gemm.hip) with fp16 operands: out[M, N] = act(x[M, K] . w[N, K]^T +
 * bias[n] + residual[M, N]), fp32 accumulation, one rounding; bias / residual fp16, optional.  K % 8 == 0.
 * One of the implementations the host layer's measured dispatch (functions/linear.py: dense_auto) picks from,
 * next to bevops_tsgemm_f16 and bevops_linear_bias_act. */
int bevops_tile_gemm_f16(const void *x, const void *weight, const void *bias, const void *residual,
                         void *out, long long M, int N, int K, int relu, void *stream);
/* Convolution on channels-last fp16 activations as an implicit GEMM on the same tiled skeleton (no column
 * buffer, no strided copy): kernel ksize x ksize in {1, 3}, pad ksize / 2, any stride.  x [B, H, W, Cin],
 * weight_taps [Cout][ksize][ksize][Cin] (= weight.permute(0, 2, 3, 1)), out [B, Hout, Wout, Cout] =
 * act(conv(x) + bias[n] + residual); bias / residual fp16, optional.  Cin % 32 == 0.  The plain 3x3
 * convolutions of the re-hosted backbone / neck (ResNet stages without DCN: resnet.py:106-260; FPN output
 * and extra convolutions: necks/fpn.py:140-155) and the stride-2 1x1 convolutions at the head of a stage,
 * with their shift and ReLU in the epilogue. */
int bevops_conv_tile_f16(const void *x, const void *weight_taps, const void *bias, const void *residual,
                         void *out, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int relu,
                         void *stream);
/* The 3x3 / stride 1 / pad 1 convolution of a 64-channel layer (conv2 of the ResNet stage-1 bottlenecks,
 * det2trt/models/backbones/resnet.py:106-260) with BOTH operands in LDS: the 64 x 576 weight matrix resident per
 * persistent block, 16 x 16-pixel output tiles whose 18 x 18 input pixels are staged once (a tap is an LDS address
 * offset).  Same operands and layouts as bevops_conv_tile_f16 (no identity rows); results bit-identical to it.
 * NOT_SUPPORTED unless Cin == Cout == 64. */
int bevops_conv3x3_c64_f16(const void *x, const void *weight_taps, const void *bias, void *out, int B, int H, int W,
                           int Cin, int Cout, int relu, void *stream);
/* Decoder reference-point refinement (det2trt/models/modules/decoder.py:24-40, 93-103) as one launch:
 *   new_reference_points[q] = sigmoid((tmp[q][0], tmp[q][1], tmp[q][4]) + inverse_sigmoid(reference_points[q]))
 * on fp16 tensors, every step rounded to binary16 as the framework's op sequence rounds it (the refined points are the
 * next layer's sampling locations: bit-exact, tests/test_refine_gpu.py).  log and sigmoid are binary16 -> binary16
 * functions of one argument: the caller hands them in as tables of 65 536 fp16 entries (entry i = f(value with bit
 * pattern i)), filled by whatever defines its reference -- the host layer fills them with the framework's own ops.
 * reference_xy [n, 2] (optional) receives the (x, y) columns contiguously.  tmp [n, tmp_stride], tmp_stride >= 5. */
int bevops_refine_reference_points(int dtype, const void *tmp, const void *reference_points, void *new_reference_points,
                                   void *reference_xy, int num_query, int tmp_stride, const void *log_table,
                                   const void *sigmoid_table, void *stream);
/* The head's box decoding on the stacked decoder levels (det2trt/models/dense_heads/bevformer_head.py:247-282 with
 * mmdet's inverse_sigmoid): x / y / z columns (0, 1, 4) = sigmoid(reg + inverse_sigmoid(ref)) * scale + offset, the
 * other seven columns copied; every step rounded to binary16 as the framework's op sequence rounds it, log / sigmoid
 * from the caller's tables (see bevops_refine_reference_points).  regs, out [count, 10]; refs [count, 3]. */
int bevops_decode_boxes(int dtype, const void *regs, const void *refs, void *out, int count, float scale_x, float offset_x,
                        float scale_y, float offset_y, float scale_z, float offset_z, const void *log_table,
                        const void *sigmoid_table, void *stream);
/* The same convolution as an INT8 layer (`Conv2dQ`, det2trt/models/utils/register.py:79): fp16 activation
 * quantised with scale_a inside the operand load (as bevops_linear_int8_fused), int8 weights in the taps-major
 * layout, int32 sums, de-quantising epilogue with fp32 bias / fp16 identity / ReLU, fp16 out.  Cin % 64 == 0. */
int bevops_conv_tile_int8_fused(const void *x_f16, float scale_a, const void *w_q_taps, const float *w_scales,
                                float scale_w, const float *bias, const void *residual, void *out, int B, int H,
                                int W, int Cin, int Cout, int ksize, int stride, int relu, void *stream);
/* The INT8 engine's int8 ACTIVATION CHAIN (not plugins -- what TensorRT builds from the
 * reference's `Conv2dQ` / `LinearQ` layers, det2trt/models/utils/register.py:78-84, selected by
 * configs/bevformer/plugin/bevformer_base_trt_p2_q.py, when it keeps the tensors BETWEEN the layers of a ResNet
 * bottleneck in int8): every producer requantises with its consumer's calibrated input scale, so a layer moves one
 * byte per element in and one out.
 *   bevops_linear_int8_chain: bevops_linear_int8 / _fused behind one entry, plus int8 identity rows:
 *                         a [M, K] BEVOPS_I8 (quantised with scale_a) or BEVOPS_F16 (quantised in the operand load);
 *                         residual [M, N] BEVOPS_F16, or BEVOPS_I8 with scale_res (int8 `a` only); out fp16, or int8
 *                         requantised with scale_out: q = clamp(rne(v * (1 / scale_out)), -127, 127).
 *   bevops_conv_tile_int8:   bevops_conv_tile_int8_fused on an activation that already IS int8 [B, H, W, Cin]
 *                         (ksize in {1, 3}, pad ksize / 2, any stride, Cin % 64 == 0), fp16 or int8 out [B, Ho, Wo,
 *                         Cout]; `residual` must be NULL.
 *   bevops_bias_relu_maxpool_nhwc_int8: bevops_bias_relu_maxpool_nhwc leaving as int8 with scale_out (the stem
 *                         epilogue: first tensor of the chain).
 *   bevops_mdconv_forward_int8_nhwc: the DCNv2 block of the chain.  Its arithmetic is the INT8 plugin's
 *                         (modulatedDeformableConv2dKernel.cu:463-607,897-978: u8 x255 area weights, dot4 blend,
 *                         T2int8(t / 255), T2int8(val * mask), s8 x s8 -> i32 GEMM, one requantisation) on
 *                         input int8 [B, H, W, Cin] and on the offsets / sigmoid(mask logits) of the raw fp16
 *                         [B, Ho, Wo, offset_mask_channels] output of the pack's offset convolution (cnn/dcn.py:
 *                         70-86), quantised with scale_offset / scale_mask while they are staged (the Q node
 *                         TensorRT places in front of the plugin); weights packed by bevops_mdconv_pack_weight(
 *                         BEVOPS_I8); output int8 [B, Ho, Wo, Cout] with the ReLU folded into the requantisation.
 *                         exact != 0: exactly that arithmetic (bit-identical to the plugin on those operands);
 *                         exact == 0 (the engine's default): the mask is folded into the four area weights before
 *                         they are quantised, a_q = u8(area_q mask 255), so a column element is requantised ONCE,
 *                         T2int8(sum a_q v_q / 255) -- the plugin's float mask multiply + second rounding, which
 *                         makes its kernel VALU-bound, disappears; within a step of the exact flavour.
 *                         `workspace` (optional, bevops_mdconv_int8_nhwc_workspace_size() bytes) holds the int32
 *                         partial sums of the split-K tail.  NOT_SUPPORTED outside (Cin / groups) % 128 == 0, one
 *                         deform group per conv group, 3 Kh Kw <= 32. */
int bevops_linear_int8_chain(const void *a, int a_dtype, float scale_a, const void *w_q, const float *w_scales,
                             float scale_w, const float *bias, const void *residual, int res_dtype, float scale_res,
                             int out_dtype, void *out, float scale_out, long long M, int N, int K, int relu,
                             void *stream);
int bevops_conv_tile_int8(const void *x_q, float scale_a, const void *w_q_taps, const float *w_scales, float scale_w,
                          const float *bias, const void *residual, int out_dtype, void *out, float scale_out, int B,
                          int H, int W, int Cin, int Cout, int ksize, int stride, int relu, void *stream);
int bevops_bias_relu_maxpool_nhwc_int8(int dtype, const void *x, const void *bias, void *out_q, float scale_out,
                                       int n, int h, int w, int channels, void *stream);
size_t bevops_mdconv_int8_nhwc_workspace_size(void);
int bevops_mdconv_forward_int8_nhwc(const void *input_nhwc, float scale_in, const void *offset_mask_nhwc,
                                    int offset_mask_channels, float scale_offset, float scale_mask,
                                    const void *packed_weight, float scale_weight, const float *bias,
                                    void *output_nhwc, float scale_out, int relu, int exact, void *workspace,
                                    size_t workspace_bytes, int B, int Cin, int H, int W, int Cout, int Kh, int Kw,
                                    int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                                    int groups, int deform_groups, void *stream);
/* Camera-image front end of the frame loop (not a plugin): the reference's test
 * pipeline NormalizeMultiviewImage + PadMultiViewImage(size_divisor=32) + DefaultFormatBundle3D
 * (configs/bevformer/bevformer_base.py:11,228-231; third_party/bev_mmdet3d/datasets/pipelines/
 * transform_3d.py:99-150; mmcv.imnormalize) in one pass.  `images` [N, H0, W0, 3] BEVOPS_U8 or
 * BEVOPS_F32 (BGR as loaded) -> `output` [N, 3, Hp, Wp] (channels_last: [N, Hp, Wp, 3]) BEVOPS_F16
 * or BEVOPS_F32 = ((x [swapped to RGB if to_rgb] - mean[c]) * (1 / std[c])), zeros in the padding
 * rows / columns (Hp >= H0, Wp >= W0).  mean_host / std_host: 3 doubles on the HOST (mmcv keeps them
 * as float64: x - float32(mean), times float32(1 / float64(std))). */
int bevops_image_normalize_pad(int in_dtype, const void *images, int out_dtype, void *output, int N,
                               int H0, int W0, int Hp, int Wp, const double *mean_host,
                               const double *std_host, int to_rgb, int channels_last, void *stream);
/* The MSDA call in two halves, for callers that sample ONE value tensor several times or want the
 * re-layout off their critical path (not a reference plugin: the plugin's enqueue is
 * bevops_msda_forward[_ws], which does both).  `packed` = the padded head-major form of `value`
 * (csrc/msda_pad.h, msda_hm4.hip), bevops_msda_packed_size() bytes, 128-byte aligned, caller-owned;
 * it depends on (dtype, shapes, bs, heads, num_query, num_point) and, for int8, on the flavour
 * (ref_dtype).  packed_size == 0 / NOT_SUPPORTED: shape outside the head-major domain (32 channels
 * per head, the instantiated (levels x points) combinations) -- use bevops_msda_forward. */
size_t bevops_msda_packed_size(int dtype, const int32_t *spatial_shapes_host, int bs, int nk, int heads,
                               int channels, int num_levels, int num_query, int num_point);
int bevops_msda_pack_value(int dtype, int ref_dtype, const void *value, const int32_t *spatial_shapes_host,
                           void *packed, size_t packed_bytes, int bs, int nk, int heads, int channels,
                           int num_levels, int num_query, int num_point, void *stream);
int bevops_msda_forward_prepacked(int dtype, const void *packed, size_t packed_bytes,
                                  const int32_t *spatial_shapes_host, const void *reference_points,
                                  int ref_dtype, const void *sampling_offsets, const void *attention_weights,
                                  void *output, int bs, int nk, int heads, int channels, int num_levels,
                                  int num_query, int num_point, int points_per_group, float scale_value,
                                  float scale_offset, float scale_weight, float scale_out, int shared_offsets,
                                  void *stream);
/* out[M, N] = act(a[M, K] . weight[N, K]^T + bias[N] + residual[M, N]) -- the dense layers around the
 * sampler (value_proj / output_proj / FFN) and the 1x1 convolutions of the
 * channels-last backbone as ONE hipBLASLt GEMM whose epilogue carries shift + identity + ReLU
 * (D = relu(A B + C + bias)).  fp16 tensors, fp32 accumulate; `bias`, `residual` optional;
 * `out` may alias `residual`.  `workspace`: caller-lent, bevops_linear_workspace_size() bytes (may be
 * NULL/0: only workspace-free algorithms are then considered).  NOT_SUPPORTED if the library has
 * no algorithm for the shape -- the caller then runs its own GEMM + bevops_bias_act_nhwc. */
size_t bevops_linear_workspace_size(void);
int bevops_linear_bias_act(int dtype, const void *a, const void *weight, const void *bias,
                           const void *residual, void *out, long long M, int N, int K, int relu,
                           void *workspace, size_t workspace_bytes, void *stream);
/* The encoder's value projection written straight into the sampler's planes, and the fused SCA sampling on
 * them (round 3; not reference plugins).  bevops_value_proj_packed: value = x @ weight.T + bias
 * (spatial_cross_attention.py:754, x [num_cams * nk, embed], weight [embed, embed]) computed by the tall-skinny
 * MFMA GEMM whose epilogue stores every output row into the padded head-major planes (csrc/msda_pad.h) instead
 * of a [cams, nk, heads, 32] tensor -- the separate re-layout pass of the sampler disappears.  `packed` (>=
 * bevops_value_proj_packed_size bytes, 128-byte aligned) then feeds bevops_sca_forward_prepacked: the fused
 * SCA sampling of bevops_sca_forward (camera-shared offsets / logits, pairs with bev_mask == 0 skipped, masked
 * camera sum) on those planes; workspace >= bevops_sca_prepacked_workspace_size bytes.  Domain: 4 levels x 8
 * points, 32 channels per head, embed % 256 == 0; BEVOPS_NOT_SUPPORTED otherwise. */
size_t bevops_value_proj_packed_size(const int32_t *spatial_shapes_host, int num_cams, int nk, int heads, int channels,
                                     int num_levels, int num_query, int num_point);
int bevops_value_proj_packed(const void *x, const void *weight, const void *bias, const int32_t *spatial_shapes_host,
                             void *packed, size_t packed_bytes, int num_cams, int nk, int heads, int channels,
                             int num_levels, int num_query, int num_point, void *stream);
int bevops_value_pack_planes(const void *value, const int32_t *spatial_shapes_host, void *packed, size_t packed_bytes,
                             int num_cams, int nk, int heads, int channels, int num_levels, int num_query,
                             int num_point, void *stream);   /* the re-layout alone, from a projected value tensor */
size_t bevops_sca_prepacked_workspace_size(int num_cams, int heads, int channels, int num_query);
int bevops_sca_forward_prepacked(int dtype, const void *packed, size_t packed_bytes, const int32_t *spatial_shapes_host,
                                 const void *reference_points_cam, const void *sampling_offsets,
                                 const void *attention_weights, const void *bev_mask, void *output, int num_cams,
                                 int nk, int heads, int channels, int num_levels, int num_query, int num_point,
                                 int points_per_group, void *workspace, size_t workspace_bytes, void *stream);
/* The same sampling on a VISIBILITY PLAN (round 5).  bev_mask depends on the calibration matrices only
 * (modules/encoder.py:255-258), so which (camera, query) pairs are sampled is known per rig, not per call:
 * bevops_sca_plan_build turns the fp16 bev_mask [num_cams, num_query] into per-camera ascending lists of the visible
 * queries (`plan`: bevops_sca_plan_size bytes -- 64 of counts, the lists, 1 KB of builder scratch per camera --
 * 16-byte aligned, device memory the caller keeps for as long as the mask is valid; num_cams <= 16, num_query <= 65 535;
 * two launches, no host synchronisation: a frame loop whose calibration changes per frame captures it in the frame's
 * graph behind bevops_point_sampling), and bevops_sca_forward_planned gives every block of the sampling
 * kernel an EQUAL slice of the global (camera, query) sequence -- no empty blocks, no per-block compaction, the same
 * number of rounds on every CU -- with the results of bevops_sca_forward_prepacked (bit-identical: same arithmetic
 * per pair, same reduction).  The plan also marks the pairs whose query no other camera sees and whose weight is
 * exactly 1 (89 % of the visible pairs of the 6-camera rig): the sampler stores those rows straight into `output`
 * (1 * v + 0 = v) and the camera reduce touches only the others.  `plan` must have been built from the VALUES of the
 * `bev_mask` passed here (not only from its zero pattern); bevops_sca_forward_planned returns BEVOPS_BAD_PARAM unless
 * plan_bytes == bevops_sca_plan_size(num_cams, num_query) -- a plan built for another camera set or query count is
 * rejected (what the plan's CONTENT was built from cannot be checked without a host synchronisation and is not). */
size_t bevops_sca_plan_size(int num_cams, int num_query);
int bevops_sca_plan_build(int dtype, const void *bev_mask, int num_cams, int num_query, void *plan, size_t plan_bytes,
                          void *stream);
int bevops_sca_forward_planned(int dtype, const void *packed, size_t packed_bytes, const int32_t *spatial_shapes_host,
                               const void *reference_points_cam, const void *sampling_offsets,
                               const void *attention_weights, const void *bev_mask, const void *plan, size_t plan_bytes,
                               void *output, int num_cams, int nk, int heads, int channels, int num_levels,
                               int num_query, int num_point, int points_per_group, void *workspace,
                               size_t workspace_bytes, void *stream);

/* point_sampling_trt (det2trt/models/modules/encoder.py:197-259) as one launch (csrc/point_sampling.hip): the BEV
 * pillar anchors `pillars` [num_points_in_pillar, num_query, 4] (fp32 metric homogeneous points: the frame-independent
 * first half, encoder.py:199-219) projected with `lidar2img` [num_cams, 4, 4] (fp32, row-major, DEVICE memory: the
 * reference feeds it as an engine input on every frame, tools/bevformer/evaluate_trt.py:131-132), divided by the depth
 * and the image size -> reference_points_cam [num_cams, num_query, num_points_in_pillar, 2] and the visibility weights
 * bev_mask [num_cams, num_query] = (any anchor inside the image and in front of the camera) / max(number of cameras that
 * see the pillar, 1e-4), both in `out_dtype` (BEVOPS_F16 | BEVOPS_F32).  Index generation is bit-exact (SURVEY 8a row
 * a6): fp32 arithmetic in the reference's op order -- separately rounded products and sums in ascending k, IEEE
 * divisions -- rounded once to the output type.  BEVOPS_NOT_SUPPORTED unless num_points_in_pillar == 4. */
int bevops_point_sampling(int out_dtype, const float *pillars, const float *lidar2img, void *reference_points_cam,
                          void *bev_mask, int num_cams, int num_query, int num_points_in_pillar, float image_h,
                          float image_w, void *stream);

/* Hand-written tall-skinny fp16 GEMM on the matrix cores (csrc/tsgemm.hip) for the dense layers that wrap the
 * samplers (the reference runs them as cuBLAS / TensorRT layers: spatial_cross_attention.py:694-768,
 * backbones/resnet.py:326-686): out[m, n] = act(sum_k x[m, k] w[n, k] + bias[n] (+ residual[m, n])), x [M, K],
 * w [N, K], residual / out [M, N] row-major fp16, fp32 accumulation, one rounding.  Persistent blocks stream
 * their rows once; both operands reach LDS by DMA.  BEVOPS_NOT_SUPPORTED outside K % 64 == 0, N % 256 == 0
 * (the caller keeps its library GEMM). */
int bevops_tsgemm_f16(const void *x, const void *weight, const void *bias, const void *residual, void *out,
                      long long m, int n, int k, int relu, void *stream);
/* The same GEMM with the LayerNorm that follows it in every encoder / decoder block (modules/encoder.py:586-636,
 * modules/decoder.py:52-112: attention or FFN with its identity, then `norm`) evaluated in the epilogue:
 *     out = LayerNorm_N(fp16(x w^T + bias (+ residual))) * ln_weight + ln_bias,     N == 256, K % 64 == 0.
 * The row is normalised from exactly the binary16 sums the unfused pair (this GEMM, then bevops_layer_norm) would have
 * read back -- fp32 mean, centred squares, one rounding -- without the second launch and the [M, 256] round trip.
 * ln_weight / ln_bias fp16 [256], 16-byte aligned.  BEVOPS_NOT_SUPPORTED when n != 256 or k % 64 != 0. */
int bevops_tsgemm_f16_ln(const void *x, const void *weight, const void *bias, const void *residual, const void *ln_weight,
                         const void *ln_bias, float eps, void *out, long long m, int n, int k, void *stream);
/* Self-attention of a few hundred queries on the matrix cores (csrc/attention.hip; the decoder's object queries:
 * mmcv MultiheadAttention in det2trt/models/modules/decoder.py:52-112 -- 900 queries, 8 heads x 32): qkv
 * [num_query, 3, heads, 32] fp16 (the in-projection's output: q, k, v of every head side by side, 16-byte aligned) ->
 * out [num_query, heads, 32] fp16 = softmax(scale q k^T) v per head; fp32 scores / statistics / accumulators.
 * BEVOPS_NOT_SUPPORTED unless head_dim == 32 and num_query <= bevops_mha_selfattn_max_queries() (the head's keys and
 * values live in LDS). */
size_t bevops_mha_selfattn_max_queries(void);
int bevops_mha_selfattn_f16(const void *qkv, void *out, int num_query, int heads, int head_dim, float scale, void *stream);
/* The int8 activation chain's flavour of the same persistent kernel (the int8 1x1 convolutions of ResNet stages
 * 3 / 4; arguments as bevops_linear_int8_chain with an int8 activation): a_q [M, K] / w_q [N, K] int8, int32 sums,
 * fp32 bias, identity rows int8 (res_dtype BEVOPS_I8, real = q * scale_res) or fp16, output int8 (requantised with
 * scale_out) or fp16.  128 k-values per step.  BEVOPS_NOT_SUPPORTED outside K % 128 == 0, N % 256 == 0.
 * The host layer uses it for the 256-column layers with K = 1 024 (ResNet stage-3 conv1), where it measured faster
 * than the tiled int8 GEMM (profiles/r04/tsgemm_s8_ab.jsonl). */
int bevops_tsgemm_s8(const void *a_q, float scale_a, const void *w_q, const float *w_scales, float scale_w,
                     const float *bias, const void *residual, int res_dtype, float scale_res, int out_dtype,
                     void *out, float scale_out, long long m, int n, int k, int relu, void *stream);

/* The same dense layer for problems with FEW rows (the decoder's 900 object queries: decoder.py:381-471,
 * bevformer_head.py:247-282; csrc/small_gemm.hip): 32 x 64 output tiles, split-K over the four waves of a block, every
 * operand fragment requested before the first matrix instruction -- one memory round trip per launch instead of a
 * chain of dependent k-steps.  fp16 in / out, fp32 sums, bias / residual fp16, optional.  BEVOPS_NOT_SUPPORTED outside
 * K % 64 == 0, K <= 1024, M <= 65536. */
int bevops_small_gemm_f16(const void *x, const void *weight, const void *bias, const void *residual, void *out,
                          long long m, int n, int k, int relu, void *stream);

/* OPTIONAL, BLOCKING: pick the hipBLASLt algorithm for one bevops_linear_bias_act problem (same
 * arguments) by timing every supporting algorithm on `stream` with the caller's buffers, and cache
 * it for the process.  `out` is scratch and must not alias `residual`.  Synchronises the host;
 * BAD_PARAM under stream capture.  Without it the operator runs the library heuristic's choice
 * (measured 1.5-3x slower at the backbone's shapes). */
int bevops_linear_tune(int dtype, const void *a, const void *weight, const void *bias,
                       const void *residual, void *out, long long M, int N, int K, int relu,
                       void *workspace, size_t workspace_bytes, void *stream);
int bevops_mdconv_forward_packed(int dtype, const void *input, const void *offset,
                                 const void *mask, const void *packed_weight, const void *bias,
                                 void *output, void *workspace, size_t workspace_bytes, int B,
                                 int Cin, int H, int W, int Cout, int Kh, int Kw, int stride_h,
                                 int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                                 int groups, int deform_groups, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BEVOPS_H_ */
