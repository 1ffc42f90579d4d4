/* oracle/ref_capi.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * extern "C" doors onto the REFERENCE's own plugin host functions, compiled for the host
 * from the sources under /root/reference/TensorRT/plugin (see oracle/Makefile, target
 * _ref/libbevref.so, and oracle/cuda_on_cpu/).  All pointers are host pointers; layouts are
 * the ones the reference's plugins hand to these functions (file:line next to each door).
 * Nothing here is product code and nothing in the product links or loads it.
 */
#include <cuda_fp16.h>
#include <cuda_runtime.h>
#include <cublas_v2.h>

#include <vector>

#include "bevPoolKernel.h"
#include "gridSamplerKernel.h"
#include "modulatedDeformableConv2dKernel.h"
#include "multiScaleDeformableAttnKernel.h"
#include "rotateKernel.h"

extern "C" {

/* dtype codes: 0 = float, 1 = __half (scalar kernels), 2 = __half2 kernels (kCHW2 / packed) */

/* MultiScaleDeformableAttnPlugin::enqueue, multiScaleDeformableAttnPlugin.cpp:71-140 */
int bevref_msda(int dtype, const void *value, const int32_t *shapes, const void *ref, const void *off,
                const void *w, void *out, int bs, int nk, int heads, int C, int L, int nq, int P, int ppg) {
  if (dtype == 0)
    ms_deformable_im2col_cuda<float>((const float *)value, shapes, (const float *)ref, (const float *)off,
                                     (const float *)w, bs, nk, heads, C, L, nq, P, ppg, (float *)out, 0);
  else if (dtype == 1)
    ms_deformable_im2col_cuda<__half>((const __half *)value, shapes, (const __half *)ref, (const __half *)off,
                                      (const __half *)w, bs, nk, heads, C, L, nq, P, ppg, (__half *)out, 0);
  else if (dtype == 2)
    ms_deformable_im2col_cuda_h2((const __half2 *)value, shapes, (const __half2 *)ref, (const __half2 *)off,
                                 (const __half *)w, bs, nk, heads, C, L, nq, P, ppg, (__half2 *)out, 0);
  else
    return 1;
  return 0;
}

/* int8 MSDA; ref_half = 0: <float> reference points, 1: <__half2> (plugin.cpp:118-133) */
int bevref_msda_int8(int ref_half, const void *value, float s_v, const int32_t *shapes, const void *ref,
                     const void *off, float s_o, const void *w, float s_w, void *out, float s_out, int bs, int nk,
                     int heads, int C, int L, int nq, int P, int ppg) {
  if (ref_half)
    ms_deformable_im2col_cuda_int8<__half2>((const int8_4 *)value, s_v, shapes, (const __half2 *)ref,
                                            (const int8_4 *)off, s_o, (const int8_4 *)w, s_w, bs, nk, heads, C, L,
                                            nq, P, ppg, (int8_4 *)out, s_out, 0);
  else
    ms_deformable_im2col_cuda_int8<float>((const int8_4 *)value, s_v, shapes, (const float *)ref,
                                          (const int8_4 *)off, s_o, (const int8_4 *)w, s_w, bs, nk, heads, C, L, nq,
                                          P, ppg, (int8_4 *)out, s_out, 0);
  return 0;
}

/* RotatePlugin::enqueue, rotatePlugin.cpp:75-115.  dims = {C, H, W}; interp 0 bilinear, 1 nearest.
 * dtype 2: img/out in kCHW2 ([C/2][H][W][2]); angle/center __half. */
int bevref_rotate(int dtype, void *out, const void *img, const void *angle, const void *center, int C, int H, int W,
                  int interp) {
  int dims[3] = {C, H, W};
  const RotateInterpolation m = interp ? RotateInterpolation::Nearest : RotateInterpolation::Bilinear;
  if (dtype == 0)
    rotate<float>((float *)out, (float *)img, (float *)angle, (float *)center, dims, m, 0);
  else if (dtype == 1)
    rotate<__half>((__half *)out, (__half *)img, (__half *)angle, (__half *)center, dims, m, 0);
  else if (dtype == 2)
    rotate_h2((__half2 *)out, (__half2 *)img, (__half *)angle, (__half *)center, dims, m, 0);
  else
    return 1;
  return 0;
}

/* int8 rotate: img/out in kCHW4 ([C/4][H][W][4]); angle/center float (angle_half = 0) or __half */
int bevref_rotate_int8(int angle_half, void *out, float s_out, const void *img, float s_in, const void *angle,
                       const void *center, int C, int H, int W, int interp) {
  int dims[3] = {C, H, W};
  const RotateInterpolation m = interp ? RotateInterpolation::Nearest : RotateInterpolation::Bilinear;
  if (angle_half)
    rotate_int8((int8_4 *)out, s_out, (const int8_4 *)img, s_in, (const __half *)angle, (const __half *)center, dims,
                m, 0);
  else
    rotate_int8((int8_4 *)out, s_out, (const int8_4 *)img, s_in, (const float *)angle, (const float *)center, dims,
                m, 0);
  return 0;
}

/* GridSamplerPlugin::enqueue, gridSamplerPlugin.cpp:110-155.  nb_dims 4 or 5; dims arrays NCHW / NCDHW. */
int bevref_grid_sample(int dtype, void *out, const void *in, const void *grid, int *out_dims, int *in_dims,
                       int *grid_dims, int nb_dims, int interp, int pad, int align) {
  const GridSamplerInterpolation im = (GridSamplerInterpolation)interp;
  const GridSamplerPadding pm = (GridSamplerPadding)pad;
  if (dtype == 0)
    grid_sample<float>((float *)out, (const float *)in, (const float *)grid, out_dims, in_dims, grid_dims, nb_dims, im,
                       pm, align != 0, 0);
  else if (dtype == 1)
    grid_sample<__half>((__half *)out, (const __half *)in, (const __half *)grid, out_dims, in_dims, grid_dims, nb_dims,
                        im, pm, align != 0, 0);
  else if (dtype == 2)
    grid_sample<__half2>((__half2 *)out, (const __half2 *)in, (const __half2 *)grid, out_dims, in_dims, grid_dims,
                         nb_dims, im, pm, align != 0, 0);
  else
    return 1;
  return 0;
}

int bevref_grid_sample_int8(void *out, float s_out, const void *in, float s_in, const void *grid, float s_grid,
                            int *out_dims, int *in_dims, int *grid_dims, int nb_dims, int interp, int pad, int align) {
  grid_sample_int8((int8_4 *)out, s_out, (const int8_4 *)in, s_in, (const int8_4 *)grid, s_grid, out_dims, in_dims,
                   grid_dims, nb_dims, (GridSamplerInterpolation)interp, (GridSamplerPadding)pad, align != 0, 0);
  return 0;
}

/* BEVPoolPlugin::enqueue, bevPoolPlugin.cpp:66-112.  num_points = number of OUTPUT elements. */
int bevref_bev_pool_v2(int dtype, int c, int n_intervals, int num_points, const void *depth, const void *feat,
                       const int *ranks_depth, const int *ranks_feat, const int *ranks_bev,
                       const int *interval_starts, const int *interval_lengths, void *out) {
  if (dtype == 0)
    bev_pool_v2<float>(c, n_intervals, num_points, (const float *)depth, (const float *)feat, ranks_depth, ranks_feat,
                       ranks_bev, interval_starts, interval_lengths, (float *)out, 0);
  else if (dtype == 1)
    bev_pool_v2<__half>(c, n_intervals, num_points, (const __half *)depth, (const __half *)feat, ranks_depth,
                        ranks_feat, ranks_bev, interval_starts, interval_lengths, (__half *)out, 0);
  else if (dtype == 2)
    bev_pool_v2_h2(c, n_intervals, num_points, (const __half *)depth, (const __half2 *)feat, ranks_depth, ranks_feat,
                   ranks_bev, interval_starts, interval_lengths, (__half2 *)out, 0);
  else
    return 1;
  return 0;
}

int bevref_bev_pool_v2_int8(int c, int n_intervals, int num_points, const void *depth, float s_d, const void *feat,
                            float s_f, const int *ranks_depth, const int *ranks_feat, const int *ranks_bev,
                            const int *interval_starts, const int *interval_lengths, void *out, float s_out) {
  bev_pool_v2_int8(c, n_intervals, num_points, (const int8_t *)depth, s_d, (const int8_4 *)feat, s_f, ranks_depth,
                   ranks_feat, ranks_bev, interval_starts, interval_lengths, (int8_4 *)out, s_out, 0);
  return 0;
}

/* ModulatedDeformableConv2dPlugin::enqueue + getWorkspaceSize,
 * modulatedDeformableConv2dPlugin.cpp:73-198.  bias may be NULL. */
int bevref_mdconv(int dtype, const void *x, const void *weight, const void *bias, const void *offset,
                  const void *mask, void *out, int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride,
                  int pad, int dil, int group, int dgroup) {
  const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
  const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
  const size_t esz = dtype == 0 ? 4 : 2;
  const size_t cin = dtype == 0 ? (size_t)Cin : (size_t)(Cin + 1) / 2 * 2;
  std::vector<char> ws(((cin * kh * kw * Ho * Wo * esz + 15) / 16) * 16 + 64);
  const int step = B < 32 ? B : 32;
  if (dtype == 0)
    ModulatedDeformConvForwardCUDAKernel<float>((const float *)x, (const float *)weight, (const float *)bias,
                                                (const float *)offset, (const float *)mask, (float *)out, ws.data(), B,
                                                Cin, H, W, Cout, kw, kh, stride, stride, pad, pad, dil, dil, group,
                                                dgroup, step, nullptr, 0);
  else if (dtype == 1)
    ModulatedDeformConvForwardCUDAKernel<__half>((const __half *)x, (const __half *)weight, (const __half *)bias,
                                                 (const __half *)offset, (const __half *)mask, (__half *)out,
                                                 ws.data(), B, Cin, H, W, Cout, kw, kh, stride, stride, pad, pad, dil,
                                                 dil, group, dgroup, step, nullptr, 0);
  else if (dtype == 2)
    ModulatedDeformConvForwardCUDAKernel<__half2>((const __half2 *)x, (const __half2 *)weight, (const __half2 *)bias,
                                                  (const __half2 *)offset, (const __half2 *)mask, (__half2 *)out,
                                                  ws.data(), B, Cin, H, W, Cout, kw, kh, stride, stride, pad, pad,
                                                  dil, dil, group, dgroup, step, nullptr, 0);
  else
    return 1;
  return 0;
}

/* int8 DCN: x and weight in kCHW4, offset / mask / out linear int8, fp32 bias (may be NULL) */
int bevref_mdconv_int8(const void *x, float s_in, const void *weight, float s_w, const float *bias,
                       const void *offset, float s_off, const void *mask, float s_mask, void *out, float s_out, int B,
                       int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil, int group,
                       int dgroup) {
  const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
  const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
  const size_t hw4 = ((size_t)Ho * Wo + 3) / 4 * 4;
  size_t sz = (size_t)(Cin + 3) / 4 * 4 * kh * kw * hw4 + (size_t)Cout / group * hw4 * 4;
  std::vector<char> ws((sz + 15) / 16 * 16 + 64);
  const int step = B < 32 ? B : 32;
  ModulatedDeformConvForwardCUDAKernel_int8<float>((const int8_4 *)x, s_in, (const int8_4 *)weight, s_w, bias,
                                                   (const int8_t *)offset, s_off, (const int8_t *)mask, s_mask,
                                                   (int8_t *)out, s_out, ws.data(), B, Cin, H, W, Cout, kw, kh, stride,
                                                   stride, pad, pad, dil, dil, group, dgroup, step, nullptr, 0);
  return 0;
}

const char *bevref_version() { return "bevref: reference plugin kernels on the host (cuda_on_cpu)"; }
}
