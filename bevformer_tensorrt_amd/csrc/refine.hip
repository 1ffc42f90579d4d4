// Decoder reference-point refinement (det2trt/models/modules/decoder.py:93-103 with its own inverse_sigmoid, :24-40):
//     new_ref = sigmoid(cat(tmp[..., :2], tmp[..., 4:5]) + log(clamp(ref, eps, 1 - eps) / (1 - clamp(ref, eps, 1 - eps))))
// on binary16 tensors, as ONE launch instead of the framework's eight per decoder layer (cat, clamp, rsub, div, log,
// add, sigmoid, + the contiguous copy of the (x, y) columns the next layer's sampler takes): 42 launches of 4-5 us per
// frame.  The refined points are SAMPLING LOCATIONS of the next layer (index generation bit-exact),
// so every step is the framework's: operands widened to fp32, ONE operation, rounded back to binary16 -- clamp against
// the fp32 images of the Python scalars 1e-5 and 1 - 1e-5, 1 - x, IEEE division, log, the sum, sigmoid.  The two
// transcendental steps are binary16 -> binary16 functions of ONE argument, and this compiler's logf is not the
// framework binary's (its log of 0.0895385742 lands on the other side of a binary16 tie: one reference point in 15 361
// came out three ulps off; round 4's one-launch refinement was removed for the same reason), so they are TABLES: 65 536
// entries each, filled once per process by the framework's own torch.log / torch.sigmoid on every binary16 value
// (functions/refine.py) -- bit-exact by construction, whatever device library either side was built with.
// tests/test_refine_gpu.py compares with the framework's op sequence on EVERY binary16 reference point in [0, 1] and on
// every finite binary16 regression value.
#include "common.h"

namespace bevops {
namespace {

__device__ __forceinline__ float r16(float v) { return __half2float(__float2half_rn(v)); }   // one binary16 rounding

__device__ __forceinline__ float table16(const __half *__restrict__ tab, float v) {
  return __half2float(tab[__half_as_ushort(__float2half_rn(v))]);
}

__device__ __forceinline__ float refine_one(float t, float ref, const __half *__restrict__ log_tab,
                                            const __half *__restrict__ sig_tab) {
#pragma clang fp contract(off) reciprocal(off)
  const float lo = 1e-5f, hi = (float)(1.0 - 1e-5);
  const float x = r16(fminf(fmaxf(ref, lo), hi));       // torch.clamp(min=eps, max=1 - eps) on a half tensor
  // (a NaN reference point stays NaN in the framework's clamp: fmaxf / fminf would drop it)
  const float xc = ref != ref ? ref : x;
  const float om = r16(1.0f - xc);                        // 1 - x
  const float q = r16(__fdiv_rn(xc, om));                 // x / (1 - x)
  const float l = table16(log_tab, q);                    // torch.log (q is a binary16 value)
  const float s = r16(t + l);                             // tmp + inverse_sigmoid(ref)
  return table16(sig_tab, s);                             // torch.sigmoid
}

__global__ __launch_bounds__(256) void refine_reference_points_kernel(const __half *__restrict__ tmp,
                                                                      const __half *__restrict__ ref,
                                                                      __half *__restrict__ new_ref,
                                                                      __half *__restrict__ ref_xy, int n, int tmp_stride,
                                                                      const __half *__restrict__ log_tab,
                                                                      const __half *__restrict__ sig_tab) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  const __half *t = tmp + (size_t)q * tmp_stride;
  const float v0 = refine_one(__half2float(t[0]), __half2float(ref[3 * q]), log_tab, sig_tab);
  const float v1 = refine_one(__half2float(t[1]), __half2float(ref[3 * q + 1]), log_tab, sig_tab);
  const float v2 = refine_one(__half2float(t[4]), __half2float(ref[3 * q + 2]), log_tab, sig_tab);
  new_ref[3 * q] = __float2half_rn(v0);
  new_ref[3 * q + 1] = __float2half_rn(v1);
  new_ref[3 * q + 2] = __float2half_rn(v2);
  if (ref_xy) {
    ref_xy[2 * q] = __float2half_rn(v0);
    ref_xy[2 * q + 1] = __float2half_rn(v1);
  }
}

// The head's box decoding (bevformer_head.py:247-282 with mmdet's inverse_sigmoid, :6,254: clamp to [0, 1], then each
// factor to >= eps) on the stacked decoder levels, same construction: per element
//   ref' = log(max(clamp(ref, 0, 1), eps) / max(1 - clamp(ref, 0, 1), eps))
//   x = sigmoid(reg[0] + ref'[0]) * sx + ox,  y = sigmoid(reg[1] + ref'[1]) * sy + oy,  z = sigmoid(reg[4] + ref'[2]) * sz + oz
// every step rounded to binary16 as the framework's ~20 launches round it; the other seven columns are copied.
__device__ __forceinline__ float clamp_keep_nan(float v, float lo, float hi) {
  return v != v ? v : fminf(fmaxf(v, lo), hi);
}
__device__ __forceinline__ float max_keep_nan(float v, float lo) { return v != v ? v : fmaxf(v, lo); }

__device__ __forceinline__ float decode_one(float reg, float ref, float scale, float off, const __half *__restrict__ log_tab,
                                            const __half *__restrict__ sig_tab) {
#pragma clang fp contract(off) reciprocal(off)
  const float c = r16(clamp_keep_nan(ref, 0.f, 1.f));
  const float x1 = r16(max_keep_nan(c, 1e-5f));
  const float x2 = r16(max_keep_nan(r16(1.0f - c), 1e-5f));
  const float l = table16(log_tab, r16(__fdiv_rn(x1, x2)));
  const float sg = table16(sig_tab, r16(reg + l));
  return r16(r16(sg * scale) + off);
}

__global__ __launch_bounds__(256) void decode_boxes_kernel(const __half *__restrict__ regs, const __half *__restrict__ refs,
                                                           __half *__restrict__ out, int n, float sx, float ox, float sy,
                                                           float oy, float sz, float oz, const __half *__restrict__ log_tab,
                                                           const __half *__restrict__ sig_tab) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  const __half *r = regs + (size_t)q * 10;
  __half o[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) o[k] = r[k];
  o[0] = __float2half_rn(decode_one(__half2float(r[0]), __half2float(refs[3 * q]), sx, ox, log_tab, sig_tab));
  o[1] = __float2half_rn(decode_one(__half2float(r[1]), __half2float(refs[3 * q + 1]), sy, oy, log_tab, sig_tab));
  o[4] = __float2half_rn(decode_one(__half2float(r[4]), __half2float(refs[3 * q + 2]), sz, oz, log_tab, sig_tab));
#pragma unroll
  for (int k = 0; k < 10; ++k) out[(size_t)q * 10 + k] = o[k];
}

}  // namespace
}  // namespace bevops

using namespace bevops;

// regs [count, 10], refs [count, 3] -> out [count, 10] (fp16; out may alias regs): the head's decoded boxes of all stacked
// levels.  (scale, offset) per axis = (pc_range[3 + a] - pc_range[a], pc_range[a]) as fp32.  Tables as below.
extern "C" int bevops_decode_boxes(int dtype, const void *regs, const void *refs, void *out, int count, float scale_x,
                                   float offset_x, float scale_y, float offset_y, float scale_z, float offset_z,
                                   const void *log_table, const void *sigmoid_table, void *stream) {
  if (dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (!regs || !refs || !out || !log_table || !sigmoid_table || count <= 0) return BEVOPS_BAD_PARAM;
  hipLaunchKernelGGL(decode_boxes_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const __half *>(regs), static_cast<const __half *>(refs), static_cast<__half *>(out), count,
                     scale_x, offset_x, scale_y, offset_y, scale_z, offset_z, static_cast<const __half *>(log_table),
                     static_cast<const __half *>(sigmoid_table));
  return launch_status();
}

// tmp [n, tmp_stride >= 5] (the regression branch's output: columns 0, 1 and 4 are used), reference_points [n, 3],
// new_reference_points [n, 3], reference_xy [n, 2] (may be null): all fp16.  log_table / sigmoid_table: fp16 [65 536],
// entry i = the function of the binary16 value with bit pattern i (the caller's definition of log and sigmoid).
extern "C" int bevops_refine_reference_points(int dtype, const void *tmp, const void *reference_points,
                                              void *new_reference_points, void *reference_xy, int num_query, int tmp_stride,
                                              const void *log_table, const void *sigmoid_table, void *stream) {
  if (dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (!tmp || !reference_points || !new_reference_points || !log_table || !sigmoid_table || num_query <= 0 || tmp_stride < 5)
    return BEVOPS_BAD_PARAM;
  hipLaunchKernelGGL(refine_reference_points_kernel, dim3((unsigned)((num_query + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const __half *>(tmp),
                     static_cast<const __half *>(reference_points), static_cast<__half *>(new_reference_points),
                     static_cast<__half *>(reference_xy), num_query, tmp_stride, static_cast<const __half *>(log_table),
                     static_cast<const __half *>(sigmoid_table));
  return launch_status();
}
