// INT8 dense layers: the reference builds its INT8 engines with `LinearQ` / `Conv2dQ`
// = pytorch_quantization's QuantLinear / QuantConv2d (det2trt/models/utils/register.py:78-84; used
// through LINEAR_LAYERS.get at modules/spatial_cross_attention.py:58,329): per-tensor symmetric
// fake-quantisation of the layer input and of the weight (amax / 127, round-half-even, clamp to
// +-127), which TensorRT turns into an INT8 GEMM.  Here that GEMM is explicit:
//   bevops_quantize_rows : x fp16 [rows, K] -> int8, q = clamp(rne(x / s), -127, 127)
//   bevops_linear_int8   : out[M, N] = act((sum_k a_q[m, k] w_q[n, k]) * (s_a * s_w[n]) + bias[n]
//                          + residual[m, n]) on v_mfma_i32_32x32x32_i8 (int32 accumulate: the integer sum
//                          is exact, unlike the fp32 sum of the fake-quantised products), fp16 out --
//                          or int8 out requantised with s_out for a following int8 layer.
//   bevops_linear_int8_fused : the same with the fp16 activation quantised inside the GEMM's operand load.
// The GEMMs live in tile_gemm.hip; this file holds the stand-alone quantise pass.
// Not a reference plugin (TensorRT owns these layers there).
#include "common.h"

namespace bevops {
namespace {

__global__ __launch_bounds__(256) void quantize_rows_kernel(const __half *__restrict__ x, int8_t *__restrict__ q,
                                                            size_t nvec, float scale) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const uint4 v = reinterpret_cast<const uint4 *>(x)[i];
  const float f[8] = {h2f_lo(v.x), h2f_hi(v.x), h2f_lo(v.y), h2f_hi(v.y), h2f_lo(v.z), h2f_hi(v.z), h2f_lo(v.w), h2f_hi(v.w)};
  int8_t r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = (int8_t)(int)fminf(fmaxf(rintf(f[k] / scale), -127.f), 127.f);   // x / s as the host calibrators quantise
  reinterpret_cast<uint2 *>(q)[i] = *reinterpret_cast<const uint2 *>(r);
}

// int8 -> fp16 with one per-tensor scale: out = fp16(q * scale) (product in fp32, one rounding) -- what
// `q.to(fp16) * scale` computes, in one pass instead of a conversion and a scaling pass
__global__ __launch_bounds__(256) void dequantize_rows_kernel(const int8_t *__restrict__ q, __half *__restrict__ out,
                                                              size_t nvec, float scale) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const uint2 v = reinterpret_cast<const uint2 *>(q)[i];
  float f[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[k] = (float)(int8_t)((v.x >> (8 * k)) & 0xffu) * scale;
    f[4 + k] = (float)(int8_t)((v.y >> (8 * k)) & 0xffu) * scale;
  }
  uint4 o;
  o.x = pack_h2(f[0], f[1]); o.y = pack_h2(f[2], f[3]); o.z = pack_h2(f[4], f[5]); o.w = pack_h2(f[6], f[7]);
  reinterpret_cast<uint4 *>(out)[i] = o;
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" int bevops_quantize_rows(int dtype, const void *x, void *q, size_t count, float scale, void *stream) {
  if (!x || !q || !(scale > 0.f)) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16 || count % 8 != 0 || !aligned16(x) || (reinterpret_cast<uintptr_t>(q) & 7u))
    return BEVOPS_NOT_SUPPORTED;
  if (count == 0) return BEVOPS_SUCCESS;
  const size_t nvec = count / 8;
  hipLaunchKernelGGL(quantize_rows_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const __half *)x, (int8_t *)q, nvec, scale);
  return launch_status();
}


extern "C" int bevops_dequantize_rows(int dtype, const void *q, void *out, size_t count, float scale, void *stream) {
  if (!q || !out) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16 || count % 8 != 0 || !aligned16(out) || (reinterpret_cast<uintptr_t>(q) & 7u))
    return BEVOPS_NOT_SUPPORTED;
  if (count == 0) return BEVOPS_SUCCESS;
  const size_t nvec = count / 8;
  hipLaunchKernelGGL(dequantize_rows_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const int8_t *)q, (__half *)out, nvec, scale);
  return launch_status();
}
