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
// 128 x 128 x 64 tiles, register-staged operands, 4 waves of 64 x 64 (2 x 2 MFMA blocks each).
// Not a reference plugin (TensorRT owns these layers there).
#include "common.h"

namespace bevops {
namespace {

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));

constexpr int kQM = 128, kQN = 128, kQK = 64, kQLd = kQK + 16;  // +16 bytes: conflict-free b128 reads

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

// OUT8: int8 output requantised with 1 / s_out (round half even, clamp +-127); else fp16
template <bool OUT8>
__global__ __launch_bounds__(256) void gemm_s8_kernel(const int8_t *__restrict__ A, const int8_t *__restrict__ W,
                                                      const float *__restrict__ wscale, float s_aw,
                                                      const float *__restrict__ bias, const __half *__restrict__ res,
                                                      void *__restrict__ out, int M, int N, int K, int relu,
                                                      float inv_s_out) {
  __shared__ __attribute__((aligned(16))) int8_t As[kQM][kQLd];
  __shared__ __attribute__((aligned(16))) int8_t Ws[kQN][kQLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * kQM, n0 = blockIdx.x * kQN;
  const int r0 = tid >> 2, kc = (tid & 3) * 16, r1 = r0 + 64;  // 128 rows x 4 chunks of 16 B
  i32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
  const int nk = (K + kQK - 1) / kQK;
  uint4 ra0, ra1, rb0, rb1;
  auto ld = [&](const int8_t *p, bool ok) { return ok ? *reinterpret_cast<const uint4 *>(p) : make_uint4(0, 0, 0, 0); };
  auto gload = [&](int kt) {
    const int k = kt * kQK + kc;
    const bool kok = k < K;  // K % 16 == 0 (host check)
    ra0 = ld(A + (size_t)(m0 + r0) * K + k, kok && m0 + r0 < M);
    ra1 = ld(A + (size_t)(m0 + r1) * K + k, kok && m0 + r1 < M);
    rb0 = ld(W + (size_t)(n0 + r0) * K + k, kok && n0 + r0 < N);
    rb1 = ld(W + (size_t)(n0 + r1) * K + k, kok && n0 + r1 < N);
  };
  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    *reinterpret_cast<uint4 *>(&As[r0][kc]) = ra0;
    *reinterpret_cast<uint4 *>(&As[r1][kc]) = ra1;
    *reinterpret_cast<uint4 *>(&Ws[r0][kc]) = rb0;
    *reinterpret_cast<uint4 *>(&Ws[r1][kc]) = rb1;
    __syncthreads();
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kk = ks * 32 + (lane >> 5) * 16;
      i32x4_t a[2], b[2];
      // MFMA operand A = the weight rows (output columns n), B = the activation rows (m): the
      // 32x32 result block then has n along the 4-row groups and m along the lanes -- transposed
      // below so that a lane's 4 consecutive accumulator rows are 4 consecutive n of one m
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const i32x4_t *>(&Ws[wn * 64 + i * 32 + (lane & 31)][kk]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = *reinterpret_cast<const i32x4_t *>(&As[wm * 64 + j * 32 + (lane & 31)][kk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // acc[i][j][r]: n = n0 + wn*64 + i*32 + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), m = m0 + wm*64 + j*32 + (lane & 31):
  // four consecutive r = four consecutive n of one output row -> one 8-byte (fp16) / 4-byte (int8) store
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wm * 64 + j * 32 + (lane & 31);
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + i * 32 + 8 * g + 4 * (lane >> 5);
        if (n >= N) continue;   // N % 4 == 0 (host check): the group of 4 is all in or all out
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float sc = wscale ? s_aw * wscale[n + c] : s_aw;
          v[c] = (float)acc[i][j][4 * g + c] * sc + (bias ? bias[n + c] : 0.f);
        }
        if (res) {
          const uint2 rr = *reinterpret_cast<const uint2 *>(res + (size_t)m * N + n);
          v[0] += h2f_lo(rr.x); v[1] += h2f_hi(rr.x); v[2] += h2f_lo(rr.y); v[3] += h2f_hi(rr.y);
        }
        if (relu) {
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
        }
        if constexpr (OUT8) {
          unsigned pk = 0;
#pragma unroll
          for (int c = 0; c < 4; ++c)
            pk |= ((unsigned)(int)fminf(fmaxf(rintf(v[c] * inv_s_out), -127.f), 127.f) & 0xffu) << (8 * c);
          *reinterpret_cast<unsigned *>(static_cast<int8_t *>(out) + (size_t)m * N + n) = pk;
        } else {
          uint2 o;
          o.x = pack_h2(v[0], v[1]);
          o.y = pack_h2(v[2], v[3]);
          *reinterpret_cast<uint2 *>(static_cast<__half *>(out) + (size_t)m * N + n) = o;
        }
      }
  }
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

extern "C" int bevops_linear_int8(const void *a_q, float scale_a, const void *w_q, const float *w_scales,
                                  float scale_w, const float *bias, const void *residual, int out_dtype,
                                  void *out, float scale_out, long long M, int N, int K, int relu, void *stream) {
  if (!a_q || !w_q || !out || M < 0 || N <= 0 || K <= 0) return BEVOPS_BAD_PARAM;
  if (!(scale_a > 0.f) || (!w_scales && !(scale_w > 0.f))) return BEVOPS_BAD_PARAM;
  if (K % 16 != 0 || N % 4 != 0 || !aligned16(a_q) || !aligned16(w_q) || (reinterpret_cast<uintptr_t>(out) & 7u) ||
      (residual && (reinterpret_cast<uintptr_t>(residual) & 7u)) || M > 0x7fffffffLL)
    return BEVOPS_NOT_SUPPORTED;
  if (out_dtype == BEVOPS_I8 && !(scale_out > 0.f)) return BEVOPS_BAD_PARAM;
  if (M == 0) return BEVOPS_SUCCESS;
  const dim3 grid((unsigned)((N + kQN - 1) / kQN), (unsigned)((M + kQM - 1) / kQM));
  const float s_aw = w_scales ? scale_a : scale_a * scale_w;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (out_dtype == BEVOPS_F16)
    hipLaunchKernelGGL((gemm_s8_kernel<false>), grid, dim3(256), 0, st, (const int8_t *)a_q, (const int8_t *)w_q,
                       w_scales, s_aw, bias, (const __half *)residual, out, (int)M, N, K, relu, 0.f);
  else if (out_dtype == BEVOPS_I8)
    hipLaunchKernelGGL((gemm_s8_kernel<true>), grid, dim3(256), 0, st, (const int8_t *)a_q, (const int8_t *)w_q,
                       w_scales, s_aw, bias, (const __half *)residual, out, (int)M, N, K, relu, 1.0f / scale_out);
  else
    return BEVOPS_NOT_SUPPORTED;
  return launch_status();
}
