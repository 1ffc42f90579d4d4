// Dense layers with FEW rows (the decoder's 900 object queries, SURVEY.md 8a-5: decoder.py:381-471 and the branches of
// bevformer_head.py:247-282; not reference plugins -- TensorRT owns these layers there):
//     out[m, n] = act( sum_k x[m, k] w[n, k] + bias[n] (+ residual[m, n]) ),   fp16 in / out, fp32 sums.
// At M = 900 the tiled GEMMs (tile_gemm.hip, tsgemm.hip) run 16-32 blocks that each walk a chain of 8-16 dependent
// k-steps (load -> LDS -> barrier -> MFMA, ~1.5 us per step): 15-16 us for 0.1-0.2 GFLOP, sixty times per frame.
// Such a layer has no bandwidth or matrix-core problem, only that chain.  Here there is NO k-loop pipeline at all:
//   * block tile 32 rows x 64 columns (M = 900, N = 256: 116 blocks), 256 threads = 4 waves, each wave owns a QUARTER of
//     K (split-K inside the block);
//   * a wave loads its operands straight from global memory in matrix-instruction fragment layout -- lane (l & 31) is
//     the row, (l >> 5) the 8-value half of a 16-value k-step: one 16-byte buffer load per lane, fragment and step --
//     and issues ALL of them before the first v_mfma_f32_32x32x16_f16: one memory round trip per launch;
//   * the four waves' fp32 partial tiles meet in LDS (32 KB), every thread then owns 8 consecutive columns of one row:
//     bias, identity, ReLU in fp32, one rounding, one 16-byte store.
// Rows / columns past M / N read as zero through the buffer descriptors' range check and are not stored.
// Domain: K % 64 == 0 (four k-quarters of whole 16-value steps), K <= 1024 (operand registers), 16-byte aligned rows.
#include "common.h"

namespace bevops {
namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kSgM = 32, kSgN = 64, kSgStride = kSgN + 4;   // partial rows padded by 4 floats (bank spread)
constexpr unsigned kSgOob = 0xFFFFFF00u;

struct SmallArgs {
  const __half *x, *w, *bias, *res;
  __half *out;
  int M, N, K, relu;
};

// STEPS = k-steps (of 16 values) per wave = K / 64
template <int STEPS>
__global__ __launch_bounds__(256) void small_gemm_f16_kernel(SmallArgs p) {
  __shared__ __attribute__((aligned(16))) float part[4][kSgM][kSgStride];   // 34.8 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (p.N + kSgN - 1) / kSgN;
  const int m0 = ((int)blockIdx.x / tiles_n) * kSgM, n0 = ((int)blockIdx.x % tiles_n) * kSgN;
  const int K = p.K;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__half *>(p.x), 0, (unsigned)((size_t)p.M * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__half *>(p.w), 0, (unsigned)((size_t)p.N * K * 2), 0x00020000);
  const int r = lane & 31, half = lane >> 5;
  const int kq = wave * (K / 4) + half * 8;          // first k of this lane's fragments in the wave's quarter
  const unsigned xo = m0 + r < p.M ? (unsigned)(((size_t)(m0 + r) * K + kq) * 2) : kSgOob;
  const unsigned wo0 = n0 + r < p.N ? (unsigned)(((size_t)(n0 + r) * K + kq) * 2) : kSgOob;
  const unsigned wo1 = n0 + 32 + r < p.N ? (unsigned)(((size_t)(n0 + 32 + r) * K + kq) * 2) : kSgOob;
  // every operand fragment of the wave's quarter, requested before any is used
  u32x4 fx[STEPS], fw0[STEPS], fw1[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    fx[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)xo, s * 32, 0);
    fw0[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)wo0, s * 32, 0);
    fw1[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)wo1, s * 32, 0);
  }
  // the identity rows of the epilogue do not depend on the sums either
  const int er = tid >> 3, ec = (tid & 7) * 8;       // epilogue role: row er, columns ec .. ec + 7 of the tile
  const int em = m0 + er, en = n0 + ec;
  const bool vec = (p.N & 7) == 0;
  uint4 rres = make_uint4(0u, 0u, 0u, 0u);
  if (p.res && vec && em < p.M && en < p.N)
    rres = *reinterpret_cast<const uint4 *>(p.res + (size_t)em * p.N + en);
  f32x16_t acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  // operand A = weight rows (output columns), B = activation rows: a lane's 4 consecutive accumulator values are 4
  // consecutive n of one output row m (the fragment convention of tile_gemm.hip)
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const f16x8_t b = __builtin_bit_cast(f16x8_t, fx[s]);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, fw0[s]), b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, fw1[s]), b, acc1, 0, 0, 0);
  }
  // acc{cb}[4 g + c]: m = r, n = cb * 32 + 8 g + 4 half + c
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    *reinterpret_cast<f32x4 *>(&part[wave][r][8 * g + 4 * half]) =
        f32x4{acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]};
    *reinterpret_cast<f32x4 *>(&part[wave][r][32 + 8 * g + 4 * half]) =
        f32x4{acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
  }
  __syncthreads();
  if (em >= p.M || en >= p.N) return;
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = 0.f;
#pragma unroll
  for (int w4 = 0; w4 < 4; ++w4) {      // fixed order: the result does not depend on which wave finished first
    const f32x4 lo = *reinterpret_cast<const f32x4 *>(&part[w4][er][ec]);
    const f32x4 hi = *reinterpret_cast<const f32x4 *>(&part[w4][er][ec + 4]);
#pragma unroll
    for (int c = 0; c < 4; ++c) { v[c] += lo[c]; v[4 + c] += hi[c]; }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (p.bias && en + c < p.N) v[c] += __half2float(p.bias[en + c]);
  if (p.res) {
    if (vec) {
      v[0] += h2f_lo(rres.x); v[1] += h2f_hi(rres.x); v[2] += h2f_lo(rres.y); v[3] += h2f_hi(rres.y);
      v[4] += h2f_lo(rres.z); v[5] += h2f_hi(rres.z); v[6] += h2f_lo(rres.w); v[7] += h2f_hi(rres.w);
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (en + c < p.N) v[c] += __half2float(p.res[(size_t)em * p.N + en + c]);
    }
  }
  if (p.relu) {
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.f);
  }
  __half *o = p.out + (size_t)em * p.N + en;
  if (vec) {
    uint4 q;
    q.x = pack_h2(v[0], v[1]); q.y = pack_h2(v[2], v[3]); q.z = pack_h2(v[4], v[5]); q.w = pack_h2(v[6], v[7]);
    *reinterpret_cast<uint4 *>(o) = q;
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (en + c < p.N) o[c] = __float2half_rn(v[c]);
  }
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" int bevops_small_gemm_f16(const void *x, const void *weight, const void *bias, const void *residual,
                                     void *out, long long M, int N, int K, int relu, void *stream) {
  if (!x || !weight || !out || M < 0 || N <= 0 || K <= 0) return BEVOPS_BAD_PARAM;
  if (K % 64 != 0 || K > 1024 || M > 65536 || !aligned16(x) || !aligned16(weight) || !aligned16(out) ||
      (residual && !aligned16(residual)) || (bias && (reinterpret_cast<uintptr_t>(bias) & 1u)))
    return BEVOPS_NOT_SUPPORTED;
  if ((unsigned long long)M * K * 2 >= kSgOob || (unsigned long long)N * K * 2 >= kSgOob) return BEVOPS_NOT_SUPPORTED;
  if (M == 0) return BEVOPS_SUCCESS;
  SmallArgs p{static_cast<const __half *>(x), static_cast<const __half *>(weight), static_cast<const __half *>(bias),
              static_cast<const __half *>(residual), static_cast<__half *>(out), (int)M, N, K, relu};
  const long long blocks = ((M + kSgM - 1) / kSgM) * ((N + kSgN - 1) / kSgN);
  if (blocks > 0x7fffffffLL) return BEVOPS_NOT_SUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)blocks), blk(256);
  switch (K / 64) {
#define BEVOPS_SG(S_) case S_: hipLaunchKernelGGL(small_gemm_f16_kernel<S_>, grid, blk, 0, st, p); break
    BEVOPS_SG(1); BEVOPS_SG(2); BEVOPS_SG(3); BEVOPS_SG(4); BEVOPS_SG(5); BEVOPS_SG(6); BEVOPS_SG(7); BEVOPS_SG(8);
    BEVOPS_SG(9); BEVOPS_SG(10); BEVOPS_SG(11); BEVOPS_SG(12); BEVOPS_SG(13); BEVOPS_SG(14); BEVOPS_SG(15); BEVOPS_SG(16);
#undef BEVOPS_SG
    default: return BEVOPS_NOT_SUPPORTED;
  }
  return launch_status();
}
