// Self-attention of the decoder's object queries (mmcv MultiheadAttention inside DetrTransformerDecoderLayer,
// det2trt/models/modules/decoder.py:52-112: 900 queries, 8 heads x 32 channels) as ONE small kernel on the matrix
// cores.  Not a reference plugin (TensorRT fuses this itself); the framework's fused kernel takes 29 us for these
// 0.8 GFLOP, six times per frame.
//
//   out[i, h, :] = sum_j softmax_j( scale * <q[i, h, :], k[j, h, :]> ) v[j, h, :]        qkv [n, 3, heads, 32] fp16
//
// MI355X mapping.  A block = one head x 32 queries, EIGHT waves that share the queries and split the KEYS (a wave walks
// an eighth of the 32-key blocks).  With the head's K / V image filling the LDS there is one block per CU, and nothing hides
// a wave's chain fragment read -> MFMA -> softmax -> MFMA of a key block: so the chain is made short and 232 blocks fill the
// chip.  Measured under graph replay (profiles/r06/attn_time.jsonl): four waves x 32 queries each over all keys, 64 blocks:
// 21.4 us; keys split over 4 / 8 / 16 waves: 10.9 / 8.8 / 9.1 us; the framework's fused kernel: 29.2 us.  The head's K
// (80-byte rows: conflict-free 16-byte fragment reads) and V (64-byte rows) are staged in LDS (n <= 1 024 keys: 144 KB),
// then per 32 keys, with v_mfma_f32_32x32x16_f16:
//   S^T[key, query] = K Q^T                  (A = K rows from LDS, B = the Q fragment, resident in registers)
//   online softmax down the KEY axis: in the C layout a lane holds 16 keys of ONE query (its partner lane + 32 the
//     other 16), so the running maximum / sum are lane-local plus one exchange with the partner, and the rescaling
//     of the output accumulator is a per-lane factor
//   O^T[d, query] += V^T P^T                 (B = the probabilities straight out of the lane's OWN registers: the key
//     order of the k index is permuted to the C layout's -- (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -- and the A operand
//     V^T picks its eight keys per lane from LDS in the same order, 2 bytes at a time)
// and at the end the waves' (maximum, sum, accumulator) meet in LDS: wave w < 4 rescales and adds the eight partial
// results of channels 8 w .. 8 w + 7 and stores them.  fp32 scores, maxima, sums and output accumulators; probabilities
// rounded to binary16 for the second product (as every fused attention does); exp2 with the scale folded in.
#include "common.h"

namespace bevops {
namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kAtD = 32;            // channels per head
constexpr int kAtKRow = 80;         // LDS bytes per K row (64 + 16: the 16-byte fragment reads of 16 lanes hit 64 banks)
constexpr int kAtVRow = 64;
constexpr int kAtWaves = 8;          // waves per block: the same 32 queries, an eighth of the keys each
constexpr int kAtThreads = 64 * kAtWaves;
constexpr int kAtMaxN = 1024;

__global__ __launch_bounds__(kAtThreads) void mha_selfattn_f16_kernel(const __half *__restrict__ qkv,
                                                                     __half *__restrict__ out, int n, int heads,
                                                                     float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [n_pad rows of K][n_pad rows of V]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = blockIdx.y;
  const int n_pad = (n + 31) & ~31;
  char *Ks = smem, *Vs = smem + (size_t)n_pad * kAtKRow;
  const size_t row = (size_t)3 * heads * kAtD;                  // halfs per query row of qkv
  const __half *kbase = qkv + ((size_t)heads + head) * kAtD, *vbase = qkv + ((size_t)2 * heads + head) * kAtD;
  // ---- stage K and V of the head: a thread moves 16 bytes (8 channels) of one key per pass; rows past n are zero
  for (int i = tid; i < n_pad * 4; i += kAtThreads) {
    const int key = i >> 2, c = i & 3;
    uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
    if (key < n) {
      kv = *reinterpret_cast<const uint4 *>(kbase + (size_t)key * row + c * 8);
      vv = *reinterpret_cast<const uint4 *>(vbase + (size_t)key * row + c * 8);
    }
    *reinterpret_cast<uint4 *>(Ks + key * kAtKRow + c * 16) = kv;
    *reinterpret_cast<uint4 *>(Vs + key * kAtVRow + c * 16) = vv;
  }
  // ---- the wave's Q fragment (B operand: column = query lane & 31, k = channels 16 t + 8 (lane >> 5) ..)
  const int hi = lane >> 5;
  const int q_idx = blockIdx.x * 32 + (lane & 31);
  f16x8_t qf[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (q_idx < n) v = *reinterpret_cast<const uint4 *>(qkv + (size_t)q_idx * row + head * kAtD + 16 * t + 8 * hi);
    qf[t] = __builtin_bit_cast(f16x8_t, v);
  }
  __syncthreads();
  // this wave's key blocks
  const int nblk = n_pad >> 5, per = (nblk + kAtWaves - 1) / kAtWaves;
  const int kb_begin = min(wave * per, nblk) * 32, kb_end = min((wave + 1) * per, nblk) * 32;

  f32x16_t acc;                       // O^T: row = channel (r & 3) + 8 (r >> 2) + 4 hi, column = this lane's query
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int d_lane = lane & 31;       // A-operand row of the second product: channel
  for (int kb = kb_begin; kb < kb_end; kb += 32) {
    // S^T block: 32 keys x 32 queries
    f32x16_t s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f16x8_t kf = *reinterpret_cast<const f16x8_t *>(Ks + (kb + (lane & 31)) * kAtKRow + (16 * t + 8 * hi) * 2);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[t], s, 0, 0, 0);
    }
    // scores in log2 units; keys past n (only in the last block) are out
    float mx = -INFINITY;
    if (kb + 32 > n) {                  // (wave-uniform: the one block with keys past n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * hi;
        s[r] = key < n ? s[r] * scale_log2e : -INFINITY;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] *= scale_log2e;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                 // the partner lane holds the query's other 16 keys
    const float m_new = fmaxf(m_run, mx);
    // (v_exp_f32 itself: the arguments are <= 0, results below the normal range may flush to 0 -- they are weights of
    // at most 2^-126 of the row's largest one)
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // (first block: exp2(-inf) = 0)
    m_run = m_new;
    float psum = 0.f;
    float p[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
      psum += p[r];
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] *= alpha;
    // O^T += V^T P^T, 16 keys per instruction: k index 8 hi + i  <->  key (i & 3) + 8 (i >> 2) + 4 hi + 16 u
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f16x8_t pf, vf;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        pf[i] = (_Float16)p[8 * u + i];
        const int key = kb + (i & 3) + 8 * (i >> 2) + 4 * hi + 16 * u;
        vf[i] = *reinterpret_cast<const _Float16 *>(Vs + key * kAtVRow + d_lane * 2);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, acc, 0, 0, 0);
    }
  }
  // ---- the waves' partial results meet in LDS (the K / V images are dead): [wave][18][64 lanes] floats
  __syncthreads();
  float *xs = reinterpret_cast<float *>(smem);
  {
    float *mine = xs + wave * 18 * 64 + lane;
    mine[0] = m_run;
    mine[64] = l_run;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[(2 + r) * 64] = acc[r];
  }
  __syncthreads();
  if (wave >= 4) return;               // waves 0-3 finish: four accumulator registers (= 4 channels per lane) each
  float m_all = -INFINITY;
#pragma unroll
  for (int w = 0; w < kAtWaves; ++w) m_all = fmaxf(m_all, xs[w * 18 * 64 + lane]);      // (equal in a lane and its partner)
  float l_tot = 0.f, o4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < kAtWaves; ++w) {
    const float *src = xs + w * 18 * 64;
    const float f = __builtin_amdgcn_exp2f(src[lane] - m_all);                   // a wave without keys: 2^-inf = 0
    l_tot += (src[64 + lane] + src[64 + (lane ^ 32)]) * f;
#pragma unroll
    for (int e = 0; e < 4; ++e) o4[e] += src[(2 + 4 * wave + e) * 64 + lane] * f;
  }
  const float inv = 1.f / l_tot;
  if (q_idx < n) {    // this wave stores channels 8 wave + 4 hi .. + 3 of its lanes' queries: 8 bytes
    __half *o = out + ((size_t)q_idx * heads + head) * kAtD + 8 * wave + 4 * hi;
    uint2 w2;
    w2.x = pack_h2(o4[0] * inv, o4[1] * inv);
    w2.y = pack_h2(o4[2] * inv, o4[3] * inv);
    *reinterpret_cast<uint2 *>(o) = w2;
  }
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" size_t bevops_mha_selfattn_max_queries(void) { return kAtMaxN; }

extern "C" int bevops_mha_selfattn_f16(const void *qkv, void *out, int num_query, int heads, int head_dim, float scale,
                                       void *stream) {
  if (!qkv || !out || num_query <= 0 || heads <= 0 || !(scale > 0.f)) return BEVOPS_BAD_PARAM;
  if (head_dim != kAtD || num_query > kAtMaxN || heads > 65535) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(qkv) || (reinterpret_cast<uintptr_t>(out) & 7u)) return BEVOPS_BAD_PARAM;
  const int n_pad = (num_query + 31) & ~31;
  const size_t lds = max((size_t)n_pad * (kAtKRow + kAtVRow), (size_t)kAtWaves * 18 * 64 * sizeof(float));
  if (!ensure_dynamic_lds<mha_selfattn_f16_kernel>(lds)) return BEVOPS_FAILURE;
  const dim3 grid((unsigned)((num_query + 31) / 32), (unsigned)heads);
  hipLaunchKernelGGL(mha_selfattn_f16_kernel, grid, dim3(kAtThreads), lds, static_cast<hipStream_t>(stream),
                     static_cast<const __half *>(qkv), static_cast<__half *>(out), num_query, heads,
                     scale * 1.4426950408889634f);
  return launch_status();
}
