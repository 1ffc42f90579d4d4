// Dense layers of the re-hosted network as ONE tiled matrix-core GEMM skeleton in three operand flavours
// (not reference plugins -- TensorRT owns these layers there):
//     out[m, n] = act( (sum_k a[m, k] w[n, k]) * scale[n] + bias[n] (+ residual[m, n]) )
//   S8    a int8 [M, K], w int8 [N, K]: v_mfma_i32_32x32x32_i8, int32 sums (exact)      bevops_linear_int8
//   F16Q  a fp16, quantised with 1 / s_a ON ITS WAY into LDS, then as S8                  bevops_linear_int8_fused
//   F16   a fp16, w fp16: v_mfma_f32_32x32x16_f16, fp32 sums                             bevops_tile_gemm_f16
// a, w row-major (nn.Linear / 1x1-convolution weight layout).  These layers are memory-bound at batch 1 (M is
// 8 700 .. 556 800 pixel or query rows, K and N are 64 .. 2 048): what matters is that the activation rows
// stream from HBM once at full rate while enough independent work is resident to cover the latency.
//
// MI355X mapping.  128 x 128 output tiles (128 x 64 for layers with N <= 64), 256 threads = 4 waves of 64 x 64
// (2 x 2 MFMA blocks of 32 x 32; 64 x 32 in the narrow flavour),
// 64 BYTES of k per step and row in every flavour (64 int8 / 32 fp16 values: the LDS images, the 16-byte
// fragment reads and the staging loads are the same code).  Operands are register-staged through buffer loads
// (rows past M / N read as zero, no branches) in TWO register sets -- the loads of two steps are in flight per block
// (one set in the F16Q flavour, whose activation loads are twice as wide) -- and written into the OTHER of two LDS
// images while the current one is multiplied: one barrier per step.  40 KB LDS and <= 158 VGPRs keep THREE blocks on a
// CU -- their prologues, k-loops and epilogues interleave, which is what covers the HBM latency (a persistent
// one-block-per-CU kernel with DMA operands, tsgemm.hip, is faster only for 256-column layers with K >= 256).
// Tiles are numbered so that the column tiles of one row tile run on the same XCD back to back (block b runs on
// XCD b % 8): the activation rows come from HBM once and from that XCD's L2 afterwards.
// F16Q: q = clamp(rne(x * (1 / s)), -127, 127); the product and the add of 1.5 * 2^23 are ONE fused
// multiply-add (v_fma_mix_f32 on the fp16 halves: single rounding), the clamp one v_med3_i32 on the float's bits,
// the integer its low mantissa byte: 2.75 VALU operations per element.  x * fl(1 / s) against the fl(x / s) of
// bevops_quantize_rows: the two can disagree (by one step) only where x / s is within 1e-5 of a rounding tie
// (4.7e-5 of the elements of a Gaussian tensor; tests/test_linear_q_gpu.py emulates this quantiser bit for bit).
// Epilogue: the identity rows are requested before the k-loop ends (they do not depend on it); the raw sums go
// through LDS per wave (32 rows x 64 columns at a time) so that a thread owns 8 consecutive columns of a row:
// scale, bias, identity, ReLU in fp32 on 16-byte accesses, ONE rounding to fp16 (or the requantisation to int8
// for a following int8 layer).
// INT8 chain (bevops_linear_int8_chain, bevops_conv_tile_int8): activations that already ARE int8 (the producer's
// epilogue requantised them with this layer's input scale), the identity rows int8 with their own scale, the
// output int8 with the consumer's scale -- one byte per element in and out of every layer of a ResNet bottleneck.
// Convolution mode (bevops_conv_tile_f16, bevops_conv_tile_int8_fused, bevops_conv_tile_int8): the same kernel as an implicit GEMM over
// channels-last activations --
// output row m is pixel (b, yo, xo), its k-values are [tap][Cin]; a tap moves the row's base address by a
// block-uniform delta and a per-row validity bit (zero padding) selects the beyond-the-buffer address; a stride
// only changes the row -> pixel map.  No column buffer, no sub-sampled copy.
#include <type_traits>

#include "common.h"

namespace bevops {
namespace {

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

enum { kS8 = 0, kF16Q = 1, kF16 = 2 };

constexpr int kTM = 128;   // (bytes of k per row and step: template parameter KB; LDS rows are KB + 16: conflict-free b128 reads)
constexpr float kQMagic = 12582912.f;          // 1.5 * 2^23: bits 0x4B400000, low byte 0
constexpr int kQMagicBits = 0x4B400000;
constexpr int kEpiStride = 64 * 4 + 16;        // staging row: 64 x 4 bytes + pad
constexpr unsigned kOob = 0xFFFFFF00u;         // beyond any buffer: reads as zero

__device__ __forceinline__ unsigned quant4(unsigned h01, unsigned h23, float r) {
  int b0 = __float_as_int(__builtin_fmaf(h2f_lo(h01), r, kQMagic));
  int b1 = __float_as_int(__builtin_fmaf(h2f_hi(h01), r, kQMagic));
  int b2 = __float_as_int(__builtin_fmaf(h2f_lo(h23), r, kQMagic));
  int b3 = __float_as_int(__builtin_fmaf(h2f_hi(h23), r, kQMagic));
  b0 = min(max(b0, kQMagicBits - 127), kQMagicBits + 127);
  b1 = min(max(b1, kQMagicBits - 127), kQMagicBits + 127);
  b2 = min(max(b2, kQMagicBits - 127), kQMagicBits + 127);
  b3 = min(max(b3, kQMagicBits - 127), kQMagicBits + 127);
  // low bytes of b0..b3 -> one word
  return __builtin_amdgcn_perm((unsigned)b1, (unsigned)b0, 0x0c0c0400u) |
         __builtin_amdgcn_perm((unsigned)b3, (unsigned)b2, 0x04000c0cu);
}

__device__ __forceinline__ int sum_bits(float v) { return __float_as_int(v); }
__device__ __forceinline__ int sum_bits(int v) { return v; }

__device__ __forceinline__ uint4 quant16(const uint4 &lo, const uint4 &hi, float r) {
  return make_uint4(quant4(lo.x, lo.y, r), quant4(lo.z, lo.w, r), quant4(hi.x, hi.y, r), quant4(hi.z, hi.w, r));
}

struct TileArgs {
  const void *a, *w, *bias, *res;   // bias: fp32 (S8 / F16Q) or fp16 (F16); res fp16
  const float *wscale;              // per output channel (S8 / F16Q) or null
  void *out;
  float inv_sa, s_aw, inv_s_out;
  float s_res;                      // RES8: scale of the int8 identity rows
  int M, N, K, relu, tiles_n, tiles_total;
  unsigned a_bytes;                 // size of the activation buffer (range check of its loads)
  // conv_cin > 0: implicit convolution over channels-last [B, Hin, Win, Cin] activations: kernel ks x ks
  // (1 or 3), pad ks / 2, stride `conv_s`, output rows [B, Hout, Wout]
  int conv_cin, conv_hin, conv_win, conv_hout, conv_wout, conv_s, conv_ks;
};

// NI: 32-column MFMA blocks per wave: 2 -> 128-column tiles, 1 -> 64-column tiles (layers with N <= 64: no
// matrix work on columns that do not exist)
// RES8: the identity rows are int8 [M, N] (real = q * s_res) instead of fp16
// Tiles are 128 rows (two 32-row MFMA blocks per wave, three blocks per CU) and a step holds 64 bytes of k per row.
// Two other builds were measured in round 4 and are no longer in this file (history; profiles/r04/tile_rows_ab.jsonl,
// tile_wide_ab.jsonl): 64-row tiles (four blocks per CU: 3-10 % slower on every layer) and 128-byte steps for the int8
// chain's plain GEMMs (bit-identical, slower: a step costs its latency whatever it holds).
constexpr int tile_lds_bytes(int tm) {
  return 2 * (tm + 128) * (64 + 16) > 4 * 32 * kEpiStride ? 2 * (tm + 128) * (64 + 16) : 4 * 32 * kEpiStride;
}
template <int MODE, bool OUT8, int NI, bool CONV, bool RES8 = false>
__global__ __launch_bounds__(256, 3) void tile_gemm_kernel(TileArgs p) {
  constexpr int MJ = 2;                          // 32-row MFMA blocks per wave
  constexpr int KB = 64;                         // bytes of k per row and step
  constexpr int kTN = 64 * NI;
  constexpr int TM = 64 * MJ;                    // rows per tile
  constexpr int kTLd = KB + 16;                  // LDS row: the step's bytes + 16 (conflict-free 16-byte fragment reads)
  constexpr int kHalves = KB / 64;               // 64-byte halves of a step: a staging thread takes one 16-byte chunk of each
  // [image][A rows | W rows][KB + 16]; 40 KB static
  __shared__ __attribute__((aligned(16))) char smem[tile_lds_bytes(TM)];
  constexpr int kAB = MODE == kS8 ? 1 : 2;     // bytes per activation element in memory
  constexpr int kWB = MODE == kF16 ? 2 : 1;    // bytes per weight element
  constexpr int kAV = MODE == kF16Q ? 2 : 1;   // 16-byte loads per activation row and step
  constexpr int kStepK = MODE == kF16 ? KB / 2 : KB;       // k-values per step
  const int M = p.M, N = p.N, K = p.K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: the 8 XCDs take contiguous runs of the (row tile, column tile) sequence
  const int per_xcd = (p.tiles_total + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= p.tiles_total) return;
  const int m0 = (logical / p.tiles_n) * TM, n0 = (logical % p.tiles_n) * kTN;
  const int r0 = tid >> 2, r1 = r0 + 64;       // 128 rows x 4 chunks of 16 bytes of k
  const int kce = (tid & 3) * (MODE == kF16 ? 8 : 16);   // this thread's first k-value inside a (64-byte half of a) step
  typename std::conditional<MODE == kF16, f32x16_t, i32x16_t>::type acc[NI][MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
  const int nk = (K + kStepK - 1) / kStepK;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(p.a), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(p.w), 0, (unsigned)((size_t)N * K * kWB), 0x00020000);
  // conv mode (implicit GEMM): output row m is pixel (b, y, x); its k-values are [tap 9][Cin] with tap (dy, dx)
  // read from pixel (y + dy - 1, x + dx - 1) of the same image, zero outside it.  A step never straddles a tap
  // (Cin % 32 == 0, host check): per step the tap moves the row's base address by a block-uniform delta and
  // a per-row validity bit decides between that address and the beyond-the-buffer one.
  const int cin = p.conv_cin;
  constexpr bool conv = CONV;
  const int taps = p.conv_ks * p.conv_ks, pad = p.conv_ks >> 1;
  unsigned a_off0, a_off1 = kOob, tapmask0 = 0, tapmask1 = 0;   // (the second row of a staging thread: 128-row tiles only)
  if (conv) {
    auto place = [&](int m, unsigned &off, unsigned &mk) {
      off = kOob; mk = 0;
      if (m >= M) return;
      const int per = p.conv_hout * p.conv_wout;
      const int b = m / per, pix = m - b * per;
      const int yo = pix / p.conv_wout, xo = pix - yo * p.conv_wout;
      const int y = yo * p.conv_s, x = xo * p.conv_s;           // centre tap in the input image
      off = (unsigned)((((size_t)b * p.conv_hin + y) * p.conv_win + x) * cin + kce) * kAB;
      for (int t = 0; t < taps; ++t) {
        const int yy = y + t / p.conv_ks - pad, xx = x + t % p.conv_ks - pad;
        mk |= (yy >= 0 && yy < p.conv_hin && xx >= 0 && xx < p.conv_win) ? (1u << t) : 0u;
      }
    };
    place(m0 + r0, a_off0, tapmask0);
    place(m0 + r1, a_off1, tapmask1);
  } else {
    a_off0 = m0 + r0 < M ? (unsigned)(((size_t)(m0 + r0) * K + kce) * kAB) : kOob;
    a_off1 = m0 + r1 < M ? (unsigned)(((size_t)(m0 + r1) * K + kce) * kAB) : kOob;
  }
  int g_tap = 0, g_c = 0;                      // conv mode: the (tap, channel) position of the NEXT gload
  const unsigned w_off0 = n0 + r0 < N ? (unsigned)(((size_t)(n0 + r0) * K + kce) * kWB) : kOob;
  const unsigned w_off1 = (NI == 2 && n0 + r1 < N) ? (unsigned)(((size_t)(n0 + r1) * K + kce) * kWB) : kOob;
  // register staging: kDepth sets (set = step parity).  Two sets keep the loads of TWO steps in flight per
  // block (the step's latency is what bounds the long-K layers); the fused-quantise flavour stages twice the
  // activation bytes per step and stays at one set (168 VGPRs = three blocks per CU)
  constexpr int kDepth = MODE == kF16Q ? 1 : 2;
  uint4 ra0[kDepth][kAV * kHalves], ra1[kDepth][kAV * kHalves], rb0[kDepth][kHalves], rb1[kDepth][kHalves];
  auto bload = [](const __amdgpu_buffer_rsrc_t &rs, unsigned voff, int soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0));
  };
  auto gload = [&](int kt, auto setc) {
    constexpr int S = decltype(setc)::value;
    const int ks = kt * kStepK;
    const bool kok = ks + kce < K;   // K is a multiple of a thread's chunk (host check)
    if constexpr (CONV) {
      const int delta = ((g_tap / p.conv_ks - pad) * p.conv_win + (g_tap % p.conv_ks - pad)) * cin * kAB;
      const unsigned v0 = (tapmask0 >> g_tap) & 1u ? a_off0 + (unsigned)delta : kOob;
      const unsigned v1 = (tapmask1 >> g_tap) & 1u ? a_off1 + (unsigned)delta : kOob;
#pragma unroll
      for (int h = 0; h < kAV; ++h) {
        ra0[S][h] = bload(rs_a, v0 + 16u * h, g_c * kAB);
        ra1[S][h] = bload(rs_a, v1 + 16u * h, g_c * kAB);
      }
      g_c += kStepK;
      if (g_c >= cin) { g_c = 0; ++g_tap; }
    } else {
#pragma unroll
      for (int h = 0; h < kAV; ++h) {
        ra0[S][h] = bload(rs_a, kok ? a_off0 + 16u * h : kOob, ks * kAB);
        ra1[S][h] = bload(rs_a, kok ? a_off1 + 16u * h : kOob, ks * kAB);
      }
    }
    rb0[S][0] = bload(rs_w, kok ? w_off0 : kOob, ks * kWB);
    if constexpr (NI == 2) rb1[S][0] = bload(rs_w, kok ? w_off1 : kOob, ks * kWB);
  };
  const int lchunk = (tid & 3) * 16;           // byte position of the thread's chunk in an LDS row
  auto lstore = [&](int buf, auto setc) {
    constexpr int S = decltype(setc)::value;
    char *As = smem + buf * (TM + 128) * kTLd, *Ws = As + TM * kTLd;
    if constexpr (MODE == kF16Q) {
      *reinterpret_cast<uint4 *>(As + r0 * kTLd + lchunk) = quant16(ra0[S][0], ra0[S][1], p.inv_sa);
      *reinterpret_cast<uint4 *>(As + r1 * kTLd + lchunk) = quant16(ra1[S][0], ra1[S][1], p.inv_sa);
      *reinterpret_cast<uint4 *>(Ws + r0 * kTLd + lchunk) = rb0[S][0];
      if constexpr (NI == 2) *reinterpret_cast<uint4 *>(Ws + r1 * kTLd + lchunk) = rb1[S][0];
    } else {
#pragma unroll
      for (int hh = 0; hh < kHalves; ++hh) {
        *reinterpret_cast<uint4 *>(As + r0 * kTLd + 64 * hh + lchunk) = ra0[S][hh];
        *reinterpret_cast<uint4 *>(As + r1 * kTLd + 64 * hh + lchunk) = ra1[S][hh];
        *reinterpret_cast<uint4 *>(Ws + r0 * kTLd + 64 * hh + lchunk) = rb0[S][hh];
        if constexpr (NI == 2) *reinterpret_cast<uint4 *>(Ws + r1 * kTLd + 64 * hh + lchunk) = rb1[S][hh];
      }
    }
  };
  // ---- epilogue roles, fixed before the loop so that the identity rows can be requested early
  constexpr int kCH = 4 * NI;                    // 8-column chunks per row of the wave's 32 NI columns
  constexpr int kIT = kCH / 2;                   // read-back passes over the wave's 32 staged rows (64 / kCH rows each)
  const int c8 = lane & (kCH - 1);               // this lane's 8-column chunk
  const int ncol = n0 + wn * 32 * NI + c8 * 8;
  const bool col_ok = ncol < N;
  const bool vec = (N & 7) == 0;                 // rows 16-byte aligned and the chunk all in or all out; else per element
  const __half *res = static_cast<const __half *>(p.res);
  constexpr int kRB = RES8 ? 1 : 2;              // bytes per identity element
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__half *>(res), 0, res ? (unsigned)((size_t)M * N * kRB) : 0u, 0x00020000);
  uint4 rres[MJ][kIT];
  auto res_request = [&](int j) {
#pragma unroll
    for (int it = 0; it < kIT; ++it) {
      const int m = m0 + wm * 32 * MJ + j * 32 + it * (64 / kCH) + lane / kCH;
      const unsigned off = (m < M && col_ok) ? (unsigned)(((size_t)m * N + ncol) * kRB) : kOob;
      if constexpr (RES8) {   // 8 identity bytes of this lane's 8 columns
        const uint2 q = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs_r, (int)off, 0, 0));
        rres[j][it] = make_uint4(q.x, q.y, 0u, 0u);
      } else {
        rres[j][it] = bload(rs_r, off, 0);
      }
    }
  };
  const bool res_vec = res != nullptr && vec;

  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, kDepth - 1>;      // kDepth == 1: the one set
  // step s is staged in set s % kDepth; with two sets the loads of step s + 2 are issued as soon as set s % 2
  // has been written to LDS, i.e. two steps of multiply ahead of their use
  auto compute = [&](int kt) {
    const char *As = smem + (kt & 1) * (TM + 128) * kTLd, *Ws = As + TM * kTLd;
#pragma unroll
    for (int ks = 0; ks < KB / 32; ++ks) {
      const int kk = ks * 32 + (lane >> 5) * 16;
      // MFMA operand A = the weight rows (output columns n), B = the activation rows (m): a lane's 4
      // consecutive accumulator rows are then 4 consecutive n of one output row m
      i32x4_t a[NI], b[MJ];
#pragma unroll
      for (int i = 0; i < NI; ++i)
        a[i] = *reinterpret_cast<const i32x4_t *>(Ws + (wn * 32 * NI + i * 32 + (lane & 31)) * kTLd + kk);
#pragma unroll
      for (int j = 0; j < MJ; ++j)
        b[j] = *reinterpret_cast<const i32x4_t *>(As + (wm * 32 * MJ + j * 32 + (lane & 31)) * kTLd + kk);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          if constexpr (MODE == kF16)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a[i]),
                                                               __builtin_bit_cast(f16x8_t, b[j]), acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
  };
  // (loads past the last step are issued too -- beyond-the-buffer addresses, they read as zero -- and so are
  // the LDS writes of a step that does not exist, into the image nobody reads: without branches around them the
  // compiler's wait counts are exact, vmcnt(4..7) in front of the LDS writes instead of a drain to 0)
  gload(0, Set0{});
  lstore(0, Set0{});
  gload(1, Set1{});
  if constexpr (kDepth == 2) gload(2, Set0{});
  if (res_vec) res_request(0);
  __syncthreads();
  // one step: image (kt + 1) & 1 was last read in step kt - 1, which every wave left through the barrier
  auto step = [&](int kt, auto next_set) {
    lstore((kt + 1) & 1, next_set);           // step kt + 1 sits in set (kt + 1) % kDepth
    gload(kt + 1 + kDepth, next_set);         // ... which is free again: reload it
    compute(kt);
    __syncthreads();
  };
  if constexpr (kDepth == 2) {
    for (int kt = 0; kt < nk; kt += 2) {      // (an odd step count runs one step on zeros: adds nothing)
      step(kt, Set1{});
      step(kt + 1, Set0{});
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) step(kt, Set0{});
  }
  // ---- epilogue.  acc[i][j][4 g + c]: n = n0 + wn*32*NI + i*32 + 8 g + 4 (lane >> 5) + c, m = m0 + wm*32*MJ + j*32 + (lane & 31)
  // (the barrier that ended the last step also freed both LDS images)
  if (res_vec) res_request(1);
  char *stage = smem + wave * 32 * kEpiStride;
  float sc[8], bs[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const bool ok = ncol + c < N;
    if constexpr (MODE == kF16) {
      sc[c] = 1.f;
      bs[c] = (p.bias && ok) ? __half2float(static_cast<const __half *>(p.bias)[ncol + c]) : 0.f;
    } else {
      sc[c] = (p.wscale && ok) ? p.s_aw * p.wscale[ncol + c] : p.s_aw;
      bs[c] = (p.bias && ok) ? static_cast<const float *>(p.bias)[ncol + c] : 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<i32x4_t *>(stage + (lane & 31) * kEpiStride + (i * 32 + 8 * g + 4 * (lane >> 5)) * 4) =
            i32x4_t{sum_bits(acc[i][j][4 * g]), sum_bits(acc[i][j][4 * g + 1]), sum_bits(acc[i][j][4 * g + 2]),
                    sum_bits(acc[i][j][4 * g + 3])};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < kIT; ++it) {
      const int row = it * (64 / kCH) + lane / kCH;
      const int m = m0 + wm * 32 * MJ + j * 32 + row;
      const i32x4_t lo = *reinterpret_cast<const i32x4_t *>(stage + row * kEpiStride + c8 * 32);
      const i32x4_t hi = *reinterpret_cast<const i32x4_t *>(stage + row * kEpiStride + c8 * 32 + 16);
      if (m >= M || !col_ok) continue;
      float v[8];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if constexpr (MODE == kF16) {
          const int l = lo[c], h = hi[c];          // (element reads first: a bit cast of a vector ELEMENT expression reads element 0)
          v[c] = __int_as_float(l) + bs[c];
          v[4 + c] = __int_as_float(h) + bs[4 + c];
        } else {
          v[c] = (float)lo[c] * sc[c] + bs[c];
          v[4 + c] = (float)hi[c] * sc[4 + c] + bs[4 + c];
        }
      }
      if (res) {
        if constexpr (RES8) {
          if (vec) {
            const uint4 q = rres[j][it];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              v[c] += (float)(int)(signed char)((q.x >> (8 * c)) & 0xffu) * p.s_res;
              v[4 + c] += (float)(int)(signed char)((q.y >> (8 * c)) & 0xffu) * p.s_res;
            }
          } else {
#pragma unroll
            for (int c = 0; c < 8; ++c)
              if (ncol + c < N) v[c] += (float)reinterpret_cast<const int8_t *>(res)[(size_t)m * N + ncol + c] * p.s_res;
          }
        } else if (vec) {
          const uint4 q = rres[j][it];
          v[0] += h2f_lo(q.x); v[1] += h2f_hi(q.x); v[2] += h2f_lo(q.y); v[3] += h2f_hi(q.y);
          v[4] += h2f_lo(q.z); v[5] += h2f_hi(q.z); v[6] += h2f_lo(q.w); v[7] += h2f_hi(q.w);
        } else {
#pragma unroll
          for (int c = 0; c < 8; ++c)
            if (ncol + c < N) v[c] += __half2float(res[(size_t)m * N + ncol + c]);
        }
      }
      if (p.relu) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.f);
      }
      if constexpr (OUT8) {
        int8_t *o8 = static_cast<int8_t *>(p.out) + (size_t)m * N + ncol;
        unsigned pk[2] = {0, 0};
#pragma unroll
        for (int c = 0; c < 8; ++c)
          pk[c >> 2] |= ((unsigned)(int)fminf(fmaxf(rintf(v[c] * p.inv_s_out), -127.f), 127.f) & 0xffu) << (8 * (c & 3));
        if (vec) {
          *reinterpret_cast<uint2 *>(o8) = make_uint2(pk[0], pk[1]);
        } else {
#pragma unroll
          for (int c = 0; c < 8; ++c)
            if (ncol + c < N) o8[c] = (int8_t)((pk[c >> 2] >> (8 * (c & 3))) & 0xffu);
        }
      } else {
        __half *oh = static_cast<__half *>(p.out) + (size_t)m * N + ncol;
        if (vec) {
          uint4 o;
          o.x = pack_h2(v[0], v[1]); o.y = pack_h2(v[2], v[3]); o.z = pack_h2(v[4], v[5]); o.w = pack_h2(v[6], v[7]);
          *reinterpret_cast<uint4 *>(oh) = o;
        } else {
#pragma unroll
          for (int c = 0; c < 8; ++c)
            if (ncol + c < N) oh[c] = __float2half_rn(v[c]);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

struct ConvGeom { int cin = 0, hin = 0, win = 0, hout = 0, wout = 0, stride = 1, ks = 1; size_t in_elems = 0; };

template <int MODE>
int launch_tile_gemm(const void *a, float scale_a, const void *w, const float *w_scales, float scale_w,
                     const void *bias, const void *residual, int out_dtype, void *out, float scale_out, long long M,
                     int N, int K, int relu, void *stream, const ConvGeom &cg = ConvGeom(), int res_dtype = BEVOPS_F16,
                     float scale_res = 1.f) {
  if (!a || !w || !out || M < 0 || N <= 0 || K <= 0) return BEVOPS_BAD_PARAM;
  if (MODE != kF16 && (!(scale_a > 0.f) || (!w_scales && !(scale_w > 0.f)))) return BEVOPS_BAD_PARAM;
  constexpr int kChunk = MODE == kF16 ? 8 : 16;   // k-values a staging thread handles per step
  const bool res8 = residual != nullptr && res_dtype == BEVOPS_I8;
  if (residual && res_dtype != BEVOPS_I8 && res_dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (res8 && (MODE != kS8 || cg.cin > 0)) return BEVOPS_NOT_SUPPORTED;   // int8 identity rows: the int8-chain GEMM only
  if (res8 && !(scale_res > 0.f)) return BEVOPS_BAD_PARAM;
  if (K % kChunk != 0 || !aligned16(a) || !aligned16(w) ||
      (residual && (reinterpret_cast<uintptr_t>(residual) & (res8 ? 7u : 15u))) ||
      (reinterpret_cast<uintptr_t>(out) & (out_dtype == BEVOPS_I8 ? 7u : 15u)) || M > 0x7fffffffLL)
    return BEVOPS_NOT_SUPPORTED;
  if (out_dtype == BEVOPS_I8 && (MODE == kF16 || !(scale_out > 0.f))) return BEVOPS_BAD_PARAM;
  if (out_dtype != BEVOPS_I8 && out_dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  // operands (and the identity) are addressed through 32-bit buffer offsets
  const unsigned long long a_bytes = (cg.cin > 0 ? (unsigned long long)cg.in_elems : (unsigned long long)M * K) *
                                     (MODE == kS8 ? 1 : 2);
  if (a_bytes >= kOob || (unsigned long long)N * K * (MODE == kF16 ? 2 : 1) >= kOob ||
      (residual && (unsigned long long)M * N * (res8 ? 1 : 2) >= kOob))
    return BEVOPS_NOT_SUPPORTED;
  if (M == 0) return BEVOPS_SUCCESS;
  TileArgs p;
  p.a = a; p.w = w; p.bias = bias; p.res = residual; p.wscale = w_scales; p.out = out;
  p.inv_sa = MODE == kF16 ? 1.f : 1.0f / scale_a;
  p.s_aw = MODE == kF16 ? 1.f : (w_scales ? scale_a : scale_a * scale_w);
  p.inv_s_out = out_dtype == BEVOPS_I8 ? 1.0f / scale_out : 0.f;
  p.s_res = scale_res;
  p.M = (int)M; p.N = N; p.K = K; p.relu = relu;
  p.a_bytes = (unsigned)a_bytes;
  p.conv_cin = cg.cin; p.conv_hin = cg.hin; p.conv_win = cg.win; p.conv_hout = cg.hout; p.conv_wout = cg.wout;
  p.conv_s = cg.stride; p.conv_ks = cg.ks;
  const bool narrow = N <= 64;                    // 64-column tiles: no matrix work on columns that do not exist
  const int tn = narrow ? 64 : 128;
  p.tiles_n = (N + tn - 1) / tn;
  // 128-row tiles, 64-byte k-steps.  Two other builds were measured in round 4 and removed from the library in round
  // 5 (their template parameters with them): 64-row tiles (four blocks per CU: 3-10 % SLOWER on every
  // base-model layer and flavour, profiles/r04/tile_rows_ab.jsonl) and 128-byte k-steps for the int8 chain's plain
  // GEMMs (bit-identical, slower: a step costs 0.53 us whatever it holds, profiles/r04/tile_wide_ab.jsonl)
  const int tm = kTM;
  const long long tiles = (long long)p.tiles_n * ((M + tm - 1) / tm);
  if (tiles > 0x3fffffffLL) return BEVOPS_NOT_SUPPORTED;
  p.tiles_total = (int)tiles;
  const dim3 grid((unsigned)((tiles + 7) / 8 * 8));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool conv = cg.cin > 0, out8 = out_dtype == BEVOPS_I8;
#define BEVOPS_TG(OUT8_, CONV_, RES8_)                                                                                 \
  do {                                                                                                                 \
    if (narrow) hipLaunchKernelGGL((tile_gemm_kernel<MODE, OUT8_, 1, CONV_, RES8_>), grid, dim3(256), 0, st, p);      \
    else hipLaunchKernelGGL((tile_gemm_kernel<MODE, OUT8_, 2, CONV_, RES8_>), grid, dim3(256), 0, st, p);             \
    return launch_status();                                                                                            \
  } while (0)
  if constexpr (MODE == kF16) {
    if (conv) BEVOPS_TG(false, true, false);
    BEVOPS_TG(false, false, false);
  } else if constexpr (MODE == kF16Q) {
    if (conv && out8) return BEVOPS_NOT_SUPPORTED;
    if (conv) BEVOPS_TG(false, true, false);
    if (out8) BEVOPS_TG(true, false, false);
    BEVOPS_TG(false, false, false);
  } else {   // kS8: the int8 chain -- every combination of convolution mode, int8 output and int8 identity rows it uses
    if (conv && out8) BEVOPS_TG(true, true, false);
    if (conv) BEVOPS_TG(false, true, false);
    if (out8 && res8) BEVOPS_TG(true, false, true);
    if (out8) BEVOPS_TG(true, false, false);
    if (res8) BEVOPS_TG(false, false, true);
    BEVOPS_TG(false, false, false);
  }
#undef BEVOPS_TG
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" int bevops_linear_int8(const void *a_q, float scale_a, const void *w_q, const float *w_scales,
                                  float scale_w, const float *bias, const void *residual, int out_dtype,
                                  void *out, float scale_out, long long M, int N, int K, int relu, void *stream) {
  return launch_tile_gemm<kS8>(a_q, scale_a, w_q, w_scales, scale_w, bias, residual, out_dtype, out, scale_out, M, N, K,
                               relu, stream);
}

extern "C" int bevops_linear_int8_fused(const void *x_f16, float scale_a, const void *w_q, const float *w_scales,
                                        float scale_w, const float *bias, const void *residual, int out_dtype,
                                        void *out, float scale_out, long long M, int N, int K, int relu, void *stream) {
  return launch_tile_gemm<kF16Q>(x_f16, scale_a, w_q, w_scales, scale_w, bias, residual, out_dtype, out, scale_out, M, N,
                                 K, relu, stream);
}

extern "C" int bevops_tile_gemm_f16(const void *x, const void *weight, const void *bias, const void *residual,
                                    void *out, long long M, int N, int K, int relu, void *stream) {
  return launch_tile_gemm<kF16>(x, 1.f, weight, nullptr, 1.f, bias, residual, BEVOPS_F16, out, 1.f, M, N, K, relu, stream);
}

static int conv_geometry(ConvGeom &cg, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int step_k) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || stride <= 0) return BEVOPS_BAD_PARAM;
  if ((ksize != 1 && ksize != 3) || Cin % step_k != 0) return BEVOPS_NOT_SUPPORTED;   // a k-step must not straddle two taps
  cg.cin = Cin; cg.hin = H; cg.win = W; cg.stride = stride; cg.ks = ksize;
  const int pad = ksize / 2;
  cg.hout = (H + 2 * pad - ksize) / stride + 1;
  cg.wout = (W + 2 * pad - ksize) / stride + 1;
  cg.in_elems = (size_t)B * H * W * Cin;
  return BEVOPS_SUCCESS;
}

extern "C" int bevops_conv_tile_int8_fused(const void *x_f16, float scale_a, const void *w_q_taps, const float *w_scales,
                                           float scale_w, const float *bias, const void *residual, void *out, int B,
                                           int H, int W, int Cin, int Cout, int ksize, int stride, int relu,
                                           void *stream) {
  ConvGeom cg;
  const int rc = conv_geometry(cg, B, H, W, Cin, Cout, ksize, stride, 64);
  if (rc != BEVOPS_SUCCESS) return rc;
  return launch_tile_gemm<kF16Q>(x_f16, scale_a, w_q_taps, w_scales, scale_w, bias, residual, BEVOPS_F16, out, 1.f,
                                 (long long)B * cg.hout * cg.wout, Cout, ksize * ksize * Cin, relu, stream, cg);
}

extern "C" int bevops_conv_tile_f16(const void *x, const void *weight_taps, const void *bias, const void *residual,
                                    void *out, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int relu,
                                    void *stream) {
  ConvGeom cg;
  const int rc = conv_geometry(cg, B, H, W, Cin, Cout, ksize, stride, 32);
  if (rc != BEVOPS_SUCCESS) return rc;
  return launch_tile_gemm<kF16>(x, 1.f, weight_taps, nullptr, 1.f, bias, residual, BEVOPS_F16, out, 1.f,
                                (long long)B * cg.hout * cg.wout, Cout, ksize * ksize * Cin, relu, stream, cg);
}

extern "C" int bevops_linear_int8_chain(const void *a, int a_dtype, float scale_a, const void *w_q, const float *w_scales,
                                        float scale_w, const float *bias, const void *residual, int res_dtype,
                                        float scale_res, int out_dtype, void *out, float scale_out, long long M, int N,
                                        int K, int relu, void *stream) {
  if (a_dtype == BEVOPS_I8)
    return launch_tile_gemm<kS8>(a, scale_a, w_q, w_scales, scale_w, bias, residual, out_dtype, out, scale_out, M, N, K,
                                 relu, stream, ConvGeom(), res_dtype, scale_res);
  if (a_dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (residual && res_dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  return launch_tile_gemm<kF16Q>(a, scale_a, w_q, w_scales, scale_w, bias, residual, out_dtype, out, scale_out, M, N, K,
                                 relu, stream);
}

extern "C" int bevops_conv_tile_int8(const void *x_q, float scale_a, const void *w_q_taps, const float *w_scales,
                                     float scale_w, const float *bias, const void *residual, int out_dtype, void *out,
                                     float scale_out, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                                     int relu, void *stream) {
  ConvGeom cg;
  const int rc = conv_geometry(cg, B, H, W, Cin, Cout, ksize, stride, 64);
  if (rc != BEVOPS_SUCCESS) return rc;
  return launch_tile_gemm<kS8>(x_q, scale_a, w_q_taps, w_scales, scale_w, bias, residual, out_dtype, out, scale_out,
                               (long long)B * cg.hout * cg.wout, Cout, ksize * ksize * Cin, relu, stream, cg);
}
