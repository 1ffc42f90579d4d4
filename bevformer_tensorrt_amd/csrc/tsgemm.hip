// Tall-skinny fp16 GEMM on the matrix cores for the dense layers that wrap the samplers (SURVEY.md 8a-5:
// value_proj / output_proj / FFN of the encoder, the 1x1 convolutions of the channels-last backbone):
//     out[m, n] = act( sum_k x[m, k] * w[n, k] + bias[n] (+ residual[m, n]) ),   M >> N, K
// x [M, K] and w [N, K] row-major fp16 (nn.Linear / 1x1-conv weight layout), fp32 accumulation, ONE rounding
// to fp16.  These layers are memory-bound at batch 1 (the 1024 -> 256 convolution of ResNet stage 3 moves
// 89 MB for 18 GFLOP): what matters is that every CU streams its share of x exactly once, at full rate,
// with the weight slices coming from L2.  The library's tile shapes leave a third of the chip idle on
// M = 34 800 (272 tiles of 128 rows on 256 CUs -> two rounds); here
//   * the rows are split in units of 32 over ONE persistent block per CU (a CU gets 4 or 5 units at
//     M = 34 800; 85 % balance instead of 53 %), processed as tiles of up to 160 rows x 256 columns;
//   * both operands go global -> LDS by DMA (buffer_load ... lds, no VGPRs), 64 k-values per step, two
//     LDS stages, one barrier per step; LDS rows are 128 bytes with the 16-byte chunks XOR-swizzled (applied
//     to the DMA's SOURCE address, the DMA being lane-linear), so the ds_read_b128 fragment reads are
//     conflict-free (the scheme of the DCNv2 kernel, mdconv.hip);
//   * wave w owns columns 32 w .. 32 w + 31 of the tile for ALL its row units (v_mfma_f32_32x32x16_f16, up
//     to 5 accumulator tiles per wave): one weight fragment per k-substep is reused by 5 matrix
//     instructions;
//   * the epilogue goes through LDS in fp32 (two column halves): threads then own 16 contiguous bytes
//     of an output row, so bias / residual / ReLU are evaluated in fp32 on coalesced 16-byte accesses and
//     rounded once -- or, for the encoder's value projection, the row is written straight into the padded
//     head-major planes the SCA sampler reads (msda_pad.h: pixel-pair entries of the big levels, row-major
//     pixels of the staged ones), which removes the separate re-layout pass (70 us per SCA call).
// N > 256: grid.y walks the 256-column chunks.  Domain: K % 64 == 0, N % 256 == 0 (other layers stay on
// hipBLASLt).
#include <type_traits>

#include "common.h"
#include "mdconv.h"
#include "msda_pad.h"

namespace bevops {
namespace {

constexpr int kTsBN = 256;      // columns per tile
constexpr int kTsG = 5;         // row units (of 32) per tile
constexpr int kTsThreads = 512;
constexpr int kTsW = kTsBN * 128;          // weight stage image (32 KB)
constexpr int kTsX = kTsG * 32 * 128;      // activation stage image (20 KB)
constexpr int kTsStage = kTsW + kTsX;
constexpr int kTsEpiStride = 128 * 4 + 16; // fp32 staging row (128 columns) + bank pad
constexpr int kTsStages = 3;               // two k-steps of DMA in flight behind the one being multiplied
constexpr int kTsLds = kTsStages * kTsStage;   // 156 KB; the epilogue staging (160 x 528 = 82.5 KB) reuses it

struct TsPacked {   // EPI == 1: destination of the encoder's value projection
  char *gset, *sset;   // (EPI == 2: the LayerNorm's weight and bias, fp16 [N])
  Hm3Tab t;
  int nk, heads;    // rows per camera, heads (N == heads * 32)
  float eps;        // EPI == 2
};

// sum over the 16 lanes of a DPP row (= the 16 threads that share an output row in the epilogue), in every lane
__device__ __forceinline__ float row16_sum(float v) {
  v = quad_sum(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
}

typedef __attribute__((address_space(3))) void lds_void_t;

// EPI 0: out = act(acc + bias (+ residual)) -> [M, N] fp16.  EPI 1: acc + bias -> packed planes.
// EPI 2 (round 6, N == 256): out = LayerNorm(fp16(acc + bias + residual)) -- the layer the encoder / decoder blocks put
// behind output_proj and the FFN's second linear (modules/encoder.py:586-636: attention or FFN, then norm).  A thread
// owns the same 8 columns of the same rows in both 128-column halves of the epilogue, so the rounded sums of a row stay
// in registers (5 rows x 2 halves x 16 bytes) and its mean / variance are two 16-lane DPP reductions: the row is
// normalised from exactly the binary16 values the unfused pair (GEMM, then bevops_layer_norm) would have read back,
// two passes (mean, then centred squares), one rounding; no second launch, no 20 MB written and re-read.
template <int EPI>
__global__ __launch_bounds__(kTsThreads) void tsgemm_f16_kernel(const __half *__restrict__ x,
                                                                const __half *__restrict__ w,
                                                                const __half *__restrict__ bias,
                                                                const __half *__restrict__ res,
                                                                __half *__restrict__ out, int M, int N, int K,
                                                                int relu, int units_total, TsPacked pk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.y * kTsBN;
  // this block's row units: a contiguous range, the first (units_total % grid) blocks take one more
  const int nb = gridDim.x, bi = blockIdx.x;
  const int per = units_total / nb, extra = units_total % nb;
  const int u_begin = bi * per + min(bi, extra);
  const int u_end = u_begin + per + (bi < extra ? 1 : 0);
  if (u_begin >= u_end) return;

  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__half *>(x), 0, (unsigned)((size_t)M * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__half *>(w), 0, (unsigned)((size_t)N * K * 2), 0x00020000);
  // DMA roles.  Weight tile: 32 pieces of 8 rows, 4 per wave; activation tile: up to 20 pieces, piece
  // wave + 8 j.  lane -> (row in piece, 16-byte chunk); the swizzle sits on the source chunk
  const unsigned prow = (unsigned)(lane >> 3), pchunk = (unsigned)(lane & 7);
  unsigned w_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned row = (unsigned)((wave * 4 + j) * 8) + prow;
    w_off[j] = (unsigned)(((size_t)(n0 + row) * K) * 2) + ((pchunk ^ swz8(row)) << 4);
  }
  const unsigned hi = (unsigned)(lane >> 5);
  const unsigned fa = (unsigned)(wave * 32 + (lane & 31));   // weight fragment row
  const int nk = K / 64;

  // column constants of this lane: acc[g][4 rq + e] is column wave * 32 + 8 rq + 4 hi + e
  float bcol[16];
#pragma unroll
  for (int rq = 0; rq < 4; ++rq)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      bcol[4 * rq + e] = bias ? __half2float(bias[n0 + wave * 32 + 8 * rq + 4 * (int)hi + e]) : 0.f;

  for (int u0 = u_begin; u0 < u_end; u0 += kTsG) {
    const int G = min(kTsG, u_end - u0);
    const int r0 = u0 * 32;
    const int pieces_x = G * 4;
    unsigned x_off[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const unsigned row = (unsigned)((wave + 8 * j) * 8) + prow;
      x_off[j] = (unsigned)(((size_t)(r0 + row) * K) * 2) + ((pchunk ^ swz8(row)) << 4);
    }
    // every resident block walks the same weight matrix: start each one at a different k-slice (and wrap), so
    // that the CUs of an XCD do not pull the same 32 KB from the same L2 channels in the same step (the fp32
    // summation order then depends on the block index only)
    const int k_rot = bi % nk;
    auto dma = [&](int step, int buf) {
      char *wd = smem + buf * kTsStage + wave * 4096;
      int kstep = step + k_rot;
      if (kstep >= nk) kstep -= nk;
      const int soff = kstep * 128;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t *)(wd + j * 1024), 16, (int)w_off[j], soff, 0, 0);
      char *xd = smem + buf * kTsStage + kTsW;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (wave + 8 * j < pieces_x)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t *)(xd + (wave + 8 * j) * 1024), 16,
                                                   (int)x_off[j], soff, 0, 0);
    };
    f32x16 acc[kTsG];
#pragma unroll
    for (int g = 0; g < kTsG; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

    // three LDS stages: the DMA of steps s + 1 and s + 2 is in flight while step s is multiplied (x comes
    // from HBM: one step of cover leaves its latency exposed on every step).  A wave waits for ITS pieces
    // of step s (vmcnt counts its own DMA instructions: 4 weight pieces + 0..3 activation pieces per
    // step), the barrier then makes everybody's pieces visible -- and also says that stage (s + 2) % 3,
    // last read in step s - 1, is free again.
    const int my_dma = 4 + (wave < pieces_x ? 1 : 0) + (wave + 8 < pieces_x ? 1 : 0) + (wave + 16 < pieces_x ? 1 : 0);
    auto wait_keep_one_step = [&]() {
      switch (my_dma) {
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      }
    };
    // the k loop for a compile-time unit count (a run-time `g < G` inside the unrolled multiply loop is a
    // branch per matrix instruction: the fragment reads then wait out their LDS latency one by one)
    auto kloop = [&](auto gc) __attribute__((always_inline)) {
      constexpr int GG = decltype(gc)::value;
      dma(0, 0);
      if (nk > 1) dma(1, 1);
      for (int s = 0; s < nk; ++s) {
        const int buf = s % kTsStages;
        if (s + 1 < nk) wait_keep_one_step();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 2 < nk) dma(s + 2, (s + 2) % kTsStages);
        const char *Wb = smem + buf * kTsStage;
        const char *Xb = Wb + kTsW;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const unsigned c = 2u * ks + hi;
          const f16x8 a = *reinterpret_cast<const f16x8 *>(Wb + fa * 128 + ((c ^ swz8(fa)) << 4));
          f16x8 b[GG];
#pragma unroll
          for (int g = 0; g < GG; ++g) {
            const unsigned xr = (unsigned)(g * 32 + (lane & 31));
            b[g] = *reinterpret_cast<const f16x8 *>(Xb + xr * 128 + ((c ^ swz8(xr)) << 4));
          }
#pragma unroll
          for (int g = 0; g < GG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[g], acc[g], 0, 0, 0);
        }
      }
    };
    switch (G) {
      case 1: kloop(std::integral_constant<int, 1>{}); break;
      case 2: kloop(std::integral_constant<int, 2>{}); break;
      case 3: kloop(std::integral_constant<int, 3>{}); break;
      case 4: kloop(std::integral_constant<int, 4>{}); break;
      default: kloop(std::integral_constant<int, 5>{}); break;
    }
    __builtin_amdgcn_s_barrier();   // every wave is done with the stages: the epilogue reuses them
    // ---- epilogue through LDS (fp32), two halves of 128 columns: waves 0..3, then waves 4..7
    const int rows = min(G * 32, M - r0);
    uint4 keep[2][kTsG];          // EPI == 2: the rounded sums of this thread's (row, 8 columns), per half and row pass
    float rsum[kTsG];
    if constexpr (EPI == 2) {
#pragma unroll
      for (int i = 0; i < kTsG; ++i) rsum[i] = 0.f;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((wave >> 2) == half) {
        const int cw = (wave & 3) * 32;
#pragma unroll
        for (int g = 0; g < kTsG; ++g) {
          if (g < G) {
            char *rowp = smem + (g * 32 + (lane & 31)) * kTsEpiStride;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
              float4 v = make_float4(acc[g][4 * rq] + bcol[4 * rq], acc[g][4 * rq + 1] + bcol[4 * rq + 1],
                                     acc[g][4 * rq + 2] + bcol[4 * rq + 2], acc[g][4 * rq + 3] + bcol[4 * rq + 3]);
              *reinterpret_cast<float4 *>(rowp + (cw + 8 * rq + 4 * (int)hi) * 4) = v;
            }
          }
        }
      }
      __builtin_amdgcn_s_barrier();
      // thread -> (row, 8 columns): 16 chunks per row of 128 columns, 32 rows per pass
      auto row_pass = [&](int r, auto itc) __attribute__((always_inline)) {
        constexpr int it = decltype(itc)::value;
        (void)it;
        const int c8 = tid & 15;
        const float4 lo = *reinterpret_cast<const float4 *>(smem + r * kTsEpiStride + c8 * 32);
        const float4 hi4 = *reinterpret_cast<const float4 *>(smem + r * kTsEpiStride + c8 * 32 + 16);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
        const int col = n0 + half * 128 + c8 * 8;
        const size_t m = (size_t)(r0 + r);
        if constexpr (EPI == 2) {
          if (res) {
            const uint4 q = *reinterpret_cast<const uint4 *>(res + m * N + col);
            v[0] += h2f_lo(q.x); v[1] += h2f_hi(q.x); v[2] += h2f_lo(q.y); v[3] += h2f_hi(q.y);
            v[4] += h2f_lo(q.z); v[5] += h2f_hi(q.z); v[6] += h2f_lo(q.w); v[7] += h2f_hi(q.w);
          }
          uint4 o;
          o.x = pack_h2(v[0], v[1]); o.y = pack_h2(v[2], v[3]); o.z = pack_h2(v[4], v[5]); o.w = pack_h2(v[6], v[7]);
          keep[half][it] = o;
          rsum[it] += (h2f_lo(o.x) + h2f_hi(o.x)) + (h2f_lo(o.y) + h2f_hi(o.y)) + (h2f_lo(o.z) + h2f_hi(o.z)) +
                      (h2f_lo(o.w) + h2f_hi(o.w));
        } else if constexpr (EPI == 0) {
          if (res) {
            const uint4 q = *reinterpret_cast<const uint4 *>(res + m * N + col);
            v[0] += h2f_lo(q.x); v[1] += h2f_hi(q.x); v[2] += h2f_lo(q.y); v[3] += h2f_hi(q.y);
            v[4] += h2f_lo(q.z); v[5] += h2f_hi(q.z); v[6] += h2f_lo(q.w); v[7] += h2f_hi(q.w);
          }
          if (relu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
          }
          uint4 o;
          o.x = pack_h2(v[0], v[1]); o.y = pack_h2(v[2], v[3]); o.z = pack_h2(v[4], v[5]); o.w = pack_h2(v[6], v[7]);
          *reinterpret_cast<uint4 *>(out + m * N + col) = o;
        } else {
          // row m = pixel `s` of camera `cam`; column chunk -> (head, channels 8 q .. 8 q + 7)
          const int cam = (int)(m / (size_t)pk.nk), s = (int)(m - (size_t)cam * pk.nk);
          const int head = col >> 5, q8 = (col & 31) >> 3;
          int l = 0;
#pragma unroll
          for (int k = 1; k < kHm3MaxLevels; ++k)
            if (k < pk.t.L && s >= pk.t.src0[k]) l = k;
          const int rel = s - pk.t.src0[l];
          const int y = rel / pk.t.W[l], xx = rel - y * pk.t.W[l];
          const int wp = pk.t.W[l] + 1;
          const int f = pk.t.ent0[l] + (y + 1) * wp + xx;        // padded entry of this pixel
          const size_t plane = (size_t)cam * pk.heads + head;
          const unsigned h0 = pack_h2(v[0], v[1]), h1 = pack_h2(v[2], v[3]), h2 = pack_h2(v[4], v[5]), h3 = pack_h2(v[6], v[7]);
          if (l >= pk.t.ls) {   // staged level: row-major 64 B per pixel
            *reinterpret_cast<uint4 *>(pk.sset + (plane * pk.t.s_entries + f) * kLdsPixBytes + q8 * 16) =
                make_uint4(h0, h1, h2, h3);
          } else {
            // big level: entry f = (this pixel, right neighbour), entry f - 1 = (left neighbour, this pixel);
            // this thread writes the halves that hold ITS pixel: the .lo lanes of entry f and the .hi lanes
            // of entry f - 1 (2-byte lanes of 4-byte words: two 16-byte read-modify-free stores are not
            // possible, so the neighbour's values come from LDS)
            float nb[8];
            const bool has_r = xx + 1 < pk.t.W[l] && r + 1 < rows;
            const bool own_r = xx + 1 < pk.t.W[l];
            if (has_r) {
              const float4 a4 = *reinterpret_cast<const float4 *>(smem + (r + 1) * kTsEpiStride + c8 * 32);
              const float4 b4 = *reinterpret_cast<const float4 *>(smem + (r + 1) * kTsEpiStride + c8 * 32 + 16);
              nb[0] = a4.x; nb[1] = a4.y; nb[2] = a4.z; nb[3] = a4.w; nb[4] = b4.x; nb[5] = b4.y; nb[6] = b4.z; nb[7] = b4.w;
            } else {
#pragma unroll
              for (int k = 0; k < 8; ++k) nb[k] = 0.f;
            }
            char *e = pk.gset + (plane * pk.t.g_entries + f) * kEntBytes + q8 * 32;
            if (has_r || !own_r) {   // the whole entry is known here: (mine, right) or (mine, pad)
              uint4 o0, o1;
              o0.x = pack_h2(v[0], nb[0]); o0.y = pack_h2(v[1], nb[1]); o0.z = pack_h2(v[2], nb[2]); o0.w = pack_h2(v[3], nb[3]);
              o1.x = pack_h2(v[4], nb[4]); o1.y = pack_h2(v[5], nb[5]); o1.z = pack_h2(v[6], nb[6]); o1.w = pack_h2(v[7], nb[7]);
              *reinterpret_cast<uint4 *>(e) = o0;
              *reinterpret_cast<uint4 *>(e + 16) = o1;
            } else {                 // right neighbour lives in the next tile: only my halves (2-byte stores)
              unsigned short *e16 = reinterpret_cast<unsigned short *>(e);
              const unsigned hv[4] = {h0, h1, h2, h3};
#pragma unroll
              for (int k = 0; k < 4; ++k) { e16[4 * k] = (unsigned short)(hv[k] & 0xffffu); e16[4 * k + 2] = (unsigned short)(hv[k] >> 16); }
            }
            if (xx == 0) {           // entry f - 1 is the pad before the row: (0, mine)
              uint4 o0, o1;
              o0.x = h0 << 16; o0.y = h0 & 0xffff0000u; o0.z = h1 << 16; o0.w = h1 & 0xffff0000u;
              o1.x = h2 << 16; o1.y = h2 & 0xffff0000u; o1.z = h3 << 16; o1.w = h3 & 0xffff0000u;
              *reinterpret_cast<uint4 *>(e - kEntBytes) = o0;
              *reinterpret_cast<uint4 *>(e - kEntBytes + 16) = o1;
            } else if (r == 0) {     // left neighbour lives in the previous tile: my halves of entry f - 1
              unsigned short *e16 = reinterpret_cast<unsigned short *>(e - kEntBytes);
              const unsigned hv[4] = {h0, h1, h2, h3};
#pragma unroll
              for (int k = 0; k < 4; ++k) { e16[4 * k + 1] = (unsigned short)(hv[k] & 0xffffu); e16[4 * k + 3] = (unsigned short)(hv[k] >> 16); }
            }
          }
        }
      };
      if constexpr (EPI == 2) {   // compile-time pass index: the kept values live in registers
        const int rb = tid >> 4;
        if (rb < rows) row_pass(rb, std::integral_constant<int, 0>{});
        if (rb + 32 < rows) row_pass(rb + 32, std::integral_constant<int, 1>{});
        if (rb + 64 < rows) row_pass(rb + 64, std::integral_constant<int, 2>{});
        if (rb + 96 < rows) row_pass(rb + 96, std::integral_constant<int, 3>{});
        if (rb + 128 < rows) row_pass(rb + 128, std::integral_constant<int, 4>{});
      } else {
        for (int r = tid >> 4; r < rows; r += kTsThreads / 16) row_pass(r, std::integral_constant<int, 0>{});
      }
      __builtin_amdgcn_s_barrier();
    }
    if constexpr (EPI == 2) {
      const int c8 = tid & 15;
      const __half *gam = reinterpret_cast<const __half *>(pk.gset), *bet = reinterpret_cast<const __half *>(pk.sset);
      int it = 0;
#pragma unroll
      for (int r = tid >> 4; r < kTsG * 32; r += kTsThreads / 16, ++it) {
        if (r >= rows) break;
        const float mean = row16_sum(rsum[it]) * (1.f / 256.f);
        float f[16];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint4 o = keep[h][it];
          f[8 * h + 0] = h2f_lo(o.x) - mean; f[8 * h + 1] = h2f_hi(o.x) - mean;
          f[8 * h + 2] = h2f_lo(o.y) - mean; f[8 * h + 3] = h2f_hi(o.y) - mean;
          f[8 * h + 4] = h2f_lo(o.z) - mean; f[8 * h + 5] = h2f_hi(o.z) - mean;
          f[8 * h + 6] = h2f_lo(o.w) - mean; f[8 * h + 7] = h2f_hi(o.w) - mean;
        }
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sq = fmaf(f[k], f[k], sq);
        const float rstd = rsqrtf(row16_sum(sq) * (1.f / 256.f) + pk.eps);
        const size_t m = (size_t)(r0 + r);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int col = h * 128 + c8 * 8;
          const uint4 g4 = *reinterpret_cast<const uint4 *>(gam + col), b4 = *reinterpret_cast<const uint4 *>(bet + col);
          const float gg[8] = {h2f_lo(g4.x), h2f_hi(g4.x), h2f_lo(g4.y), h2f_hi(g4.y),
                               h2f_lo(g4.z), h2f_hi(g4.z), h2f_lo(g4.w), h2f_hi(g4.w)};
          const float bb[8] = {h2f_lo(b4.x), h2f_hi(b4.x), h2f_lo(b4.y), h2f_hi(b4.y),
                               h2f_lo(b4.z), h2f_hi(b4.z), h2f_lo(b4.w), h2f_hi(b4.w)};
          float y[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) y[k] = fmaf(f[8 * h + k] * rstd, gg[k], bb[k]);
          uint4 o;
          o.x = pack_h2(y[0], y[1]); o.y = pack_h2(y[2], y[3]); o.z = pack_h2(y[4], y[5]); o.w = pack_h2(y[6], y[7]);
          *reinterpret_cast<uint4 *>(out + m * N + col) = o;
        }
      }
    }
  }
}

// EPI 1: the entries of the padded sets that hold no pixel at all -- per plane the leading / trailing entry,
// the pad rows above and below each level, the pad pixel after each row of a staged level, and the pad
// after the LAST row of a big level (the pad after any other row of a big level is the entry
// (0, first pixel of the next row): the GEMM epilogue writes it).  grid (x, planes).
__global__ __launch_bounds__(256) void tsgemm_pad_zero_kernel(TsPacked pk, int planes) {
  const Hm3Tab &t = pk.t;
  const int plane = blockIdx.y;
  if (plane >= planes) return;
  for (int l = 0; l < t.L; ++l) {
    const bool staged = l >= t.ls;
    const int wp = t.W[l] + 1, H = t.H[l];
    const int ebytes = staged ? kLdsPixBytes : kEntBytes;
    const int chunks = ebytes / 16;
    char *base = (staged ? pk.sset + (size_t)plane * t.s_entries * kLdsPixBytes
                         : pk.gset + (size_t)plane * t.g_entries * kEntBytes);
    // pad entries of this level: row 0 (wp entries, but for a big level the LAST one pairs with row 1's
    // first pixel and is written by the epilogue), row H + 1 (wp), and column W of rows 1..H (big level:
    // rows 1..H-1 pair with the next row's first pixel -> epilogue; row H's pad -> zero here)
    const int total = 2 * wp + H;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total * chunks; i += gridDim.x * 256) {
      const int e = i / chunks, c = i - e * chunks;
      int yp, xx;
      if (e < wp) { yp = 0; xx = e; }
      else if (e < 2 * wp) { yp = H + 1; xx = e - wp; }
      else { yp = e - 2 * wp + 1; xx = t.W[l]; }
      if (!staged && xx == t.W[l] && yp < H) continue;   // (0, first pixel of row yp + 1): the epilogue's
      *reinterpret_cast<uint4 *>(base + (size_t)(t.ent0[l] + yp * wp + xx) * ebytes + c * 16) = make_uint4(0, 0, 0, 0);
    }
  }
  // leading entry 0 and the trailing entry of each set
  if (blockIdx.x == 0 && threadIdx.x < 16) {
    const int c = threadIdx.x & 7, which = threadIdx.x >> 3;
    char *g = pk.gset + (size_t)plane * t.g_entries * kEntBytes;
    *reinterpret_cast<uint4 *>(g + (which ? (size_t)(t.g_entries - 1) * kEntBytes : 0) + c * 16) = make_uint4(0, 0, 0, 0);
    if (t.s_entries && c < 4) {
      char *s = pk.sset + (size_t)plane * t.s_entries * kLdsPixBytes;
      *reinterpret_cast<uint4 *>(s + (which ? (size_t)(t.s_entries - 1) * kLdsPixBytes : 0) + c * 16) = make_uint4(0, 0, 0, 0);
    }
  }
}

// ---- the int8 activation chain's flavour (quantization.Int8ChainBackbone: the 1x1 convolutions of ResNet stages 3 / 4):
//     out = requant( act( (sum_k a[m, k] w[n, k]) * s_a * s_w[n] + bias[n] (+ identity[m, n]) ) )
// a [M, K], w [N, K] int8 row-major, int32 sums (exact), fp32 bias, identity rows int8 (with their own scale) or fp16,
// output int8 (requantised with the consumer's scale) or fp16.  The same persistent skeleton as tsgemm_f16_kernel --
// one block per CU, tiles of up to 160 rows x 256 columns, both operands global -> LDS by DMA in three stages, XOR
// swizzled 128-byte rows -- with 128 k-values per step instead of 64 (the LDS images, the DMA roles and the 16-byte
// fragment reads are byte-for-byte the fp16 kernel's: a v_mfma_i32_32x32x32_i8 operand is 16 consecutive bytes of k,
// lanes 32..63 the second 16 of a 32-value sub-step).  Measured (profiles/r04/tsgemm_s8_ab.jsonl): faster than the
// tiled int8 GEMM on the 256-column layers with K = 1 024 (stage-3 conv1: 21.5 vs 24.6 us), slower for N > 256 (every
// 256-column chunk re-reads the activation rows): functions/int8_chain.py picks it for the former only.
typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x16v __attribute__((ext_vector_type(16)));

struct TsS8Args {
  const int8_t *a, *w;
  const float *bias, *wscale;   // fp32 [N] or null
  const void *res;              // int8 or fp16 [M, N] or null
  void *out;                    // int8 or fp16 [M, N]
  float s_aw, s_res, inv_s_out;
  int M, N, K, relu, units_total, res_i8, out_i8;
};

__global__ __launch_bounds__(kTsThreads) void tsgemm_s8_kernel(const TsS8Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = p.M, N = p.N, K = p.K;
  const int n0 = blockIdx.y * kTsBN;
  const int nb = gridDim.x, bi = blockIdx.x;
  const int per = p.units_total / nb, extra = p.units_total % nb;
  const int u_begin = bi * per + min(bi, extra);
  const int u_end = u_begin + per + (bi < extra ? 1 : 0);
  if (u_begin >= u_end) return;
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(p.a), 0, (unsigned)((size_t)M * K), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(p.w), 0, (unsigned)((size_t)N * K), 0x00020000);
  const unsigned prow = (unsigned)(lane >> 3), pchunk = (unsigned)(lane & 7);
  unsigned w_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned row = (unsigned)((wave * 4 + j) * 8) + prow;
    w_off[j] = (unsigned)((size_t)(n0 + row) * K) + ((pchunk ^ swz8(row)) << 4);
  }
  const unsigned hi = (unsigned)(lane >> 5);
  const unsigned fa = (unsigned)(wave * 32 + (lane & 31));
  const int nk = K / 128;
  // column constants of this lane: acc[g][4 rq + e] is column wave * 32 + 8 rq + 4 hi + e
  float scol[16], bcol[16];
#pragma unroll
  for (int rq = 0; rq < 4; ++rq)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = n0 + wave * 32 + 8 * rq + 4 * (int)hi + e;
      scol[4 * rq + e] = p.wscale ? p.s_aw * p.wscale[col] : p.s_aw;
      bcol[4 * rq + e] = p.bias ? p.bias[col] : 0.f;
    }
  for (int u0 = u_begin; u0 < u_end; u0 += kTsG) {
    const int G = min(kTsG, u_end - u0);
    const int r0 = u0 * 32;
    const int pieces_x = G * 4;
    unsigned x_off[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const unsigned row = (unsigned)((wave + 8 * j) * 8) + prow;
      x_off[j] = (unsigned)((size_t)(r0 + row) * K) + ((pchunk ^ swz8(row)) << 4);
    }
    const int k_rot = bi % nk;   // (integer sums: the order of the k-slices does not change the result)
    auto dma = [&](int step, int buf) {
      char *wd = smem + buf * kTsStage + wave * 4096;
      int kstep = step + k_rot;
      if (kstep >= nk) kstep -= nk;
      const int soff = kstep * 128;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_t *)(wd + j * 1024), 16, (int)w_off[j], soff, 0, 0);
      char *xd = smem + buf * kTsStage + kTsW;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (wave + 8 * j < pieces_x)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_t *)(xd + (wave + 8 * j) * 1024), 16,
                                                   (int)x_off[j], soff, 0, 0);
    };
    i32x16v acc[kTsG];
#pragma unroll
    for (int g = 0; g < kTsG; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = 0;
    const int my_dma = 4 + (wave < pieces_x ? 1 : 0) + (wave + 8 < pieces_x ? 1 : 0) + (wave + 16 < pieces_x ? 1 : 0);
    auto wait_keep_one_step = [&]() {
      switch (my_dma) {
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      }
    };
    auto kloop = [&](auto gc) __attribute__((always_inline)) {
      constexpr int GG = decltype(gc)::value;
      dma(0, 0);
      if (nk > 1) dma(1, 1);
      for (int s = 0; s < nk; ++s) {
        const int buf = s % kTsStages;
        if (s + 1 < nk) wait_keep_one_step();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 2 < nk) dma(s + 2, (s + 2) % kTsStages);
        const char *Wb = smem + buf * kTsStage;
        const char *Xb = Wb + kTsW;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const unsigned c = 2u * ks + hi;
          const i32x4v a = *reinterpret_cast<const i32x4v *>(Wb + fa * 128 + ((c ^ swz8(fa)) << 4));
          i32x4v b[GG];
#pragma unroll
          for (int g = 0; g < GG; ++g) {
            const unsigned xr = (unsigned)(g * 32 + (lane & 31));
            b[g] = *reinterpret_cast<const i32x4v *>(Xb + xr * 128 + ((c ^ swz8(xr)) << 4));
          }
#pragma unroll
          for (int g = 0; g < GG; ++g) acc[g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[g], acc[g], 0, 0, 0);
        }
      }
    };
    switch (G) {
      case 1: kloop(std::integral_constant<int, 1>{}); break;
      case 2: kloop(std::integral_constant<int, 2>{}); break;
      case 3: kloop(std::integral_constant<int, 3>{}); break;
      case 4: kloop(std::integral_constant<int, 4>{}); break;
      default: kloop(std::integral_constant<int, 5>{}); break;
    }
    __builtin_amdgcn_s_barrier();
    // ---- epilogue through LDS (fp32: sums already scaled and shifted), two halves of 128 columns
    const int rows = min(G * 32, M - r0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((wave >> 2) == half) {
        const int cw = (wave & 3) * 32;
#pragma unroll
        for (int g = 0; g < kTsG; ++g) {
          if (g < G) {
            char *rowp = smem + (g * 32 + (lane & 31)) * kTsEpiStride;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
              float4 v = make_float4((float)acc[g][4 * rq] * scol[4 * rq] + bcol[4 * rq],
                                     (float)acc[g][4 * rq + 1] * scol[4 * rq + 1] + bcol[4 * rq + 1],
                                     (float)acc[g][4 * rq + 2] * scol[4 * rq + 2] + bcol[4 * rq + 2],
                                     (float)acc[g][4 * rq + 3] * scol[4 * rq + 3] + bcol[4 * rq + 3]);
              *reinterpret_cast<float4 *>(rowp + (cw + 8 * rq + 4 * (int)hi) * 4) = v;
            }
          }
        }
      }
      __builtin_amdgcn_s_barrier();
      for (int r = tid >> 4; r < rows; r += kTsThreads / 16) {
        const int c8 = tid & 15;
        const float4 lo = *reinterpret_cast<const float4 *>(smem + r * kTsEpiStride + c8 * 32);
        const float4 hi4 = *reinterpret_cast<const float4 *>(smem + r * kTsEpiStride + c8 * 32 + 16);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
        const int col = n0 + half * 128 + c8 * 8;
        const size_t m = (size_t)(r0 + r);
        if (p.res) {
          if (p.res_i8) {
            const uint2 q = *reinterpret_cast<const uint2 *>(static_cast<const int8_t *>(p.res) + m * N + col);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              v[c] += (float)(int)(signed char)((q.x >> (8 * c)) & 0xffu) * p.s_res;
              v[4 + c] += (float)(int)(signed char)((q.y >> (8 * c)) & 0xffu) * p.s_res;
            }
          } else {
            const uint4 q = *reinterpret_cast<const uint4 *>(static_cast<const __half *>(p.res) + m * N + col);
            v[0] += h2f_lo(q.x); v[1] += h2f_hi(q.x); v[2] += h2f_lo(q.y); v[3] += h2f_hi(q.y);
            v[4] += h2f_lo(q.z); v[5] += h2f_hi(q.z); v[6] += h2f_lo(q.w); v[7] += h2f_hi(q.w);
          }
        }
        if (p.relu) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        if (p.out_i8) {
          unsigned pk[2] = {0, 0};
#pragma unroll
          for (int c = 0; c < 8; ++c)
            pk[c >> 2] |= ((unsigned)(int)fminf(fmaxf(rintf(v[c] * p.inv_s_out), -127.f), 127.f) & 0xffu) << (8 * (c & 3));
          *reinterpret_cast<uint2 *>(static_cast<int8_t *>(p.out) + m * N + col) = make_uint2(pk[0], pk[1]);
        } else {
          uint4 o;
          o.x = pack_h2(v[0], v[1]); o.y = pack_h2(v[2], v[3]); o.z = pack_h2(v[4], v[5]); o.w = pack_h2(v[6], v[7]);
          *reinterpret_cast<uint4 *>(static_cast<__half *>(p.out) + m * N + col) = o;
        }
      }
      __builtin_amdgcn_s_barrier();
    }
  }
}

inline int ts_grid_x(int units, int chunks_n) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    hipDeviceProp_t p;
    static int cached[16] = {0};
    if (cached[dev & 15] == 0 && hipGetDeviceProperties(&p, dev) == hipSuccess) cached[dev & 15] = p.multiProcessorCount;
    if (cached[dev & 15] > 0) cus = cached[dev & 15];
  }
  (void)chunks_n;
  return units < cus ? units : cus;
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" int bevops_tsgemm_f16(const void *x, const void *weight, const void *bias, const void *residual,
                                 void *out, long long m, int n, int k, int relu, void *stream) {
  if (!x || !weight || !out || m <= 0 || n <= 0 || k <= 0) return BEVOPS_BAD_PARAM;
  if (k % 64 != 0 || n % kTsBN != 0) return BEVOPS_NOT_SUPPORTED;
  if ((double)m * k * 2 >= 4294967040.0 || (double)n * k * 2 >= 4294967040.0 || m > 0x7fffffff) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(x) || !aligned16(weight) || !aligned16(out) || (bias && !aligned16(bias)) ||
      (residual && !aligned16(residual)))
    return BEVOPS_BAD_PARAM;
  if (!ensure_dynamic_lds<tsgemm_f16_kernel<0>>(kTsLds)) return BEVOPS_FAILURE;
  const int units = (int)((m + 31) / 32);
  const dim3 grid((unsigned)ts_grid_x(units, n / kTsBN), (unsigned)(n / kTsBN));
  TsPacked none{};
  hipLaunchKernelGGL(tsgemm_f16_kernel<0>, grid, dim3(kTsThreads), kTsLds, static_cast<hipStream_t>(stream),
                     (const __half *)x, (const __half *)weight, (const __half *)bias, (const __half *)residual,
                     (__half *)out, (int)m, n, k, relu, units, none);
  return launch_status();
}

// out = LayerNorm_256(fp16(x @ weight.T + bias + residual)) * ln_weight + ln_bias in ONE launch (EPI 2 above).
// N must be 256 (the embedding width of the encoder / decoder blocks), K % 64 == 0; fp16 operands, fp32 statistics.
extern "C" int bevops_tsgemm_f16_ln(const void *x, const void *weight, const void *bias, const void *residual,
                                    const void *ln_weight, const void *ln_bias, float eps, void *out, long long m, int n,
                                    int k, void *stream) {
  if (!x || !weight || !out || !ln_weight || !ln_bias || m <= 0 || n <= 0 || k <= 0 || !(eps >= 0.f)) return BEVOPS_BAD_PARAM;
  if (k % 64 != 0 || n != kTsBN) return BEVOPS_NOT_SUPPORTED;
  if ((double)m * k * 2 >= 4294967040.0 || m > 0x7fffffff) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(x) || !aligned16(weight) || !aligned16(out) || (bias && !aligned16(bias)) ||
      (residual && !aligned16(residual)) || !aligned16(ln_weight) || !aligned16(ln_bias))
    return BEVOPS_BAD_PARAM;
  if (!ensure_dynamic_lds<tsgemm_f16_kernel<2>>(kTsLds)) return BEVOPS_FAILURE;
  const int units = (int)((m + 31) / 32);
  const dim3 grid((unsigned)ts_grid_x(units, 1), 1u);
  TsPacked pk{};
  pk.gset = const_cast<char *>(static_cast<const char *>(ln_weight));
  pk.sset = const_cast<char *>(static_cast<const char *>(ln_bias));
  pk.eps = eps;
  hipLaunchKernelGGL(tsgemm_f16_kernel<2>, grid, dim3(kTsThreads), kTsLds, static_cast<hipStream_t>(stream),
                     (const __half *)x, (const __half *)weight, (const __half *)bias, (const __half *)residual,
                     (__half *)out, (int)m, n, k, 0, units, pk);
  return launch_status();
}

// int8 chain flavour of the persistent tall-skinny GEMM (see tsgemm_s8_kernel).  a_q [M, K] / w_q [N, K] int8;
// w_scales fp32 [N] or NULL (then scale_w); bias fp32 [N] or NULL; residual [M, N] int8 (res_dtype BEVOPS_I8, real =
// q * scale_res) or fp16 or NULL; out [M, N] int8 (requantised with scale_out) or fp16.  Domain: K % 128 == 0,
// N % 256 == 0, 16-byte aligned operands (8-byte for the int8 identity / output rows).
extern "C" int bevops_tsgemm_s8(const void *a_q, float scale_a, const void *w_q, const float *w_scales, float scale_w,
                                const float *bias, const void *residual, int res_dtype, float scale_res, int out_dtype,
                                void *out, float scale_out, long long m, int n, int k, int relu, void *stream) {
  if (!a_q || !w_q || !out || m <= 0 || n <= 0 || k <= 0) return BEVOPS_BAD_PARAM;
  if (!(scale_a > 0.f) || (!w_scales && !(scale_w > 0.f))) return BEVOPS_BAD_PARAM;
  if (k % 128 != 0 || n % kTsBN != 0) return BEVOPS_NOT_SUPPORTED;
  if (out_dtype != BEVOPS_I8 && out_dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (residual && res_dtype != BEVOPS_I8 && res_dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (out_dtype == BEVOPS_I8 && !(scale_out > 0.f)) return BEVOPS_BAD_PARAM;
  if (residual && res_dtype == BEVOPS_I8 && !(scale_res > 0.f)) return BEVOPS_BAD_PARAM;
  if ((double)m * k >= 4294967040.0 || (double)n * k >= 4294967040.0 || m > 0x7fffffff) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(a_q) || !aligned16(w_q) || (reinterpret_cast<uintptr_t>(out) & (out_dtype == BEVOPS_I8 ? 7u : 15u)) ||
      (residual && (reinterpret_cast<uintptr_t>(residual) & (res_dtype == BEVOPS_I8 ? 7u : 15u))) ||
      (bias && (reinterpret_cast<uintptr_t>(bias) & 3u)) || (w_scales && (reinterpret_cast<uintptr_t>(w_scales) & 3u)))
    return BEVOPS_BAD_PARAM;
  if (!ensure_dynamic_lds<tsgemm_s8_kernel>(kTsLds)) return BEVOPS_FAILURE;
  const int units = (int)((m + 31) / 32);
  const dim3 grid((unsigned)ts_grid_x(units, n / kTsBN), (unsigned)(n / kTsBN));
  TsS8Args p;
  p.a = static_cast<const int8_t *>(a_q); p.w = static_cast<const int8_t *>(w_q);
  p.bias = bias; p.wscale = w_scales; p.res = residual; p.out = out;
  p.s_aw = w_scales ? scale_a : scale_a * scale_w;
  p.s_res = scale_res;
  p.inv_s_out = out_dtype == BEVOPS_I8 ? 1.0f / scale_out : 0.f;
  p.M = (int)m; p.N = n; p.K = k; p.relu = relu; p.units_total = units;
  p.res_i8 = residual && res_dtype == BEVOPS_I8 ? 1 : 0;
  p.out_i8 = out_dtype == BEVOPS_I8 ? 1 : 0;
  hipLaunchKernelGGL(tsgemm_s8_kernel, grid, dim3(kTsThreads), kTsLds, static_cast<hipStream_t>(stream), p);
  return launch_status();
}

// The encoder's value projection (spatial_cross_attention.py:754: value = self.value_proj(value)) written
// straight into the padded head-major planes of the SCA sampler: packed = [big set][staged set][visibility
// bytes], the layout bevops_msda_forward_prepacked consumes for the 4-level x 8-point shape.
extern "C" size_t bevops_value_proj_packed_size(const int32_t *spatial_shapes_host, int num_cams, int nk, int heads,
                                                int channels, int num_levels, int num_query, int num_point) {
  if (!spatial_shapes_host || num_cams <= 0 || nk <= 0) return 0;
  return msda_hm5_workspace_bytes(spatial_shapes_host, num_cams, heads, channels, num_levels, num_query, num_point);
}

extern "C" int bevops_value_proj_packed(const void *x, const void *weight, const void *bias,
                                        const int32_t *spatial_shapes_host, void *packed, size_t packed_bytes,
                                        int num_cams, int nk, int heads, int channels, int num_levels,
                                        int num_query, int num_point, void *stream) {
  if (!x || !weight || !spatial_shapes_host || !packed) return BEVOPS_BAD_PARAM;
  if (num_cams <= 0 || nk <= 0 || heads <= 0 || channels != 32 || (heads * channels) % kTsBN != 0) return BEVOPS_NOT_SUPPORTED;
  long total = 0;
  for (int l = 0; l < num_levels; ++l) total += (long)spatial_shapes_host[2 * l] * spatial_shapes_host[2 * l + 1];
  if (total != nk) return BEVOPS_BAD_PARAM;
  TsPacked pk{};
  size_t g_room = 0, s_bytes = 0;
  if (!msda_hm5_layout(spatial_shapes_host, num_cams, heads, channels, num_levels, num_query, num_point, &pk.t, &g_room,
                       &s_bytes))
    return BEVOPS_NOT_SUPPORTED;
  if (packed_bytes < g_room + s_bytes || (reinterpret_cast<uintptr_t>(packed) & 127u)) return BEVOPS_BAD_PARAM;
  const int n = heads * channels, k = n;   // embed -> embed
  const long long m = (long long)num_cams * nk;
  if ((double)m * k * 2 >= 4294967040.0) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(x) || !aligned16(weight) || (bias && !aligned16(bias))) return BEVOPS_BAD_PARAM;
  pk.gset = static_cast<char *>(packed);
  pk.sset = pk.gset + g_room;
  pk.nk = nk;
  pk.heads = heads;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(tsgemm_pad_zero_kernel, dim3(4, (unsigned)(num_cams * heads)), dim3(256), 0, st, pk, num_cams * heads);
  if (!ensure_dynamic_lds<tsgemm_f16_kernel<1>>(kTsLds)) return BEVOPS_FAILURE;
  const int units = (int)((m + 31) / 32);
  const dim3 grid((unsigned)ts_grid_x(units, n / kTsBN), (unsigned)(n / kTsBN));
  hipLaunchKernelGGL(tsgemm_f16_kernel<1>, grid, dim3(kTsThreads), kTsLds, st, (const __half *)x, (const __half *)weight,
                     (const __half *)bias, (const __half *)nullptr, (__half *)nullptr, (int)m, n, k, 0, units, pk);
  return launch_status();
}

// The same planes from an already projected value tensor [num_cams, nk, heads, 32] (the re-layout pass of the
// drop-in call as an entry of its own): what bevops_value_proj_packed must reproduce byte for byte when
// `value` is its own GEMM's output.
extern "C" int bevops_value_pack_planes(const void *value, const int32_t *spatial_shapes_host, void *packed,
                                        size_t packed_bytes, int num_cams, int nk, int heads, int channels,
                                        int num_levels, int num_query, int num_point, void *stream) {
  if (!value || !spatial_shapes_host || !packed) return BEVOPS_BAD_PARAM;
  TsPacked pk{};
  size_t g_room = 0, s_bytes = 0;
  if (!msda_hm5_layout(spatial_shapes_host, num_cams, heads, channels, num_levels, num_query, num_point, &pk.t, &g_room,
                       &s_bytes))
    return BEVOPS_NOT_SUPPORTED;
  if (packed_bytes < g_room + s_bytes || (reinterpret_cast<uintptr_t>(packed) & 127u)) return BEVOPS_BAD_PARAM;
  msda_hm3_repack_launch(value, static_cast<char *>(packed), static_cast<char *>(packed) + g_room, &pk.t, num_cams, nk,
                         heads, static_cast<hipStream_t>(stream));
  return launch_status();
}
