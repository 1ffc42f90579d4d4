// conv3x3_c32: the DCNv2 pack's `conv_offset` (cnn/dcn.py:62-70: a plain 3x3 / stride 1 / pad 1
// convolution producing the 27 offset + mask channels that feed modulated_deformable_conv2d) on
// channels-last activations, fp16, with its bias in the epilogue.  The library convolution spends
// 48 us per stage-3 call on this 32-channel output (+ 10 us for the bias pass + 9 us for its own
// tensor set-up) -- as much as the deformable convolution it prepares.  As an implicit GEMM it is
// M = pixels, N = 32, K = 9 * Cin: N is exactly one 32-wide MFMA tile, and ALL weights of a
// 256-channel phase (9 * 256 * 32 * 2 B = 144 KiB) fit the CU's LDS.
//
// MI355X mapping: a block of 2 / 5 / 8 waves fills LDS once with the packed weights of a phase (straight
// 16-byte copies of a pre-swizzled image, conflict-free ds_read_b128 afterwards); each of its waves
// owns one tile of 32 consecutive pixels: accumulators = one v_mfma_f32_32x32x16_f16 tile (weights
// are the A operand, the image is the B operand, so a lane ends up with 4 x 4 consecutive output
// channels of one pixel = 8-byte stores into the [pixel][32] result).  The image operand never
// touches LDS: in NHWC the 16 k-values of a lane are 16 contiguous bytes of a pixel's channel row,
// fetched through a buffer descriptor (out-of-image taps read as zero by the range check, no
// branches), a ring of DEPTH (tap, 64-channel chunk) steps ahead of the MFMAs.
// Not a reference plugin by itself: part of ModulatedDeformConv2dPackPlugin.forward.
#include "common.h"

namespace bevops {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

int g_conv_variant = 0;                          // bevops_conv3x3_c32_set_variant
constexpr int kTile = 32;                       // pixels per wave; WPB waves (= tiles) per block
constexpr int kDepth = 4;                       // (tap, chunk) steps in flight per wave
constexpr unsigned kOob = 0xFFFFFF00u;          // beyond any buffer: reads as zero

// packed weight image, per phase of CP = 64 * CCP channels:
//   [tap 9][chunk CCP][j 4][hi 2][m 32][8 halves]   (16-byte groups; lane (m, hi) of k-step j reads group
//   ((tap * CCP + chunk) * 4 + j) * 64 + hi * 32 + m), holding channels chunk*64 + hi*32 + j*8 .. +7
__global__ __launch_bounds__(256) void pack_conv3x3_c32_kernel(const __half *__restrict__ w, __half *__restrict__ dst,
                                                                int cout, int Cin, int CP) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // one output half each
  const size_t total = (size_t)9 * Cin * 32;
  if (idx >= total) return;
  const int e = (int)(idx & 7);
  const int m = (int)((idx >> 3) & 31);
  const int hi = (int)((idx >> 8) & 1);
  const int j = (int)((idx >> 9) & 3);
  const size_t rest = idx >> 11;  // (phase * 9 + tap) * CCP + chunk
  const int CCP = CP / 64;
  const int chunk = (int)(rest % CCP);
  const int tap = (int)((rest / CCP) % 9);
  const int phase = (int)(rest / ((size_t)CCP * 9));
  const int c = phase * CP + chunk * 64 + hi * 32 + j * 8 + e;
  dst[idx] = m < cout ? w[((size_t)m * Cin + c) * 9 + tap] : __float2half(0.f);
}

// KS (round 5): K split across waves.  With one wave per 32-pixel tile a CU holds 2 .. 8 waves, each a chain of 9 * CCP
// dependent steps whose image loads come from the fabric (the activation was written by another XCD a moment ago):
// too few requests in flight, 29 us per stage-3 call against a ~10 us byte / matrix floor.  KS = 3 gives every tile
// three waves -- wave (tile, part) takes the taps of kernel ROW part (dy = part - 1; taps 3 part .. 3 part + 2)
// -- so three times the loads are in flight per CU; the partial tiles meet in LDS (the weight image is dead by then)
// and part 0 stores.  fp32 partial sums: the summation order differs from KS = 1 in the last bits only.
// 4 x 4 transpose across the lanes of a quad: in: r[i] of lane c = M[i][c]; out: r[i] of lane c = M[c][i].  Two stages
// of "send the element my partner needs, take its" (partner c ^ 2, then c ^ 1) through DPP quad permutes.
__device__ __forceinline__ void quad_transpose4(unsigned &r0, unsigned &r1, unsigned &r2, unsigned &r3, bool c2, bool c1) {
  auto xchg = [](unsigned &x, unsigned &y, bool upper, auto perm) __attribute__((always_inline)) {
    const unsigned send = upper ? x : y;         // lanes of the lower half hand over y, the upper ones x
    const unsigned recv = perm(send);
    if (upper) x = recv; else y = recv;
  };
  auto p2 = [](unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true); };   // quad_perm [2,3,0,1]
  auto p1 = [](unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); };   // quad_perm [1,0,3,2]
  xchg(r0, r2, c2, p2);
  xchg(r1, r3, c2, p2);
  xchg(r0, r1, c1, p1);
  xchg(r2, r3, c1, p1);
}

// KS == 3 also changes HOW the image operand is fetched (round 5, after counters: profiles/r05/conv_offset_pmc.txt).
// With lane = (k-half hi, pixel n) every 16-byte load of a wave instruction falls into another pixel row: 64 separate
// 64-byte sectors per instruction, 8.7 M L1 accesses per stage-3 call = 40 k cycles of tag look-ups per CU -- the
// whole 28 us; the matrix cores, the L2 and the fabric idle.  Now the four lanes of a quad (hi, a) read the four
// 16-byte slabs of ONE 64-byte sector: load i of a step fetches pixel 4 a + i (both hi halves: its whole 128-byte
// line), 16 sectors per instruction, and a 4 x 4 transpose across the quad's lanes (16 dwords, 48 VALU operations per
// step on an otherwise idle VALU) hands lane (hi, 4 a + b) the 64 bytes of pixel 4 a + b the MFMA operand layout wants.
template <int CCP, int WPB, int KS = 1>  // 64-channel chunks per phase (1, 2 or 4); tiles per block; waves per tile
__global__ __launch_bounds__(WPB * KS * 64) void conv3x3_c32_kernel(const __half *__restrict__ x,
                                                               const __half *__restrict__ wp,
                                                               const __half *__restrict__ bias,
                                                               __half *__restrict__ out, int B, int H, int W, int Cin,
                                                               int phases) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(KS == 1 || KS == 3, "one wave per tile, or one per kernel row");
  constexpr int kThreads = WPB * KS * 64, kTilesPerBlock = WPB;
  constexpr int kTaps = 9 / KS;             // taps this wave walks
  constexpr int kSteps = kTaps * CCP;       // ... and its (tap, chunk) steps per phase
  constexpr int kGroups = 9 * CCP * 4 * 64;  // 16-byte groups per phase (the whole weight image)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile_w = KS == 1 ? wave : wave % WPB, part = KS == 1 ? 0 : wave / WPB;
  const int n = lane & 31, hi = lane >> 5;
  const long npix = (long)B * H * W;
  const long pix = ((long)blockIdx.x * kTilesPerBlock + tile_w) * kTile + n;
  const bool live = pix < npix;
  int pb = 0, ph = 0, pw = 0;
  if (live) {
    pb = (int)(pix / ((long)H * W));
    const int r = (int)(pix - (long)pb * H * W);
    ph = r / W;
    pw = r - ph * W;
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__half *>(x), 0, (unsigned)((size_t)npix * Cin * 2), 0x00020000);
  // byte offset of (tap, channel 0) for this lane's pixel, or kOob
  // KS == 1: slot i = tap i of this lane's pixel n.  KS == 3: slot [i][k] = tap (row part - 1, column i - 1) of pixel
  // 4 a + k of the tile (a = (lane & 31) >> 2), this lane's 16-byte slab b = lane & 3 of the k-half hi
  constexpr int kSub = KS == 1 ? 1 : 4;
  unsigned toff[kTaps][kSub];
  if constexpr (KS == 1) {
#pragma unroll
    for (int i = 0; i < kTaps; ++i) {
      const int hh = ph + i / 3 - 1, ww = pw + i % 3 - 1;
      const bool ok = live && hh >= 0 && hh < H && ww >= 0 && ww < W;
      toff[i][0] = ok ? (unsigned)((((size_t)pb * H + hh) * W + ww) * Cin * 2) + (unsigned)(hi * 64) : kOob;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long pk = ((long)blockIdx.x * kTilesPerBlock + tile_w) * kTile + (n & ~3) + k;
      const bool lk = pk < npix;
      int kb = 0, kh = 0, kw = 0;
      if (lk) {
        kb = (int)(pk / ((long)H * W));
        const int r = (int)(pk - (long)kb * H * W);
        kh = r / W;
        kw = r - kh * W;
      }
#pragma unroll
      for (int i = 0; i < kTaps; ++i) {
        const int hh = kh + part - 1, ww = kw + i - 1;
        const bool ok = lk && hh >= 0 && hh < H && ww >= 0 && ww < W;
        toff[i][k] = ok ? (unsigned)((((size_t)kb * H + hh) * W + ww) * Cin * 2) + (unsigned)(hi * 64 + (n & 3) * 16) : kOob;
      }
    }
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int phase = 0; phase < phases; ++phase) {
    if (phase) __syncthreads();  // everyone is done reading the previous phase's weights
    {
      // all of a thread's 16-byte groups are requested before the first one is written: one round
      // trip to L2 for the whole 144 KiB image instead of one per group
      const uint4 *src = reinterpret_cast<const uint4 *>(wp) + (size_t)phase * kGroups;
      uint4 *dst = reinterpret_cast<uint4 *>(smem);
      constexpr int kIters = (kGroups + kThreads - 1) / kThreads;
      constexpr int kBatch = 16;  // groups requested together per thread (64 VGPRs)
#pragma unroll
      for (int k0 = 0; k0 < kIters; k0 += kBatch) {
        uint4 tmp[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
          const int i = tid + (k0 + k) * kThreads;
          tmp[k] = (k0 + k < kIters && i < kGroups) ? src[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
          const int i = tid + (k0 + k) * kThreads;
          if (k0 + k < kIters && i < kGroups) dst[i] = tmp[k];
        }
      }
    }
    __syncthreads();
    const unsigned cbase = (unsigned)(phase * CCP * 128);  // byte offset of the phase's first channel
    u32x4 ring[kDepth][4];
    auto issue = [&](int s, int slot) {
      // KS == 1: tap-major (s = tap * CCP + chunk).  KS == 3: this wave owns kernel ROW part; s = chunk * 3 + column,
      // the three columns of a chunk back to back -- they read the same lines shifted by a pixel, which are then L1 hits
      const int chunk = KS == 1 ? s % CCP : s / 3;
      const unsigned so = cbase + (unsigned)(chunk * 128);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned vo = KS == 1 ? toff[s / CCP][0] + 16u * j : toff[s % 3][KS == 1 ? 0 : j];
        ring[slot][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, (int)so, 0);
      }
    };
#pragma unroll
    for (int s = 0; s < kDepth && s < kSteps; ++s) issue(s, s);
    // keep the ring kDepth steps ahead: the scheduler would otherwise sink the loads next to their use
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      const int slot = s % kDepth;
      // weight groups of global step (tap, chunk): tap = s / CCP (KS == 1) or 3 part + s % 3, chunk s / 3
      const int gs = KS == 1 ? s : (3 * part + s % 3) * CCP + s / 3;
      const f16x8 *ag = reinterpret_cast<const f16x8 *>(smem) + (size_t)gs * 256 + hi * 32 + n;
      if constexpr (KS != 1) {   // ring[slot][k] = slab b of pixel 4 a + k  ->  slab k of pixel 4 a + b
        const bool c2 = (lane & 2) != 0, c1 = (lane & 1) != 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          unsigned r0 = ring[slot][0][d], r1 = ring[slot][1][d], r2 = ring[slot][2][d], r3 = ring[slot][3][d];
          quad_transpose4(r0, r1, r2, r3, c2, c1);
          ring[slot][0][d] = r0; ring[slot][1][d] = r1; ring[slot][2][d] = r2; ring[slot][3][d] = r3;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f16x8 a = ag[j * 64];
        const f16x8 b = __builtin_bit_cast(f16x8, ring[slot][j]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
      }
      if (s + kDepth < kSteps) issue(s + kDepth, slot);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (KS > 1) {
    // the row partials of a tile meet in LDS: [tile][part - 1][quad of accumulators 4][lane 64] float4 (a lane's
    // 16-byte pieces of one quad are consecutive across the wave: conflict-free), parts 1 .. KS - 1 write, part 0 adds
    __syncthreads();   // every wave is through with the weight image
    float4 *red = reinterpret_cast<float4 *>(smem) + (size_t)tile_w * (KS - 1) * 256;
    if (part > 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        red[((part - 1) * 4 + g) * 64 + lane] = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    }
    __syncthreads();
    if (part > 0) return;
#pragma unroll
    for (int q = 0; q < KS - 1; ++q)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = red[(q * 4 + g) * 64 + lane];
        acc[4 * g] += v.x; acc[4 * g + 1] += v.y; acc[4 * g + 2] += v.z; acc[4 * g + 3] += v.w;
      }
  }
  if (!live) return;
  // C/D layout: column (pixel) = lane & 31, row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  __half *op = out + (size_t)pix * 32;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c0 = 8 * g + 4 * hi;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = acc[4 * g + k] + (bias ? __half2float(bias[c0 + k]) : 0.f);
    u32x2 o;
    o.x = pack_h2(v[0], v[1]);
    o.y = pack_h2(v[2], v[3]);
    *reinterpret_cast<u32x2 *>(op + c0) = o;
  }
}

// ---- "rows" variant (bevops_conv3x3_c32_set_variant(1); A/B only, see the dispatch below) ----------
// The tile kernel pulls every tap of every pixel from the fabric: 9 x the image bytes, at the
// ~11 B/clk a CU gets for data another XCD wrote.  The 9 taps of a run of consecutive pixels
// [P0, P0 + 128) only touch the contiguous flattened range [P0 - W - 1, P0 + 128 + W + 1): this
// variant stages that range ONCE per 64-channel chunk in LDS (coalesced 128-byte rows, 16-byte
// chunks XOR-swizzled by the row so the fragment reads are conflict-free) next to the chunk's
// 36 KiB of weights, and both MFMA operands come from LDS.  <= 80 KiB per block at W <= 100, so two
// blocks share a CU and one computes while the other loads.
constexpr int kRowsWaves = 4;                      // 128 pixels per block
constexpr int kRowsThreads = kRowsWaves * 64;
constexpr int kWGroups = 9 * 4 * 64;               // weight groups (16 B) of one 64-channel chunk

__global__ __launch_bounds__(kRowsThreads) void conv3x3_c32_rows_kernel(const __half *__restrict__ x,
                                                                        const __half *__restrict__ wp,
                                                                        const __half *__restrict__ bias,
                                                                        __half *__restrict__ out, int B, int H, int W,
                                                                        int Cin, int CCP) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4 *wl = reinterpret_cast<uint4 *>(smem);                // [tap 9][j 4][hi 2][m 32] x 16 B
  char *al = smem + (size_t)kWGroups * 16;                     // (R + 1) rows x 128 B; row R stays zero
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, hi = lane >> 5;
  const int R = 32 * kRowsWaves + 2 * W + 2;
  const long npix = (long)B * H * W;
  const long P0 = (long)blockIdx.x * (32 * kRowsWaves);
  const long pix = P0 + wave * 32 + n;
  const bool live = pix < npix;
  int ph = 0, pw = 0;
  if (live) {
    const int r = (int)(pix % ((long)H * W));
    ph = r / W;
    pw = r - ph * W;
  }
  // LDS byte offset of the (row, 16-byte chunk 0) a tap reads for this lane, or the zero row
  unsigned trow[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    const bool ok = live && ph + dy >= 0 && ph + dy < H && pw + dx >= 0 && pw + dx < W;
    trow[t] = (unsigned)(ok ? wave * 32 + n + dy * W + dx + W + 1 : R);
  }
  if (tid < 8) reinterpret_cast<uint4 *>(al + (size_t)R * 128)[tid] = make_uint4(0, 0, 0, 0);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__half *>(x), 0, (unsigned)((size_t)npix * Cin * 2), 0x00020000);
  const unsigned row_bytes = (unsigned)Cin * 2u;
  const long first = P0 - W - 1;  // flattened pixel of staged row 0 (may be negative / beyond the end)

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int nchunks = Cin / 64;
  const int stage_items = R * 8;  // 16-byte pieces of the activation range
  for (int cc = 0; cc < nchunks; ++cc) {
    __syncthreads();  // the previous chunk's fragments have been read
    // weights of this chunk: 9 taps x 256 groups, strided by CCP * 256 in the packed image
    {
      const int phase = cc / CCP, chunk = cc - phase * CCP;
      const uint4 *src = reinterpret_cast<const uint4 *>(wp) + (size_t)phase * (9 * CCP * 256) + (size_t)chunk * 256;
#pragma unroll
      for (int k = 0; k < kWGroups / kRowsThreads; ++k) {  // 9 iterations of 256 threads
        const int i = tid + k * kRowsThreads;              // = tap * 256 + (i & 255)
        wl[i] = src[(size_t)(i >> 8) * (CCP * 256) + (i & 255)];
      }
    }
    // activation range of this chunk: piece e = row * 8 + chunk16; rows outside the tensor read as zero
    for (int e0 = 0; e0 < stage_items; e0 += kRowsThreads * 4) {
      u32x4 tmp[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = e0 + tid + k * kRowsThreads;
        const int row = e >> 3, c16 = e & 7;
        const long p = first + row;
        const bool ok = e < stage_items && p >= 0 && p < npix;
        const unsigned vo = ok ? (unsigned)((size_t)p * row_bytes) + (unsigned)(cc * 128 + c16 * 16) : kOob;
        tmp[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = e0 + tid + k * kRowsThreads;
        if (e < stage_items) {
          const int row = e >> 3, c16 = e & 7;
          *reinterpret_cast<u32x4 *>(al + (size_t)row * 128 + (size_t)((c16 ^ (row & 7)) * 16)) = tmp[k];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const unsigned row = trow[t];
      const char *rb = al + (size_t)row * 128;
      const unsigned sw = row & 7u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f16x8 a = reinterpret_cast<const f16x8 *>(wl)[t * 256 + j * 64 + hi * 32 + n];
        const f16x8 b = *reinterpret_cast<const f16x8 *>(rb + (((unsigned)(hi * 4 + j) ^ sw) * 16));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
      }
    }
  }
  if (!live) return;
  __half *op = out + (size_t)pix * 32;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c0 = 8 * g + 4 * hi;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = acc[4 * g + k] + (bias ? __half2float(bias[c0 + k]) : 0.f);
    u32x2 o;
    o.x = pack_h2(v[0], v[1]);
    o.y = pack_h2(v[2], v[3]);
    *reinterpret_cast<u32x2 *>(op + c0) = o;
  }
}

int launch_conv_rows(const __half *x, const __half *wp, const __half *bias, __half *out, int B, int H, int W,
                     int Cin, int CCP, hipStream_t st) {
  const size_t lds = (size_t)kWGroups * 16 + ((size_t)32 * kRowsWaves + 2 * W + 3) * 128;
  if (lds > 160 * 1024) return BEVOPS_NOT_SUPPORTED;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_c32_rows_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return BEVOPS_FAILURE;
  const long npix = (long)B * H * W;
  const long blocks = (npix + 32 * kRowsWaves - 1) / (32 * kRowsWaves);
  if (blocks > 0x7FFFFFFFL) return BEVOPS_NOT_SUPPORTED;
  hipLaunchKernelGGL(conv3x3_c32_rows_kernel, dim3((unsigned)blocks), dim3(kRowsThreads), lds, st, x, wp, bias, out,
                     B, H, W, Cin, CCP);
  return launch_status();
}

template <int CCP, int WPB, int KS = 1>
int launch_conv(const __half *x, const __half *wp, const __half *bias, __half *out, int B, int H, int W, int Cin,
                int phases, hipStream_t st) {
  size_t lds = (size_t)9 * CCP * 4 * 64 * 16;
  const size_t red = (size_t)WPB * (KS - 1) * 4096;   // the row partials reuse the (dead) weight image
  if (red > lds) lds = red;
  static bool ready = false;  // attribute set once per process (idempotent; benign if raced)
  if (!ready) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_c32_kernel<CCP, WPB, KS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return BEVOPS_FAILURE;
    ready = true;
  }
  const long npix = (long)B * H * W;
  const long blocks = (npix + kTile * WPB - 1) / (kTile * WPB);
  if (blocks > 0x7FFFFFFFL) return BEVOPS_NOT_SUPPORTED;
  hipLaunchKernelGGL((conv3x3_c32_kernel<CCP, WPB, KS>), dim3((unsigned)blocks), dim3(WPB * KS * 64), lds, st, x, wp,
                     bias, out, B, H, W, Cin, phases);
  return launch_status();
}

// ---- "resident" build (round 6; Cin == 256: the 23 + 3 offset convolutions of ResNet-101 stage 3 at base / small) -----
// The tile kernel above reads every pixel's 512-byte channel row nine times through L1 (once per tap); at ~11 bytes
// per clock and CU that is what its 18 us are.  Here the roles of the two operands are swapped:
//   * the WEIGHTS live in registers: wave `part` (0 .. 2) owns kernel row `part`; its 3 taps x 256 channels x 32 outputs are
//     48 MFMA A-operand fragments = 192 registers per lane, loaded once per (persistent) block from the same packed
//     image the tile kernel uses;
//   * the IMAGE goes through LDS: an 8 x 8-pixel output tile needs 10 x 10 input pixels (51 KB, 1.56 x the unique
//     bytes instead of 9 x), brought in with LDS-DMA (54 one-KiB pieces per tile, no registers; every wave issues a
//     share, the block's fourth wave does nothing else) into one of TWO buffers -- the pixels of tile t + 1 land
//     while tile t is multiplied.  Pixel rows are
//     padded to 528 bytes and tile rows to 5 504 (= 32 banks mod 64), which makes the 16-byte fragment reads of the
//     instruction's lane groups conflict-free;
//   * k order per wave (chunk, column, 16-channel step) and the combination of the three row partials ((p0 + p1) + p2,
//     then the bias, one rounding) are the tile kernel's: the results are bit-identical to it.
constexpr int kRT = 8, kRH = kRT + 2;
constexpr int kRPix = 256 * 2 + 16;                 // LDS bytes per staged pixel
constexpr int kRRow = kRH * kRPix + 224;            // ... per staged tile row: 5 504
constexpr int kRPieces = (kRH * kRRow + 1023) / 1024;   // 54 DMA pieces of 1 KiB
constexpr int kRBuf = kRPieces * 1024;              // 55 296
constexpr int kRRed = 3 * 2 * 4 * 64 * 16;          // row partials: [part][pixel block][quad][lane] x 16 B = 24 576
constexpr int kRLds = 2 * kRBuf + kRRed;            // 135 168

typedef __attribute__((address_space(3))) void lds_void_r;

__global__ __launch_bounds__(256) void conv3x3_c32_resident_kernel(const __half *__restrict__ x,
                                                                   const __half *__restrict__ wp,
                                                                   const __half *__restrict__ bias,
                                                                   __half *__restrict__ out, int H, int W, int tiles_x,
                                                                   int tiles_img, int tiles_total, unsigned x_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *Rs = smem + 2 * kRBuf;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hi = lane >> 5;
  const int g = (int)gridDim.x;
  int t = blockIdx.x;
  if (t >= tiles_total) return;
  auto tile_origin = [&](int tt, int &b, int &ty0, int &tx0) {
    b = tt / tiles_img;
    const int rem = tt - b * tiles_img;
    const int ty = rem / tiles_x;
    ty0 = ty * kRT;
    tx0 = (rem - ty * tiles_x) * kRT;
  };
  // LDS-DMA of a tile's 10 x 10 pixels: 54 one-KiB pieces.  ONE wave issuing all of them takes longer than the multiply
  // (an LDS-DMA instruction costs its wave 60-185 cycles at issue: 26.7 us per launch that way), so every wave issues a
  // share: the three multiplying waves 12 pieces each, spread between their matrix instructions, the fourth wave 18.
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half *>(x), 0, x_bytes, 0x00020000);
  auto dma_piece = [&](int b, int ty0, int tx0, int buf, int p) __attribute__((always_inline)) {
    // this lane's 16 bytes of piece p sit at byte d of the padded tile image: (row, pixel, channel byte) or padding
    const int d = p * 1024 + lane * 16;
    const int row = d / kRRow, rem = d - row * kRRow;
    const int px = rem / kRPix, c = rem - px * kRPix;
    const int y = ty0 + row - 1, xx = tx0 + px - 1;
    const bool ok = row < kRH && px < kRH && c < 512 && y >= 0 && y < H && xx >= 0 && xx < W;
    const unsigned off = ok ? (unsigned)((((size_t)b * H + y) * W + xx) * 512 + c) : kOob;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_r *)(smem + buf * kRBuf + p * 1024), 16, (int)off, 0, 0, 0);
  };
  constexpr int kMine = 12;      // pieces per multiplying wave (6 / 12 / 16 measured: 15.4 / 15.9 / 17.2 us at base, 12.1 / 11.8 / 12.9 at small); the fourth wave takes 36 .. 53
  {
    int b, ty0, tx0;
    tile_origin(t, b, ty0, tx0);
    if (wave == 3) {
#pragma unroll
      for (int p = 3 * kMine; p < kRPieces; ++p) dma_piece(b, ty0, tx0, 0, p);
    } else {
#pragma unroll
      for (int p = 0; p < kMine; ++p) dma_piece(b, ty0, tx0, 0, wave * kMine + p);
    }
  }
  if (wave == 3) {
    // ================= the fourth wave: its share of the next tile's pieces, nothing else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();      // P
    for (int it = 0; t < tiles_total; t += g, ++it) {
      if (t + g < tiles_total) {        // (that buffer was last read a tile ago: barrier Y)
        int b, ty0, tx0;
        tile_origin(t + g, b, ty0, tx0);
#pragma unroll
        for (int p = 3 * kMine; p < kRPieces; ++p) dma_piece(b, ty0, tx0, (it + 1) & 1, p);
      }
      __syncthreads();    // Y
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();    // X
    }
    return;
  }
  // ================= waves 0 .. 2: kernel row `part`, weights in registers
  const int part = wave;
  f16x8 wr[4][3][4];      // [chunk][column][16-channel step]
#pragma unroll
  for (int chunk = 0; chunk < 4; ++chunk)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        wr[chunk][dx][j] = reinterpret_cast<const f16x8 *>(wp)[((((3 * part + dx) * 4 + chunk) * 4 + j) * 64) + hi * 32 + n];
  float bcol[16];
  {
    u32x2 braw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) braw[q] = bias ? *reinterpret_cast<const u32x2 *>(bias + 8 * q + 4 * hi) : u32x2{0u, 0u};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bcol[4 * q] = h2f_lo(braw[q].x); bcol[4 * q + 1] = h2f_hi(braw[q].x);
      bcol[4 * q + 2] = h2f_lo(braw[q].y); bcol[4 * q + 3] = h2f_hi(braw[q].y);
    }
  }
  // fragment base of pixel block pb (tile rows 4 pb .. 4 pb + 3): lane n -> pixel (4 pb + (n >> 3), n & 7); the tap of
  // kernel row `part`, column dx reads staged pixel (row + part, column + dx)
  unsigned xb[2];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) xb[pb] = (unsigned)((4 * pb + (n >> 3) + part) * kRRow + (n & 7) * kRPix + hi * 64);
  float4 *red = reinterpret_cast<float4 *>(Rs);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();        // P
  for (int it = 0; t < tiles_total; t += g, ++it) {
    const char *Xs = smem + (it & 1) * kRBuf;
    const bool more = t + g < tiles_total;
    int nb = 0, ny0 = 0, nx0 = 0;
    if (more) tile_origin(t + g, nb, ny0, nx0);
    f32x16 acc[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pb][r] = 0.f;
    // 48 substeps (chunk, column, step), the two pixel blocks' fragments requested two substeps ahead
    f16x8 fb[3][2];
    auto fetch = [&](int sidx, int slot) __attribute__((always_inline)) {
      const int chunk = sidx / 12, dx = (sidx / 4) % 3, j = sidx & 3;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
        fb[slot][pb] = *reinterpret_cast<const f16x8 *>(Xs + xb[pb] + dx * kRPix + chunk * 128 + j * 16);
    };
    fetch(0, 0);
    fetch(1, 1);
#pragma unroll
    for (int sidx = 0; sidx < 48; ++sidx) {
      if (sidx + 2 < 48) fetch(sidx + 2, (sidx + 2) % 3);
      if (sidx % (48 / kMine) == 1 && sidx / (48 / kMine) < kMine && more) dma_piece(nb, ny0, nx0, (it + 1) & 1, part * kMine + sidx / (48 / kMine));
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
        acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[sidx / 12][(sidx / 4) % 3][sidx & 3], fb[sidx % 3][pb], acc[pb], 0, 0, 0);
      if (sidx + 2 < 48) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
    __syncthreads();      // Y: the staged pixels are free; the previous tile's partials have been read
    // wave 0 finishes pixel block 0, wave 1 pixel block 1: everybody hands over the blocks it does not finish
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
      if (part != pb) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          red[((part * 2 + pb) * 4 + q) * 64 + lane] = make_float4(acc[pb][4 * q], acc[pb][4 * q + 1], acc[pb][4 * q + 2], acc[pb][4 * q + 3]);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the next tile have landed
    __syncthreads();      // X
    if (part < 2) {
      const int pb = part;
      int b, ty0, tx0;
      tile_origin(t, b, ty0, tx0);
      const int y = ty0 + 4 * pb + (n >> 3), xx = tx0 + (n & 7);
      f32x16 a = acc[0];
      if (pb == 1) a = acc[1];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4] = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
        if (pb == 0) {          // (p0 + p1) + p2
          const float4 p1 = red[((1 * 2 + 0) * 4 + q) * 64 + lane], p2 = red[((2 * 2 + 0) * 4 + q) * 64 + lane];
          v[0] = (v[0] + p1.x) + p2.x; v[1] = (v[1] + p1.y) + p2.y; v[2] = (v[2] + p1.z) + p2.z; v[3] = (v[3] + p1.w) + p2.w;
        } else {
          const float4 p0 = red[((0 * 2 + 1) * 4 + q) * 64 + lane], p2 = red[((2 * 2 + 1) * 4 + q) * 64 + lane];
          v[0] = (p0.x + v[0]) + p2.x; v[1] = (p0.y + v[1]) + p2.y; v[2] = (p0.z + v[2]) + p2.z; v[3] = (p0.w + v[3]) + p2.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bcol[4 * q + e];
        if (y < H && xx < W) {
          u32x2 o;
          o.x = pack_h2(v[0], v[1]);
          o.y = pack_h2(v[2], v[3]);
          *reinterpret_cast<u32x2 *>(out + (((size_t)b * H + y) * W + xx) * 32 + 8 * q + 4 * hi) = o;
        }
      }
    }
  }
}

int launch_conv_resident(const __half *x, const __half *wp, const __half *bias, __half *out, int B, int H, int W,
                         hipStream_t st) {
  int cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return BEVOPS_FAILURE;
  const int tiles_x = (W + kRT - 1) / kRT, tiles_y = (H + kRT - 1) / kRT;
  const long total = (long)B * tiles_x * tiles_y;
  if (total > (1l << 30)) return BEVOPS_NOT_SUPPORTED;
  if (!ensure_dynamic_lds<conv3x3_c32_resident_kernel>(kRLds)) return BEVOPS_FAILURE;
  const int blocks = (int)(total < cus ? total : cus);
  hipLaunchKernelGGL(conv3x3_c32_resident_kernel, dim3((unsigned)blocks), dim3(256), kRLds, st, x, wp, bias, out, H, W,
                     tiles_x, tiles_x * tiles_y, (int)total, (unsigned)((size_t)B * H * W * 512));
  return launch_status();
}

// The image operand comes from the fabric (it was written by another XCD a moment ago) at ~11 B/clk
// per CU whatever the block does, so the launch is as fast as its busiest CU: the LDS image allows
// one block per CU, hence tiles per block = ceil(tiles / CUs), from {2, 5, 8} waves.
template <int CCP>
int launch_conv_any(const __half *x, const __half *wp, const __half *bias, __half *out, int B, int H, int W, int Cin,
                    int phases, hipStream_t st) {
  int cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return BEVOPS_FAILURE;
  const long tiles = ((long)B * H * W + kTile - 1) / kTile;
  const long need = (tiles + cus - 1) / cus;
  // three waves per tile (one per kernel row) wherever the block still fits 16 waves; variant 2 = one wave per tile
  const bool split = g_conv_variant != 2;     // (variant 3: this kernel with three waves per tile, as rounds 5's default)
  if (need <= 2) return split ? launch_conv<CCP, 2, 3>(x, wp, bias, out, B, H, W, Cin, phases, st)
                              : launch_conv<CCP, 2>(x, wp, bias, out, B, H, W, Cin, phases, st);
  if (need <= 5) return split ? launch_conv<CCP, 5, 3>(x, wp, bias, out, B, H, W, Cin, phases, st)
                              : launch_conv<CCP, 5>(x, wp, bias, out, B, H, W, Cin, phases, st);
  return launch_conv<CCP, 8>(x, wp, bias, out, B, H, W, Cin, phases, st);
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" size_t bevops_conv3x3_c32_packed_weight_size(int dtype, int Cin) {
  if (dtype != BEVOPS_F16 || Cin <= 0 || Cin % 64 != 0 || Cin == 192 || (Cin > 256 && Cin % 256 != 0)) return 0;
  return (size_t)9 * Cin * 32 * 2;
}

extern "C" int bevops_conv3x3_c32_pack_weight(int dtype, const void *weight, void *packed, int Cout, int Cin,
                                              void *stream) {
  if (!weight || !packed || Cout <= 0 || Cout > 32) return BEVOPS_BAD_PARAM;
  if (bevops_conv3x3_c32_packed_weight_size(dtype, Cin) == 0) return BEVOPS_NOT_SUPPORTED;
  const int CP = Cin < 256 ? Cin : 256;
  const size_t total = (size_t)9 * Cin * 32;
  hipLaunchKernelGGL(pack_conv3x3_c32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const __half *)weight, (__half *)packed, Cout, Cin, CP);
  return launch_status();
}

extern "C" int bevops_conv3x3_c32_forward_nhwc(int dtype, const void *input_nhwc, const void *packed_weight,
                                               const void *bias32, void *output_nhwc, int B, int H, int W, int Cin,
                                               void *stream) {
  if (!input_nhwc || !packed_weight || !output_nhwc || B <= 0 || H <= 0 || W <= 0) return BEVOPS_BAD_PARAM;
  if (bevops_conv3x3_c32_packed_weight_size(dtype, Cin) == 0) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(input_nhwc) || !aligned16(packed_weight) || !aligned16(output_nhwc)) return BEVOPS_BAD_PARAM;
  if ((size_t)B * H * W * Cin * 2 >= 0xFFFFFF00ull) return BEVOPS_NOT_SUPPORTED;  // 32-bit buffer offsets
  const int CP = Cin < 256 ? Cin : 256;
  const int phases = Cin / CP;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const __half *x = (const __half *)input_nhwc, *wp = (const __half *)packed_weight, *b = (const __half *)bias32;
  __half *o = (__half *)output_nhwc;
  // rows-in-LDS variant (variant 1): measured on hardware in round 2 -- 27.2 vs 29.6 us at the base stage-3
  // shape in isolation (profiles/r02/int8_dcn_time.jsonl), but inside the model, where the activation was
  // just written by the previous kernel, the frame is SLOWER with it (base 15.88 vs 15.46 ms, small 9.49 vs
  // 9.21 ms, profiles/r02/model_bench_conv_ab.jsonl): it stays an A/B switch, the tile kernel the default
  if (g_conv_variant == 1) {
    const int rc = launch_conv_rows(x, wp, b, o, B, H, W, Cin, CP / 64, st);
    if (rc != BEVOPS_NOT_SUPPORTED) return rc;
  }
  // round 6: weights in registers, image tile in LDS (Cin == 256; bit-identical to the tile kernel); variants 2 / 3 = the
  // tile kernel (one wave / three waves per tile)
  if (g_conv_variant == 0 && Cin == 256 && (reinterpret_cast<uintptr_t>(b) & 7u) == 0)
    return launch_conv_resident(x, wp, b, o, B, H, W, st);
  switch (CP / 64) {
    case 1: return launch_conv_any<1>(x, wp, b, o, B, H, W, Cin, phases, st);
    case 2: return launch_conv_any<2>(x, wp, b, o, B, H, W, Cin, phases, st);
    case 3: return BEVOPS_NOT_SUPPORTED;
    default: return launch_conv_any<4>(x, wp, b, o, B, H, W, Cin, phases, st);
  }
}

extern "C" int bevops_conv3x3_c32_set_variant(int variant) {
  const int prev = g_conv_variant;
  g_conv_variant = variant;
  return prev;
}
