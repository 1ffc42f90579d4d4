// stem7x7_pool: the ResNet stem of the re-hosted backbone -- conv 7x7 / stride 2 / pad 3 (3 -> 64 channels, folded BN
// shift) -> ReLU -> max_pool 3x3 / stride 2 / pad 1 (backbones/resnet.py: conv1 -> norm1 -> relu -> maxpool) -- as ONE
// kernel from the planar camera images to the pooled channels-last activation.  The library form writes the
// convolution's [6, 464, 800, 64] fp16 output (285 MB), reads it back for the pooling pass and needs the images
// channels-last first: 0.25 ms + 2 x 0.04 ms of tensor set-up + 0.10 ms pooling + a 53 MB re-layout per frame.  Here
// the convolution output never leaves the registers.
//
// MI355X mapping.  Implicit GEMM with the WEIGHTS as the A operand (64 channels = two 32-row tiles, all 11 k-steps of
// both tiles live in 88 registers of every wave for the whole kernel) and 32 consecutive convolution pixels of one
// output row as the B operand of v_mfma_f32_32x32x16_f16.  K is ordered (ky, c, kx'): a lane's 8 k-values are the 8
// consecutive input halves x = 2 ox - 4 .. 2 ox + 3 of one (ky, c) image row -- kx' = 0 is a zero-weight slot that
// makes the run start on an even x (a dword of the planar image and of its LDS copy), kx' = 1 .. 7 are the 7 taps --
// so a k-step (16 k-values = the two lane halves) covers two (ky, c) rows, 21 rows = 10.5 k-steps, and the spare row
// of k-step 10 carries the BIAS: its B operand reads a row of ones, its weight slot 0 holds the shift.  A block of four
// waves stages the (4 R + 7) x 3 input rows of its R = 4 pooled rows x 60 pooled columns once (LDS rows of 160
// dwords: the two (ky, c) rows of a k-step sit 32 banks apart, and a half-wave's 32 pixels read consecutive dwords --
// conflict-free), then every wave walks its own strip of 15 pooled columns down the 2 R + 1 convolution rows: 22
// MFMAs per row, running maximum over the three rows of a pooling window in fp32 registers (the lane keeps the same
// (pixel, channel) elements in every row), the two x-neighbours of a window through two wave-wide DPP shifts, ReLU,
// one rounding, 16 consecutive channels = 32 bytes per lane and tile stored straight into the [pixel][64] result.
// Not a reference plugin: part of the re-hosted backbone.
#include "common.h"

namespace bevops {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) unsigned lds_u32;

constexpr int kR = 4;                          // pooled rows per block
constexpr int kStrip = 15;                     // pooled columns per wave (30 + 2 convolution columns)
constexpr int kWaves = 4;                      // strips per block
constexpr int kConvRows = 2 * kR + 1;          // convolution rows a block walks
constexpr int kInRows = 2 * (kConvRows - 1) + 7;   // input rows under them
constexpr int kRows = kInRows * 3;             // LDS rows [input row][channel]; one more row of ones follows
constexpr int kRowDw = 160;                    // dwords per LDS row (>= 125 used; 160 = 32 mod 64 banks)
constexpr int kUsedDw = 2 * kWaves * kStrip + 5;   // 125: dwords a block's strips read of a row
constexpr int kSteps = 11;                     // k-steps of 16
constexpr int kLdsBytes = (kRows + 1) * kRowDw * 4;
constexpr size_t kPackedHalves = (size_t)kSteps * 2 * 64 * 8;

// packed A operand: [k-step 11][channel tile 2][lane 64][8 halves].  Lane (m, hi) of tile t holds, for image row
// r = 2 s + hi = 3 ky + c, the slots kx' = 0 .. 7 (0: zero, 1 .. 7: taps kx = 0 .. 6) of output channel
// co = 32 t + 16 ((m >> 2) & 1) + 4 (m >> 3) + (m & 3) -- the row order that leaves lane (pixel, hi) of the result
// with the 16 CONSECUTIVE channels 32 t + 16 hi .. + 15; r = 21 (the spare half of the last k-step): slot 0 = bias.
__global__ __launch_bounds__(256) void stem_pack_kernel(const __half *__restrict__ w, const __half *__restrict__ bias,
                                                        __half *__restrict__ dst) {
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= (unsigned)kPackedHalves) return;
  const int e = (int)(idx & 7u), lane = (int)((idx >> 3) & 63u), t = (int)((idx >> 9) & 1u), s = (int)(idx >> 10);
  const int m = lane & 31, hi = lane >> 5;
  const int co = 32 * t + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);
  const int r = 2 * s + hi;
  __half v = __float2half(0.f);
  if (r < 21) {
    const int ky = r / 3, c = r % 3;
    if (e >= 1) v = w[((co * 3 + c) * 7 + ky) * 7 + (e - 1)];
  } else if (e == 0 && bias) {
    v = bias[co];
  }
  dst[idx] = v;
}

// lane i takes lane i + 1's value (the last lane of the wave: unspecified, never used)
template <int SHUF>
__device__ __forceinline__ float next_lane(float v) {
  if constexpr (SHUF == 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));  // wave_shl:1
  else
    return __shfl_down(v, 1, 64);
}

template <bool OUT8, int SHUF>
__global__ __launch_bounds__(256, 2) void stem7x7_pool_kernel(const __half *__restrict__ x,
                                                              const __half *__restrict__ wp, void *__restrict__ out_,
                                                              int B, int H, int W, int Hc, int Wc, int Hp, int Wp,
                                                              int nbx, int nby, float inv_s_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned *lds = reinterpret_cast<unsigned *>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bx = (int)(blockIdx.x % (unsigned)nbx);
  const int by = (int)((blockIdx.x / (unsigned)nbx) % (unsigned)nby);
  const int b = (int)(blockIdx.x / ((unsigned)nbx * (unsigned)nby));
  const int PX0 = bx * (kWaves * kStrip), PY0 = by * kR;
  const int IXE = 4 * PX0 - 6, IY0 = 4 * PY0 - 5;   // input (x, y) of LDS (dword 0, row 0)

  // ---- the A operand first (its 22 loads fly while the image rows are staged), then stage the input rows (zero
  // outside the image = the convolution's padding) -- all of a wave's loads are issued before the first LDS write
  f16x8 wa[kSteps][2];
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    wa[s][0] = *reinterpret_cast<const f16x8 *>(wp + ((size_t)(s * 2 + 0) * 64 + lane) * 8);
    wa[s][1] = *reinterpret_cast<const f16x8 *>(wp + ((size_t)(s * 2 + 1) * 64 + lane) * 8);
  }
  constexpr int kRowsPerWave = (kRows + kWaves - 1) / kWaves;
  unsigned stage[kRowsPerWave][2];
#pragma unroll
  for (int i = 0; i < kRowsPerWave; ++i) {
    const int row = wave + kWaves * i;
    const int ly = row / 3, c = row - ly * 3;
    const int y = IY0 + ly;
    const bool row_ok = row < kRows && y >= 0 && y < H;
    const __half *src = x + (((size_t)b * 3 + c) * H + (row_ok ? y : 0)) * W;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int xg = IXE + 2 * (lane + 64 * k);
      stage[i][k] = 0u;
      if (row_ok && xg >= 0 && xg < W) stage[i][k] = *reinterpret_cast<const unsigned *>(src + xg);
    }
  }
#pragma unroll
  for (int i = 0; i < kRowsPerWave; ++i) {
    const int row = wave + kWaves * i;
    if (row < kRows) {
      lds[row * kRowDw + lane] = stage[i][0];
      if (lane + 64 < kUsedDw) lds[row * kRowDw + lane + 64] = stage[i][1];
    }
  }
  if (tid < kUsedDw) lds[kRows * kRowDw + tid] = 0x3c003c00u;   // binary16 1.0 | 1.0
  __syncthreads();

  const int n = lane & 31, hi = lane >> 5;
  const int PXs = PX0 + wave * kStrip;            // first pooled column of this wave's strip
  if (PXs >= Wp) return;                          // (after the only barrier)
  const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
  const unsigned col = (unsigned)(2 * kStrip * wave + n) * 4u;                 // byte offset of the lane's first dword
  const unsigned ones_a = lbase + (unsigned)(kRows * kRowDw) * 4u + col;
  // convolution column of this lane, for the pooling pad: cx in [0, Wc) or excluded
  const int cx = 2 * PXs - 1 + n;
  const bool col_ok = cx >= 0 && cx < Wc;
  const bool edge = (2 * PXs - 1 < 0) || (2 * PXs + 30 >= Wc);
  const float kLow = -3.0e38f;

  f32x16 vm0, vm1;
#pragma unroll
  for (int q = 0; q < 16; ++q) { vm0[q] = kLow; vm1[q] = kLow; }
#pragma unroll 1
  for (int j = 0; j < kConvRows; ++j) {
    const int cy = 2 * PY0 - 1 + j;
    const bool valid = cy >= 0 && cy < Hc;
    f32x16 a0, a1;
#pragma unroll
    for (int q = 0; q < 16; ++q) { a0[q] = 0.f; a1[q] = 0.f; }
    if (valid) {
      const unsigned row_a = lbase + (unsigned)((6 * j + hi) * kRowDw) * 4u + col;
      // all 11 B operands of the row are requested before the first MFMA waits for one
      unsigned bw[kSteps][4];
#pragma unroll
      for (int s = 0; s < kSteps; ++s) {
        unsigned at = row_a + (unsigned)(2 * s * kRowDw) * 4u;
        if (s == kSteps - 1) at = hi ? ones_a : at;
        bw[s][0] = *(const lds_u32 *)(size_t)(at);
        bw[s][1] = *(const lds_u32 *)(size_t)(at + 4u);
        bw[s][2] = *(const lds_u32 *)(size_t)(at + 8u);
        bw[s][3] = *(const lds_u32 *)(size_t)(at + 12u);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < kSteps; ++s) {
        const f16x8 bv = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(bw[s]));
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[s][0], bv, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[s][1], bv, a1, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) { vm0[q] = fmaxf(vm0[q], a0[q]); vm1[q] = fmaxf(vm1[q], a1[q]); }
    }
    if (j >= 2 && (j & 1) == 0) {
      const int py = PY0 + (j >> 1) - 1;
      if (py < Hp) {
        // x-pooling: lane n (even) <- max over lanes n, n + 1, n + 2 of its half-wave
        float r0[16], r1[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          float u0 = vm0[q], u1 = vm1[q];
          if (edge && !col_ok) { u0 = kLow; u1 = kLow; }
          const float p0 = fmaxf(u0, next_lane<SHUF>(u0)), p1 = fmaxf(u1, next_lane<SHUF>(u1));
          r0[q] = fmaxf(fmaxf(p0, next_lane<SHUF>(p0)), 0.f);      // (ReLU; the bias is inside the sums)
          r1[q] = fmaxf(fmaxf(p1, next_lane<SHUF>(p1)), 0.f);
        }
        const int px = PXs + (n >> 1);
        if ((n & 1) == 0 && n <= 2 * (kStrip - 1) && px < Wp) {
          const size_t o_at = (((size_t)b * Hp + py) * Wp + px) * 64 + 16 * hi;
          if constexpr (OUT8) {
            int8_t *o = static_cast<int8_t *>(out_) + o_at;
            uint4 q0, q1;
            unsigned *pq0 = reinterpret_cast<unsigned *>(&q0), *pq1 = reinterpret_cast<unsigned *>(&q1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              unsigned d0 = 0u, d1 = 0u;
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                d0 |= ((unsigned)(int)fminf(rintf(r0[4 * g + i] * inv_s_out), 127.f) & 0xffu) << (8 * i);
                d1 |= ((unsigned)(int)fminf(rintf(r1[4 * g + i] * inv_s_out), 127.f) & 0xffu) << (8 * i);
              }
              pq0[g] = d0; pq1[g] = d1;
            }
            *reinterpret_cast<uint4 *>(o) = q0;
            *reinterpret_cast<uint4 *>(o + 32) = q1;
          } else {
            __half *o = static_cast<__half *>(out_) + o_at;
            uint4 h;
            h.x = pack_h2(r0[0], r0[1]); h.y = pack_h2(r0[2], r0[3]); h.z = pack_h2(r0[4], r0[5]); h.w = pack_h2(r0[6], r0[7]);
            *reinterpret_cast<uint4 *>(o) = h;
            h.x = pack_h2(r0[8], r0[9]); h.y = pack_h2(r0[10], r0[11]); h.z = pack_h2(r0[12], r0[13]); h.w = pack_h2(r0[14], r0[15]);
            *reinterpret_cast<uint4 *>(o + 8) = h;
            h.x = pack_h2(r1[0], r1[1]); h.y = pack_h2(r1[2], r1[3]); h.z = pack_h2(r1[4], r1[5]); h.w = pack_h2(r1[6], r1[7]);
            *reinterpret_cast<uint4 *>(o + 32) = h;
            h.x = pack_h2(r1[8], r1[9]); h.y = pack_h2(r1[10], r1[11]); h.z = pack_h2(r1[12], r1[13]); h.w = pack_h2(r1[14], r1[15]);
            *reinterpret_cast<uint4 *>(o + 40) = h;
          }
        }
      }
      // the window's last row is the next window's first
#pragma unroll
      for (int q = 0; q < 16; ++q) { vm0[q] = valid ? a0[q] : kLow; vm1[q] = valid ? a1[q] : kLow; }
    }
  }
}

thread_local int g_stem_variant = 0;   // bevops_stem_set_variant: 1 = the x-neighbours through ds_bpermute (A/B partner)

template <bool OUT8>
int stem_launch(const __half *x, const __half *wp, void *out, int n, int h, int w, float inv_s, hipStream_t st) {
  const int Hc = (h - 1) / 2 + 1, Wc = (w - 1) / 2 + 1;
  const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
  const int nbx = (Wp + kWaves * kStrip - 1) / (kWaves * kStrip), nby = (Hp + kR - 1) / kR;
  const size_t blocks = (size_t)n * nbx * nby;
  if (blocks == 0) return BEVOPS_SUCCESS;
  if (blocks > 0x7fffffffull) return BEVOPS_NOT_SUPPORTED;
  if (g_stem_variant == 1)
    hipLaunchKernelGGL((stem7x7_pool_kernel<OUT8, 1>), dim3((unsigned)blocks), dim3(256), kLdsBytes, st, x, wp, out, n, h,
                       w, Hc, Wc, Hp, Wp, nbx, nby, inv_s);
  else
    hipLaunchKernelGGL((stem7x7_pool_kernel<OUT8, 0>), dim3((unsigned)blocks), dim3(256), kLdsBytes, st, x, wp, out, n, h,
                       w, Hc, Wc, Hp, Wp, nbx, nby, inv_s);
  return launch_status();
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" size_t bevops_stem_packed_size(void) { return kPackedHalves * sizeof(__half); }

extern "C" int bevops_stem_pack(int dtype, const void *weight, const void *bias, void *packed, void *stream) {
  if (!weight || !packed) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (reinterpret_cast<uintptr_t>(packed) & 15u) return BEVOPS_BAD_PARAM;
  hipLaunchKernelGGL(stem_pack_kernel, dim3((unsigned)((kPackedHalves + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const __half *>(weight),
                     static_cast<const __half *>(bias), static_cast<__half *>(packed));
  return launch_status();
}

extern "C" int bevops_stem_conv_pool(int dtype, int out_dtype, const void *x, const void *packed, void *out, int n,
                                     int h, int w, float scale_out, void *stream) {
  if (!x || !packed || !out || n < 0 || h <= 0 || w <= 0) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16 || (out_dtype != BEVOPS_F16 && out_dtype != BEVOPS_I8)) return BEVOPS_NOT_SUPPORTED;
  if ((w & 1) || (reinterpret_cast<uintptr_t>(x) & 3u) || (reinterpret_cast<uintptr_t>(packed) & 15u) ||
      (reinterpret_cast<uintptr_t>(out) & 15u))
    return BEVOPS_NOT_SUPPORTED;   // dword-aligned image rows (an even width), 16-byte stores
  if (out_dtype == BEVOPS_I8 && !(scale_out > 0.f)) return BEVOPS_BAD_PARAM;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (out_dtype == BEVOPS_I8)
    return stem_launch<true>(static_cast<const __half *>(x), static_cast<const __half *>(packed), out, n, h, w,
                             1.f / scale_out, st);
  return stem_launch<false>(static_cast<const __half *>(x), static_cast<const __half *>(packed), out, n, h, w, 0.f, st);
}

extern "C" int bevops_stem_set_variant(int variant) {
  const int prev = g_stem_variant;
  g_stem_variant = variant;
  return prev;
}
