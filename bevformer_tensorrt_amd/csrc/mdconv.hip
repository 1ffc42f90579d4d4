// Modulated deformable convolution (DCNv2) forward for MI355X (gfx950).
// Replaces ModulatedDeformableConv2dPlugin::enqueue / getWorkspaceSize
// (TensorRT/plugin/modulated_deformable_conv2d/modulatedDeformableConv2dPlugin.cpp:73-197)
// and ModulatedDeformConvForwardCUDAKernel<T> (modulatedDeformableConv2dKernel.cu:695-978).
// Numerical contract: SURVEY.md Appendix A.5.
//
// The reference runs, per image and per group, a channel-planar im2col (one thread per
// (c_in, pixel), nine scattered 2-byte gathers each) followed by one cuBLAS GEMM, with a
// per-image column buffer.  MI355X-first restructuring:
//   1. x is re-laid out once to channels-last (NHWC) so that a bilinear corner is ONE
//      contiguous 16-byte load of 8 channels per lane (64..512 B per pixel across lanes);
//   2. weights are re-packed to [Cout][tap][Cin/g] so both GEMM operands are K-contiguous
//      with K ordered (tap, c_in): the sampling footprint (4 corner indices + weights*mask)
//      of a (pixel, tap) is computed once and reused across all input channels;
//   3. ONE batched GEMM over all images of the call (N = B*Ho*Wo columns) on the matrix
//      cores: 128x128x32 tiles, v_mfma_f32_32x32x16_f16, fp32 accumulate, bias fused in
//      the epilogue which scatters straight into NCHW.  (fp32 I/O uses an fp32 FMA tile
//      kernel -- exact fp32, parity path.)
// Workspace (caller-owned, bevops_mdconv_workspace_size): NHWC copy + packed weights +
// columns [G][N][K*K*Cin/g].
// This file: fp32 / fp16 kernels, the variant switches, workspace sizing and the fp entry points; the INT8
// flavour lives in mdconv_s8.hip; mdconv.h holds what the two share.
#include <type_traits>

#include "mdconv.h"

namespace bevops {
// A/B switches of bevops_mdconv_set_variant (thread-local; read by mdconv_s8.hip too)
thread_local int g_mdconv_variant = 0;
thread_local bool g_mdconv_no_tail = false;
thread_local bool g_mdconv_wide = false;  // variant 5: 1024-thread blocks, 128-pixel tiles whatever the tile count
thread_local int g_mdconv_rotate = 0;   // fp16 LDS-DMA kernel: 0 = round 6's wave order; variant 7 -> 1: halves rotated by half an iteration
                                        // (round 2); variant 13 -> 2: one order for all waves (rounds 2-5, the A/B partner)
namespace {

// fp16, HW % 8 == 0 and C % 8 == 0: 64 channels x 64 pixels per block, 16-byte global accesses on
// both sides (8 pixels of a channel in, 8 channels of a pixel out), 2-byte transposed LDS writes
__global__ __launch_bounds__(256) void nchw_to_nhwc_f16v_kernel(const __half *__restrict__ in,
                                                                __half *__restrict__ out, int C, int HW) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64][72];  // [pixel][channel], padded rows
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const __half *ib = in + (size_t)b * C * HW;
  __half *ob = out + (size_t)b * C * HW;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int v = threadIdx.x + it * 256;  // 64 channels x 8 pixel-vectors
    const int c = v >> 3, pv = (v & 7) * 8;
    if (c0 + c < C && p0 + pv < HW) {
      const uint4 q = *reinterpret_cast<const uint4 *>(ib + (size_t)(c0 + c) * HW + p0 + pv);
      const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        tile[pv + 2 * k][c] = (unsigned short)(w[k] & 0xffffu);
        tile[pv + 2 * k + 1][c] = (unsigned short)(w[k] >> 16);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int v = threadIdx.x + it * 256;  // 64 pixels x 8 channel-vectors
    const int p = v >> 3, cv = (v & 7) * 8;
    if (p0 + p < HW && c0 + cv < C)
      *reinterpret_cast<uint4 *>(ob + (size_t)(p0 + p) * C + c0 + cv) = *reinterpret_cast<const uint4 *>(&tile[p][cv]);
  }
}

// fp16, HW % 4 == 0 and C % 8 == 0: 64 channels x 128 pixels per block; a thread loads a 4 x 4 block of halves
// (8 bytes from each of 4 channel rows, lanes along the pixels: 256 contiguous bytes per row and half-wave),
// transposes it with 8 v_perm and writes 4 x 8 bytes (4 channels of one pixel) into the [pixel][channel] tile --
// a quarter of the LDS write instructions of the 2-byte version above; the tile leaves as 16-byte vectors,
// 128 contiguous bytes per pixel.
__global__ __launch_bounds__(256) void nchw_to_nhwc_f16q_kernel(const __half *__restrict__ in, __half *__restrict__ out,
                                                                int C, int HW) {
  constexpr int kRow = 128 + 16;   // bytes per tile row
  __shared__ __attribute__((aligned(16))) unsigned char tile[128 * kRow];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 128, c0 = blockIdx.y * 64;
  const __half *ib = in + (size_t)b * C * HW;
  __half *ob = out + (size_t)b * C * HW;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int v = threadIdx.x + it * 256;      // 16 channel quads x 32 pixel quads
    const int pq = v & 31, cq = v >> 5;
    const int p = p0 + pq * 4, c = c0 + cq * 4;
    uint2 r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = make_uint2(0u, 0u);
    if (p < HW && c < C) {                      // HW % 4 == 0, C % 4 == 0: the 4 x 4 block is all in or all out
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = *reinterpret_cast<const uint2 *>(ib + (size_t)(c + k) * HW + p);
    }
    // r[k] = pixels p .. p+3 of channel c + k  ->  o[j] = channels c .. c+3 of pixel p + j
    uint2 o[4];
    o[0] = make_uint2(__builtin_amdgcn_perm(r[1].x, r[0].x, 0x05040100u), __builtin_amdgcn_perm(r[3].x, r[2].x, 0x05040100u));
    o[1] = make_uint2(__builtin_amdgcn_perm(r[1].x, r[0].x, 0x07060302u), __builtin_amdgcn_perm(r[3].x, r[2].x, 0x07060302u));
    o[2] = make_uint2(__builtin_amdgcn_perm(r[1].y, r[0].y, 0x05040100u), __builtin_amdgcn_perm(r[3].y, r[2].y, 0x05040100u));
    o[3] = make_uint2(__builtin_amdgcn_perm(r[1].y, r[0].y, 0x07060302u), __builtin_amdgcn_perm(r[3].y, r[2].y, 0x07060302u));
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<uint2 *>(&tile[(pq * 4 + j) * kRow + cq * 8]) = o[j];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = threadIdx.x + it * 256;      // 128 pixels x 8 chunks of 8 channels
    const int ch = v & 7, pl = v >> 3;
    if (p0 + pl < HW && c0 + ch * 8 < C)
      *reinterpret_cast<uint4 *>(ob + (size_t)(p0 + pl) * C + c0 + ch * 8) =
          *reinterpret_cast<const uint4 *>(&tile[pl * kRow + ch * 16]);
  }
}

// ---- 3. deformable + modulated im2col on NHWC --------------------------------------
// thread = (global pixel n, tap, channel vector); columns [G][N][KK][cin_g]
template <typename T, int V>
__global__ __launch_bounds__(256) void im2col_nhwc_kernel(const T *__restrict__ xt,
                                                          const T *__restrict__ offset,
                                                          const T *__restrict__ mask,
                                                          T *__restrict__ col, ConvDims d) {
  const int vec_per_pix = d.Cin / V;
  const int KK = d.Kh * d.Kw;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int cv = (int)(idx % vec_per_pix);
  const size_t r = idx / vec_per_pix;
  const int t = (int)(r % KK);
  const size_t n = r / KK;
  const int HoWo = d.Ho * d.Wo;
  const size_t N = (size_t)d.B * HoWo;
  if (n >= N) return;
  const int b = (int)(n / HoWo);
  const int pix = (int)(n - (size_t)b * HoWo);
  const int ho = pix / d.Wo, wo = pix - ho * d.Wo;
  const int c = cv * V;
  const int dg = c / (d.Cin / d.DG);
  const int i = t / d.Kw, j = t - i * d.Kw;
  // offset [B][DG][2*KK][Ho][Wo] (h then w), mask [B][DG][KK][Ho][Wo]  (kernel.cu:283-300)
  const size_t obase = (((size_t)b * d.DG + dg) * 2 * KK) * HoWo + pix;
  const float off_h = tof(offset[obase + (size_t)(2 * t) * HoWo]);
  const float off_w = tof(offset[obase + (size_t)(2 * t + 1) * HoWo]);
  const float m = tof(mask[(((size_t)b * d.DG + dg) * KK + t) * HoWo + pix]);
  float h_im, w_im;
  {
#pragma clang fp contract(off)
    h_im = (float)(ho * d.sh - d.ph + i * d.dh) + off_h;
    w_im = (float)(wo * d.sw - d.pw + j * d.dw) + off_w;
  }
  float acc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = 0.f;
  if (h_im > -1.f && w_im > -1.f && h_im < (float)d.H && w_im < (float)d.W) {
#pragma clang fp contract(off)
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int h0 = (int)hf, w0 = (int)wf;
    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
    const float wt[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
    const bool ok[4] = {h0 >= 0 && w0 >= 0, h0 >= 0 && w0 + 1 <= d.W - 1,
                        h0 + 1 <= d.H - 1 && w0 >= 0, h0 + 1 <= d.H - 1 && w0 + 1 <= d.W - 1};
    const int hs[4] = {h0, h0, h0 + 1, h0 + 1}, ws[4] = {w0, w0 + 1, w0, w0 + 1};
    const T *xb = xt + (size_t)b * d.H * d.W * d.Cin + c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!ok[q]) continue;
      const T *p = xb + ((size_t)hs[q] * d.W + ws[q]) * d.Cin;
      if constexpr (sizeof(T) == 2 && V == 8) {
        const uint4 v = *reinterpret_cast<const uint4 *>(p);
        acc[0] += wt[q] * h2f_lo(v.x); acc[1] += wt[q] * h2f_hi(v.x);
        acc[2] += wt[q] * h2f_lo(v.y); acc[3] += wt[q] * h2f_hi(v.y);
        acc[4] += wt[q] * h2f_lo(v.z); acc[5] += wt[q] * h2f_hi(v.z);
        acc[6] += wt[q] * h2f_lo(v.w); acc[7] += wt[q] * h2f_hi(v.w);
      } else if constexpr (sizeof(T) == 4 && V == 4) {
        const float4 v = *reinterpret_cast<const float4 *>(p);
        acc[0] += wt[q] * v.x; acc[1] += wt[q] * v.y; acc[2] += wt[q] * v.z; acc[3] += wt[q] * v.w;
      } else {
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] += wt[q] * tof(p[k]);
      }
    }
  }
  const int cin_g = d.Cin / d.G;
  const int g = c / cin_g, cg = c - g * cin_g;
  T *o = col + (((size_t)g * N + n) * KK + t) * cin_g + cg;
  if constexpr (sizeof(T) == 2 && V == 8) {
    uint4 v;
    v.x = pack_h2(acc[0] * m, acc[1] * m); v.y = pack_h2(acc[2] * m, acc[3] * m);
    v.z = pack_h2(acc[4] * m, acc[5] * m); v.w = pack_h2(acc[6] * m, acc[7] * m);
    *reinterpret_cast<uint4 *>(o) = v;
  } else if constexpr (sizeof(T) == 4 && V == 4) {
    *reinterpret_cast<float4 *>(o) = make_float4(acc[0] * m, acc[1] * m, acc[2] * m, acc[3] * m);
  } else {
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = fromf<T>(acc[k] * m);
  }
}

// ---- 4a. fp16 GEMM on the matrix cores:  C[m][n] = sum_k A[m][k] * B[n][k] ------------
// A = packed weights of one group [M][K], B = columns [N][K]; epilogue adds bias and
// scatters to NCHW.  128x128x32 tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32x16.

__device__ __forceinline__ uint4 ld16_guard(const __half *p, bool ok) {
  return ok ? *reinterpret_cast<const uint4 *>(p) : make_uint4(0, 0, 0, 0);
}

__global__ __launch_bounds__(256) void gemm_tn_f16_kernel(const __half *__restrict__ A,
                                                          const __half *__restrict__ Bm,
                                                          const __half *__restrict__ bias,
                                                          __half *__restrict__ out, int M, int N,
                                                          int K, GemmEpi e) {
  __shared__ __attribute__((aligned(16))) __half As[kBM][kLd];
  __shared__ __attribute__((aligned(16))) __half Bs[kBN][kLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
  // global->LDS mapping: 128 rows x 4 chunks of 16 B; thread handles chunks tid and tid+256
  const int r0 = tid >> 2, kc = (tid & 3) * 8;
  const int r1 = r0 + 64;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (K + kBK - 1) / kBK;
  uint4 ra0, ra1, rb0, rb1;
  auto gload = [&](int kt) {
    const int k = kt * kBK + kc;
    const bool kok = k < K;  // K % 8 == 0 is guaranteed by the launcher
    ra0 = ld16_guard(A + (size_t)(m0 + r0) * K + k, kok && m0 + r0 < M);
    ra1 = ld16_guard(A + (size_t)(m0 + r1) * K + k, kok && m0 + r1 < M);
    rb0 = ld16_guard(Bm + (size_t)(n0 + r0) * K + k, kok && n0 + r0 < N);
    rb1 = ld16_guard(Bm + (size_t)(n0 + r1) * K + k, kok && n0 + r1 < N);
  };
  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    *reinterpret_cast<uint4 *>(&As[r0][kc]) = ra0;
    *reinterpret_cast<uint4 *>(&As[r1][kc]) = ra1;
    *reinterpret_cast<uint4 *>(&Bs[r0][kc]) = rb0;
    *reinterpret_cast<uint4 *>(&Bs[r1][kc]) = rb1;
    __syncthreads();
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kk = ks * 16 + (lane >> 5) * 8;
      f16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const f16x8 *>(&As[wm * 64 + i * 32 + (lane & 31)][kk]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = *reinterpret_cast<const f16x8 *>(&Bs[wn * 64 + j * 32 + (lane & 31)][kk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + (lane & 31);
    if (n >= N) continue;
    const int b = n / e.HoWo, pix = n - b * e.HoWo;
    __half *ob = out + ((size_t)b * e.Cout + e.co0) * e.HoWo + pix;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M) {
          float v = acc[i][j][r];
          if (bias) v += __half2float(bias[e.co0 + m]);
          ob[(size_t)m * e.HoWo] = __float2half_rn(v);
        }
      }
  }
}

// ---- 4a'. fp16 GEMM for ragged K (K % 8 != 0: rows are not 16-byte aligned) -----------
// One thread per output element, 2-byte loads, fp32 accumulate.  Only reached by shapes such as
// Cin/groups = 4 with a 3x3 kernel (K = 36); the reference's half kernel accepts them, so the
// drop-in must too.  Lanes run along the pixels so the column reads of a k step are one row apart.
__global__ __launch_bounds__(256) void gemm_tn_f16_ragged_kernel(const __half *__restrict__ A,
                                                                 const __half *__restrict__ B,
                                                                 const __half *__restrict__ bias,
                                                                 __half *__restrict__ out, int M, int N, int K,
                                                                 GemmEpi e) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M * N) return;
  const int n = (int)(idx % N), m = (int)(idx / N);
  const __half *a = A + (size_t)m * K, *b = B + (size_t)n * K;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(__half2float(a[k]), __half2float(b[k]), acc);
  if (bias) acc += __half2float(bias[e.co0 + m]);
  const int bi = n / e.HoWo, pix = n - bi * e.HoWo;
  out[((size_t)bi * e.Cout + e.co0 + m) * e.HoWo + pix] = __float2half_rn(acc);
}

// ---- 4b. fp32 GEMM (exact fp32 FMA tiles; parity path) -------------------------------
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(const float *__restrict__ A,
                                                          const float *__restrict__ Bm,
                                                          const float *__restrict__ bias,
                                                          float *__restrict__ out, int M, int N,
                                                          int K, GemmEpi e) {
  constexpr int T = 64, BK = 16;
  __shared__ float As[BK][T + 4], Bs[BK][T + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
  const int lr = tid >> 2, lk = (tid & 3) * 4;  // 64 rows x 4 chunks of 4 floats
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += BK) {
    float av[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = k0 + lk + q;
      av[q] = (m0 + lr < M && k < K) ? A[(size_t)(m0 + lr) * K + k] : 0.f;
      bv[q] = (n0 + lr < N && k < K) ? Bm[(size_t)(n0 + lr) * K + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      As[lk + q][lr] = av[q];
      Bs[lk + q][lr] = bv[q];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BK; ++k) {
      const float4 a = *reinterpret_cast<const float4 *>(&As[k][ty * 4]);
      const float4 b = *reinterpret_cast<const float4 *>(&Bs[k][tx * 4]);
      const float aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(aa[i], bb[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + tx * 4 + j;
    if (n >= N) continue;
    const int b = n / e.HoWo, pix = n - b * e.HoWo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + ty * 4 + i;
      if (m < M)
        out[((size_t)b * e.Cout + e.co0 + m) * e.HoWo + pix] = acc[i][j] + (bias ? bias[e.co0 + m] : 0.f);
    }
  }
}


// ---- 5. fused deformable implicit GEMM (fp16) ------------------------------------------
// out[b, co, pix] = bias[co] + sum_{tap, ci} W[co][tap][ci] * col(pix, tap, ci), where the
// column element is produced on the fly: the B-tile loader bilinearly samples the NHWC image
// (4 x 128-byte lines per (pixel, tap, 64-channel chunk)), scales by the mask and writes the
// MFMA operand tile straight into LDS -- no column buffer in HBM (160 MB written + read per
// stage-3 call in the two-kernel version).
//   block tile 256 (Cout) x 64 (pixels) x 64 (one tap, 64 input channels); 4 waves, each
//   64 x 64 = 2x2 v_mfma_f32_32x32x16_f16 tiles x 4 k-substeps; footprint (4 corner indices
//   + weights*mask) computed once per (pixel, tap) and reused across the channel chunks;
//   bilinear blend in packed fp16 (v_pk_fma_f16, as the reference's half2 kernel does,
//   modulatedDeformableConv2dKernel.cu:390-461), accumulation in fp32 on the matrix cores.
// Arithmetic intensity is capped by the gather at BM/4 flop per gathered byte, hence the
// full-Cout M tile.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_fma(unsigned a, unsigned w, unsigned c) {
  const f16x2 r = __builtin_bit_cast(f16x2, a) * __builtin_bit_cast(f16x2, w) + __builtin_bit_cast(f16x2, c);
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned pk_mul(unsigned a, unsigned w) {
  const f16x2 r = __builtin_bit_cast(f16x2, a) * __builtin_bit_cast(f16x2, w);
  return __builtin_bit_cast(unsigned, r);
}

// THREADS = 256: 4 waves, each 64 x 64 outputs (the r01c kernel, 206 VGPR, 8 waves per CU).
// THREADS = 512: 8 waves in a 4 (Cout) x 2 (pixels) grid, each 64 x 32 outputs: half the
// accumulators, prefetch registers and blend work per thread -> 16 waves per CU to hide the
// gather / weight-load latency between the two barriers of a k-step.
template <int THREADS>
__global__ __launch_bounds__(THREADS, 2) void dcn_fused_f16_kernel(
    const __half *__restrict__ xt, const __half *__restrict__ offset,
    const __half *__restrict__ mask, const __half *__restrict__ wt,
    const __half *__restrict__ bias, __half *__restrict__ out, ConvDims d, int g) {
  constexpr int NW = THREADS / 64;        // waves
  constexpr int WN = NW / 4;              // waves along the pixel dimension (1 or 2)
  constexpr int NJ = kFN / WN / 32;       // 32-pixel MFMA tiles per wave (2 or 1)
  constexpr int LPP = THREADS / kFN;      // producer lanes per pixel (4 or 8)
  constexpr int NB = kFK / LPP / 8;       // 16-byte vectors per corner per thread (2 or 1)
  constexpr int RPP = THREADS / 8;        // weight rows per loader pass (32 or 64)
  constexpr int NA = kFM / RPP;           // loader passes (8 or 4)
  __shared__ __attribute__((aligned(16))) __half As[kFM][kFLd];
  __shared__ __attribute__((aligned(16))) __half Bs[kFN][kFLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int KK = d.Kh * d.Kw, cin_g = d.Cin / d.G, cout_g = d.Cout / d.G;
  const int HoWo = d.Ho * d.Wo;
  const int N = d.B * HoWo;
  // consecutive pixel tiles share input lines (3x3 footprints, neighbouring rows): keep them on
  // one XCD so its 4 MiB L2 holds ~1/8 of the images + the weights instead of all of it
  const int n0 = (int)xcd_remap(blockIdx.x, gridDim.x) * kFN, m0 = blockIdx.y * kFM;
  const int Kg = KK * cin_g;
  const __half *A = wt + (size_t)g * cout_g * Kg;

  // B-producer role: pixel n0 + tid / LPP, channels [cq * 8 * NB, +8 * NB) of the 64-chunk
  const int pn = n0 + tid / LPP, cq = tid % LPP;
  const bool pvalid = pn < N;
  const int pb = pvalid ? pn / HoWo : 0;
  const int ppix = pvalid ? pn - pb * HoWo : 0;
  const int pho = ppix / d.Wo, pwo = ppix - pho * d.Wo;
  // A-loader role: rows (tid >> 3) + RPP*i, 16-byte chunk (tid & 7)
  const int ar = tid >> 3, ac = (tid & 7) * 8;
  // both operands are fetched through buffer descriptors: per-thread byte offsets that only
  // change with the tap (image) or never (weights) plus a scalar offset per k-step -- no
  // 64-bit address arithmetic in the k-loop; rows past Cout read as zero (range check)
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__half *>(xt), 0, (unsigned)((size_t)d.B * d.H * d.W * d.Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__half *>(A), 0, (unsigned)((size_t)cout_g * Kg * 2), 0x00020000);
  const unsigned ximg_off = (unsigned)(((size_t)pb * d.H * d.W * d.Cin + g * cin_g + cq * 8 * NB) * 2);
  unsigned a_off[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int r = m0 + ar + RPP * i;
    a_off[i] = r < cout_g ? (unsigned)(((size_t)r * Kg + ac) * 2) : 0xFFFFFFF0u;
  }

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int chunks = cin_g / kFK;
  const int nsteps = KK * chunks;
  int fidx[4];
  unsigned fw[4];  // half2 (w, w) = bilinear weight * mask
  int cur_tap = -1, cur_dg = -1;
  uint4 ra[NA], rb[4][NB];

  // all offsets / masks of the tile's 64 pixels (3 * KK values per deform group) are fetched
  // once into LDS: a tap change then costs ALU only instead of a dependent global round trip
  // in front of the gather
  constexpr int kOmMax = 64;
  __shared__ __half Om[kOmMax][kFN];
  const int om_per_dg = 3 * KK;
  const bool om_ok = d.DG * om_per_dg <= kOmMax;
  if (om_ok) {
    for (int idx = tid; idx < d.DG * om_per_dg * kFN; idx += THREADS) {
      const int p = idx % kFN, t = idx / kFN;
      const int dgi = t / om_per_dg, tt = t - dgi * om_per_dg;
      const int n = n0 + p;
      __half v = __float2half(0.f);
      if (n < N) {
        const int b = n / HoWo, pix = n - b * HoWo;
        v = tt < 2 * KK ? offset[(((size_t)b * d.DG + dgi) * 2 * KK + tt) * HoWo + pix]
                        : mask[(((size_t)b * d.DG + dgi) * KK + (tt - 2 * KK)) * HoWo + pix];
      }
      Om[t][p] = v;
    }
    __syncthreads();
  }

  auto prefetch = [&](int step) {
    const int tap = step / chunks, c0 = (step - tap * chunks) * kFK;
    const int dg = (g * cin_g + c0) / (d.Cin / d.DG);
    if (tap != cur_tap || dg != cur_dg) {
      cur_tap = tap;
      cur_dg = dg;
      const int i = tap / d.Kw, j = tap - i * d.Kw;
      float off_h, off_w, m;
      if (om_ok) {
        const int p = tid / LPP;
        off_h = __half2float(Om[dg * om_per_dg + 2 * tap][p]);
        off_w = __half2float(Om[dg * om_per_dg + 2 * tap + 1][p]);
        m = __half2float(Om[dg * om_per_dg + 2 * KK + tap][p]);
      } else {
        const size_t ob = (((size_t)pb * d.DG + dg) * 2 * KK) * HoWo + ppix;
        off_h = __half2float(offset[ob + (size_t)(2 * tap) * HoWo]);
        off_w = __half2float(offset[ob + (size_t)(2 * tap + 1) * HoWo]);
        m = __half2float(mask[(((size_t)pb * d.DG + dg) * KK + tap) * HoWo + ppix]);
      }
      const float h_im = (float)(pho * d.sh - d.ph + i * d.dh) + off_h;
      const float w_im = (float)(pwo * d.sw - d.pw + j * d.dw) + off_w;
      const bool in = pvalid && h_im > -1.f && w_im > -1.f && h_im < (float)d.H && w_im < (float)d.W;
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h0 = (int)hf, w0 = (int)wf;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const float wq[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
      const int hs[4] = {h0, h0, h0 + 1, h0 + 1}, ws[4] = {w0, w0 + 1, w0, w0 + 1};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = in && hs[q] >= 0 && hs[q] <= d.H - 1 && ws[q] >= 0 && ws[q] <= d.W - 1;
        fidx[q] = (int)(ximg_off + (unsigned)(ok ? hs[q] * d.W + ws[q] : 0) * (unsigned)(d.Cin * 2));
        const float w = ok ? wq[q] * m : 0.f;
        fw[q] = pack_h2(w, w);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int v = 0; v < NB; ++v)
        rb[q][v] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, fidx[q] + 16 * v, c0 * 2, 0));
    const int a_s = (tap * cin_g + c0) * 2;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      ra[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)a_off[i], a_s, 0));
  };

#ifdef DCN_PROFILE
  unsigned long long tacc[5] = {0, 0, 0, 0, 0}, t0 = __builtin_amdgcn_s_memtime(), t1;
#define TICK(i) { t1 = __builtin_amdgcn_s_memtime(); tacc[i] += t1 - t0; t0 = t1; }
#else
#define TICK(i)
#endif
  prefetch(0);
  TICK(0)
  for (int step = 0; step < nsteps; ++step) {
    // blend the 4 corners (packed fp16), 8 * NB channels per thread
    uint4 bl[NB];
#pragma unroll
    for (int v = 0; v < NB; ++v) {
      bl[v].x = pk_mul(rb[0][v].x, fw[0]); bl[v].y = pk_mul(rb[0][v].y, fw[0]);
      bl[v].z = pk_mul(rb[0][v].z, fw[0]); bl[v].w = pk_mul(rb[0][v].w, fw[0]);
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        bl[v].x = pk_fma(rb[q][v].x, fw[q], bl[v].x); bl[v].y = pk_fma(rb[q][v].y, fw[q], bl[v].y);
        bl[v].z = pk_fma(rb[q][v].z, fw[q], bl[v].z); bl[v].w = pk_fma(rb[q][v].w, fw[q], bl[v].w);
      }
    }
    TICK(1)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<uint4 *>(&As[ar + RPP * i][ac]) = ra[i];
#pragma unroll
    for (int v = 0; v < NB; ++v) *reinterpret_cast<uint4 *>(&Bs[tid / LPP][cq * 8 * NB + 8 * v]) = bl[v];
    __syncthreads();
    TICK(2)
    if (step + 1 < nsteps) prefetch(step + 1);
    TICK(3)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = ks * 16 + (lane >> 5) * 8;
      f16x8 a[2], b[NJ];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const f16x8 *>(&As[wm * 64 + i * 32 + (lane & 31)][kk]);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        b[j] = *reinterpret_cast<const f16x8 *>(&Bs[wn * (kFN / WN) + j * 32 + (lane & 31)][kk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    TICK(4)
  }
#ifdef DCN_PROFILE
  if ((tid & 63) == 0 && blockIdx.x < 64) {
    unsigned long long *dbg = reinterpret_cast<unsigned long long *>(const_cast<__half *>(wt)) + 1000000 +
                              (blockIdx.x * (THREADS / 64) + wave) * 8;
    for (int i = 0; i < 5; ++i) dbg[i] = tacc[i];
  }
#endif
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = n0 + wn * (kFN / WN) + j * 32 + (lane & 31);
    if (n >= N) continue;
    const int b = n / HoWo, pix = n - b * HoWo;
    __half *ob = out + ((size_t)b * d.Cout + g * cout_g) * HoWo + pix;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < cout_g) {
          float v = acc[i][j][r];
          if (bias) v += __half2float(bias[g * cout_g + m]);
          ob[(size_t)m * HoWo] = __float2half_rn(v);
        }
      }
  }
}


// ---- 5c. pipelined fused kernel: weights by LDS-DMA, two LDS buffers, one barrier per k-step ----
// The register-staged kernel above spends a k-step as  wait loads -> blend -> barrier -> 40 KB of
// ds_write_b128 (13 cycles per wave-instruction) -> barrier -> MFMA: the phase probe
// (tools/dcn_phase_probe.py) showed 1/3 of the time in the write + barrier phase and the matrix
// cores 12 % busy.  Here the weight tile (80 % of the staged bytes) goes global -> LDS directly
// (buffer_load_dwordx4 ... lds, no VGPRs, no ds_write) one step ahead into the other LDS
// buffer while the MFMAs of the current step run; only the blended pixel tile (8 KB) is
// written by the threads.  LDS image: 128-byte rows (64 k-values), 16-byte chunks XOR-swizzled
// by (row ^ row >> 3) & 7 -- the DMA is lane-linear, so the swizzle is applied to the SOURCE
// address; fragment reads (ds_read_b128) are bank-conflict free.
//   512 threads = 8 waves, 4 (Cout) x 2 (pixels), wave tile 64 x 32; 80 KB LDS -> 2 blocks per CU.
// Domain: one deform group per conv group (dg constant over a block's k-loop); else the
// register-staged kernel runs.
// 4 consecutive output channels m..m+3 of pixel (b, pix): bias, optional ReLU, NCHW or NHWC store
__device__ __forceinline__ void dcn_store4(__half *__restrict__ out, const __half *__restrict__ bias,
                                           const float (&v)[4], int b, int pix, int m, int g,
                                           int cout_g, int Cout, int HoWo, const TailPlan &tp) {
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[e] = v[e];
    if (bias && m + e < cout_g) o[e] += __half2float(bias[g * cout_g + m + e]);
    if (tp.relu) o[e] = fmaxf(o[e], 0.f);
  }
  if (tp.out_nhwc) {
    __half *p = out + ((size_t)b * HoWo + pix) * Cout + g * cout_g + m;
    if (m + 3 < cout_g && (((g * cout_g + m) | Cout) & 3) == 0) {
      uint2 w;
      w.x = pack_h2(o[0], o[1]);
      w.y = pack_h2(o[2], o[3]);
      *reinterpret_cast<uint2 *>(p) = w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (m + e < cout_g) p[e] = __float2half_rn(o[e]);
    }
  } else {
    __half *p = out + ((size_t)b * Cout + g * cout_g + m) * HoWo + pix;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (m + e < cout_g) p[(size_t)e * HoWo] = __float2half_rn(o[e]);
  }
}

// SCHED: 0 = every wave runs a step's segments in one order (rounds 2-5; tp.rotate == 1: upper half rotated by half an
// iteration); 4 = round 6's default, see the loop
template <int WN, int SCHED = 0>
__global__ __launch_bounds__(256 * WN, WN == 2 ? 2 : 1) void dcn_glds_f16_kernel(
    const __half *__restrict__ xt, const __half *__restrict__ offset,
    const __half *__restrict__ mask, const __half *__restrict__ wt,
    const __half *__restrict__ bias, __half *__restrict__ out, ConvDims d, int g, TailPlan tp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [A0][A1][B0][B1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int KK = d.Kh * d.Kw, cin_g = d.Cin / d.G, cout_g = d.Cout / d.G;
  const int HoWo = d.Ho * d.Wo;
  const int N = d.B * HoWo;
  // block -> (pixel tile, k-range)
  const int n_tail_blocks = tp.tail_tiles * tp.split;
  const bool is_tail = (int)blockIdx.x < n_tail_blocks;
  int ntile, part = 0;
  if (is_tail) {
    ntile = tp.main_tiles + (int)blockIdx.x / tp.split;
    part = (int)blockIdx.x % tp.split;
  } else {
    ntile = (int)xcd_remap(blockIdx.x - n_tail_blocks, tp.main_tiles);
  }
  constexpr int kN = Glds<WN>::kN, kGB = Glds<WN>::kB, kTh = Glds<WN>::kThreads, kPc = Glds<WN>::kPieces;
  const int n0 = ntile * kN, m0 = blockIdx.y * kFM;
  const int Kg = KK * cin_g;
  const __half *A = wt + (size_t)g * cout_g * Kg;
  const int dg = (g * cin_g) / (d.Cin / d.DG);

  // pixel-producer role: pixel n0 + (tid >> 3), 8 channels (16 B) cq of the 64-chunk
  const int pp = tid >> 3, cq = tid & 7;
  const int pn = n0 + pp;
  const bool pvalid = pn < N;
  const int pb = pvalid ? pn / HoWo : 0;
  const int ppix = pvalid ? pn - pb * HoWo : 0;
  const int pho = ppix / d.Wo, pwo = ppix - pho * d.Wo;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__half *>(xt), 0, (unsigned)((size_t)d.B * d.H * d.W * d.Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__half *>(A), 0, (unsigned)((size_t)cout_g * Kg * 2), 0x00020000);
  const unsigned ximg_off = (unsigned)(((size_t)pb * d.H * d.W * d.Cin + g * cin_g + cq * 8) * 2);
  const unsigned b_dst = (unsigned)(pp * 128 + ((cq ^ swz8(pp)) << 4));
  // weight DMA role: piece j of this wave = rows (wave*4 + j)*8 .. +8, lane -> (row, chunk)
  // SCHED 4: the LOWER half of the waves issues ALL the weight pieces (two waves' worth each), the upper half none --
  // an upper wave then has no vector-memory instruction in front of its matrix segment (phase probe, round 6: with the
  // 16 waves' loads arbitrated oldest first, even the two DMA pieces of an upper wave sat ~1 000 cycles behind the lower
  // waves' gathers before its MFMAs could start)
  constexpr int kPcS = SCHED == 4 ? 2 * kPc : kPc;
  unsigned a_off[kPcS];
#pragma unroll
  for (int j = 0; j < kPcS; ++j) {
    const unsigned row = (unsigned)((wave * kPcS + j) * 8 + (lane >> 3));
    const unsigned chunk = (lane & 7u) ^ swz8(row);
    a_off[j] = (m0 + (int)row) < cout_g ? (unsigned)(((size_t)(m0 + row) * Kg) * 2 + chunk * 16) : 0xFFFFFFF0u;
  }
  // fragment read offsets inside a buffer (k-substep ks adds chunk 2*ks: XOR into the swizzle)
  unsigned fa[2], fb;
#pragma unroll
  for (int i = 0; i < 2; ++i) fa[i] = (unsigned)(wm * 64 + i * 32 + (lane & 31));
  fb = (unsigned)(wn * 32 + (lane & 31));
  const unsigned hi = lane >> 5;

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int chunks = cin_g / kFK;
  // offsets / mask of the pixel, one tap ahead in registers
  const size_t ob = (((size_t)pb * d.DG + dg) * 2 * KK) * HoWo + ppix;
  const size_t mb = (((size_t)pb * d.DG + dg) * KK) * HoWo + ppix;
  const size_t omb = ((size_t)pb * HoWo + ppix) * (size_t)tp.om_channels + (size_t)dg * 3 * KK;
  auto load_om = [&](int tap, __half &oh, __half &ow, __half &mm) {
    if (tp.om_channels) {  // channels-last: one pixel's values are contiguous; mask = sigmoid(logit)
      const unsigned o2 = *reinterpret_cast<const unsigned *>(offset + omb + 2 * tap);
      oh = __ushort_as_half((unsigned short)(o2 & 0xffffu));
      ow = __ushort_as_half((unsigned short)(o2 >> 16));
      mm = offset[omb + 2 * KK + tap];   // raw mask logit: the sigmoid is applied where the value is consumed
    } else {
      oh = offset[ob + (size_t)(2 * tap) * HoWo];
      ow = offset[ob + (size_t)(2 * tap + 1) * HoWo];
      mm = mask[mb + (size_t)tap * HoWo];
    }
  };
  const int taps_per_part = is_tail ? KK / tp.split : KK;
  // all resident blocks walk the same 1.2 MB weight matrix: start each tile at a different tap
  // (and wrap) so that they do not all pull the same 32 KB slice from the same L2 channels in
  // the same k-step.  The fp32 summation order then depends on the tile index only.
  // (only while all taps' slices fit an XCD's L2 together: 4.7 MB at stage 4 would thrash it)
  const bool stagger = (size_t)cout_g * Kg * 2 <= (size_t)(2 << 20);
  const int tap_begin = is_tail ? part * taps_per_part : (stagger ? ntile % KK : 0);
  const int n_my_steps = taps_per_part * chunks;
  __half n_oh, n_ow, n_mm;
  load_om(tap_begin, n_oh, n_ow, n_mm);
  int fidx[4];
  unsigned fw[4];
  uint4 rb[4];
  auto footprint = [&](int tap) {
    const float off_h = __half2float(n_oh), off_w = __half2float(n_ow);
    const float m = tp.om_channels ? __half2float(__float2half_rn(1.f / (1.f + __expf(-__half2float(n_mm))))) : __half2float(n_mm);
    load_om(tap + 1 < KK ? tap + 1 : 0, n_oh, n_ow, n_mm);
    const int i = tap / d.Kw, j = tap - i * d.Kw;
    const float h_im = (float)(pho * d.sh - d.ph + i * d.dh) + off_h;
    const float w_im = (float)(pwo * d.sw - d.pw + j * d.dw) + off_w;
    const bool in = pvalid && h_im > -1.f && w_im > -1.f && h_im < (float)d.H && w_im < (float)d.W;
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int h0 = (int)hf, w0 = (int)wf;
    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
    const float wq[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
    const int hs[4] = {h0, h0, h0 + 1, h0 + 1}, ws[4] = {w0, w0 + 1, w0, w0 + 1};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = in && hs[q] >= 0 && hs[q] <= d.H - 1 && ws[q] >= 0 && ws[q] <= d.W - 1;
      fidx[q] = (int)(ximg_off + (unsigned)(ok ? hs[q] * d.W + ws[q] : 0) * (unsigned)(d.Cin * 2));
      const float w = ok ? wq[q] * m : 0.f;
      fw[q] = pack_h2(w, w);
    }
  };
  typedef __attribute__((address_space(3))) void lds_void;
  // The gathers run TWO steps ahead of the MFMAs (their corners are blended one step ahead), the weight
  // DMA one step ahead; each pipeline walks (tap, chunk) with its own counters.  Iteration `step` (tiles of
  // `step` in buffer step & 1, raw corners of step + 1 in rb, a step old):
  //   blend + ds_write of the step + 1 pixel row [rb free] | weight DMA of step + 1 | footprint + gathers
  //   of step + 2 | 8 MFMAs | s_waitcnt vmcnt(4): the DMA is older than the 4 gathers, which stay in flight
  //   across the barrier (r01: gathers issued and consumed inside one step, their round trip exposed).
  int g_tap = tap_begin, g_chunk = 0, w_tap = tap_begin, w_chunk = 0;
  unsigned c_fw[4] = {0u, 0u, 0u, 0u};   // blend weights of the corners held in rb
  auto gather_next = [&]() {
    if (g_chunk == 0) footprint(g_tap);   // (also requests the offsets / mask of the tap after it: younger than the
                                          // step's weight DMA, older than the gathers -- retired by the same vmcnt(4))
    const int c0 = g_chunk * kFK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      rb[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, fidx[q], c0 * 2, 0));
      c_fw[q] = fw[q];
    }
    if (++g_chunk == chunks) { g_chunk = 0; g_tap = g_tap + 1 < KK ? g_tap + 1 : 0; }
  };
  auto weights_next = [&](int buf) {
    const int a_s = (w_tap * cin_g + w_chunk * kFK) * 2;
    char *adst = smem + buf * kGA + wave * (kPcS * 1024);
#pragma unroll
    for (int j = 0; j < kPcS; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void *)(adst + j * 1024), 16, (int)a_off[j], a_s, 0, 0);
    if (++w_chunk == chunks) { w_chunk = 0; w_tap = w_tap + 1 < KK ? w_tap + 1 : 0; }
  };
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned b_lds = (unsigned)(size_t)((lds_char *)smem) + (unsigned)(2 * kGA) + b_dst;
  auto blend_store = [&](int buf) {
    u32x4_t bl;
    bl.x = pk_mul(rb[0].x, c_fw[0]); bl.y = pk_mul(rb[0].y, c_fw[0]);
    bl.z = pk_mul(rb[0].z, c_fw[0]); bl.w = pk_mul(rb[0].w, c_fw[0]);
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      bl.x = pk_fma(rb[q].x, c_fw[q], bl.x); bl.y = pk_fma(rb[q].y, c_fw[q], bl.y);
      bl.z = pk_fma(rb[q].z, c_fw[q], bl.z); bl.w = pk_fma(rb[q].w, c_fw[q], bl.w);
    }
    // hand-written store (see dcn_glds_s8_kernel): a compiler-visible LDS store would drain vmcnt to 0
    asm volatile("ds_write_b128 %0, %1" ::"v"(b_lds + (unsigned)(buf * kGB)), "v"(bl) : "memory");
  };
  constexpr bool kLowerDma = SCHED == 4;

  // prologue: step 0 -> buffer 0, corners of step 1 in flight
  gather_next();
  blend_store(0);
  __builtin_amdgcn_sched_barrier(0);
  if (!kLowerDma || __builtin_amdgcn_readfirstlane(wave) < kTh / 128) weights_next(0);
  if (n_my_steps > 1) gather_next();
  if (n_my_steps > 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  auto mfma_step = [&](int buf) {
    const char *Ab = smem + buf * kGA;
    const char *Bb = smem + 2 * kGA + buf * kGB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned c = 2u * ks + hi;
      f16x8 a[2], b;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const f16x8 *>(Ab + fa[i] * 128 + ((c ^ swz8(fa[i])) << 4));
      b = *reinterpret_cast<const f16x8 *>(Bb + fb * 128 + ((c ^ swz8(fb)) << 4));
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b, acc[i], 0, 0, 0);
    }
  };
  // tp.rotate: the upper half of the waves runs the same loop rotated by half an iteration (its barrier sits
  // between the blend and the MFMAs, MFMA(0) is peeled): between two barriers one half issues its loads while
  // the other half reads fragments and feeds the matrix cores (see dcn_glds_s8_kernel)
  const bool upper = __builtin_amdgcn_readfirstlane(wave) >= kTh / 128;
  const bool late = SCHED == 0 && tp.rotate == 1 && upper;
  // tp.rotate == 2 (round 6): OPPOSED halves.  Every wave of the block issues 6 vector-memory instructions per step
  // (2 weight DMA pieces, 4 corner gathers): 96 wave-instructions of 1 KB through a 64 B/clk L1 path = 1 536 cycles in
  // which, with all 16 waves in the same order, nobody is in its matrix segment (and the round-2 rotation above keeps
  // "loads, then MFMAs" in both halves).  Here the upper half runs  DMA -> MFMA(step) -> blend(step + 1) -> gathers
  // while the lower half runs  blend -> DMA -> gathers -> MFMA(step): between two barriers one half occupies the L1
  // path while the other occupies the matrix cores and the LDS read path, then they swap.  Same barrier per step, same
  // buffers (a step consumes buffer step & 1 and produces the other), same arithmetic and summation order.
  if constexpr (SCHED == 4) {
    // Round 6 (default).  The phase probe of the one-order loop (s_memtime stamps between the segments,
    // profiles/r06/dcn_phase_probe.jsonl) showed where a step goes: its 96 vector-memory wave-instructions (32 weight
    // DMA pieces, 64 corner gathers: 768 cache lines) leave the CU's L1 path at ~0.4 lines per clock, arbitrated oldest
    // wave first -- waves 12-15 sit in their load-issue segment for 58 % of the loop (waves 0-3: 27 %, the rest of
    // their time at the barrier), and nobody is in a matrix segment meanwhile.  Even the two DMA pieces of an upper
    // wave queue ~1 000 cycles behind the lower waves' gathers.  Here the LOWER half of the waves issues all the
    // weight pieces (kPcS = two waves' worth each) and the upper half none:
    //   lower half: blend -> DMA (all pieces) -> gathers -> MFMA       upper half: MFMA -> blend -> gathers
    // so an upper wave has no vector-memory instruction in front of its matrix segment.  One loop body, the segments a
    // half does not run at a position skipped by a wave-uniform branch (two copies of the loop, one per half, cost the
    // 1 024-thread build ten spilled registers).  Same barrier per step, same buffers, same arithmetic and summation
    // order: bit-identical.  Measured with it (and removed): both halves issuing their own pieces in opposed order
    // (+- 1 %), raised priority in the matrix segment (+- 1 %), a second set of corner registers so that a gather has
    // two iterations to land (no gain): design/dcn.md.
    for (int step = 0; step < n_my_steps; ++step) {
      const bool more1 = step + 1 < n_my_steps, more2 = step + 2 < n_my_steps;
      if (!upper && more1) blend_store((step + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (!upper && more1) weights_next((step + 1) & 1);
      if (!upper && more2) gather_next();
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(step & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (upper && more1) blend_store((step + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (upper && more2) gather_next();
      __builtin_amdgcn_sched_barrier(0);
      if (more2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
  if (late) {
    if (n_my_steps > 1) weights_next(1);
    mfma_step(0);
  }
  for (int step = 0; step < n_my_steps; ++step) {
    const bool more1 = step + 1 < n_my_steps, more2 = step + 2 < n_my_steps;
    if (more1) blend_store((step + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    if (late) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (late ? more2 : more1) weights_next(late ? (step & 1) : ((step + 1) & 1));
    if (more2) gather_next();
    __builtin_amdgcn_sched_barrier(0);
    if (late ? more1 : true) mfma_step(late ? ((step + 1) & 1) : (step & 1));
    if (!late) {
      if (more2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  }   // (SCHED < 2)
  if (is_tail) {  // fp32 partials, thread-private order (the finish kernel uses the same mapping)
    float4 *pp = reinterpret_cast<float4 *>(tp.partial) +
                 ((((size_t)part * tp.tail_tiles + (ntile - tp.main_tiles)) * gridDim.y + blockIdx.y) * 8) * kTh + tid;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pp[(i * 4 + r) * kTh] = make_float4(acc[i][4 * r], acc[i][4 * r + 1], acc[i][4 * r + 2], acc[i][4 * r + 3]);
    return;
  }
  {
    const int n = n0 + wn * 32 + (lane & 31);
    if (n < N) {
      const int b = n / HoWo, pix = n - b * HoWo;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int m = m0 + wm * 64 + i * 32 + 8 * rq + 4 * (lane >> 5);
          const float v[4] = {acc[i][4 * rq], acc[i][4 * rq + 1], acc[i][4 * rq + 2], acc[i][4 * rq + 3]};
          dcn_store4(out, bias, v, b, pix, m, g, cout_g, d.Cout, HoWo, tp);
        }
    }
  }
}


// grid (tail tiles, Cout tiles, 8): block z sums accumulator quad z = i*4 + r of every thread
template <int WN>
__global__ __launch_bounds__(256 * WN) void dcn_tail_finish_kernel(const __half *__restrict__ bias,
                                                              __half *__restrict__ out, ConvDims d,
                                                              int g, TailPlan tp) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int cout_g = d.Cout / d.G, HoWo = d.Ho * d.Wo, N = d.B * HoWo;
  const int ntile = tp.main_tiles + blockIdx.x;
  constexpr int kTh = Glds<WN>::kThreads;
  const int n0 = ntile * Glds<WN>::kN, m0 = blockIdx.y * kFM;
  const int quad = blockIdx.z, i = quad >> 2, rq = quad & 3;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int part = 0; part < tp.split; ++part) {
    const float4 v = reinterpret_cast<const float4 *>(tp.partial)[
        ((((size_t)part * tp.tail_tiles + blockIdx.x) * gridDim.y + blockIdx.y) * 8 + quad) * kTh + tid];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  const int n = n0 + wn * 32 + (lane & 31);
  if (n >= N) return;
  const int b = n / HoWo, pix = n - b * HoWo;
  const float av[4] = {a.x, a.y, a.z, a.w};
  dcn_store4(out, bias, av, b, pix, m0 + wm * 64 + i * 32 + 8 * rq + 4 * (lane >> 5), g, cout_g, d.Cout, HoWo, tp);
}


thread_local bool g_mdconv_old_copy = false;   // variant 12: the r01 NCHW -> NHWC copy kernel (A/B)

template <int WN>
int glds_resident_blocks() {
  static thread_local int cached_dev = -1, cached = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev == cached_dev) return cached;
  int per_cu = 0, cus = 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_glds_f16_kernel<WN>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, Glds<WN>::kLds) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dcn_glds_f16_kernel<WN>, Glds<WN>::kThreads,
                                                   Glds<WN>::kLds) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return 0;
  cached_dev = dev;
  cached = per_cu * cus;
  return cached;
}

// launch of the LDS-DMA kernel with its tail plan
template <int WN>
int launch_glds(const __half *xt, const void *offset, const void *mask, const __half *wt, const void *bias,
                void *output, const ConvDims &d, int g, char *part_ws, size_t part_room, bool nhwc_io,
                bool relu, bool allow_tail, int om_channels, hipStream_t st) {
  const int KK = d.Kh * d.Kw, cout_g = d.Cout / d.G;
  const size_t N = (size_t)d.B * d.Ho * d.Wo;
  const dim3 grid((unsigned)((N + Glds<WN>::kN - 1) / Glds<WN>::kN), (cout_g + kFM - 1) / kFM);
  const int slots = glds_resident_blocks<WN>();
  if (slots <= 0) return BEVOPS_FAILURE;
  // tail plan: leftover tiles of a sparsely filled last round are split along K
  TailPlan tp{0, 1, (int)grid.x, nullptr, nhwc_io ? 1 : 0, relu ? 1 : 0, om_channels, g_mdconv_rotate};
  const int blocks = (int)(grid.x * grid.y);
  if (allow_tail && blocks > slots && grid.y == 1) {
    const int left = blocks % slots;
    int split = 0;
    for (int f = KK; f >= 2; --f)
      if (KK % f == 0 && left * f <= slots) { split = f; break; }
    const size_t need = (size_t)split * left * Glds<WN>::kThreads * 32 * sizeof(float);
    if (left > 0 && left * 2 <= slots && split >= 2 && part_room >= need) {
      tp.tail_tiles = left;
      tp.split = split;
      tp.main_tiles = (int)grid.x - left;
      tp.partial = reinterpret_cast<float *>(part_ws);
    }
  }
  const dim3 grid2((unsigned)(tp.tail_tiles * tp.split + tp.main_tiles), grid.y);
  // tp.rotate: 0 = round 6's order (SCHED 4), 1 = the round-2 rotation, 2 = one order for all waves (rounds 2-5)
  auto kern = tp.rotate == 0 ? dcn_glds_f16_kernel<WN, 4> : dcn_glds_f16_kernel<WN, 0>;
  if (tp.rotate == 0) {
    static thread_local int lds_set[2] = {0, 0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (lds_set[WN == 4] != dev + 1) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              Glds<WN>::kLds) != hipSuccess)
        return BEVOPS_FAILURE;
      lds_set[WN == 4] = dev + 1;
    }
  }
  hipLaunchKernelGGL(kern, grid2, dim3(Glds<WN>::kThreads), Glds<WN>::kLds, st, xt,
                     (const __half *)offset, (const __half *)mask, wt, (const __half *)bias, (__half *)output, d,
                     g, tp);
  if (tp.tail_tiles)
    hipLaunchKernelGGL(dcn_tail_finish_kernel<WN>, dim3((unsigned)tp.tail_tiles, grid.y, 8),
                       dim3(Glds<WN>::kThreads), 0, st, (const __half *)bias, (__half *)output, d, g, tp);
  return launch_status();
}

template <typename T>
int run(const void *input, const void *offset, const void *mask, const void *weight,
        const void *bias, void *output, void *workspace, const ConvDims &d, hipStream_t st,
        bool weight_is_packed = false, bool nhwc_io = false, bool relu = false, int om_channels = 0) {
  const WsLayout w = ws_layout(d, sizeof(T));
  char *ws = static_cast<char *>(workspace);
  T *xt = reinterpret_cast<T *>(ws + w.xt);
  T *wt = reinterpret_cast<T *>(ws + w.wt);
  T *col = reinterpret_cast<T *>(ws + w.col);
  const int HW = d.H * d.W, KK = d.Kh * d.Kw, cin_g = d.Cin / d.G, cout_g = d.Cout / d.G;
  const size_t N = (size_t)d.B * d.Ho * d.Wo;
  if (N > 0x7FFFFFFFull || (size_t)d.B * d.Cin * HW > 0x7FFFFFFF00ull) return BEVOPS_NOT_SUPPORTED;
  const bool fits32 = (size_t)d.B * d.Cin * HW * 2 < 0xFFFFFF00ull && (size_t)d.Cout * cin_g * KK * 2 < 0xFFFFFF00ull;
  if (nhwc_io)  // the caller's tensor already is the [B, H, W, Cin] image the gather wants
    xt = const_cast<T *>(static_cast<const T *>(input));
  else if (sizeof(T) == 2 && HW % 4 == 0 && d.Cin % 8 == 0 && aligned16(input) && !g_mdconv_old_copy)
    hipLaunchKernelGGL(nchw_to_nhwc_f16q_kernel, dim3((HW + 127) / 128, (d.Cin + 63) / 64, d.B), dim3(256), 0, st,
                       (const __half *)input, (__half *)xt, d.Cin, HW);
  else if (sizeof(T) == 2 && HW % 8 == 0 && d.Cin % 8 == 0 && aligned16(input))   // variant 12: the r01 copy kernel (A/B)
    hipLaunchKernelGGL(nchw_to_nhwc_f16v_kernel, dim3((HW + 63) / 64, (d.Cin + 63) / 64, d.B), dim3(256), 0, st,
                       (const __half *)input, (__half *)xt, d.Cin, HW);
  else
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3((HW + 31) / 32, (d.Cin + 31) / 32, d.B), dim3(256),
                       0, st, (const T *)input, xt, d.Cin, HW);
  const size_t wtot = (size_t)d.Cout * cin_g * KK;
  if (weight_is_packed)  // [Cout][tap][Cin/groups] image made by bevops_mdconv_pack_weight
    wt = const_cast<T *>(static_cast<const T *>(weight));
  else
    hipLaunchKernelGGL((repack_weight_kernel<T>), dim3((unsigned)((wtot + 255) / 256)), dim3(256), 0, st,
                       (const T *)weight, wt, d.Cout, cin_g, KK, KK * cin_g);
  if constexpr (sizeof(T) == 2) {
    // fused implicit GEMM: a 64-channel K chunk must sit inside one group and one deform group
    if (g_mdconv_variant != 1 && fits32 && cin_g % kFK == 0 && (d.Cin / d.DG) % kFK == 0) {
      const dim3 grid((unsigned)((N + kFN - 1) / kFN), (cout_g + kFM - 1) / kFM);
      for (int g = 0; g < d.G; ++g) {
        const bool one_dg = cin_g <= d.Cin / d.DG && (g * cin_g) / (d.Cin / d.DG) == (g * cin_g + cin_g - 1) / (d.Cin / d.DG);
        if ((nhwc_io || relu || om_channels) && !(g_mdconv_variant == 0 && one_dg)) return BEVOPS_NOT_SUPPORTED;
        if (g_mdconv_variant == 0 && one_dg) {
          // 128-pixel tiles (weights fetched once per 128 pixels) when they still give every CU a block
          int cus = 0, dev = 0;
          if (hipGetDevice(&dev) != hipSuccess ||
              hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return BEVOPS_FAILURE;
          const size_t wide_blocks = ((N + Glds<4>::kN - 1) / Glds<4>::kN) * ((cout_g + kFM - 1) / kFM);
          // ... or at least every second CU: base stage 4 (136 blocks of 128 pixels) 105 us vs 118 us with
          // 272 blocks of 64 pixels (profiles/r02/dcn_time.jsonl) -- 16 waves on half the CUs beat 8 waves on all
          const bool wide = g_mdconv_wide || (!g_mdconv_no_tail && wide_blocks * 2 >= (size_t)cus);
          const int rc = wide
                             ? launch_glds<4>((const __half *)xt, offset, mask, (const __half *)wt, bias, output, d, g,
                                              ws + w.col, w.total - w.col, nhwc_io, relu, !g_mdconv_no_tail, om_channels, st)
                             : launch_glds<2>((const __half *)xt, offset, mask, (const __half *)wt, bias, output, d, g,
                                              ws + w.col, w.total - w.col, nhwc_io, relu, !g_mdconv_no_tail, om_channels, st);
          if (rc != BEVOPS_SUCCESS) return rc;
        } else if (g_mdconv_variant == 2)  // A/B: the 4-wave block of r01c
          hipLaunchKernelGGL(dcn_fused_f16_kernel<256>, grid, dim3(256), 0, st, (const __half *)xt,
                             (const __half *)offset, (const __half *)mask, (const __half *)wt,
                             (const __half *)bias, (__half *)output, d, g);
        else
          hipLaunchKernelGGL(dcn_fused_f16_kernel<512>, grid, dim3(512), 0, st, (const __half *)xt,
                             (const __half *)offset, (const __half *)mask, (const __half *)wt,
                             (const __half *)bias, (__half *)output, d, g);
      }
      return launch_status();
    }
  }
  if (nhwc_io || relu || om_channels) return BEVOPS_NOT_SUPPORTED;  // only the fused fp16 kernel has these
  constexpr int VMAX = sizeof(T) == 2 ? 8 : 4;
  const bool vec = cin_g % VMAX == 0 && (d.Cin / d.DG) % VMAX == 0;
  {
    const int V = vec ? VMAX : 1;
    const size_t threads = N * KK * (d.Cin / V);
    const size_t blocks = (threads + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return BEVOPS_NOT_SUPPORTED;
    if (vec)
      hipLaunchKernelGGL((im2col_nhwc_kernel<T, VMAX>), dim3((unsigned)blocks), dim3(256), 0, st, xt,
                         (const T *)offset, (const T *)mask, col, d);
    else
      hipLaunchKernelGGL((im2col_nhwc_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, st, xt,
                         (const T *)offset, (const T *)mask, col, d);
  }
  const int Kg = KK * cin_g;
  for (int g = 0; g < d.G; ++g) {
    const GemmEpi e{d.Ho * d.Wo, d.Cout, g * cout_g};
    const T *Ag = wt + (size_t)g * cout_g * Kg;
    const T *Bg = col + (size_t)g * N * Kg;
    if constexpr (sizeof(T) == 2) {
      if (Kg % 8 == 0) {
        hipLaunchKernelGGL(gemm_tn_f16_kernel, dim3((unsigned)((N + kBN - 1) / kBN), (cout_g + kBM - 1) / kBM),
                           dim3(256), 0, st, (const __half *)Ag, (const __half *)Bg, (const __half *)bias,
                           (__half *)output, cout_g, (int)N, Kg, e);
        continue;
      }
      const size_t outs = (size_t)cout_g * N;  // ragged K: scalar fallback
      if ((outs + 255) / 256 > 0x7FFFFFFFull) return BEVOPS_NOT_SUPPORTED;
      hipLaunchKernelGGL(gemm_tn_f16_ragged_kernel, dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, st,
                         (const __half *)Ag, (const __half *)Bg, (const __half *)bias, (__half *)output, cout_g,
                         (int)N, Kg, e);
    } else {
      hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3((unsigned)((N + 63) / 64), (cout_g + 63) / 64), dim3(256),
                         0, st, (const float *)Ag, (const float *)Bg, (const float *)bias, (float *)output,
                         cout_g, (int)N, Kg, e);
    }
  }
  return launch_status();
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" int bevops_mdconv_set_variant(int variant) {
  const int prev = g_mdconv_no_tail ? 4 : (g_mdconv_wide ? 5 : g_mdconv_variant);
  g_mdconv_no_tail = variant == 4;
  g_mdconv_wide = variant == 5;
  g_mdconv_rotate = variant == 7 ? 1 : (variant == 13 ? 2 : 0);
  g_mdconv_old_copy = variant == 12;
  g_mdconv_variant = (variant == 4 || variant == 5 || variant == 7 || variant == 12 || variant == 13) ? 0 : variant;
  return prev;
}

extern "C" size_t bevops_mdconv_workspace_size(int dtype, int B, int Cin, int H, int W, int Cout,
                                               int Kh, int Kw, int stride_h, int stride_w,
                                               int pad_h, int pad_w, int dil_h, int dil_w,
                                               int groups, int deform_groups) {
  ConvDims d;
  if (!make_dims(d, B, Cin, H, W, Cout, Kh, Kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w,
                 groups, deform_groups))
    return 0;
  if (dtype != BEVOPS_F32 && dtype != BEVOPS_F16 && dtype != BEVOPS_I8) return 0;
  return ws_layout(d, dtype == BEVOPS_F32 ? 4 : (dtype == BEVOPS_F16 ? 2 : 1)).total;
}

extern "C" size_t bevops_mdconv_packed_weight_size(int dtype, int Cout, int Cin_per_group, int Kh,
                                                   int Kw) {
  if ((dtype != BEVOPS_F32 && dtype != BEVOPS_F16 && dtype != BEVOPS_I8) || Cout <= 0 || Cin_per_group <= 0 ||
      Kh <= 0 || Kw <= 0)
    return 0;
  const size_t kg = (size_t)Cin_per_group * Kh * Kw;
  if (dtype == BEVOPS_I8) return (size_t)Cout * ((kg + 15) & ~size_t(15));   // rows padded to 16 bytes (zero filled)
  return (size_t)Cout * kg * (dtype == BEVOPS_F32 ? 4 : 2);
}

extern "C" int bevops_mdconv_pack_weight(int dtype, const void *weight, void *packed, int Cout,
                                         int Cin_per_group, int Kh, int Kw, void *stream) {
  if (!weight || !packed) return BEVOPS_BAD_PARAM;
  if (bevops_mdconv_packed_weight_size(dtype, Cout, Cin_per_group, Kh, Kw) == 0) return BEVOPS_BAD_PARAM;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int KK = Kh * Kw;
  const size_t wtot = (size_t)Cout * Cin_per_group * KK;
  const dim3 grid((unsigned)((wtot + 255) / 256));
  if (dtype == BEVOPS_I8) {
    const int Kp = (KK * Cin_per_group + 15) & ~15;
    if (Kp != KK * Cin_per_group && hipMemsetAsync(packed, 0, (size_t)Cout * Kp, st) != hipSuccess) return BEVOPS_FAILURE;
    hipLaunchKernelGGL((repack_weight_kernel<int8_t>), grid, dim3(256), 0, st, (const int8_t *)weight,
                       (int8_t *)packed, Cout, Cin_per_group, KK, Kp);
  } else if (dtype == BEVOPS_F32)
    hipLaunchKernelGGL((repack_weight_kernel<float>), grid, dim3(256), 0, st, (const float *)weight,
                       (float *)packed, Cout, Cin_per_group, KK, KK * Cin_per_group);
  else
    hipLaunchKernelGGL((repack_weight_kernel<__half>), grid, dim3(256), 0, st, (const __half *)weight,
                       (__half *)packed, Cout, Cin_per_group, KK, KK * Cin_per_group);
  return launch_status();
}

static int mdconv_forward_impl(int dtype, const void *input, const void *offset, const void *mask,
                               const void *weight, const void *bias, void *output, void *workspace,
                               size_t workspace_bytes, int B, int Cin, int H, int W, int Cout, int Kh,
                               int Kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h,
                               int dil_w, int groups, int deform_groups, void *stream, bool packed,
                               bool nhwc_io = false, bool relu = false, int om_channels = 0) {
  if (!input || !offset || (!mask && !om_channels) || !weight || !output || !workspace) return BEVOPS_BAD_PARAM;
  if (om_channels && (om_channels < deform_groups * 3 * Kh * Kw || (om_channels & 1))) return BEVOPS_BAD_PARAM;
  ConvDims d;
  if (!make_dims(d, B, Cin, H, W, Cout, Kh, Kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w,
                 groups, deform_groups))
    return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F32 && dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (workspace_bytes < ws_layout(d, dtype == BEVOPS_F32 ? 4 : 2).total) return BEVOPS_BAD_PARAM;
  if (!aligned16(workspace) || (packed && !aligned16(weight))) return BEVOPS_BAD_PARAM;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == BEVOPS_F32)
    return run<float>(input, offset, mask, weight, bias, output, workspace, d, st, packed, nhwc_io, relu, om_channels);
  return run<__half>(input, offset, mask, weight, bias, output, workspace, d, st, packed, nhwc_io, relu, om_channels);
}

extern "C" int bevops_mdconv_forward_nhwc(int dtype, const void *input_nhwc, const void *offset,
                                          const void *mask, const void *packed_weight,
                                          const void *bias, void *output_nhwc, int relu,
                                          int offset_mask_channels, void *workspace,
                                          size_t workspace_bytes, int B, int Cin,
                                          int H, int W, int Cout, int Kh, int Kw, int stride_h,
                                          int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                                          int groups, int deform_groups, void *stream) {
  if (dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (input_nhwc && !aligned16(input_nhwc)) return BEVOPS_BAD_PARAM;
  return mdconv_forward_impl(dtype, input_nhwc, offset, mask, packed_weight, bias, output_nhwc,
                             workspace, workspace_bytes, B, Cin, H, W, Cout, Kh, Kw, stride_h,
                             stride_w, pad_h, pad_w, dil_h, dil_w, groups, deform_groups, stream, true,
                             true, relu != 0, offset_mask_channels);
}

extern "C" int bevops_mdconv_forward_packed(int dtype, const void *input, const void *offset,
                                            const void *mask, const void *packed_weight,
                                            const void *bias, void *output, void *workspace,
                                            size_t workspace_bytes, int B, int Cin, int H, int W,
                                            int Cout, int Kh, int Kw, int stride_h, int stride_w,
                                            int pad_h, int pad_w, int dil_h, int dil_w, int groups,
                                            int deform_groups, void *stream) {
  return mdconv_forward_impl(dtype, input, offset, mask, packed_weight, bias, output, workspace,
                             workspace_bytes, B, Cin, H, W, Cout, Kh, Kw, stride_h, stride_w, pad_h,
                             pad_w, dil_h, dil_w, groups, deform_groups, stream, true);
}

extern "C" int bevops_mdconv_forward(int dtype, const void *input, const void *offset,
                                     const void *mask, const void *weight, const void *bias,
                                     void *output, void *workspace, size_t workspace_bytes, int B,
                                     int Cin, int H, int W, int Cout, int Kh, int Kw, int stride_h,
                                     int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                                     int groups, int deform_groups, void *stream) {
  return mdconv_forward_impl(dtype, input, offset, mask, weight, bias, output, workspace, workspace_bytes,
                             B, Cin, H, W, Cout, Kh, Kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w,
                             groups, deform_groups, stream, false);
}
