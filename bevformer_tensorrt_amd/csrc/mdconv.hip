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
#include <type_traits>

#include "common.h"

namespace bevops {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvDims {
  int B, Cin, H, W, Cout, Kh, Kw, sh, sw, ph, pw, dh, dw, G, DG, Ho, Wo;
};

template <typename T> __device__ __forceinline__ float tof(T v);
template <> __device__ __forceinline__ float tof<float>(float v) { return v; }
template <> __device__ __forceinline__ float tof<__half>(__half v) { return __half2float(v); }
template <typename T> __device__ __forceinline__ T fromf(float v);
template <> __device__ __forceinline__ float fromf<float>(float v) { return v; }
template <> __device__ __forceinline__ __half fromf<__half>(float v) { return __float2half_rn(v); }

// ---- 1. NCHW -> NHWC ------------------------------------------------------------
// `flip` (int8 only): XOR applied to every byte -- 0x80 makes the copy hold v + 128 as u8 (the
// fused int8 kernel's operand form, dcn_fused_s8_kernel)
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const T *__restrict__ in,
                                                           T *__restrict__ out, int C, int HW, int flip = 0) {
  __shared__ T tile[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const T *ib = in + (size_t)b * C * HW;
  T *ob = out + (size_t)b * C * HW;
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int c = c0 + ty + r, p = p0 + tx;
    if (c < C && p < HW) {
      if constexpr (sizeof(T) == 1) tile[ty + r][tx] = (T)(ib[(size_t)c * HW + p] ^ (T)flip);
      else tile[ty + r][tx] = ib[(size_t)c * HW + p];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const int p = p0 + ty + r, c = c0 + tx;
    if (c < C && p < HW) ob[(size_t)p * C + c] = tile[tx][ty + r];
  }
}

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

// int8, HW % 4 == 0 and C % 16 == 0: 128 channels x 128 pixels per block.  A thread loads a 4 x 4 byte block
// (4 pixels of 4 channel rows, lanes along the pixels: 128 contiguous bytes per row and half-wave), transposes
// it in registers and writes 4 dwords (4 channels of one pixel each) into the [pixel][channel] tile; the tile
// leaves as 16-byte vectors, 128 contiguous bytes per pixel.  (The byte-wise 32 x 32 kernel above took 12 us
// for the 8.9 MB stage-3 image -- longer than the fp16 copy of twice the bytes.)
__global__ __launch_bounds__(256) void nchw_to_nhwc_s8v_kernel(const int8_t *__restrict__ in, int8_t *__restrict__ out,
                                                               int C, int HW, unsigned flip4) {
  constexpr int kRow = 128 + 16;   // bytes per tile row (16-byte aligned rows for the b128 reads)
  __shared__ __attribute__((aligned(16))) unsigned char tile[128 * kRow];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 128, c0 = blockIdx.y * 128;
  const int8_t *ib = in + (size_t)b * C * HW;
  int8_t *ob = out + (size_t)b * C * HW;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = threadIdx.x + it * 256;      // 32 channel quads x 32 pixel quads
    const int pq = v & 31, cq = v >> 5;
    const int p = p0 + pq * 4, c = c0 + cq * 4;
    unsigned r[4] = {0u, 0u, 0u, 0u};
    if (p < HW && c < C) {                      // HW % 4 == 0, C % 4 == 0: the 4 x 4 block is all in or all out
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = *reinterpret_cast<const unsigned *>(ib + (size_t)(c + k) * HW + p) ^ flip4;
    }
    // r[k] = 4 pixels of channel c + k  ->  o[j] = 4 channels of pixel p + j
    const unsigned a = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u), e = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);
    const unsigned f = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u), g = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
    const unsigned o[4] = {__builtin_amdgcn_perm(f, a, 0x05040100u), __builtin_amdgcn_perm(f, a, 0x07060302u),
                           __builtin_amdgcn_perm(g, e, 0x05040100u), __builtin_amdgcn_perm(g, e, 0x07060302u)};
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<unsigned *>(&tile[(pq * 4 + j) * kRow + cq * 4]) = o[j];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = threadIdx.x + it * 256;      // 128 pixels x 8 chunks of 16 channels
    const int ch = v & 7, pl = v >> 3;
    if (p0 + pl < HW && c0 + ch * 16 < C)
      *reinterpret_cast<uint4 *>(ob + (size_t)(p0 + pl) * C + c0 + ch * 16) =
          *reinterpret_cast<const uint4 *>(&tile[pl * kRow + ch * 16]);
  }
}

// ---- 2. weight [Cout][Cin/g][KK] -> [Cout][KK][Cin/g] ------------------------------
template <typename T>
__global__ __launch_bounds__(256) void repack_weight_kernel(const T *__restrict__ w,
                                                            T *__restrict__ wt, int Cout,
                                                            int cin_g, int KK, int row_stride) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)Cout * cin_g * KK;
  if (i >= total) return;
  const int ci = (int)(i % cin_g);
  const int t = (int)((i / cin_g) % KK);
  const size_t co = i / ((size_t)cin_g * KK);
  wt[co * row_stride + (size_t)t * cin_g + ci] = w[(co * cin_g + ci) * KK + t];
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
constexpr int kBM = 128, kBN = 128, kBK = 32, kLd = kBK + 8;  // +8 halves: conflict-free b128 reads

struct GemmEpi {
  int HoWo, Cout, co0;  // out[(n / HoWo) * Cout + co0 + m][n % HoWo]
};

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
constexpr int kFM = 256, kFN = 64, kFK = 64, kFLd = kFK + 8;
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
constexpr int kGA = kFM * kFK * 2;  // 32 KB weight tile image
// WN = wave columns (each 32 pixels): 2 -> 512 threads, 64-pixel tile, 80 KB LDS, 2 blocks per CU;
//                                      4 -> 1024 threads, 128-pixel tile, 96 KB LDS, 1 block per CU
//                                           (the weight tile is fetched once per 128 pixels)
template <int WN> struct Glds {
  static constexpr int kN = 32 * WN;          // pixels per tile
  static constexpr int kThreads = 256 * WN;
  static constexpr int kB = kN * kFK * 2;     // pixel tile image bytes
  static constexpr int kLds = 2 * (kGA + kB);
  static constexpr int kPieces = 32 / (4 * WN);  // 1 KB weight DMA pieces per wave
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned swz8(unsigned r) { return (r ^ (r >> 3)) & 7u; }

// Tail handling: when the tile count leaves a sparsely filled last round (base stage 3: 544 tiles
// on 512 resident blocks -- a round costs the same ~36 dependent k-steps however few blocks run),
// the leftover `tail_tiles` tiles are split `split` ways along K (whole taps) into short blocks
// that come FIRST in the grid; each writes its fp32 partial accumulators to the workspace and
// dcn_tail_finish_kernel adds them in a fixed order (deterministic, no atomics).
struct TailPlan {
  int tail_tiles, split, main_tiles;  // grid.x = tail_tiles * split + main_tiles
  float *partial;                     // [split][tail_tiles][Cout tiles][8 quads][threads] float4
  int out_nhwc, relu;                 // epilogue: output layout [B,Ho,Wo,Cout], fused ReLU
  int om_channels;                    // > 0: `offset` is the raw [B,Ho,Wo,om_channels] output of the pack's
                                      // offset convolution (2*KK offsets, then KK mask logits): sigmoid here
  int rotate;                         // fp16 kernel: wave halves in opposite phase order (A/B switch, variant 7;
                                      // measured 4-6 % slower at both ResNet-101 shapes, profiles/r02)
};

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

template <int WN>
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
  unsigned a_off[kPc];
#pragma unroll
  for (int j = 0; j < kPc; ++j) {
    const unsigned row = (unsigned)((wave * kPc + j) * 8 + (lane >> 3));
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
    char *adst = smem + buf * kGA + wave * (kPc * 1024);
#pragma unroll
    for (int j = 0; j < kPc; ++j)
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

  // prologue: step 0 -> buffer 0, corners of step 1 in flight
  gather_next();
  blend_store(0);
  __builtin_amdgcn_sched_barrier(0);
  weights_next(0);
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
  const bool late = tp.rotate && __builtin_amdgcn_readfirstlane(wave) >= kTh / 128;
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


// ---- 6. INT8 flavour (modulatedDeformableConv2dKernel.cu:190-257,463-607,897-978) ----------
// im2col on the NHWC int8 image (16 channels per lane = one 16-byte load per corner; unsigned
// x255 area weights, int32 4-corner dot, T2int8(t/255), then T2int8(val * mask)), one batched
// int8 GEMM on the matrix cores (v_mfma_i32_32x32x32_i8, int32 accumulate), epilogue
// T2int8((acc * s_in*s_w + bias) / s_out) scattered to NCHW.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int q_away(float a) {
  a = fminf(fmaxf(a, -128.f), 127.f);
  return (int)(a + (a > 0.f ? 0.5f : -0.5f));
}
__device__ __forceinline__ int u8w(float a) { return (int)fminf(fmaxf(rintf(a * 255.f), 0.f), 255.f); }

template <int V>
__global__ __launch_bounds__(256) void im2col_nhwc_s8_kernel(const int8_t *__restrict__ xt,
                                                             const int8_t *__restrict__ offset,
                                                             const int8_t *__restrict__ mask,
                                                             int8_t *__restrict__ col, ConvDims d,
                                                             float s_off, float s_mask, int kp) {
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
  const size_t obase = (((size_t)b * d.DG + dg) * 2 * KK) * HoWo + pix;
  float h_im, w_im, m;
  {
#pragma clang fp contract(off)
    const float off_h = (float)offset[obase + (size_t)(2 * t) * HoWo] * s_off;
    const float off_w = (float)offset[obase + (size_t)(2 * t + 1) * HoWo] * s_off;
    m = (float)mask[(((size_t)b * d.DG + dg) * KK + t) * HoWo + pix] * s_mask;
    h_im = off_h + (float)(ho * d.sh - d.ph + i * d.dh);
    w_im = off_w + (float)(wo * d.sw - d.pw + j * d.dw);
  }
  int acc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = 0;
  const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)d.H && w_im < (float)d.W;
  if (in) {
#pragma clang fp contract(off)
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int h0 = (int)hf, w0 = (int)wf;
    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
    const int aw[4] = {u8w(hh * hw), u8w(hh * lw), u8w(lh * hw), u8w(lh * lw)};
    const bool ok[4] = {h0 >= 0 && w0 >= 0, h0 >= 0 && w0 + 1 < d.W, h0 + 1 < d.H && w0 >= 0,
                        h0 + 1 < d.H && w0 + 1 < d.W};
    const int hs[4] = {h0, h0, h0 + 1, h0 + 1}, ws[4] = {w0, w0 + 1, w0, w0 + 1};
    const int8_t *xb = xt + (size_t)b * d.H * d.W * d.Cin + c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!ok[q]) continue;
      const int8_t *p = xb + ((size_t)hs[q] * d.W + ws[q]) * d.Cin;
      int8_t v[V];
      if constexpr (V == 16) *reinterpret_cast<uint4 *>(v) = *reinterpret_cast<const uint4 *>(p);
      else if constexpr (V == 4) *reinterpret_cast<unsigned *>(v) = *reinterpret_cast<const unsigned *>(p);
      else v[0] = p[0];
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += (int)v[k] * aw[q];
    }
  }
  int8_t res[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
#pragma clang fp contract(off)
    const int val = in ? q_away((float)acc[k] * (1 / 255.f)) : 0;
    res[k] = (int8_t)q_away((float)val * m);
  }
  const int cin_g = d.Cin / d.G;
  const int g = c / cin_g, cg = c - g * cin_g;
  int8_t *o = col + ((size_t)g * N + n) * kp + (size_t)t * cin_g + cg;
  if constexpr (V == 16) *reinterpret_cast<uint4 *>(o) = *reinterpret_cast<const uint4 *>(res);
  else if constexpr (V == 4) *reinterpret_cast<unsigned *>(o) = *reinterpret_cast<const unsigned *>(res);
  else o[0] = res[0];
}

// C[m][n] = sum_k A[m][k] * B[n][k] on int8, 128x128x64 tiles, requantising epilogue
constexpr int kIK = 64, kILd = kIK + 16;  // bytes per LDS row (+16: conflict-free b128 reads)
__global__ __launch_bounds__(256) void gemm_tn_s8_kernel(const int8_t *__restrict__ A,
                                                         const int8_t *__restrict__ Bm,
                                                         const float *__restrict__ bias,
                                                         int8_t *__restrict__ out, int M, int N, int K,
                                                         GemmEpi e, float s_iw, float s_out) {
  __shared__ __attribute__((aligned(16))) int8_t As[kBM][kILd];
  __shared__ __attribute__((aligned(16))) int8_t Bs[kBN][kILd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
  const int r0 = tid >> 2, kc = (tid & 3) * 16, r1 = r0 + 64;  // 128 rows x 4 chunks of 16 B
  i32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
  const int nk = (K + kIK - 1) / kIK;
  uint4 ra0, ra1, rb0, rb1;
  auto ld = [&](const int8_t *p, bool ok) { return ok ? *reinterpret_cast<const uint4 *>(p) : make_uint4(0, 0, 0, 0); };
  auto gload = [&](int kt) {
    const int k = kt * kIK + kc;
    const bool kok = k < K;  // K % 16 == 0 guaranteed by the launcher
    ra0 = ld(A + (size_t)(m0 + r0) * K + k, kok && m0 + r0 < M);
    ra1 = ld(A + (size_t)(m0 + r1) * K + k, kok && m0 + r1 < M);
    rb0 = ld(Bm + (size_t)(n0 + r0) * K + k, kok && n0 + r0 < N);
    rb1 = ld(Bm + (size_t)(n0 + r1) * K + k, kok && n0 + r1 < N);
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
      const int kk = ks * 32 + (lane >> 5) * 16;
      i32x4_t a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const i32x4_t *>(&As[wm * 64 + i * 32 + (lane & 31)][kk]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = *reinterpret_cast<const i32x4_t *>(&Bs[wn * 64 + j * 32 + (lane & 31)][kk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + (lane & 31);
    if (n >= N) continue;
    const int b = n / e.HoWo, pix = n - b * e.HoWo;
    int8_t *ob = out + ((size_t)b * e.Cout + e.co0) * e.HoWo + pix;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M) {
#pragma clang fp contract(off)
          const float v = ((float)acc[i][j][r] * s_iw + (bias ? bias[e.co0 + m] : 0.f)) / s_out;
          ob[(size_t)m * e.HoWo] = (int8_t)q_away(v);
        }
      }
  }
}

// ---- 6b. fused deformable implicit GEMM, int8 ------------------------------------------------
// The int8 flavour of section 5: the column element T2int8(T2int8(sum_corners v a / 255) * mask)
// (modulatedDeformableConv2dKernel.cu:463-548) is produced by the B-tile loader straight into
// LDS and consumed by v_mfma_i32_32x32x32_i8 -- no 80 MB column buffer written and re-read per
// stage-3 call.  Block tile 256 (Cout) x 64 pixels x 64 (one tap, 64 input channels); 4 waves of
// 64 x 64 outputs.  Producer thread = (pixel, 16-channel quarter): four 16-byte corner loads from
// the u8-BIASED channels-last copy (v + 128), 4x4 byte transposes, per channel ONE unsigned dot4
// whose addend -128 * sum(a) removes the bias, the exact integer T2int8(. / 255) of msda_hm4.hip
// (saturating v_mad_i32_i24, result in the top byte; hand-placed: DOT -> other VALU hazard), then
// the reference's float multiply by the mask and round-half-away, 16 result bytes = one
// ds_write_b128.  Out-of-range corners get area weight 0 (so they leave the dot and the bias sum).
// Bit-identical to im2col_nhwc_s8_kernel + gemm_tn_s8_kernel (tests/test_mdconv_gpu.py).
constexpr int kSM = 256, kSN = 64, kSK = 64, kSLd = kSK + 16;

__device__ __forceinline__ void s8_quad(const unsigned (&v)[4], unsigned aw, int neg, int magic, int half,
                                        int (&x)[4]) {
  asm("v_dot4_u32_u8 %0, %4, %8, %9\n\t"
      "v_dot4_u32_u8 %1, %5, %8, %9\n\t"
      "v_dot4_u32_u8 %2, %6, %8, %9\n\t"
      "v_dot4_u32_u8 %3, %7, %8, %9\n\t"
      "v_mad_i32_i24 %0, %0, %10, %11 clamp\n\t"
      "v_mad_i32_i24 %1, %1, %10, %11 clamp\n\t"
      "v_mad_i32_i24 %2, %2, %10, %11 clamp\n\t"
      "v_mad_i32_i24 %3, %3, %10, %11 clamp"
      : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(aw), "v"(neg), "v"(magic), "v"(half));
}
// 4x4 byte transpose: r[k] = 4 channels of corner k  ->  o[c] = channel c of corners 0..3
__device__ __forceinline__ void s8_transpose(unsigned r0, unsigned r1, unsigned r2, unsigned r3, unsigned (&o)[4]) {
  const unsigned a = __builtin_amdgcn_perm(r1, r0, 0x05010400u);
  const unsigned b = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
  const unsigned c = __builtin_amdgcn_perm(r3, r2, 0x05010400u);
  const unsigned e = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
  o[0] = __builtin_amdgcn_perm(c, a, 0x05040100u);
  o[1] = __builtin_amdgcn_perm(c, a, 0x07060302u);
  o[2] = __builtin_amdgcn_perm(e, b, 0x05040100u);
  o[3] = __builtin_amdgcn_perm(e, b, 0x07060302u);
}

// Schedule of a k-step (one set of operand registers, 168 VGPRs = 3 blocks per CU, so the 544 tiles
// of the base stage-3 call run in ONE round; with double-buffered operands the kernel needs 200
// VGPRs = 512 slots and the 32 left-over tiles cost a second round -- measured 124 us, no better
// than the im2col + GEMM pair, profiles/r02):
//   corners of this step arrive -> 4 x (byte transpose + dot / requantise)      [corner registers free]
//   -> gathers of the NEXT step issued -> mask multiply + rounding of the 16 values (~100 VALU)
//   -> barrier, weight + pixel tile to LDS, barrier -> weight loads of the next step issued -> MFMAs.
// A gather's round trip hides behind ~100 VALU instructions, two barriers and the 8 MFMAs; the
// weight loads (L2 hits, coalesced) behind the MFMAs and the next step's front half.
__global__ __launch_bounds__(256, 3) void dcn_fused_s8_kernel(
    const int8_t *__restrict__ xt, const int8_t *__restrict__ offset, const int8_t *__restrict__ mask,
    const int8_t *__restrict__ wt, const float *__restrict__ bias, int8_t *__restrict__ out, ConvDims d, int g,
    int Kp, float s_off, float s_mask, float s_iw, float s_out) {
  __shared__ __attribute__((aligned(16))) int8_t As[kSM][kSLd];
  __shared__ __attribute__((aligned(16))) int8_t Bs[kSN][kSLd];
  constexpr int kOmMax = 64;
  __shared__ int8_t Om[kOmMax][kSN];
  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int KK = d.Kh * d.Kw, cin_g = d.Cin / d.G, cout_g = d.Cout / d.G;
  const int HoWo = d.Ho * d.Wo;
  const int N = d.B * HoWo;
  const int n0 = (int)xcd_remap(blockIdx.x, gridDim.x) * kSN, m0 = blockIdx.y * kSM;
  const int8_t *A = wt + (size_t)g * cout_g * Kp;
  // B-producer role: pixel n0 + tid / 4, channels [cq * 16, +16) of the 64-channel chunk
  const int pp = tid >> 2, cq = tid & 3;
  const int pn = n0 + pp;
  const bool pvalid = pn < N;
  const int pb = pvalid ? pn / HoWo : 0;
  const int ppix = pvalid ? pn - pb * HoWo : 0;
  const int pho = ppix / d.Wo, pwo = ppix - pho * d.Wo;
  // A-loader role: rows (tid >> 2) + 64 i, 16-byte chunk (tid & 3)
  const int ar = tid >> 2, ac = (tid & 3) * 16;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(xt), 0, (unsigned)((size_t)d.B * d.H * d.W * d.Cin), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(A), 0, (unsigned)((size_t)cout_g * Kp), 0x00020000);
  const unsigned ximg_off = (unsigned)((size_t)pb * d.H * d.W * d.Cin + g * cin_g + cq * 16);
  unsigned a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = m0 + ar + 64 * i;
    a_off[i] = r < cout_g ? (unsigned)((size_t)r * Kp + ac) : 0xFFFFFFF0u;
  }
  i32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

  const int chunks = cin_g / kSK;
  const int nsteps = KK * chunks;
  int fidx[4];
  unsigned faw = 0;   // 4 packed u8 area weights (0 for out-of-range corners)
  int fneg = 0;       // -128 * their sum
  float fm = 0.f;     // mask value
  int cur_tap = -1, cur_dg = -1;
  uint4 ra[4], rb[4];

  const int om_per_dg = 3 * KK;
  const bool om_ok = d.DG * om_per_dg <= kOmMax;
  if (om_ok) {
    for (int idx = tid; idx < d.DG * om_per_dg * kSN; idx += 256) {
      const int p = idx % kSN, t = idx / kSN;
      const int dgi = t / om_per_dg, tt = t - dgi * om_per_dg;
      const int n = n0 + p;
      int8_t v = 0;
      if (n < N) {
        const int b = n / HoWo, pix = n - b * HoWo;
        v = tt < 2 * KK ? offset[(((size_t)b * d.DG + dgi) * 2 * KK + tt) * HoWo + pix]
                        : mask[(((size_t)b * d.DG + dgi) * KK + (tt - 2 * KK)) * HoWo + pix];
      }
      Om[t][p] = v;
    }
    __syncthreads();
  }

  // footprint of (pixel, tap): corner offsets, packed area weights, mask -- recomputed when the tap
  // (or the deform group) of `step` differs from the current one
  auto footprint = [&](int step) {
    const int tap = step / chunks, c0 = (step - tap * chunks) * kSK;
    const int dg = (g * cin_g + c0) / (d.Cin / d.DG);
    if (tap == cur_tap && dg == cur_dg) return;
    cur_tap = tap;
    cur_dg = dg;
    const int i = tap / d.Kw, j = tap - i * d.Kw;
    int qoh, qow, qm;
    if (om_ok) {
      qoh = Om[dg * om_per_dg + 2 * tap][pp];
      qow = Om[dg * om_per_dg + 2 * tap + 1][pp];
      qm = Om[dg * om_per_dg + 2 * KK + tap][pp];
    } else {
      const size_t ob = (((size_t)pb * d.DG + dg) * 2 * KK) * HoWo + ppix;
      qoh = offset[ob + (size_t)(2 * tap) * HoWo];
      qow = offset[ob + (size_t)(2 * tap + 1) * HoWo];
      qm = mask[(((size_t)pb * d.DG + dg) * KK + tap) * HoWo + ppix];
    }
    float h_im, w_im;
    {
#pragma clang fp contract(off)
      const float off_h = (float)qoh * s_off, off_w = (float)qow * s_off;
      fm = (float)qm * s_mask;
      h_im = off_h + (float)(pho * d.sh - d.ph + i * d.dh);
      w_im = off_w + (float)(pwo * d.sw - d.pw + j * d.dw);
    }
    const bool in = pvalid && h_im > -1.f && w_im > -1.f && h_im < (float)d.H && w_im < (float)d.W;
    unsigned aw[4] = {0u, 0u, 0u, 0u};
    int hs[4] = {0, 0, 0, 0}, ws[4] = {0, 0, 0, 0};
    if (in) {
#pragma clang fp contract(off)
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h0 = (int)hf, w0 = (int)wf;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool ok[4] = {h0 >= 0 && w0 >= 0, h0 >= 0 && w0 + 1 < d.W, h0 + 1 < d.H && w0 >= 0,
                          h0 + 1 < d.H && w0 + 1 < d.W};
      const int a4[4] = {u8w(hh * hw), u8w(hh * lw), u8w(lh * hw), u8w(lh * lw)};
      const int hq[4] = {h0, h0, h0 + 1, h0 + 1}, wq[4] = {w0, w0 + 1, w0, w0 + 1};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        aw[q] = ok[q] ? (unsigned)a4[q] : 0u;
        hs[q] = ok[q] ? hq[q] : 0;
        ws[q] = ok[q] ? wq[q] : 0;
      }
    }
    faw = aw[0] | (aw[1] << 8) | (aw[2] << 16) | (aw[3] << 24);
    fneg = -(int)((aw[0] + aw[1] + aw[2] + aw[3]) << 7);
#pragma unroll
    for (int q = 0; q < 4; ++q) fidx[q] = (int)(ximg_off + (unsigned)(hs[q] * d.W + ws[q]) * (unsigned)d.Cin);
  };
  auto gather = [&](int step) {
    const int tap = step / chunks, c0 = (step - tap * chunks) * kSK;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      rb[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, fidx[q], c0, 0));
  };
  auto weights = [&](int step) {
    const int tap = step / chunks, c0 = (step - tap * chunks) * kSK;
    const int a_s = tap * cin_g + c0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      ra[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)a_off[i], a_s, 0));
  };

  const int magic = 65793, half = 1 << 23;   // round(2^24 / 255): exact T2int8(t / 255), see msda_hm4.hip
  footprint(0);
  gather(0);
  weights(0);
  for (int step = 0; step < nsteps; ++step) {
    // 1. corners -> requantised bilinear sums of this thread's 16 channels (top bytes of x)
    int x[16];
    {
      const unsigned c0w[4] = {rb[0].x, rb[0].y, rb[0].z, rb[0].w}, c1w[4] = {rb[1].x, rb[1].y, rb[1].z, rb[1].w};
      const unsigned c2w[4] = {rb[2].x, rb[2].y, rb[2].z, rb[2].w}, c3w[4] = {rb[3].x, rb[3].y, rb[3].z, rb[3].w};
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        unsigned tr[4];
        s8_transpose(c0w[v], c1w[v], c2w[v], c3w[v], tr);
        int xq[4];
        s8_quad(tr, faw, fneg, magic, half, xq);
#pragma unroll
        for (int c = 0; c < 4; ++c) x[4 * v + c] = xq[c];
      }
    }
    const float m_cur = fm;
    __builtin_amdgcn_sched_barrier(0);
    // 2. the next step's gathers go out now (corner registers are free)
    if (step + 1 < nsteps) {
      footprint(step + 1);
      gather(step + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    // 3. mask multiply + round half away, pack 16 bytes
    unsigned res[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      int rq[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma clang fp contract(off)
        const float val = (float)(x[4 * v + c] >> 24);   // T2int8(sum / 255)
        rq[c] = q_away(val * m_cur);                     // T2int8(val * mask)
      }
      res[v] = ((unsigned)rq[0] & 0xffu) | (((unsigned)rq[1] & 0xffu) << 8) | (((unsigned)rq[2] & 0xffu) << 16) |
               ((unsigned)rq[3] << 24);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4 *>(&As[ar + 64 * i][ac]) = ra[i];
    *reinterpret_cast<uint4 *>(&Bs[pp][cq * 16]) = make_uint4(res[0], res[1], res[2], res[3]);
    __syncthreads();
    if (step + 1 < nsteps) weights(step + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kk = ks * 32 + (lane >> 5) * 16;
      i32x4_t a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const i32x4_t *>(&As[wm * 64 + i * 32 + (lane & 31)][kk]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = *reinterpret_cast<const i32x4_t *>(&Bs[j * 32 + (lane & 31)][kk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + j * 32 + (lane & 31);
    if (n >= N) continue;
    const int b = n / HoWo, pix = n - b * HoWo;
    int8_t *ob = out + ((size_t)b * d.Cout + g * cout_g) * HoWo + pix;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < cout_g) {
#pragma clang fp contract(off)
          const float v = ((float)acc[i][j][r] * s_iw + (bias ? bias[g * cout_g + m] : 0.f)) / s_out;
          ob[(size_t)m * HoWo] = (int8_t)q_away(v);
        }
      }
  }
}

// ---- 6c. the int8 flavour of section 5c: weights by LDS-DMA, two LDS buffers, ONE barrier per k-step ----
// The register-staged kernel above is latency-bound (two barriers, 64 B of ds_write_b128 per thread and a
// dependent gather round trip per 64-channel step: ~6 000 clk per step for ~2 400 clk of VALU work).  Here
// a k-step is one tap x 128 input channels = a 128-byte LDS row per weight row / pixel, i.e. exactly the
// LDS image, XOR swizzle, DMA piece mapping and fragment reads of dcn_glds_f16_kernel with
// v_mfma_i32_32x32x32_i8 consuming 32 bytes of k where the fp16 kernel consumes 16 halves.  Schedule of
// iteration `step` (tiles of `step` in buffer step & 1, raw corners of step + 1 in registers, a step old):
//   corners of step + 1 -> byte transposes, dots, mask multiply + rounding -> ds_write_b128 of the step + 1
//   pixel row | weight DMA of step + 1 -> other buffer | gathers of step + 2 issued | 8 MFMAs |
//   s_waitcnt vmcnt(4) (the DMA is older than the 4 gathers, which stay in flight across the barrier) |
//   barrier -- the upper half of the waves runs the MFMAs first and the producer work after them.
// The int32 partial sums of the split-K tail are exact in any order.  Bit-identical to the im2col + GEMM pair.
constexpr int kSOmRows = 32;   // 3 * Kh * Kw int8 rows of offsets / mask per pixel tile
template <int WN> struct GldsS8 {
  static constexpr int kLds = Glds<WN>::kLds + kSOmRows * Glds<WN>::kN;
};

// T2int8(T2int8(sum / 255) * mask) of 4 channels whose first requantisation sits in the top byte of x[c]
__device__ __forceinline__ unsigned s8_mask4(const int (&x)[4], float m) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  // (no SDWA byte-select convert here: hand-placed directly behind the v_mad_i32_i24 of s8_quad it read stale
  // registers -- the hazard recogniser does not see inline asm -- and it measured no faster where it was legal)
  float val[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) val[c] = (float)(x[c] >> 24);
  int q[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {   // two channels per packed fp32 multiply / add (separate roundings, as the reference)
    f32x2_t a;
    {
#pragma clang fp contract(off)
      a = f32x2_t{val[2 * h], val[2 * h + 1]} * f32x2_t{m, m};
    }
    const float ax = __builtin_amdgcn_fmed3f(a.x, -128.f, 127.f), ay = __builtin_amdgcn_fmed3f(a.y, -128.f, 127.f);
    // + copysign(0.5, a): a == 0 rounds to 0 with either sign, as (a > 0 ? 0.5 : -0.5) does.  (Scalars on
    // purpose: with the clamped values written back into the 2-vector, hipcc 7.2 emitted ONE v_bfi for the
    // pair and gave both lanes the first lane's sign -- caught by the bit-identity test.)
    const float hx = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, ax) & 0x80000000u) | 0x3f000000u);
    const float hy = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, ay) & 0x80000000u) | 0x3f000000u);
    q[2 * h] = (int)add_rn(ax, hx);
    q[2 * h + 1] = (int)add_rn(ay, hy);
  }
  const unsigned lo = __builtin_amdgcn_perm((unsigned)q[1], (unsigned)q[0], 0x0c0c0400u);   // bytes: q0, q1, 0, 0
  const unsigned hi = __builtin_amdgcn_perm((unsigned)q[3], (unsigned)q[2], 0x04000c0cu);   // bytes: 0, 0, q2, q3
  return lo | hi;
}

// ABL (timing experiments; 1..8 give wrong results): 1 no gathers, 2 no producer VALU, 4 no weight DMA, 8 no MFMA,
// 16 wave halves in opposite phase order (correct results; measured no faster: the kernel is VALU-bound)
template <int WN, int ABL>
__global__ __launch_bounds__(256 * WN, WN == 2 ? 2 : 1) void dcn_glds_s8_kernel(
    const int8_t *__restrict__ xt, const int8_t *__restrict__ offset, const int8_t *__restrict__ mask,
    const int8_t *__restrict__ wt, const float *__restrict__ bias, int8_t *__restrict__ out, ConvDims d, int g,
    int Kp, float s_off, float s_mask, float s_iw, float s_out, TailPlan tp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [A0][A1][B0][B1][Om]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int KK = d.Kh * d.Kw, cin_g = d.Cin / d.G, cout_g = d.Cout / d.G;
  const int HoWo = d.Ho * d.Wo;
  const int N = d.B * HoWo;
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
  constexpr int kStepK = 128;                      // input channels (= bytes) per k-step
  const int n0 = ntile * kN, m0 = blockIdx.y * kFM;
  const int8_t *A = wt + (size_t)g * cout_g * Kp;
  const int dg = (g * cin_g) / (d.Cin / d.DG);
  int8_t *Om = reinterpret_cast<int8_t *>(smem + Glds<WN>::kLds);   // [3 KK][kN]

  // pixel-producer role: pixel n0 + (tid >> 3), 16 channels cq of the 128-channel chunk
  const int pp = tid >> 3, cq = tid & 7;
  const int pn = n0 + pp;
  const bool pvalid = pn < N;
  const int pb = pvalid ? pn / HoWo : 0;
  const int ppix = pvalid ? pn - pb * HoWo : 0;
  const int pho = ppix / d.Wo, pwo = ppix - pho * d.Wo;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(xt), 0, (unsigned)((size_t)d.B * d.H * d.W * d.Cin), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(A), 0, (unsigned)((size_t)cout_g * Kp), 0x00020000);
  const unsigned ximg_off = (unsigned)((size_t)pb * d.H * d.W * d.Cin + g * cin_g + cq * 16);
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned b_lds = (unsigned)(size_t)((lds_char *)smem) + (unsigned)(2 * kGA) +
                         (unsigned)(pp * 128 + ((cq ^ swz8(pp)) << 4));   // LDS byte address of the thread's B chunk
  unsigned a_off[kPc];
#pragma unroll
  for (int j = 0; j < kPc; ++j) {
    const unsigned row = (unsigned)((wave * kPc + j) * 8 + (lane >> 3));
    const unsigned chunk = (lane & 7u) ^ swz8(row);
    a_off[j] = (m0 + (int)row) < cout_g ? (unsigned)((size_t)(m0 + row) * Kp + chunk * 16) : 0xFFFFFFF0u;
  }
  unsigned fa[2], fb;
#pragma unroll
  for (int i = 0; i < 2; ++i) fa[i] = (unsigned)(wm * 64 + i * 32 + (lane & 31));
  fb = (unsigned)(wn * 32 + (lane & 31));
  const unsigned hi = lane >> 5;

  i32x16_t acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0;

  // offsets / mask of the tile's pixels -> LDS, once
  for (int idx = tid; idx < 3 * KK * kN; idx += kTh) {
    const int p = idx % kN, t = idx / kN;
    const int n = n0 + p;
    int8_t v = 0;
    if (n < N) {
      const int b = n / HoWo, pix = n - b * HoWo;
      v = t < 2 * KK ? offset[(((size_t)b * d.DG + dg) * 2 * KK + t) * HoWo + pix]
                     : mask[(((size_t)b * d.DG + dg) * KK + (t - 2 * KK)) * HoWo + pix];
    }
    Om[t * kN + p] = v;
  }
  __syncthreads();

  const int chunks = cin_g / kStepK;
  const int taps_per_part = is_tail ? KK / tp.split : KK;
  const bool stagger = (size_t)cout_g * Kp <= (size_t)(2 << 20);
  const int tap_begin = is_tail ? part * taps_per_part : (stagger ? ntile % KK : 0);
  const int n_my_steps = taps_per_part * chunks;

  int fidx[4];
  unsigned f_aw = 0, c_aw = 0;   // packed u8 area weights of the gather / of the corners held in rb
  int f_neg = 0, c_neg = 0;      // -128 * their sum
  float f_m = 0.f, c_m = 0.f;    // mask value
  uint4 rb[4];
  if constexpr (ABL & 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) rb[q] = make_uint4(tid + q, tid * 3u, tid * 5u + q, tid * 7u);
  }
  auto footprint = [&](int tap) {
    const int i = tap / d.Kw, j = tap - i * d.Kw;
    const int qoh = Om[(2 * tap) * kN + pp], qow = Om[(2 * tap + 1) * kN + pp], qm = Om[(2 * KK + tap) * kN + pp];
    float h_im, w_im;
    {
#pragma clang fp contract(off)
      const float off_h = (float)qoh * s_off, off_w = (float)qow * s_off;
      f_m = (float)qm * s_mask;
      h_im = off_h + (float)(pho * d.sh - d.ph + i * d.dh);
      w_im = off_w + (float)(pwo * d.sw - d.pw + j * d.dw);
    }
    const bool in = pvalid && h_im > -1.f && w_im > -1.f && h_im < (float)d.H && w_im < (float)d.W;
    unsigned aw[4] = {0u, 0u, 0u, 0u};
    int hs[4] = {0, 0, 0, 0}, ws[4] = {0, 0, 0, 0};
    if (in) {
#pragma clang fp contract(off)
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h0 = (int)hf, w0 = (int)wf;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool ok[4] = {h0 >= 0 && w0 >= 0, h0 >= 0 && w0 + 1 < d.W, h0 + 1 < d.H && w0 >= 0,
                          h0 + 1 < d.H && w0 + 1 < d.W};
      const int a4[4] = {u8w(hh * hw), u8w(hh * lw), u8w(lh * hw), u8w(lh * lw)};
      const int hq[4] = {h0, h0, h0 + 1, h0 + 1}, wq[4] = {w0, w0 + 1, w0, w0 + 1};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        aw[q] = ok[q] ? (unsigned)a4[q] : 0u;
        hs[q] = ok[q] ? hq[q] : 0;
        ws[q] = ok[q] ? wq[q] : 0;
      }
    }
    f_aw = aw[0] | (aw[1] << 8) | (aw[2] << 16) | (aw[3] << 24);
    f_neg = -(int)((aw[0] + aw[1] + aw[2] + aw[3]) << 7);
#pragma unroll
    for (int q = 0; q < 4; ++q) fidx[q] = (int)(ximg_off + (unsigned)(hs[q] * d.W + ws[q]) * (unsigned)d.Cin);
  };
  // the gather pipeline and the weight pipeline each walk (tap, chunk) with their own counters
  int g_tap = tap_begin, g_chunk = 0, w_tap = tap_begin, w_chunk = 0;
  auto gather_next = [&]() {      // corners of the gather pipeline's current step -> rb; then advance
    if (g_chunk == 0) footprint(g_tap);
    if constexpr (!(ABL & 1)) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        rb[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, fidx[q], g_chunk * kStepK, 0));
    }
    if (++g_chunk == chunks) { g_chunk = 0; g_tap = g_tap + 1 < KK ? g_tap + 1 : 0; }
  };
  typedef __attribute__((address_space(3))) void lds_void;
  auto weights_next = [&](int buf) {
    const int a_s = w_tap * cin_g + w_chunk * kStepK;
    char *adst = smem + buf * kGA + wave * (kPc * 1024);
    if constexpr (!(ABL & 4)) {
#pragma unroll
      for (int j = 0; j < kPc; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void *)(adst + j * 1024), 16, (int)a_off[j], a_s, 0, 0);
    }
    if (++w_chunk == chunks) { w_chunk = 0; w_tap = w_tap + 1 < KK ? w_tap + 1 : 0; }
  };
  const int magic = 65793, half = 1 << 23;   // round(2^24 / 255): exact T2int8(t / 255), see msda_hm4.hip
  // corners in rb (footprint c_*) -> the pixel's 16 column bytes in buffer `buf`: per group of 4 channels the
  // byte transposes, the dots with the first requantisation (top byte), the mask multiply + rounding
  auto produce = [&](int buf) {
    const unsigned c0w[4] = {rb[0].x, rb[0].y, rb[0].z, rb[0].w}, c1w[4] = {rb[1].x, rb[1].y, rb[1].z, rb[1].w};
    const unsigned c2w[4] = {rb[2].x, rb[2].y, rb[2].z, rb[2].w}, c3w[4] = {rb[3].x, rb[3].y, rb[3].z, rb[3].w};
    unsigned res[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if constexpr (ABL & 2) {
        res[v] = c0w[v] ^ c1w[v] ^ c2w[v] ^ (c3w[v] + __float_as_uint(c_m));
      } else {
        unsigned tr[4];
        s8_transpose(c0w[v], c1w[v], c2w[v], c3w[v], tr);
        int xq[4];
        s8_quad(tr, c_aw, c_neg, magic, half, xq);
        res[v] = s8_mask4(xq, c_m);
      }
    }
    // hand-written store: behind a compiler-visible LDS store the waitcnt pass drains vmcnt to 0 (it cannot
    // tell the store from the rows the weight DMA is filling), which would stall on loads still in flight;
    // the s_waitcnt lgkmcnt(0) in front of the barrier covers it
    const u32x4_t rv = {res[0], res[1], res[2], res[3]};
    asm volatile("ds_write_b128 %0, %1" ::"v"(b_lds + (unsigned)(buf * kGB)), "v"(rv) : "memory");
  };
  auto next_gathers = [&]() {
    gather_next();
    c_aw = f_aw; c_neg = f_neg; c_m = f_m;
  };

  // prologue: step 0 -> buffer 0, corners of step 1 in flight
  next_gathers();
  produce(0);
  __builtin_amdgcn_sched_barrier(0);
  weights_next(0);
  if (n_my_steps > 1) next_gathers();
  if (n_my_steps > 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // iteration `step`: tiles of `step` in buffer step & 1, raw corners of step + 1 in rb (a step old).  The
  // weight DMA of step + 1 is always issued BEFORE the gathers of step + 2, so vmcnt(4) in front of the
  // barrier retires it and leaves the gathers in flight.
  // ABL & 16: the two halves of the block's waves run the step's phases in opposite order (every SIMD holds
  // two waves of each half): while the early half does its VALU work the late half runs the MFMAs + fragment
  // reads.  Measured no faster (68.6 vs 67.6 us): the producer VALU work is ~70 % of the SIMD time whatever
  // the order (profiles/r02/int8_dcn_pmc.txt), so the default keeps all waves in the same order.
  const bool late = (ABL & 16) && __builtin_amdgcn_readfirstlane(wave) >= kTh / 128;
  auto mfma_step = [&](int buf) {
    if constexpr (ABL & 8) return;
    const char *Ab = smem + buf * kGA;
    const char *Bb = smem + 2 * kGA + buf * kGB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned c = 2u * ks + hi;
      i32x4_t a[2], b;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const i32x4_t *>(Ab + fa[i] * 128 + ((c ^ swz8(fa[i])) << 4));
      b = *reinterpret_cast<const i32x4_t *>(Bb + fb * 128 + ((c ^ swz8(fb)) << 4));
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b, acc[i], 0, 0, 0);
    }
  };
  // ONE code path for both halves: the late half is the same loop rotated by half an iteration -- its barrier
  // sits between the producer work and the MFMAs, and it runs MFMA(step + 1) where the early half runs
  // MFMA(step); MFMA(0) is peeled.  Between two barriers the early half does producer(s + 1), DMA, gathers,
  // MFMA(s); the late half DMA, gathers, MFMA(s), producer(s + 1).
  if (late) {
    if (n_my_steps > 1) weights_next(1);
    mfma_step(0);
  }
  for (int step = 0; step < n_my_steps; ++step) {
    const bool more1 = step + 1 < n_my_steps, more2 = step + 2 < n_my_steps;
    if (more1) produce((step + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    if (late) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (late ? more2 : more1) weights_next(late ? (step & 1) : ((step + 1) & 1));
    if (more2) next_gathers();
    __builtin_amdgcn_sched_barrier(0);
    if (late ? more1 : true) mfma_step(late ? ((step + 1) & 1) : (step & 1));
    if (!late) {
      if (more2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  if (is_tail) {  // int32 partials (exact in any order), thread-private layout shared with the finish kernel
    int4 *pq = reinterpret_cast<int4 *>(tp.partial) +
               ((((size_t)part * tp.tail_tiles + (ntile - tp.main_tiles)) * gridDim.y + blockIdx.y) * 8) * kTh + tid;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pq[(i * 4 + r) * kTh] = make_int4(acc[i][4 * r], acc[i][4 * r + 1], acc[i][4 * r + 2], acc[i][4 * r + 3]);
    return;
  }
  const int n = n0 + wn * 32 + (lane & 31);
  if (n >= N) return;
  const int b = n / HoWo, pix = n - b * HoWo;
  int8_t *ob = out + ((size_t)b * d.Cout + g * cout_g) * HoWo + pix;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m < cout_g) {
#pragma clang fp contract(off)
        const float v = ((float)acc[i][r] * s_iw + (bias ? bias[g * cout_g + m] : 0.f)) / s_out;
        ob[(size_t)m * HoWo] = (int8_t)q_away(v);
      }
    }
}

// grid (tail tiles, Cout tiles, 8): block z sums accumulator quad z = i*4 + rq of every thread, then requantises
template <int WN>
__global__ __launch_bounds__(256 * WN) void dcn_tail_finish_s8_kernel(const float *__restrict__ bias,
                                                                 int8_t *__restrict__ out, ConvDims d, int g,
                                                                 float s_iw, float s_out, TailPlan tp) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int cout_g = d.Cout / d.G, HoWo = d.Ho * d.Wo, N = d.B * HoWo;
  const int ntile = tp.main_tiles + blockIdx.x;
  constexpr int kTh = Glds<WN>::kThreads;
  const int n0 = ntile * Glds<WN>::kN, m0 = blockIdx.y * kFM;
  const int quad = blockIdx.z, i = quad >> 2, rq = quad & 3;
  int4 a = make_int4(0, 0, 0, 0);
  for (int part = 0; part < tp.split; ++part) {
    const int4 v = reinterpret_cast<const int4 *>(tp.partial)[
        ((((size_t)part * tp.tail_tiles + blockIdx.x) * gridDim.y + blockIdx.y) * 8 + quad) * kTh + tid];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  const int n = n0 + wn * 32 + (lane & 31);
  if (n >= N) return;
  const int b = n / HoWo, pix = n - b * HoWo;
  const int av[4] = {a.x, a.y, a.z, a.w};
  int8_t *ob = out + ((size_t)b * d.Cout + g * cout_g) * HoWo + pix;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = m0 + wm * 64 + i * 32 + 8 * rq + 4 * (lane >> 5) + e;
    if (m < cout_g) {
#pragma clang fp contract(off)
      const float v = ((float)av[e] * s_iw + (bias ? bias[g * cout_g + m] : 0.f)) / s_out;
      ob[(size_t)m * HoWo] = (int8_t)q_away(v);
    }
  }
}

int run_s8(const void *input, const void *offset, const void *mask, const void *weight,
           const void *bias, void *output, void *workspace, const ConvDims &d, float s_in, float s_off,
           float s_mask, float s_w, float s_out, hipStream_t st, bool weight_is_packed);

thread_local int g_mdconv_variant = 0;
thread_local bool g_mdconv_no_tail = false;
thread_local bool g_mdconv_rotate = false;   // variant 7: fp16 LDS-DMA kernel with the wave halves in opposite phase order
thread_local bool g_mdconv_wide = false;  // variant 5: 1024-thread blocks, 128-pixel tiles  // variant 4: LDS-DMA kernel without the split-K tail

size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

bool make_dims(ConvDims &d, int B, int Cin, int H, int W, int Cout, int Kh, int Kw, int sh, int sw,
               int ph, int pw, int dh, int dw, int G, int DG) {
  if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Kh <= 0 || Kw <= 0 || sh <= 0 ||
      sw <= 0 || ph < 0 || pw < 0 || dh <= 0 || dw <= 0 || G <= 0 || DG <= 0)
    return false;
  if (Cin % G || Cout % G || Cin % DG) return false;
  const int Ho = (H + 2 * ph - (dh * (Kh - 1) + 1)) / sh + 1;
  const int Wo = (W + 2 * pw - (dw * (Kw - 1) + 1)) / sw + 1;
  if (Ho <= 0 || Wo <= 0) return false;
  d = ConvDims{B, Cin, H, W, Cout, Kh, Kw, sh, sw, ph, pw, dh, dw, G, DG, Ho, Wo};
  return true;
}

struct WsLayout {
  size_t xt, wt, col, total;
};
// int8 GEMM rows are padded to a multiple of 16 bytes (zero filled)
inline size_t kpad(const ConvDims &d, size_t es) {
  const size_t kg = (size_t)(d.Cin / d.G) * d.Kh * d.Kw;
  return es == 1 ? ((kg + 15) & ~size_t(15)) : kg;
}
WsLayout ws_layout(const ConvDims &d, size_t es) {
  WsLayout w;
  const size_t kp = kpad(d, es);
  w.xt = 0;
  w.wt = align256((size_t)d.B * d.Cin * d.H * d.W * es);
  w.col = w.wt + align256((size_t)d.Cout * kp * es);
  w.total = w.col + align256((size_t)d.B * d.Ho * d.Wo * d.G * kp * es);
  return w;
}

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
  TailPlan tp{0, 1, (int)grid.x, nullptr, nhwc_io ? 1 : 0, relu ? 1 : 0, om_channels, g_mdconv_rotate ? 1 : 0};
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
  hipLaunchKernelGGL(dcn_glds_f16_kernel<WN>, grid2, dim3(Glds<WN>::kThreads), Glds<WN>::kLds, st, xt,
                     (const __half *)offset, (const __half *)mask, wt, (const __half *)bias, (__half *)output, d,
                     g, tp);
  if (tp.tail_tiles)
    hipLaunchKernelGGL(dcn_tail_finish_kernel<WN>, dim3((unsigned)tp.tail_tiles, grid.y, 8),
                       dim3(Glds<WN>::kThreads), 0, st, (const __half *)bias, (__half *)output, d, g, tp);
  return launch_status();
}

template <int WN, int ABL>
int glds_s8_resident_blocks() {
  static thread_local int cached_dev = -1, cached = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev == cached_dev) return cached;
  int per_cu = 0, cus = 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_glds_s8_kernel<WN, ABL>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, GldsS8<WN>::kLds) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dcn_glds_s8_kernel<WN, ABL>, Glds<WN>::kThreads,
                                                   GldsS8<WN>::kLds) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return 0;
  cached_dev = dev;
  cached = per_cu * cus;
  return cached;
}

// launch of the int8 LDS-DMA kernel with its tail plan (same plan as launch_glds; int32 partials)
template <int WN, int ABL>
int launch_glds_s8(const int8_t *xt, const void *offset, const void *mask, const int8_t *wt, const void *bias,
                   void *output, const ConvDims &d, int g, int Kp, char *part_ws, size_t part_room, bool allow_tail,
                   float s_off, float s_mask, float s_iw, float s_out, hipStream_t st) {
  const int KK = d.Kh * d.Kw, cout_g = d.Cout / d.G;
  const size_t N = (size_t)d.B * d.Ho * d.Wo;
  const dim3 grid((unsigned)((N + Glds<WN>::kN - 1) / Glds<WN>::kN), (cout_g + kFM - 1) / kFM);
  const int slots = glds_s8_resident_blocks<WN, ABL>();
  if (slots <= 0) return BEVOPS_FAILURE;
  TailPlan tp{0, 1, (int)grid.x, nullptr, 0, 0, 0, 0};
  const int blocks = (int)(grid.x * grid.y);
  if (allow_tail && blocks > slots && grid.y == 1) {
    const int left = blocks % slots;
    int split = 0;
    for (int f = KK; f >= 2; --f)
      if (KK % f == 0 && left * f <= slots) { split = f; break; }
    const size_t need = (size_t)split * left * Glds<WN>::kThreads * 32 * sizeof(int);
    if (left > 0 && left * 2 <= slots && split >= 2 && part_room >= need) {
      tp.tail_tiles = left;
      tp.split = split;
      tp.main_tiles = (int)grid.x - left;
      tp.partial = reinterpret_cast<float *>(part_ws);
    }
  }
  const dim3 grid2((unsigned)(tp.tail_tiles * tp.split + tp.main_tiles), grid.y);
  hipLaunchKernelGGL((dcn_glds_s8_kernel<WN, ABL>), grid2, dim3(Glds<WN>::kThreads), GldsS8<WN>::kLds, st, xt,
                     (const int8_t *)offset, (const int8_t *)mask, wt, (const float *)bias, (int8_t *)output, d, g,
                     Kp, s_off, s_mask, s_iw, s_out, tp);
  if (tp.tail_tiles)
    hipLaunchKernelGGL(dcn_tail_finish_s8_kernel<WN>, dim3((unsigned)tp.tail_tiles, grid.y, 8),
                       dim3(Glds<WN>::kThreads), 0, st, (const float *)bias, (int8_t *)output, d, g, s_iw, s_out, tp);
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
  else if (sizeof(T) == 2 && HW % 8 == 0 && d.Cin % 8 == 0 && aligned16(input))
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
          const bool wide = g_mdconv_wide || (!g_mdconv_no_tail && wide_blocks >= (size_t)cus);
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

int run_s8(const void *input, const void *offset, const void *mask, const void *weight,
           const void *bias, void *output, void *workspace, const ConvDims &d, float s_in, float s_off,
           float s_mask, float s_w, float s_out, hipStream_t st, bool weight_is_packed) {
  const WsLayout w = ws_layout(d, 1);
  char *ws = static_cast<char *>(workspace);
  int8_t *xt = reinterpret_cast<int8_t *>(ws + w.xt);
  int8_t *wt = reinterpret_cast<int8_t *>(ws + w.wt);
  int8_t *col = reinterpret_cast<int8_t *>(ws + w.col);
  const int HW = d.H * d.W, KK = d.Kh * d.Kw, cin_g = d.Cin / d.G, cout_g = d.Cout / d.G;
  const size_t N = (size_t)d.B * d.Ho * d.Wo;
  if (N > 0x7FFFFFFFull) return BEVOPS_NOT_SUPPORTED;
  // multiScale...Plugin-style precondition (modulatedDeformableConv2dPlugin.cpp:217-219)
  if (d.Cin % 4 != 0 || cout_g % 4 != 0) return BEVOPS_NOT_SUPPORTED;
  const int Kg = KK * cin_g;
  const int Kp = (int)kpad(d, 1);
  if (Kp != Kg) {  // zero the padding columns once (packed weights + column buffer)
    if (hipMemsetAsync(ws + w.wt, 0, w.total - w.wt, st) != hipSuccess) return BEVOPS_FAILURE;
  }
  if (weight_is_packed)  // [Cout][Kp] image made by bevops_mdconv_pack_weight(BEVOPS_I8, ...), padding zeroed there
    wt = const_cast<int8_t *>(static_cast<const int8_t *>(weight));
  // fused implicit GEMM (no column buffer) when a 64-channel chunk stays inside one group and one
  // deform group; variant 6 keeps the im2col + GEMM pair (A/B reference)
  // ... and when there are enough 256 x 64 tiles to give every CU two blocks: the base stage-3 call
  // (544 tiles) 117 vs 170 us; stage 4 (272 tiles, one 4-wave block per CU) 138 vs 130 us keeps the pair
  // (profiles/r02).  Variant 8 forces the fused kernel.
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const size_t tiles = ((N + kSN - 1) / kSN) * (size_t)((cout_g + kSM - 1) / kSM);
  const bool fits32 = (size_t)d.B * d.H * d.W * d.Cin < 0xFFFFFF00ull && (size_t)cout_g * Kp < 0xFFFFFF00ull;
  // LDS-DMA pipelined kernel (section 6c): 128-channel k-steps inside one group and ONE deform group per
  // conv group; variants 8 (register-staged fused kernel) and 6 (im2col + GEMM) keep the A/B references,
  // variants 20 + mask = timing experiments with parts of the kernel removed (wrong results)
  bool one_dg = cin_g <= d.Cin / d.DG;
  for (int g = 0; g < d.G && one_dg; ++g)
    one_dg = (g * cin_g) / (d.Cin / d.DG) == (g * cin_g + cin_g - 1) / (d.Cin / d.DG);
  // ... and enough 128-pixel tiles to give every CU a block (base stage 3: 67 us vs 100 us for the
  // register-staged kernel; stage 4 with its 136 blocks stays on the im2col + GEMM pair: 121 vs 130 us,
  // profiles/r02); variant 9 forces it
  const size_t wide_blocks = ((N + Glds<4>::kN - 1) / Glds<4>::kN) * ((cout_g + kFM - 1) / kFM);
  const bool glds = g_mdconv_variant != 6 && g_mdconv_variant != 8 && fits32 && one_dg && cin_g % 128 == 0 &&
                    3 * KK <= kSOmRows && Kp == Kg && (wide_blocks >= (size_t)cus || g_mdconv_variant == 9);
  const bool fused = glds || (g_mdconv_variant != 6 && cin_g % kSK == 0 && (d.Cin / d.DG) % kSK == 0 && fits32 &&
                              (g_mdconv_variant == 8 || tiles >= (size_t)2 * cus));
  if (HW % 4 == 0 && d.Cin % 16 == 0 && aligned16(input))
    hipLaunchKernelGGL(nchw_to_nhwc_s8v_kernel, dim3((HW + 127) / 128, (d.Cin + 127) / 128, d.B), dim3(256), 0, st,
                       (const int8_t *)input, xt, d.Cin, HW, fused ? 0x80808080u : 0u);
  else
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<int8_t>), dim3((HW + 31) / 32, (d.Cin + 31) / 32, d.B), dim3(256),
                       0, st, (const int8_t *)input, xt, d.Cin, HW, fused ? 0x80 : 0);
  const size_t wtot = (size_t)d.Cout * cin_g * KK;
  if (!weight_is_packed)
    hipLaunchKernelGGL((repack_weight_kernel<int8_t>), dim3((unsigned)((wtot + 255) / 256)), dim3(256), 0, st,
                       (const int8_t *)weight, wt, d.Cout, cin_g, KK, Kp);
  if (glds) {
    const int abl = g_mdconv_variant >= 20 && g_mdconv_variant <= 36 ? g_mdconv_variant - 20 : 0;   // timing experiments
    for (int g = 0; g < d.G; ++g) {
      int rc = BEVOPS_NOT_SUPPORTED;
      char *pw = ws + w.col;
      const size_t room = w.total - w.col;
      const float s_iw = s_in * s_w;
#define BEVOPS_S8_GO(WN_, ABL_) \
  rc = launch_glds_s8<WN_, ABL_>(xt, offset, mask, wt, bias, output, d, g, Kp, pw, room, !g_mdconv_no_tail, s_off, s_mask, s_iw, s_out, st)
      const bool w4 = wide_blocks >= (size_t)cus;
      if (!w4) BEVOPS_S8_GO(2, 0);
      else
        switch (abl) {
          case 0: BEVOPS_S8_GO(4, 0); break;
          case 1: BEVOPS_S8_GO(4, 1); break;
          case 2: BEVOPS_S8_GO(4, 2); break;
          case 4: BEVOPS_S8_GO(4, 4); break;
          case 8: BEVOPS_S8_GO(4, 8); break;
          case 7: BEVOPS_S8_GO(4, 7); break;
          case 11: BEVOPS_S8_GO(4, 11); break;
          case 13: BEVOPS_S8_GO(4, 13); break;
          case 14: BEVOPS_S8_GO(4, 14); break;
          case 15: BEVOPS_S8_GO(4, 15); break;
          case 16: BEVOPS_S8_GO(4, 16); break;
          default: break;
        }
#undef BEVOPS_S8_GO
      if (rc != BEVOPS_SUCCESS) return rc;
    }
    return launch_status();
  }
  if (fused) {
    for (int g = 0; g < d.G; ++g)
      hipLaunchKernelGGL(dcn_fused_s8_kernel, dim3((unsigned)((N + kSN - 1) / kSN), (cout_g + kSM - 1) / kSM), dim3(256),
                         0, st, xt, (const int8_t *)offset, (const int8_t *)mask, wt, (const float *)bias,
                         (int8_t *)output, d, g, Kp, s_off, s_mask, s_in * s_w, s_out);
    return launch_status();
  }
  const bool v16 = cin_g % 16 == 0 && (d.Cin / d.DG) % 16 == 0;
  const bool v4 = cin_g % 4 == 0 && (d.Cin / d.DG) % 4 == 0;
  const int V = v16 ? 16 : (v4 ? 4 : 1);
  const size_t blocks = (N * KK * (d.Cin / V) + 255) / 256;
  if (blocks > 0x7FFFFFFFull) return BEVOPS_NOT_SUPPORTED;
  if (v16)
    hipLaunchKernelGGL((im2col_nhwc_s8_kernel<16>), dim3((unsigned)blocks), dim3(256), 0, st, xt,
                       (const int8_t *)offset, (const int8_t *)mask, col, d, s_off, s_mask, Kp);
  else if (v4)
    hipLaunchKernelGGL((im2col_nhwc_s8_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, st, xt,
                       (const int8_t *)offset, (const int8_t *)mask, col, d, s_off, s_mask, Kp);
  else
    hipLaunchKernelGGL((im2col_nhwc_s8_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, xt,
                       (const int8_t *)offset, (const int8_t *)mask, col, d, s_off, s_mask, Kp);
  for (int g = 0; g < d.G; ++g) {
    const GemmEpi e{d.Ho * d.Wo, d.Cout, g * cout_g};
    hipLaunchKernelGGL(gemm_tn_s8_kernel, dim3((unsigned)((N + kBN - 1) / kBN), (cout_g + kBM - 1) / kBM),
                       dim3(256), 0, st, wt + (size_t)g * cout_g * Kp, col + (size_t)g * N * Kp,
                       (const float *)bias, (int8_t *)output, cout_g, (int)N, Kp, e, s_in * s_w, s_out);
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
  g_mdconv_rotate = variant == 7;
  g_mdconv_variant = (variant == 4 || variant == 5 || variant == 7) ? 0 : variant;
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

extern "C" int bevops_mdconv_forward_int8(const void *input, float scale_in, const void *offset,
                                          float scale_offset, const void *mask, float scale_mask,
                                          const void *weight, float scale_weight, const float *bias,
                                          void *output, float scale_out, void *workspace,
                                          size_t workspace_bytes, int B, int Cin, int H, int W, int Cout,
                                          int Kh, int Kw, int stride_h, int stride_w, int pad_h, int pad_w,
                                          int dil_h, int dil_w, int groups, int deform_groups,
                                          void *stream) {
  if (!input || !offset || !mask || !weight || !output || !workspace) return BEVOPS_BAD_PARAM;
  if (!(scale_in > 0.f) || !(scale_offset > 0.f) || !(scale_mask > 0.f) || !(scale_weight > 0.f) ||
      !(scale_out > 0.f))
    return BEVOPS_BAD_PARAM;
  ConvDims d;
  if (!make_dims(d, B, Cin, H, W, Cout, Kh, Kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w,
                 groups, deform_groups))
    return BEVOPS_BAD_PARAM;
  if (workspace_bytes < ws_layout(d, 1).total || !aligned16(workspace)) return BEVOPS_BAD_PARAM;
  return run_s8(input, offset, mask, weight, bias, output, workspace, d, scale_in, scale_offset,
                scale_mask, scale_weight, scale_out, static_cast<hipStream_t>(stream), false);
}

extern "C" int bevops_mdconv_forward_int8_packed(const void *input, float scale_in, const void *offset,
                                                 float scale_offset, const void *mask, float scale_mask,
                                                 const void *packed_weight, float scale_weight, const float *bias,
                                                 void *output, float scale_out, void *workspace,
                                                 size_t workspace_bytes, int B, int Cin, int H, int W, int Cout,
                                                 int Kh, int Kw, int stride_h, int stride_w, int pad_h, int pad_w,
                                                 int dil_h, int dil_w, int groups, int deform_groups,
                                                 void *stream) {
  if (!input || !offset || !mask || !packed_weight || !output || !workspace) return BEVOPS_BAD_PARAM;
  if (!(scale_in > 0.f) || !(scale_offset > 0.f) || !(scale_mask > 0.f) || !(scale_weight > 0.f) ||
      !(scale_out > 0.f))
    return BEVOPS_BAD_PARAM;
  ConvDims d;
  if (!make_dims(d, B, Cin, H, W, Cout, Kh, Kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w,
                 groups, deform_groups))
    return BEVOPS_BAD_PARAM;
  if (workspace_bytes < ws_layout(d, 1).total || !aligned16(workspace) || !aligned16(packed_weight))
    return BEVOPS_BAD_PARAM;
  return run_s8(input, offset, mask, packed_weight, bias, output, workspace, d, scale_in, scale_offset,
                scale_mask, scale_weight, scale_out, static_cast<hipStream_t>(stream), true);
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
