// Shared pieces of the DCNv2 translation units (mdconv.hip: fp32 / fp16; mdconv_s8.hip: int8).
#ifndef BEVOPS_MDCONV_H_
#define BEVOPS_MDCONV_H_
#include <type_traits>

#include "common.h"

namespace bevops {
// A/B switches of bevops_mdconv_set_variant (defined in mdconv.hip)
extern thread_local int g_mdconv_variant;
extern thread_local bool g_mdconv_no_tail;
extern thread_local bool g_mdconv_wide;
extern thread_local int g_mdconv_rotate;   // LDS-DMA kernels' wave order: 0 default (round 6), 1 round-2 rotation (fp16), 2 one order

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

// fp16 / int8 GEMM tile (sections 4a, 6) and its scatter epilogue
constexpr int kBM = 128, kBN = 128, kBK = 32, kLd = kBK + 8;  // +8 halves: conflict-free b128 reads

struct GemmEpi {
  int HoWo, Cout, co0;  // out[(n / HoWo) * Cout + co0 + m][n % HoWo]
};

// fused implicit-GEMM block tile (section 5)
constexpr int kFM = 256, kFN = 64, kFK = 64, kFLd = kFK + 8;

// LDS image of the LDS-DMA kernels (sections 5c, 6c)
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

// ---- host side: problem description and workspace layout -------------------------------------------
inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

inline bool make_dims(ConvDims &d, int B, int Cin, int H, int W, int Cout, int Kh, int Kw, int sh, int sw,
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
inline WsLayout ws_layout(const ConvDims &d, size_t es) {
  WsLayout w;
  const size_t kp = kpad(d, es);
  w.xt = 0;
  w.wt = align256((size_t)d.B * d.Cin * d.H * d.W * es);
  w.col = w.wt + align256((size_t)d.Cout * kp * es);
  w.total = w.col + align256((size_t)d.B * d.Ho * d.Wo * d.G * kp * es);
  return w;
}

}  // namespace
}  // namespace bevops
#endif  // BEVOPS_MDCONV_H_
