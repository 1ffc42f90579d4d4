// The 3x3 convolutions of ResNet stage 1 (conv2 of its three bottlenecks: 64 -> 64 channels, stride 1, pad 1, on the
// 232 x 400 maps of six cameras; det2trt/models/backbones/resnet.py:106-260) as an implicit GEMM whose operands BOTH
// live in LDS.  Not a reference plugin (TensorRT owns the layer there).
//
// Why a kernel of its own.  As a row of tile_gemm.hip's implicit GEMM the layer takes 87 us for 41 GFLOP and 142 MB:
// with 64 channels a pixel is ONE 128-byte line, every one of the nine taps fetches it again from L1 / L2 (1.06 GB
// through the L2 -> L1 path per launch, profiles/r05/frame_pmc_fp16.txt), and K = 576 is nine short k-steps whose
// latency nothing covers.  Here
//   * the whole weight matrix (64 x 576 fp16 = 72 KB) is loaded into LDS ONCE per (persistent) block;
//   * a block walks 16 x 16-pixel output tiles; the tile's 18 x 18 input pixels (with the zero padding of the image
//     border) are staged in LDS once -- 1.27 x the unique bytes instead of 9 x -- and the next tile's pixels are
//     already in flight (registers) while this one is multiplied;
//   * a tap is then an LDS address offset: wave w owns pixel rows 4 w .. 4 w + 3 of the tile and all 64 output
//     channels -- 2 x 2 v_mfma_f32_32x32x16_f16 blocks, four operand fragments (16-byte LDS reads, rows padded to
//     144 / 1 168 bytes: conflict-free) per four matrix instructions, 36 k-substeps fully unrolled;
//   * k runs [tap][channel] with the same 16-value MFMA steps as tile_gemm's implicit GEMM and the epilogue is the
//     same arithmetic (fp32 sum + bias, ReLU, one rounding): the results are BIT-IDENTICAL to bevops_conv_tile_f16
//     (tests/test_conv_halo_gpu.py), so the dispatch may switch between the two freely;
//   * the tile's outputs go through LDS once more so that a pixel's 128 bytes leave as eight 16-byte stores of
//     neighbouring lanes.
#include <algorithm>

#include "common.h"

namespace bevops {
namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kHC = 64;                          // channels in = channels out
constexpr int kHT = 16;                          // output tile: 16 x 16 pixels
constexpr int kHH = kHT + 2;                     // staged input rows / columns
constexpr int kHPix = kHC * 2 + 16;              // LDS bytes per staged pixel (128 + 16)
constexpr int kHWRow = 9 * kHC * 2 + 16;         // LDS bytes per weight row (1 152 + 16)
constexpr int kHWBytes = kHC * kHWRow;           // 74 752
constexpr int kHXBytes = kHH * kHH * kHPix;      // 46 656
constexpr int kHLds = kHWBytes + kHXBytes;       // 121 408
constexpr int kHThreads = 256;
constexpr int kHChunks = kHH * kHH * 8;          // 16-byte chunks of a staged tile: 2 592
constexpr int kHRounds = (kHChunks + kHThreads - 1) / kHThreads;   // 11

__global__ __launch_bounds__(kHThreads) void conv3x3_c64_halo_kernel(const __half *__restrict__ x,
                                                                     const __half *__restrict__ w,
                                                                     const __half *__restrict__ bias,
                                                                     __half *__restrict__ out, int H, int W, int relu,
                                                                     int tiles_x, int tiles_img, int tiles_total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *Ws = smem, *Xs = smem + kHWBytes;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  // ---- weights [64][576] -> LDS, once
  for (int i = tid; i < kHC * 72; i += kHThreads) {
    const int row = i / 72, c = i - row * 72;
    *reinterpret_cast<uint4 *>(Ws + row * kHWRow + c * 16) = reinterpret_cast<const uint4 *>(w)[i];
  }
  // ---- this thread's chunks of a staged tile: chunk q = tid + 256 r -> (staged pixel q >> 3, 16-byte piece q & 7)
  int hyx[kHRounds];
#pragma unroll
  for (int r = 0; r < kHRounds; ++r) {
    const int q = tid + kHThreads * r, pix = q >> 3;
    const int hy = pix / kHH;
    hyx[r] = q < kHChunks ? ((hy << 8) | (pix - hy * kHH)) : -1;
  }
  auto tile_origin = [&](int t, int &b, int &ty0, int &tx0) {
    b = t / tiles_img;
    const int rem = t - b * tiles_img;
    const int ty = rem / tiles_x;
    ty0 = ty * kHT;
    tx0 = (rem - ty * tiles_x) * kHT;
  };
  uint4 pre[kHRounds];
  auto load_tile = [&](int t) {
    int b, ty0, tx0;
    tile_origin(t, b, ty0, tx0);
#pragma unroll
    for (int r = 0; r < kHRounds; ++r) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (hyx[r] >= 0) {
        const int y = ty0 + (hyx[r] >> 8) - 1, xx = tx0 + (hyx[r] & 255) - 1;
        if (y >= 0 && y < H && xx >= 0 && xx < W)
          v = *reinterpret_cast<const uint4 *>(x + (((size_t)b * H + y) * W + xx) * kHC + (tid & 7) * 8);
      }
      pre[r] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int r = 0; r < kHRounds; ++r)
      if (hyx[r] >= 0) {
        const int q = tid + kHThreads * r;
        *reinterpret_cast<uint4 *>(Xs + (q >> 3) * kHPix + (q & 7) * 16) = pre[r];
      }
  };
  // fragment bases: operand A = weight rows (output channel 32 i + l31), operand B = pixels (j: rows 4 w + 2 j + (l31 >> 4))
  const char *wa[2], *xb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) wa[i] = Ws + (32 * i + l31) * kHWRow + 16 * hi;
#pragma unroll
  for (int j = 0; j < 2; ++j) xb[j] = Xs + ((4 * wave + 2 * j + (l31 >> 4)) * kHH + (l31 & 15)) * kHPix + 16 * hi;
  // acc[i][j][4 g + e]: channel 32 i + 8 g + 4 hi + e of pixel (row 4 w + 2 j + (l31 >> 4), column l31 & 15)
  float bcol[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) bcol[i][4 * g + e] = bias ? __half2float(bias[32 * i + 8 * g + 4 * hi + e]) : 0.f;

  int t = blockIdx.x;
  if (t >= tiles_total) return;
  load_tile(t);
  store_tile();
  __syncthreads();
  for (; t < tiles_total; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < tiles_total) load_tile(tn);      // in flight while this tile is multiplied
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int xoff = ((tap / 3) * kHH + (tap % 3)) * kHPix;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        f16x8_t a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f16x8_t *>(wa[i] + tap * (kHC * 2) + kk * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f16x8_t *>(xb[j] + xoff + kk * 32);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();      // everybody is done with the staged pixels: their LDS now stages the outputs
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      char *row = Xs + ((4 * wave + 2 * j + (l31 >> 4)) * kHT + (l31 & 15)) * kHPix;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[i][j][4 * g + e] + bcol[i][4 * g + e];
            if (relu) v[e] = fmaxf(v[e], 0.f);
          }
          *reinterpret_cast<uint2 *>(row + (32 * i + 8 * g + 4 * hi) * 2) = make_uint2(pack_h2(v[0], v[1]), pack_h2(v[2], v[3]));
        }
    }
    __syncthreads();
    {
      int b, ty0, tx0;
      tile_origin(t, b, ty0, tx0);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int q = tid + kHThreads * r, p = q >> 3;
        const int y = ty0 + (p >> 4), xx = tx0 + (p & 15);
        if (y < H && xx < W)
          *reinterpret_cast<uint4 *>(out + (((size_t)b * H + y) * W + xx) * kHC + (q & 7) * 8) =
              *reinterpret_cast<const uint4 *>(Xs + p * kHPix + (q & 7) * 16);
      }
    }
    __syncthreads();      // the staged outputs are read: the next tile's pixels may land
    if (tn < tiles_total) store_tile();
    __syncthreads();
  }
}

int halo_cu_count() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    hipDeviceProp_t p;
    static int cached[16] = {0};
    if (cached[dev & 15] == 0 && hipGetDeviceProperties(&p, dev) == hipSuccess) cached[dev & 15] = p.multiProcessorCount;
    if (cached[dev & 15] > 0) cus = cached[dev & 15];
  }
  return cus;
}

}  // namespace
}  // namespace bevops

using namespace bevops;

// x [B, H, W, 64] channels-last fp16, weight_taps [64, 3, 3, 64] (taps-major, as bevops_conv_tile_f16 takes it), bias
// [64] fp16 or null -> out [B, H, W, 64] = act(conv3x3(x, stride 1, pad 1) + bias).  NOT_SUPPORTED for other channel
// counts.  Bit-identical to bevops_conv_tile_f16 on the same operands.
extern "C" int bevops_conv3x3_c64_f16(const void *x, const void *weight_taps, const void *bias, void *out, int B, int H,
                                      int W, int Cin, int Cout, int relu, void *stream) {
  if (!x || !weight_taps || !out || B <= 0 || H <= 0 || W <= 0) return BEVOPS_BAD_PARAM;
  if (Cin != kHC || Cout != kHC) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(x) || !aligned16(weight_taps) || !aligned16(out)) return BEVOPS_BAD_PARAM;
  if ((long long)B * H * W * kHC * 2 >= (1ll << 40)) return BEVOPS_NOT_SUPPORTED;
  const int tiles_x = (W + kHT - 1) / kHT, tiles_y = (H + kHT - 1) / kHT;
  const long long total = (long long)B * tiles_x * tiles_y;
  if (total > (1ll << 30)) return BEVOPS_NOT_SUPPORTED;
  if (!ensure_dynamic_lds<conv3x3_c64_halo_kernel>(kHLds)) return BEVOPS_FAILURE;
  const int blocks = (int)std::min<long long>(total, (long long)halo_cu_count());
  hipLaunchKernelGGL(conv3x3_c64_halo_kernel, dim3((unsigned)blocks), dim3(kHThreads), kHLds, static_cast<hipStream_t>(stream),
                     static_cast<const __half *>(x), static_cast<const __half *>(weight_taps),
                     static_cast<const __half *>(bias), static_cast<__half *>(out), H, W, relu, tiles_x, tiles_x * tiles_y,
                     (int)total);
  return launch_status();
}
