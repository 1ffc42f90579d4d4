// The 3x3 convolutions of ResNet stage 1 (conv2 of its three bottlenecks: 64 -> 64 channels, stride 1, pad 1, on the
// 232 x 400 maps of six cameras; det2trt/models/backbones/resnet.py:106-260) as an implicit GEMM whose operands BOTH
// live in LDS.  Not a reference plugin (TensorRT owns the layer there).
//
// Why a kernel of its own.  As a row of tile_gemm.hip's implicit GEMM the layer takes 88 us for 41 GFLOP and 142 MB:
// with 64 channels a pixel is ONE 128-byte line, every one of the nine taps fetches it again from L1 / L2 (1.06 GB
// through the L2 -> L1 path per launch, profiles/r05/frame_pmc_fp16.txt), and K = 576 is nine short k-steps whose
// latency nothing covers.  Here (47 us at the same shape, profiles/r06/conv_halo_time.jsonl)
//   * the whole weight matrix (64 x 576 fp16 = 72 KB) is loaded into LDS ONCE per (persistent) block;
//   * a block walks 16 x 16-pixel output tiles; the tile's 18 x 18 input pixels (with the zero padding of the image
//     border) are staged in LDS once -- 1.27 x the unique bytes instead of 9 x; a tap is then an LDS address offset;
//   * EIGHT waves in two roles.  Waves 0-3 multiply: wave w owns tile rows 4 w .. 4 w + 3 and all 64 output channels
//     -- 2 x 2 v_mfma_f32_32x32x16_f16 blocks, four operand fragments (16-byte LDS reads; pixel rows padded to 144,
//     tile rows to 2 816, weight rows to 1 168 bytes: conflict-free for the instruction's lane groups) per four
//     matrix instructions, 36 k-substeps unrolled with the fragments requested two substeps ahead.  Waves 4-7 move
//     data: the pixels of tile t + 1 wait in one register set for the multiply to let go of the staged tile while the
//     loads of tile t + 2 are already in flight in a second set, and the outputs of tile t - 1 leave from an LDS
//     staging region of their own.  The first build had every wave do both: a wave that issues 11 loads + 8 stores
//     per tile into a memory pipe that accepts ~11 bytes per clock and CU stalls AT ISSUE for about as long as the
//     multiply takes, and nothing overlapped (66 us; s_memtime phase stamps: profiles/r06/conv_halo_phase_probe.json
//     is the last probe build -- multiply 5.1 k cycles per tile, movers 6.5 k at the pipe's rate, output staging 2.2 k);
//   * k runs [tap][channel] with the same 16-value MFMA steps as tile_gemm's implicit GEMM and the epilogue is the
//     same arithmetic (fp32 sum + bias, ReLU, one rounding): the results are BIT-IDENTICAL to bevops_conv_tile_f16
//     (tests/test_conv_halo_gpu.py), so the dispatch may switch between the two freely.
// What bounds it now: 73.5 KB of loads + stores per tile at the per-CU memory rate is 6.5 k cycles, the multiply +
// output staging 7.3 k, and the two barriers per tile keep the roles from overlapping perfectly (LDS is full: 162 KB;
// no room for a second pixel or output buffer).
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace bevops {
namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kHC = 64;                          // channels in = channels out
constexpr int kHT = 16;                          // output tile: 16 x 16 pixels
constexpr int kHH = kHT + 2;                     // staged input rows / columns
constexpr int kHPix = kHC * 2 + 16;              // LDS bytes per staged pixel (128 + 16)
constexpr int kHRow = kHH * kHPix + 224;         // LDS bytes per staged pixel ROW: 2 816 = 0 mod 256, so that the 16 lanes of
                                                 // a fragment read that sit on the next tile row continue the bank sequence
constexpr int kHWRow = 9 * kHC * 2 + 16;         // LDS bytes per weight row (1 152 + 16)
constexpr int kHWBytes = kHC * kHWRow;           // 74 752
constexpr int kHXBytes = kHH * kHRow;            // 50 688
constexpr int kHYBytes = kHT * kHT * kHPix;      // output staging: 36 864
constexpr int kHLds = kHWBytes + kHXBytes + kHYBytes;   // 162 304 of the CU's 163 840
constexpr int kHThreads = 512;                   // waves 0-3 multiply, waves 4-7 move data
constexpr int kHRole = 256;
constexpr int kHChunks = kHH * kHH * 8;          // 16-byte chunks of a staged tile: 2 592
constexpr int kHRounds = (kHChunks + kHRole - 1) / kHRole;   // 11

__global__ __launch_bounds__(kHThreads) void conv3x3_c64_halo_kernel(const __half *__restrict__ x,
                                                                     const __half *__restrict__ w,
                                                                     const __half *__restrict__ bias,
                                                                     __half *__restrict__ out, int H, int W, int relu,
                                                                     int tiles_x, int tiles_img, int tiles_total,
                                                                     unsigned x_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *Ws = smem, *Xs = smem + kHWBytes, *Ys = Xs + kHXBytes;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const bool mover = wave >= 4;
  const int rt = tid & (kHRole - 1), cw = wave & 3;      // thread index inside its role, multiply-wave index
  // ---- weights [64][576] -> LDS, once (all eight waves)
  {
    uint4 wr[kHC * 72 / kHThreads];      // 9 loads in flight, then 9 LDS writes (a rolled loop pays one round trip each)
#pragma unroll
    for (int r = 0; r < kHC * 72 / kHThreads; ++r) wr[r] = reinterpret_cast<const uint4 *>(w)[tid + kHThreads * r];
#pragma unroll
    for (int r = 0; r < kHC * 72 / kHThreads; ++r) {
      const int i = tid + kHThreads * r, row = i / 72, c = i - row * 72;
      *reinterpret_cast<uint4 *>(Ws + row * kHWRow + c * 16) = wr[r];
    }
  }
  auto tile_origin = [&](int t, int &b, int &ty0, int &tx0) {
    b = t / tiles_img;
    const int rem = t - b * tiles_img;
    const int ty = rem / tiles_x;
    ty0 = ty * kHT;
    tx0 = (rem - ty * tiles_x) * kHT;
  };
  int t = blockIdx.x;
  if (t >= tiles_total) return;

  if (mover) {
    // ================= waves 4-7: the tile's pixels in, the tile's outputs out
    // chunk q = rt + 256 r of a staged tile -> (staged pixel q >> 3, 16-byte piece q & 7)
    int hyx[kHRounds];
#pragma unroll
    for (int r = 0; r < kHRounds; ++r) {
      const int q = rt + kHRole * r, pix = q >> 3;
      const int hy = pix / kHH;
      hyx[r] = q < kHChunks ? ((hy << 8) | (pix - hy * kHH)) : -1;
    }
    // two register sets: the pixels of tile t + 1 wait in one for the multiply to let go of the staged tile while
    // the loads of tile t + 2 are already in flight in the other -- the memory pipe never idles across the barriers
    uint4 pre[2][kHRounds];
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half *>(x), 0, x_bytes, 0x00020000);
    auto load_tile = [&](int tt, auto setc) __attribute__((always_inline)) {
      constexpr int S = decltype(setc)::value;
      int b, ty0, tx0;
      tile_origin(tt, b, ty0, tx0);
#pragma unroll
      for (int r = 0; r < kHRounds; ++r) {
        // (a buffer load: a pixel outside the image -- the zero padding -- gets the beyond-the-buffer offset and reads
        // as zero; no branch, so the wait in front of land_tile counts exactly THIS set's loads)
        const int y = ty0 + (hyx[r] >> 8) - 1, xx = tx0 + (hyx[r] & 255) - 1;
        const bool in = hyx[r] >= 0 && y >= 0 && y < H && xx >= 0 && xx < W;
        const unsigned off = in ? (unsigned)(((((size_t)b * H + y) * W + xx) * kHC + (rt & 7) * 8) * 2) : 0xFFFFFF00u;
        pre[S][r] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 0));
      }
    };
    auto land_tile = [&](auto setc) __attribute__((always_inline)) {
      constexpr int S = decltype(setc)::value;
#pragma unroll
      for (int r = 0; r < kHRounds; ++r)
        if (hyx[r] >= 0)
          *reinterpret_cast<uint4 *>(Xs + (hyx[r] >> 8) * kHRow + (hyx[r] & 255) * kHPix + (rt & 7) * 16) = pre[S][r];
    };
    auto store_outputs = [&](int tt) {
      int b, ty0, tx0;
      tile_origin(tt, b, ty0, tx0);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int q = rt + kHRole * r, p = q >> 3;
        const int y = ty0 + (p >> 4), xx = tx0 + (p & 15);
        if (y < H && xx < W)
          *reinterpret_cast<uint4 *>(out + (((size_t)b * H + y) * W + xx) * kHC + (q & 7) * 8) =
              *reinterpret_cast<const uint4 *>(Ys + p * kHPix + (q & 7) * 16);
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int g = (int)gridDim.x;
    load_tile(t, S0{});
    land_tile(S0{});
    if (t + g < tiles_total) load_tile(t + g, S1{});
    __syncthreads();
    int prev = -1;
    // one iteration: tile t is being multiplied, tile t + g waits in set N1, tile t + 2 g is requested into set N2
    auto iteration = [&](auto n1, auto n2) __attribute__((always_inline)) {
      if (prev >= 0) store_outputs(prev);        // (staged behind barrier B of the previous iteration)
      if (t + 2 * g < tiles_total) load_tile(t + 2 * g, n2);   // the memory pipe's back-pressure stalls THESE waves
      __syncthreads();      // A: the multiply is done with the staged pixels
      if (t + g < tiles_total) land_tile(n1);
      __syncthreads();      // B: next pixels and this tile's staged outputs are visible
      prev = t;
      t += g;
    };
    while (t < tiles_total) {
      iteration(S1{}, S0{});
      if (t >= tiles_total) break;
      iteration(S0{}, S1{});
    }
    store_outputs(prev);
  } else {
    // ================= waves 0-3: the multiply.  Wave cw owns tile rows 4 cw .. 4 cw + 3, all 64 output channels
    // fragment bases: operand A = weight rows (output channel 32 i + l31), operand B = pixels (j: row 4 cw + 2 j + (l31 >> 4))
    const char *wa[2], *xb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) wa[i] = Ws + (32 * i + l31) * kHWRow + 16 * hi;
#pragma unroll
    for (int j = 0; j < 2; ++j) xb[j] = Xs + (4 * cw + 2 * j + (l31 >> 4)) * kHRow + (l31 & 15) * kHPix + 16 * hi;
    // acc[i][j][4 g + e]: channel 32 i + 8 g + 4 hi + e of pixel (row 4 cw + 2 j + (l31 >> 4), column l31 & 15)
    float bcol[2][16];
    {
      uint2 braw[2][4];      // (eight independent 8-byte loads: scalar element loads are serialised by the compiler)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          braw[i][g] = bias ? *reinterpret_cast<const uint2 *>(bias + 32 * i + 8 * g + 4 * hi) : make_uint2(0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bcol[i][4 * g] = h2f_lo(braw[i][g].x);
          bcol[i][4 * g + 1] = h2f_hi(braw[i][g].x);
          bcol[i][4 * g + 2] = h2f_lo(braw[i][g].y);
          bcol[i][4 * g + 3] = h2f_hi(braw[i][g].y);
        }
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(3);      // the matrix instructions go first; the movers' issue slots are what is left
    for (; t < tiles_total; t += gridDim.x) {
      f32x16_t acc[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      // 36 k-substeps (tap = s / 4, 16 channels each), operand fragments requested TWO substeps ahead of their
      // matrix instructions: one substep of cover (4 instructions = 128 cycles) leaves the LDS latency exposed
      f16x8_t fa[3][2], fb[3][2];
      auto fetch = [&](int sidx, int slot) __attribute__((always_inline)) {
        const int tap = sidx >> 2, kk = sidx & 3;
        const int xoff = (tap / 3) * kHRow + (tap % 3) * kHPix;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[slot][i] = *reinterpret_cast<const f16x8_t *>(wa[i] + tap * (kHC * 2) + kk * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[slot][j] = *reinterpret_cast<const f16x8_t *>(xb[j] + xoff + kk * 32);
      };
      fetch(0, 0);
      fetch(1, 1);
#pragma unroll
      for (int sidx = 0; sidx < 36; ++sidx) {
        if (sidx + 2 < 36) fetch(sidx + 2, (sidx + 2) % 3);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sidx % 3][i], fb[sidx % 3][j], acc[i][j], 0, 0, 0);
        // (keep that order in the schedule: the four fragment reads of substep s + 2, then the four matrix instructions of s)
        if (sidx + 2 < 36) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      __syncthreads();      // A
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        char *row = Ys + ((4 * cw + 2 * j + (l31 >> 4)) * kHT + (l31 & 15)) * kHPix;
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
      __syncthreads();      // B
    }
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
  if (!aligned16(x) || !aligned16(weight_taps) || !aligned16(out) || (reinterpret_cast<uintptr_t>(bias) & 7u))
    return BEVOPS_BAD_PARAM;
  if ((long long)B * H * W * kHC * 2 >= 0xFFFFFF00ll) return BEVOPS_NOT_SUPPORTED;     // (32-bit buffer offsets)
  const int tiles_x = (W + kHT - 1) / kHT, tiles_y = (H + kHT - 1) / kHT;
  const long long total = (long long)B * tiles_x * tiles_y;
  if (total > (1ll << 30)) return BEVOPS_NOT_SUPPORTED;
  if (!ensure_dynamic_lds<conv3x3_c64_halo_kernel>(kHLds)) return BEVOPS_FAILURE;
  const int blocks = (int)std::min<long long>(total, (long long)halo_cu_count());
  hipLaunchKernelGGL(conv3x3_c64_halo_kernel, dim3((unsigned)blocks), dim3(kHThreads), kHLds, static_cast<hipStream_t>(stream),
                     static_cast<const __half *>(x), static_cast<const __half *>(weight_taps),
                     static_cast<const __half *>(bias), static_cast<__half *>(out), H, W, relu, tiles_x, tiles_x * tiles_y,
                     (int)total, (unsigned)((size_t)B * H * W * kHC * 2));
  return launch_status();
}
