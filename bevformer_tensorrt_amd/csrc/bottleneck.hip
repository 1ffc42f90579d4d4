// A whole ResNet bottleneck of stage 1 (blocks 2 and 3 of layer1: 256 -> 64 -> 64 -> 256 channels, stride 1, identity
// shortcut; det2trt/models/backbones/resnet.py:106-260 with the BatchNorms folded) as ONE kernel:
//     y1 = relu(conv1x1(x;  W1 [64, 256]) + b1)
//     y2 = relu(conv3x3(y1; W2 [64, 3, 3, 64], pad 1) + b2)
//     out = relu(conv1x1(y2; W3 [256, 64]) + b3 + x)
// on channels-last fp16 activations.  Not a reference plugin (TensorRT fuses these layers itself).  As three launches
// the block moves 1.16 GB (x read twice, y1 and y2 written and read: 82 + 47 + 133 us on the six 232 x 400 maps); here
// x is read once per 8 x 8-pixel tile (with its one-pixel border: 1.56 x) and only `out` is written: 0.73 GB.
//
// MI355X mapping (the roles of conv_halo.hip): eight waves.  Waves 4-7 MOVE data: the tile's 10 x 10 input pixels
// (x 512 bytes) are requested into registers while the previous tile is computed -- in four portions, one in front of
// each of the tile's barriers, so that a wave stalled at memory-instruction issue never holds the others up for long
// -- landed in LDS, and the finished outputs leave from the same LDS bytes (the epilogue of the last convolution
// overwrites the identity it has just read).  Waves 0-3 compute three implicit GEMMs on v_mfma_f32_32x32x16_f16 with
// the intermediate images in LDS:
//   1. y1 on all 100 staged pixels (the 3x3 needs its border), W1 in registers; pixels outside the image are written
//      as ZERO (they are conv2's padding, not relu(b1));
//   2. y2 on the 64 output pixels, W2 (64 x 576) resident in LDS for the whole kernel, a tap = an LDS address offset;
//   3. out on the 64 pixels x 256 channels, W3 in registers, identity from the staged x.
// Every stage keeps tile_gemm's arithmetic -- fp32 sums in ascending k (conv2: [tap][channel]), bias (+ identity) in
// fp32, ReLU, ONE rounding to binary16 -- so the result is bit-identical to the three separate hand-written kernels
// (tests/test_bottleneck_gpu.py).  LDS rows are padded (528 / 144 / 1 168 bytes, y1 tile rows 1 664) so that the 16-byte
// fragment reads of ds_read_b128's lane groups are conflict-free.
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace bevops {
namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kBT = 8, kBH = kBT + 2, kBP = kBH * kBH;      // 8 x 8 outputs, 10 x 10 = 100 staged pixels
constexpr int kBXPix = 256 * 2 + 16;                        // staged x: 528 bytes per pixel, linear pixel index
constexpr int kBYPix = 64 * 2 + 16;                         // y1 / y2: 144 bytes per pixel
constexpr int kBYRow = kBH * kBYPix + 224;                  // y1 tile row: 1 664 (= 32 banks mod 64)
constexpr int kBWRow = 9 * 64 * 2 + 16;                     // W2 row: 1 168
constexpr int kBW2 = 64 * kBWRow;                           // 74 752
constexpr int kBXs = kBP * kBXPix;                          // 52 800
constexpr int kBY1 = kBH * kBYRow;                          // 16 640
constexpr int kBY2 = kBT * kBT * kBYPix;                    // 9 216
constexpr int kBLds = kBW2 + kBXs + kBY1 + kBY2;            // 153 408
constexpr int kBThreads = 512, kBRole = 256;
constexpr int kBChunks = kBP * 32;                          // 16-byte pieces of a staged tile: 3 200
constexpr int kBRounds = (kBChunks + kBRole - 1) / kBRole;  // 13
constexpr unsigned kBOob = 0xFFFFFF00u;

struct BnArgs {
  const __half *x, *w1, *b1, *w2, *b2, *w3, *b3;
  __half *out;
  int H, W, tiles_x, tiles_img, tiles_total;
  unsigned x_bytes;
};

__global__ __launch_bounds__(kBThreads) void bottleneck_c256_64_kernel(BnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *W2s = smem, *Xs = smem + kBW2, *Y1s = Xs + kBXs, *Y2s = Y1s + kBY1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, n = lane & 31;
  const bool mover = wave >= 4;
  const int rt = tid & (kBRole - 1), cw = wave & 3;
  const int H = p.H, W = p.W, g = (int)gridDim.x;
  // ---- W2 [64][576] -> LDS, once (all eight waves)
  {
    uint4 wr[64 * 72 / kBThreads];
#pragma unroll
    for (int r = 0; r < 64 * 72 / kBThreads; ++r) wr[r] = reinterpret_cast<const uint4 *>(p.w2)[tid + kBThreads * r];
#pragma unroll
    for (int r = 0; r < 64 * 72 / kBThreads; ++r) {
      const int i = tid + kBThreads * r, row = i / 72, c = i - row * 72;
      *reinterpret_cast<uint4 *>(W2s + row * kBWRow + c * 16) = wr[r];
    }
  }
  auto tile_origin = [&](int t, int &b, int &ty0, int &tx0) {
    b = t / p.tiles_img;
    const int rem = t - b * p.tiles_img;
    const int ty = rem / p.tiles_x;
    ty0 = ty * kBT;
    tx0 = (rem - ty * p.tiles_x) * kBT;
  };
  int t = blockIdx.x;
  if (t >= p.tiles_total) return;

  if (mover) {
    // ================= waves 4-7
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half *>(p.x), 0, p.x_bytes, 0x00020000);
    // piece q = rt + 256 r of a staged tile -> (staged pixel q >> 5, 16-byte piece q & 31 = rt & 31)
    int hyx[kBRounds];
#pragma unroll
    for (int r = 0; r < kBRounds; ++r) {
      const int q = rt + kBRole * r, pix = q >> 5;
      const int hy = pix / kBH;
      hyx[r] = q < kBChunks ? ((hy << 8) | (pix - hy * kBH)) : -1;
    }
    uint4 pre[kBRounds];
    int nb = 0, ny0 = 0, nx0 = 0;
    auto request = [&](auto r0c, auto r1c) __attribute__((always_inline)) {
      constexpr int R0 = decltype(r0c)::value, R1 = decltype(r1c)::value;
#pragma unroll
      for (int r = R0; r < R1; ++r) {
        const int y = ny0 + (hyx[r] >> 8) - 1, xx = nx0 + (hyx[r] & 255) - 1;
        const bool in = hyx[r] >= 0 && y >= 0 && y < H && xx >= 0 && xx < W;
        const unsigned off = in ? (unsigned)(((((size_t)nb * H + y) * W + xx) * 256 + (rt & 31) * 8) * 2) : kBOob;
        pre[r] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 0));
      }
    };
    auto land = [&]() {
#pragma unroll
      for (int r = 0; r < kBRounds; ++r)
        if (hyx[r] >= 0) *reinterpret_cast<uint4 *>(Xs + (rt + kBRole * r) * 16 + ((rt + kBRole * r) >> 5) * 16) = pre[r];
    };
    auto store_outputs = [&](int tt) {
      int b, ty0, tx0;
      tile_origin(tt, b, ty0, tx0);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int q = rt + kBRole * r, pp = q >> 5;             // output pixel 0 .. 63, piece q & 31
        const int rr = pp >> 3, cc = pp & 7;
        const int y = ty0 + rr, xx = tx0 + cc;
        if (y < H && xx < W)
          *reinterpret_cast<uint4 *>(p.out + (((size_t)b * H + y) * W + xx) * 256 + (q & 31) * 8) =
              *reinterpret_cast<const uint4 *>(Xs + ((rr + 1) * kBH + cc + 1) * kBXPix + (q & 31) * 16);
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I4 = std::integral_constant<int, 4>;
    using I7 = std::integral_constant<int, 7>;
    using I10 = std::integral_constant<int, 10>;
    using I13 = std::integral_constant<int, kBRounds>;
    tile_origin(t, nb, ny0, nx0);
    request(I0{}, I13{});
    land();
    __syncthreads();            // B0: tile landed (and W2 in place)
    for (; t < p.tiles_total; t += g) {
      const bool more = t + g < p.tiles_total;
      if (more) tile_origin(t + g, nb, ny0, nx0);
      if (more) request(I0{}, I4{});
      __syncthreads();          // B1
      if (more) request(I4{}, I7{});
      __syncthreads();          // B2
      if (more) request(I7{}, I10{});
      __syncthreads();          // B3: the tile's outputs sit where its inner pixels were
      if (more) request(I10{}, I13{});
      store_outputs(t);
      if (more) land();
      __syncthreads();          // B0
    }
    return;
  }
  // ================= waves 0-3: three implicit GEMMs per tile
  // ---- stage 1 roles: column block cb1 (32 of y1's 64 channels), pixel blocks 2 (cw >> 1) + {0, 1} of the 128-row padded tile
  const int cb1 = cw & 1, pg1 = cw >> 1;
  f16x8_t w1f[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) w1f[s] = *reinterpret_cast<const f16x8_t *>(p.w1 + (size_t)(cb1 * 32 + n) * 256 + s * 16 + hi * 8);
  // ---- stage 3 roles: column blocks 2 cw + {0, 1} (of 8 x 32 output channels), both pixel blocks
  f16x8_t w3f[2][4];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int s = 0; s < 4; ++s) w3f[ci][s] = *reinterpret_cast<const f16x8_t *>(p.w3 + (size_t)((2 * cw + ci) * 32 + n) * 64 + s * 16 + hi * 8);
  // ---- stage 2 roles: pixel block pb2 (tile rows 4 pb2 .. + 3), column block cb2
  const int pb2 = cw >> 1, cb2 = cw & 1;
  const char *wa2 = W2s + (cb2 * 32 + n) * kBWRow + 16 * hi;
  const unsigned yb2 = (unsigned)((4 * pb2 + (n >> 3)) * kBYRow + (n & 7) * kBYPix + 16 * hi);
  auto bias4 = [&](const __half *b, int c0, float (&v)[4]) __attribute__((always_inline)) {
    const uint2 raw = b ? *reinterpret_cast<const uint2 *>(b + c0) : make_uint2(0, 0);
    v[0] = h2f_lo(raw.x); v[1] = h2f_hi(raw.x); v[2] = h2f_lo(raw.y); v[3] = h2f_hi(raw.y);
  };
  __syncthreads();              // B0
  for (; t < p.tiles_total; t += g) {
    int b, ty0, tx0;
    tile_origin(t, b, ty0, tx0);
    // ---------------- stage 1: y1[pixel][64] on the 100 staged pixels
    {
      f32x16_t acc[2];
      unsigned xo[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        const int pp = (2 * pg1 + k) * 32 + n;
        xo[k] = (unsigned)(min(pp, kBP - 1) * kBXPix + 16 * hi);      // rows past the 100th pixel: computed, never stored
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        f16x8_t xb[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) xb[k] = *reinterpret_cast<const f16x8_t *>(Xs + xo[k] + s * 32);
#pragma unroll
        for (int k = 0; k < 2; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1f[s], xb[k], acc[k], 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int pp = (2 * pg1 + k) * 32 + n;
        if (pp < kBP) {
          const int hy = pp / kBH, hx = pp - hy * kBH;
          const int y = ty0 + hy - 1, xx = tx0 + hx - 1;
          const bool in = y >= 0 && y < H && xx >= 0 && xx < W;
          char *dst = Y1s + hy * kBYRow + hx * kBYPix;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c0 = cb1 * 32 + 8 * q + 4 * hi;
            float bv[4];
            bias4(p.b1, c0, bv);
            uint2 o = make_uint2(0, 0);
            if (in) o = make_uint2(pack_h2(fmaxf(acc[k][4 * q] + bv[0], 0.f), fmaxf(acc[k][4 * q + 1] + bv[1], 0.f)),
                                   pack_h2(fmaxf(acc[k][4 * q + 2] + bv[2], 0.f), fmaxf(acc[k][4 * q + 3] + bv[3], 0.f)));
            *reinterpret_cast<uint2 *>(dst + c0 * 2) = o;
          }
        }
      }
    }
    __syncthreads();            // B1: y1 complete
    // ---------------- stage 2: y2[64 pixels][64] = conv3x3(y1)
    {
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int yoff = (tap / 3) * kBYRow + (tap % 3) * kBYPix;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const f16x8_t a = *reinterpret_cast<const f16x8_t *>(wa2 + tap * 128 + kk * 32);
          const f16x8_t bb = *reinterpret_cast<const f16x8_t *>(Y1s + yb2 + yoff + kk * 32);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, acc, 0, 0, 0);
        }
      }
      char *dst = Y2s + (pb2 * 32 + n) * kBYPix;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = cb2 * 32 + 8 * q + 4 * hi;
        float bv[4];
        bias4(p.b2, c0, bv);
        *reinterpret_cast<uint2 *>(dst + c0 * 2) =
            make_uint2(pack_h2(fmaxf(acc[4 * q] + bv[0], 0.f), fmaxf(acc[4 * q + 1] + bv[1], 0.f)),
                       pack_h2(fmaxf(acc[4 * q + 2] + bv[2], 0.f), fmaxf(acc[4 * q + 3] + bv[3], 0.f)));
      }
    }
    __syncthreads();            // B2: y2 complete
    // ---------------- stage 3: out[64 pixels][256] = relu(conv1x1(y2) + b3 + x), written over the staged x
    {
      f32x16_t acc[2][2];       // [column block][pixel block]
#pragma unroll
      for (int ci = 0; ci < 2; ++ci)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ci][k][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        f16x8_t yb[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) yb[k] = *reinterpret_cast<const f16x8_t *>(Y2s + (k * 32 + n) * kBYPix + s * 32 + 16 * hi);
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
          for (int k = 0; k < 2; ++k) acc[ci][k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3f[ci][s], yb[k], acc[ci][k], 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int pp = k * 32 + n, rr = pp >> 3, cc = pp & 7;
        char *px = Xs + ((rr + 1) * kBH + cc + 1) * kBXPix;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c0 = (2 * cw + ci) * 32 + 8 * q + 4 * hi;
            float bv[4];
            bias4(p.b3, c0, bv);
            const uint2 id = *reinterpret_cast<const uint2 *>(px + c0 * 2);
            const float v0 = acc[ci][k][4 * q] + bv[0] + h2f_lo(id.x), v1 = acc[ci][k][4 * q + 1] + bv[1] + h2f_hi(id.x);
            const float v2 = acc[ci][k][4 * q + 2] + bv[2] + h2f_lo(id.y), v3 = acc[ci][k][4 * q + 3] + bv[3] + h2f_hi(id.y);
            *reinterpret_cast<uint2 *>(px + c0 * 2) =
                make_uint2(pack_h2(fmaxf(v0, 0.f), fmaxf(v1, 0.f)), pack_h2(fmaxf(v2, 0.f), fmaxf(v3, 0.f)));
          }
      }
    }
    __syncthreads();            // B3
    __syncthreads();            // B0: outputs stored, next tile landed
  }
}

int bn_cu_count() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    hipDeviceProp_t pr;
    static int cached[16] = {0};
    if (cached[dev & 15] == 0 && hipGetDeviceProperties(&pr, dev) == hipSuccess) cached[dev & 15] = pr.multiProcessorCount;
    if (cached[dev & 15] > 0) cus = cached[dev & 15];
  }
  return cus;
}

}  // namespace
}  // namespace bevops

using namespace bevops;

// x [B, H, W, 256] channels-last fp16; w1 [64, 256], w2_taps [64, 3, 3, 64] (taps-major), w3 [256, 64]; b1 [64], b2 [64],
// b3 [256] fp16 (any may be null); out [B, H, W, 256] (may not alias x).  NOT_SUPPORTED for other channel counts.
extern "C" int bevops_bottleneck_c256_64_f16(const void *x, const void *w1, const void *b1, const void *w2_taps, const void *b2,
                                             const void *w3, const void *b3, void *out, int B, int H, int W, int Cin, int planes,
                                             void *stream) {
  if (!x || !w1 || !w2_taps || !w3 || !out || B <= 0 || H <= 0 || W <= 0 || x == out) return BEVOPS_BAD_PARAM;
  if (Cin != 256 || planes != 64) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(x) || !aligned16(w1) || !aligned16(w2_taps) || !aligned16(w3) || !aligned16(out) ||
      ((reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(b2) | reinterpret_cast<uintptr_t>(b3)) & 7u))
    return BEVOPS_BAD_PARAM;
  if ((long long)B * H * W * 512 >= 0xFFFFFF00ll) return BEVOPS_NOT_SUPPORTED;     // (32-bit buffer offsets)
  BnArgs a;
  a.x = static_cast<const __half *>(x); a.w1 = static_cast<const __half *>(w1); a.b1 = static_cast<const __half *>(b1);
  a.w2 = static_cast<const __half *>(w2_taps); a.b2 = static_cast<const __half *>(b2);
  a.w3 = static_cast<const __half *>(w3); a.b3 = static_cast<const __half *>(b3); a.out = static_cast<__half *>(out);
  a.H = H; a.W = W;
  a.tiles_x = (W + kBT - 1) / kBT;
  const int tiles_y = (H + kBT - 1) / kBT;
  a.tiles_img = a.tiles_x * tiles_y;
  const long long total = (long long)B * a.tiles_img;
  if (total > (1ll << 30)) return BEVOPS_NOT_SUPPORTED;
  a.tiles_total = (int)total;
  a.x_bytes = (unsigned)((size_t)B * H * W * 512);
  if (!ensure_dynamic_lds<bottleneck_c256_64_kernel>(kBLds)) return BEVOPS_FAILURE;
  const int blocks = (int)std::min<long long>(total, (long long)bn_cu_count());
  hipLaunchKernelGGL(bottleneck_c256_64_kernel, dim3((unsigned)blocks), dim3(kBThreads), kBLds, static_cast<hipStream_t>(stream), a);
  return launch_status();
}
