// rotate: prev_bev [C,H,W] rotated by `angle` degrees about `center`, zeros padding,
// align_corners=False, bilinear or nearest.  Replaces RotatePlugin::enqueue
// (TensorRT/plugin/rotate/rotatePlugin.cpp:75-116) and rotate<T>/rotate_h2/rotate_int8
// (rotateKernel.cu:128-748); arithmetic = the PyTorch path functions/rotate.py:12-80.
//
// MI355X mapping: the reference runs H*W threads, each looping over all C planes
// (40 000 threads for the 200x200 BEV = 2.4 waves per CU).  Here the grid is
// (pixel tiles) x (channel chunks): every thread resolves the affine source
// footprint once and moves CPT channel planes, so the 256-channel BEV launches
// ~5 000 workgroups; loads/stores are coalesced along W inside each plane.
#include "sampler.h"

namespace bevops {
namespace {

constexpr int kBlock = 256;
constexpr int kCPT = 8;  // channel planes per thread
thread_local bool g_rotate_narrow = false;   // bevops_rotate_set_variant(1): per-lane stores (the round 1-3 kernel; A/B)

template <typename A> __device__ __forceinline__ float scalar_at(const void *p, int i);
template <> __device__ __forceinline__ float scalar_at<float>(const void *p, int i) {
  return static_cast<const float *>(p)[i];
}
template <> __device__ __forceinline__ float scalar_at<__half>(const void *p, int i) {
  return __half2float(static_cast<const __half *>(p)[i]);
}

// source pixel coordinates of output pixel (w, h): functions/rotate.py:15-48,66
template <typename A>
__device__ __forceinline__ void rotate_source(const void *angle, const void *center, int w, int h, int H, int W,
                                              float &ix, float &iy) {
  float gx, gy;
  {
#pragma clang fp contract(off)
    // functions/rotate.py:15-28
    const float cx = scalar_at<A>(center, 0) - (float)(W * 0.5);
    const float cy = scalar_at<A>(center, 1) - (float)(H * 0.5);
    const float ang = -scalar_at<A>(angle, 0) * 3.14159265358979323846f / 180.f;
    float sn, cs;
    sincosf(ang, &sn, &cs);
    const float t02 = -cx * cs - cy * sn + cx;
    const float t12 = cx * sn - cy * cs + cy;
    // rescaled_theta = 2 * theta^T / (W, H)   (:44-46);  grid = base_grid @ rescaled_theta (:48)
    const float ax = 2 * cs / W, bx = 2 * sn / W, cx2 = 2 * t02 / W;
    const float ay = 2 * -sn / H, by = 2 * cs / H, cy2 = 2 * t12 / H;
    const float x = (float)(-W * 0.5 + 0.5) + (float)w, y = (float)(-H * 0.5 + 0.5) + (float)h;
    gx = (x * ax + y * bx) + cx2;
    gy = (x * ay + y * by) + cy2;
  }
  ix = gs_source_index(gx, W, BEVOPS_PAD_ZEROS, false);
  iy = gs_source_index(gy, H, BEVOPS_PAD_ZEROS, false);
}

// Channels-last variant: img / out are [H, W, C] (the layout prev_bev [H*W, 1, C] already has in the
// model, transformer.py:296-303 permutes it to [C, H, W] and back around the plugin).  A thread
// moves one 16-byte channel vector of one pixel, so a pixel's channels are one contiguous run: the
// nearest mode the model uses is a gather of whole 512-byte rows.  Same arithmetic as above.
template <typename T, typename A>
__global__ __launch_bounds__(kBlock) void rotate_hwc_kernel(const T *__restrict__ img,
                                                            const void *__restrict__ angle,
                                                            const void *__restrict__ center,
                                                            T *__restrict__ out, int C, int H, int W,
                                                            int interp) {
  constexpr int V = 16 / sizeof(T);
  const int vpp = C / V;  // vectors per pixel
  const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= (size_t)H * W * vpp) return;
  const int pix = (int)(idx / vpp), v = (int)(idx - (size_t)pix * vpp);
  const int w = pix % W, h = pix / W;
  float ix, iy;
  rotate_source<A>(angle, center, w, h, H, W, ix, iy);
  uint4 *op = reinterpret_cast<uint4 *>(out + (size_t)pix * C) + v;
  if (interp == BEVOPS_NEAREST) {
    const int o = footprint_nearest(ix, iy, H, W);
    *op = o >= 0 ? reinterpret_cast<const uint4 *>(img + (size_t)o * C)[v] : make_uint4(0, 0, 0, 0);
    return;
  }
  Footprint2D<4> f;
  footprint_bilinear(ix, iy, H, W, f);
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (f.off[k] >= 0) {
      const uint4 raw = reinterpret_cast<const uint4 *>(img + (size_t)f.off[k] * C)[v];
      const T *e = reinterpret_cast<const T *>(&raw);
#pragma unroll
      for (int j = 0; j < V; ++j) {
#pragma clang fp contract(off)
        acc[j] += ld<T>(e + j) * f.w[k];
      }
    }
  uint4 res;
  T *r = reinterpret_cast<T *>(&res);
#pragma unroll
  for (int j = 0; j < V; ++j) st<T>(r + j, acc[j], 1.f);
  *op = res;
}

// WIDE: a thread's kCPT values do not leave as kCPT stores of sizeof(T) bytes per lane (2-byte stores for fp16: the
// launch was store-ISSUE-bound, 26 % of the HBM roofline) but through a [kCPT][kBlock] LDS tile read back in 16-byte
// runs of consecutive pixels of one plane: one 16-byte store per thread (fp16) instead of eight 2-byte ones.  Needs
// plane starts 16-byte aligned (H W sizeof(T) % 16 == 0, aligned `out`): the host picks the flavour.
template <typename T, typename A, bool WIDE>
__global__ __launch_bounds__(kBlock) void rotate_kernel(const T *__restrict__ img,
                                                        const void *__restrict__ angle,
                                                        const void *__restrict__ center,
                                                        T *__restrict__ out, int C, int H, int W,
                                                        int interp, float s_in, float s_out) {
  __shared__ __attribute__((aligned(16))) T tile[WIDE ? kCPT : 1][WIDE ? kBlock : 1];
  const int p0 = blockIdx.x * kBlock;
  const int pix = p0 + threadIdx.x;
  const int HW = H * W;
  const bool valid = pix < HW;
  if (!WIDE && !valid) return;
  const int c0 = blockIdx.y * kCPT;
  const int c1 = min(c0 + kCPT, C);
  constexpr bool kInt8 = sizeof(T) == 1;
  if (valid) {
    const int w = pix % W, h = pix / W;
    float ix, iy;
    rotate_source<A>(angle, center, w, h, H, W, ix, iy);
    const T *ip = img + (size_t)c0 * HW;
    T *op = out + (size_t)c0 * HW + pix;
    auto emit = [&](int c, T v) {
      if constexpr (WIDE) tile[c - c0][threadIdx.x] = v;
      else op[(size_t)(c - c0) * HW] = v;
    };
    if (interp == BEVOPS_NEAREST) {
      const int o = footprint_nearest(ix, iy, H, W);
      const float os = kInt8 ? s_in / s_out : 1.f;
      for (int c = c0; c < c1; ++c, ip += HW) {
        T v;
        if (o >= 0) {
          if constexpr (kInt8) st<T>(&v, ld<T>(ip + o), os);
          else v = ip[o];  // pure copy: bit-exact for fp16/fp32
        } else {
          st<T>(&v, 0.f, 1.f);
        }
        emit(c, v);
      }
    } else {
      Footprint2D<4> f;
      footprint_bilinear(ix, iy, H, W, f);
      if constexpr (kInt8) {
        // rotateKernel.cu:463-540: int8 area weights x127, int32 dot, requantise
        int wq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) wq[k] = q127_rne(f.w[k]);
        const float os = (1.f / 127.f) * s_in / s_out;
        for (int c = c0; c < c1; ++c, ip += HW) {
          int t = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (f.off[k] >= 0) t += (int)ip[f.off[k]] * wq[k];
          emit(c, t2int8((float)t * os));
        }
      } else {
        for (int c = c0; c < c1; ++c, ip += HW) {
          float o = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (f.off[k] >= 0) {
#pragma clang fp contract(off)
              o += ld<T>(ip + f.off[k]) * f.w[k];
            }
          T v;
          st<T>(&v, o, 1.f);
          emit(c, v);
        }
      }
    }
  }
  if constexpr (WIDE) {
    __syncthreads();
    constexpr int VPX = 16 / sizeof(T);             // pixels per 16-byte run
    constexpr int kChunks = kBlock / VPX;           // runs per plane of this block's pixel range
    for (int idx = threadIdx.x; idx < kCPT * kChunks; idx += kBlock) {
      const int plane = idx / kChunks, chunk = idx - plane * kChunks;
      const int px = p0 + chunk * VPX;
      if (c0 + plane < c1 && px < HW)               // (H W % VPX == 0: a run is all in or all out)
        *reinterpret_cast<uint4 *>(out + (size_t)(c0 + plane) * HW + px) =
            *reinterpret_cast<const uint4 *>(&tile[plane][chunk * VPX]);
    }
  }
}

template <typename T>
int launch(const void *img, const void *angle, const void *center, int angle_dtype, void *out,
           int C, int H, int W, int interp, float s_in, float s_out, hipStream_t st) {
  const dim3 grid((unsigned)((H * W + kBlock - 1) / kBlock), (unsigned)((C + kCPT - 1) / kCPT));
  // wide (LDS-transposed) stores when every plane starts on a 16-byte boundary; variant 1 keeps the per-lane stores (A/B)
  const bool wide = !g_rotate_narrow && ((size_t)H * W * sizeof(T)) % 16 == 0 && aligned16(out);
#define BEVOPS_ROT(A_, WIDE_)                                                                                   \
  hipLaunchKernelGGL((rotate_kernel<T, A_, WIDE_>), grid, dim3(kBlock), 0, st, (const T *)img, angle, center, \
                     (T *)out, C, H, W, interp, s_in, s_out)
  if (angle_dtype == BEVOPS_F32) {
    if (wide) BEVOPS_ROT(float, true);
    else BEVOPS_ROT(float, false);
  } else {
    if (wide) BEVOPS_ROT(__half, true);
    else BEVOPS_ROT(__half, false);
  }
#undef BEVOPS_ROT
  return launch_status();
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" int bevops_rotate_forward(int dtype, const void *img, const void *angle,
                                     const void *center, int angle_dtype, void *output,
                                     int channels, int height, int width, int interpolation,
                                     float scale_in, float scale_out, void *stream) {
  if (!img || !angle || !center || !output) return BEVOPS_BAD_PARAM;
  if (channels <= 0 || height <= 0 || width <= 0) return BEVOPS_BAD_PARAM;
  if (interpolation != BEVOPS_BILINEAR && interpolation != BEVOPS_NEAREST) return BEVOPS_BAD_PARAM;
  if (angle_dtype != BEVOPS_F32 && angle_dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if ((long)height * width > 0x7FFFFFFFL || (long)channels > 65535L * kCPT)
    return BEVOPS_NOT_SUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case BEVOPS_F32:
      if (angle_dtype != BEVOPS_F32) return BEVOPS_NOT_SUPPORTED;  // rotatePlugin.cpp:152-155
      return launch<float>(img, angle, center, angle_dtype, output, channels, height, width,
                           interpolation, 1.f, 1.f, st);
    case BEVOPS_F16:
      return launch<__half>(img, angle, center, angle_dtype, output, channels, height, width,
                            interpolation, 1.f, 1.f, st);
    case BEVOPS_I8:
      if (!(scale_in > 0.f) || !(scale_out > 0.f)) return BEVOPS_BAD_PARAM;
      return launch<int8_t>(img, angle, center, angle_dtype, output, channels, height, width,
                            interpolation, scale_in, scale_out, st);
    default:
      return BEVOPS_NOT_SUPPORTED;
  }
}

extern "C" int bevops_rotate_forward_hwc(int dtype, const void *img, const void *angle, const void *center,
                                         int angle_dtype, void *output, int channels, int height, int width,
                                         int interpolation, void *stream) {
  if (!img || !angle || !center || !output) return BEVOPS_BAD_PARAM;
  if (channels <= 0 || height <= 0 || width <= 0) return BEVOPS_BAD_PARAM;
  if (interpolation != BEVOPS_BILINEAR && interpolation != BEVOPS_NEAREST) return BEVOPS_BAD_PARAM;
  if (angle_dtype != BEVOPS_F32 && angle_dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (dtype != BEVOPS_F32 && dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (dtype == BEVOPS_F32 && angle_dtype != BEVOPS_F32) return BEVOPS_NOT_SUPPORTED;
  const int V = dtype == BEVOPS_F32 ? 4 : 8;
  if (channels % V != 0) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(img) || !aligned16(output)) return BEVOPS_BAD_PARAM;
  if ((long)height * width > 0x7FFFFFFFL) return BEVOPS_NOT_SUPPORTED;
  const size_t threads = (size_t)height * width * (channels / V);
  const size_t blocks = (threads + kBlock - 1) / kBlock;
  if (blocks > 0x7FFFFFFFull) return BEVOPS_NOT_SUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)blocks), blk(kBlock);
  if (dtype == BEVOPS_F32)
    hipLaunchKernelGGL((rotate_hwc_kernel<float, float>), grid, blk, 0, st, (const float *)img, angle, center,
                       (float *)output, channels, height, width, interpolation);
  else if (angle_dtype == BEVOPS_F32)
    hipLaunchKernelGGL((rotate_hwc_kernel<__half, float>), grid, blk, 0, st, (const __half *)img, angle, center,
                       (__half *)output, channels, height, width, interpolation);
  else
    hipLaunchKernelGGL((rotate_hwc_kernel<__half, __half>), grid, blk, 0, st, (const __half *)img, angle, center,
                       (__half *)output, channels, height, width, interpolation);
  return launch_status();
}

// A/B switch (thread-local): 1 = the per-lane stores of rounds 1-3, 0 = LDS-transposed 16-byte stores where the
// plane alignment allows.  Returns the previous value.
extern "C" int bevops_rotate_set_variant(int variant) {
  const int prev = g_rotate_narrow ? 1 : 0;
  g_rotate_narrow = variant == 1;
  return prev;
}
