// Camera-image front end of the frame loop: the reference's test pipeline
// NormalizeMultiviewImage -> PadMultiViewImage(size_divisor=32) -> DefaultFormatBundle3D
// (configs/bevformer/bevformer_base.py:11,228-231; third_party/bev_mmdet3d/datasets/pipelines/
// transform_3d.py:99-150 and mmcv.imnormalize) as ONE pass: raw [N, H0, W0, 3] camera images (uint8
// or fp32, BGR as cv2 loads them) -> (x - mean[c]) * (1 / std[c]) in fp32 (optionally after the
// BGR -> RGB swap), zero padding at the bottom / right to [Hp, Wp], channel-first planes or
// channels-last rows, fp16 or fp32 out.  53 MB written per base frame instead of four
// host-side numpy passes and a host-to-device copy of fp32 planes.  Not a reference plugin.
#include "common.h"

namespace bevops {
namespace {

// float -> output type as its OWN rounding step: a plain cast lets the compiler fold the multiply and
// the conversion into one v_fma_mixlo_f16 (single rounding of the exact product), while the
// reference multiplies in float32 and converts afterwards
__device__ __forceinline__ float to_out(float v, float *) { return v; }
__device__ __forceinline__ __half to_out(float v, __half *) {
  unsigned r;
  asm("v_cvt_f16_f32 %0, %1" : "=v"(r) : "v"(v));
  return __ushort_as_half((unsigned short)r);
}

template <typename In, typename Out, bool NHWC>
__global__ __launch_bounds__(256) void image_normalize_pad_kernel(const In *__restrict__ img, Out *__restrict__ out,
                                                                  int N, int H0, int W0, int Hp, int Wp, float m0,
                                                                  float m1, float m2, float i0, float i1, float i2,
                                                                  int to_rgb) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t plane = (size_t)Hp * Wp;
  if (idx >= (size_t)N * plane) return;
  const int n = (int)(idx / plane);
  const size_t r = idx - (size_t)n * plane;
  const int y = (int)(r / Wp), x = (int)(r - (size_t)y * Wp);
  float v[3] = {0.f, 0.f, 0.f};
  if (y < H0 && x < W0) {
    const In *p = img + (((size_t)n * H0 + y) * W0 + x) * 3;
    float a = (float)p[0], b = (float)p[1], c = (float)p[2];
    if (to_rgb) { const float t = a; a = c; c = t; }
    {
#pragma clang fp contract(off)
      v[0] = (a - m0) * i0;   // cv2.subtract, then cv2.multiply by 1 / std: two roundings
      v[1] = (b - m1) * i1;
      v[2] = (c - m2) * i2;
    }
  }
  if constexpr (NHWC) {
    Out *o = out + idx * 3;
    o[0] = to_out(v[0], (Out *)nullptr); o[1] = to_out(v[1], (Out *)nullptr); o[2] = to_out(v[2], (Out *)nullptr);
  } else {
    Out *o = out + (size_t)n * 3 * plane + r;
    o[0] = to_out(v[0], (Out *)nullptr); o[plane] = to_out(v[1], (Out *)nullptr); o[2 * plane] = to_out(v[2], (Out *)nullptr);
  }
}

template <typename In, typename Out>
int launch(const void *img, void *out, int N, int H0, int W0, int Hp, int Wp, const float *mean, const float *stdinv,
           int to_rgb, int channels_last, hipStream_t st) {
  const size_t total = (size_t)N * Hp * Wp;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (channels_last)
    hipLaunchKernelGGL((image_normalize_pad_kernel<In, Out, true>), grid, dim3(256), 0, st, (const In *)img, (Out *)out, N,
                       H0, W0, Hp, Wp, mean[0], mean[1], mean[2], stdinv[0], stdinv[1], stdinv[2], to_rgb);
  else
    hipLaunchKernelGGL((image_normalize_pad_kernel<In, Out, false>), grid, dim3(256), 0, st, (const In *)img, (Out *)out, N,
                       H0, W0, Hp, Wp, mean[0], mean[1], mean[2], stdinv[0], stdinv[1], stdinv[2], to_rgb);
  return launch_status();
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" int bevops_image_normalize_pad(int in_dtype, const void *images, int out_dtype, void *output, int N,
                                          int H0, int W0, int Hp, int Wp, const double *mean_host,
                                          const double *std_host, int to_rgb, int channels_last, void *stream) {
  if (!images || !output || !mean_host || !std_host) return BEVOPS_BAD_PARAM;
  if (N <= 0 || H0 <= 0 || W0 <= 0 || Hp < H0 || Wp < W0) return BEVOPS_BAD_PARAM;
  if ((size_t)N * Hp * Wp / 256 >= 0x7fffffffull) return BEVOPS_NOT_SUPPORTED;
  float inv[3], mean[3];
  for (int c = 0; c < 3; ++c) {
    if (!(std_host[c] > 0.0)) return BEVOPS_BAD_PARAM;
    inv[c] = (float)(1.0 / std_host[c]);   // mmcv.imnormalize: stdinv = 1 / float64(std), applied in float32
    mean[c] = (float)mean_host[c];
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool u8 = in_dtype == BEVOPS_U8, f32in = in_dtype == BEVOPS_F32;
  if (!u8 && !f32in) return BEVOPS_NOT_SUPPORTED;
  if (out_dtype == BEVOPS_F16)
    return u8 ? launch<uint8_t, __half>(images, output, N, H0, W0, Hp, Wp, mean, inv, to_rgb, channels_last, st)
              : launch<float, __half>(images, output, N, H0, W0, Hp, Wp, mean, inv, to_rgb, channels_last, st);
  if (out_dtype == BEVOPS_F32)
    return u8 ? launch<uint8_t, float>(images, output, N, H0, W0, Hp, Wp, mean, inv, to_rgb, channels_last, st)
              : launch<float, float>(images, output, N, H0, W0, Hp, Wp, mean, inv, to_rgb, channels_last, st);
  return BEVOPS_NOT_SUPPORTED;
}
