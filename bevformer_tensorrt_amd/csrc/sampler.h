// Device-side coordinate / interpolation vocabulary shared by rotate.hip and
// grid_sampler.hip.  Arithmetic follows the reference's PyTorch path
// (aten grid_sampler as called from det2trt/models/functions/grid_sampler.py:28-32 and
// functions/rotate.py:66), which the plugin kernels restate in
// TensorRT/plugin/grid_sampler/gridSamplerKernel.cu:82-260,373-437,457-559.
// All coordinate math is fp32 for every storage dtype.
#pragma once
#include "common.h"

namespace bevops {

// ---- element conversion -----------------------------------------------------
template <typename T> __device__ __forceinline__ float ld(const T *p);
template <> __device__ __forceinline__ float ld<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ld<__half>(const __half *p) { return __half2float(*p); }
template <> __device__ __forceinline__ float ld<int8_t>(const int8_t *p) { return (float)*p; }

// T2int8<float>: clamp, then round half away from zero (gridSamplerKernel.cu T2int8,
// multiScaleDeformableAttnKernel.cu:44-55)
__device__ __forceinline__ int8_t t2int8(float a) {
  a = a > 127.f ? 127.f : a;
  a = a < -128.f ? -128.f : a;
  return (int8_t)(a + (a > 0.f ? 0.5f : -0.5f));
}
// half2int8(v, 1/127): RNE then clamp (gridSamplerKernel.cu half2int8)
__device__ __forceinline__ int q127_rne(float area) {
  float r = rintf(area * 127.f);
  r = r > 127.f ? 127.f : r;
  r = r < -128.f ? -128.f : r;
  return (int)r;
}

template <typename T> __device__ __forceinline__ void st(T *p, float v, float oscale);
template <> __device__ __forceinline__ void st<float>(float *p, float v, float) { *p = v; }
template <> __device__ __forceinline__ void st<__half>(__half *p, float v, float) { *p = __float2half_rn(v); }
template <> __device__ __forceinline__ void st<int8_t>(int8_t *p, float v, float oscale) { *p = t2int8(v * oscale); }

// ---- coordinates --------------------------------------------------------------
__device__ __forceinline__ float gs_unnormalize(float c, int size, bool align) {
#pragma clang fp contract(off)
  if (align) return ((c + 1.f) / 2) * (float)(size - 1);
  return ((c + 1.f) * (float)size - 1.f) / 2;
}
__device__ __forceinline__ float gs_clip(float in, int limit) {
  return fminf((float)(limit - 1), fmaxf(in, 0.f));
}
__device__ __forceinline__ float gs_reflect(float in, int twice_low, int twice_high) {
#pragma clang fp contract(off)
  if (twice_low == twice_high) return 0.f;
  const float mn = (float)twice_low / 2;
  const float span = (float)(twice_high - twice_low) / 2;
  in = fabsf(in - mn);
  const float extra = fmodf(in, span);
  const int flips = (int)floorf(in / span);
  return (flips % 2 == 0) ? extra + mn : span - extra + mn;
}
__device__ __forceinline__ float gs_safe_int(float x) {
  if (x > 2147483646.f || x < -2147483648.f || !isfinite(x)) return -100.f;
  return x;
}
__device__ __forceinline__ float gs_coord(float c, int size, int pad, bool align) {
  if (pad == BEVOPS_PAD_BORDER) {
    c = gs_clip(c, size);
  } else if (pad == BEVOPS_PAD_REFLECTION) {
    c = align ? gs_reflect(c, 0, 2 * (size - 1)) : gs_reflect(c, -1, 2 * size - 1);
    c = gs_clip(c, size);
  }
  return gs_safe_int(c);
}
__device__ __forceinline__ float gs_source_index(float c, int size, int pad, bool align) {
  return gs_coord(gs_unnormalize(c, size, align), size, pad, align);
}
__device__ __forceinline__ bool in2d(int h, int w, int H, int W) {
  return h >= 0 && h < H && w >= 0 && w < W;
}

__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2) * x - (A + 3)) * x * x + 1; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A; }
__device__ __forceinline__ void cubic_coeffs(float (&c)[4], float t) {
#pragma clang fp contract(off)
  const float A = -0.75f;
  c[0] = cubic2(t + 1.0f, A);
  c[1] = cubic1(t, A);
  const float x2 = 1.0f - t;
  c[2] = cubic1(x2, A);
  c[3] = cubic2(x2 + 1.0f, A);
}

// A 2-D sampling footprint resolved once per output pixel and reused for every
// channel plane: up to 16 (bicubic) plane offsets + weights.  Offsets < 0 mean
// "contributes zero" (out of bounds under the padding mode).
template <int TAPS>
struct Footprint2D {
  int off[TAPS];
  float w[TAPS];
};

// bilinear (4 taps) / nearest (1 tap) source index already padded (ix, iy)
__device__ __forceinline__ void footprint_bilinear(float ix, float iy, int H, int W,
                                                   Footprint2D<4> &f) {
#pragma clang fp contract(off)
  const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
  const int x1 = x0 + 1, y1 = y0 + 1;
  f.w[0] = ((float)x1 - ix) * ((float)y1 - iy);  // nw
  f.w[1] = (ix - (float)x0) * ((float)y1 - iy);  // ne
  f.w[2] = ((float)x1 - ix) * (iy - (float)y0);  // sw
  f.w[3] = (ix - (float)x0) * (iy - (float)y0);  // se
  f.off[0] = in2d(y0, x0, H, W) ? y0 * W + x0 : -1;
  f.off[1] = in2d(y0, x1, H, W) ? y0 * W + x1 : -1;
  f.off[2] = in2d(y1, x0, H, W) ? y1 * W + x0 : -1;
  f.off[3] = in2d(y1, x1, H, W) ? y1 * W + x1 : -1;
}
__device__ __forceinline__ int footprint_nearest(float ix, float iy, int H, int W) {
  const int xn = (int)rintf(ix), yn = (int)rintf(iy);  // aten: std::nearbyint (RNE)
  return in2d(yn, xn, H, W) ? yn * W + xn : -1;
}

}  // namespace bevops
