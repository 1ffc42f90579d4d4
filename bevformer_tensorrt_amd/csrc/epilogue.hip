// bias_act: x[rows, C] (+ residual[rows, C]) + bias[C], optional ReLU, in place -- the epilogue
// chain of a folded-BN convolution (conv -> +shift -> (+identity) -> ReLU) as ONE pass over the
// activation instead of the 2-3 element-wise kernels (one of them a strided broadcast add) a
// framework launches after a library convolution.  Channels-last (NHWC) activations: the channel
// is the fastest dimension, so a lane's 8 channels are fixed and its bias vector is loaded once.
// Used by the re-hosted backbone; not a reference plugin.
#include "common.h"

namespace bevops {
namespace {

__global__ __launch_bounds__(256) void bias_act_f16_kernel(__half *__restrict__ x,
                                                           const __half *__restrict__ bias,
                                                           const __half *__restrict__ res, size_t nvec,
                                                           int C, int relu) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  // the grid stride is a multiple of C/8 vectors (host guarantees), so the channel stays fixed
  float b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (bias && i < nvec) {
    const uint4 bv = *reinterpret_cast<const uint4 *>(bias + (i % (size_t)(C / 8)) * 8);
    b[0] = h2f_lo(bv.x); b[1] = h2f_hi(bv.x); b[2] = h2f_lo(bv.y); b[3] = h2f_hi(bv.y);
    b[4] = h2f_lo(bv.z); b[5] = h2f_hi(bv.z); b[6] = h2f_lo(bv.w); b[7] = h2f_hi(bv.w);
  }
  for (; i < nvec; i += stride) {
    const uint4 v = reinterpret_cast<const uint4 *>(x)[i];
    float a[8] = {h2f_lo(v.x), h2f_hi(v.x), h2f_lo(v.y), h2f_hi(v.y),
                  h2f_lo(v.z), h2f_hi(v.z), h2f_lo(v.w), h2f_hi(v.w)};
    if (res) {
      const uint4 r = reinterpret_cast<const uint4 *>(res)[i];
      a[0] += h2f_lo(r.x); a[1] += h2f_hi(r.x); a[2] += h2f_lo(r.y); a[3] += h2f_hi(r.y);
      a[4] += h2f_lo(r.z); a[5] += h2f_hi(r.z); a[6] += h2f_lo(r.w); a[7] += h2f_hi(r.w);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a[k] += b[k];
      if (relu) a[k] = fmaxf(a[k], 0.f);
    }
    uint4 o;
    o.x = pack_h2(a[0], a[1]); o.y = pack_h2(a[2], a[3]);
    o.z = pack_h2(a[4], a[5]); o.w = pack_h2(a[6], a[7]);
    reinterpret_cast<uint4 *>(x)[i] = o;
  }
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" int bevops_bias_act_nhwc(int dtype, void *x, const void *bias, const void *residual,
                                    size_t rows, int channels, int relu, void *stream) {
  if (!x || channels <= 0) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16 || channels % 8 != 0) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(x) || (bias && !aligned16(bias)) || (residual && !aligned16(residual))) return BEVOPS_BAD_PARAM;
  if (rows == 0) return BEVOPS_SUCCESS;
  const size_t nvec = rows * (size_t)(channels / 8);
  const size_t vpr = (size_t)channels / 8;
  // blocks: enough to fill the chip, grid stride a multiple of the vectors per row
  size_t blocks = (nvec + 255) / 256;
  const size_t cap = 256 * 16;
  if (blocks > cap) blocks = cap;
  // make blocks*256 a multiple of vpr: round blocks up to a multiple of vpr / gcd(vpr, 256)
  size_t g = vpr, h = 256;
  while (h) { const size_t t = g % h; g = h; h = t; }
  const size_t unit = vpr / g;
  blocks = (blocks + unit - 1) / unit * unit;
  hipLaunchKernelGGL(bias_act_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (__half *)x, (const __half *)bias, (const __half *)residual, nvec, channels, relu);
  return launch_status();
}
