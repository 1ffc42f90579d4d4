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

// stem epilogue of the channels-last backbone: max_pool2d(relu(x + bias), 3, stride 2, pad 1) in ONE pass over the
// convolution's raw output (resnet.py: conv1 -> norm1 (folded) -> relu -> maxpool).  A thread owns 8 channels of
// one output pixel: up to nine 16-byte loads, max in fp32 of the shifted values, ReLU, one rounding -- rounding is
// monotonic, so this IS max over the fp16 values relu(x + bias) the two-pass form pools.
// OUT8: the pooled value leaves as int8, q = clamp(rne(v * (1 / s_out)), -127, 127) -- the first tensor of the INT8
// engine's int8 activation chain (the requantisation of tile_gemm.hip's OUT8 epilogue).
template <bool OUT8>
__global__ __launch_bounds__(256) void bias_relu_maxpool_f16_kernel(const __half *__restrict__ x,
                                                                    const __half *__restrict__ bias,
                                                                    void *__restrict__ out_, int B, int H, int W,
                                                                    int C, int Ho, int Wo, float inv_s_out) {
  const int cv = C / 8;
  const size_t total = (size_t)B * Ho * Wo * cv;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c8 = (int)(i % cv);
  size_t pix = i / cv;
  const int xo = (int)(pix % Wo);
  pix /= Wo;
  const int yo = (int)(pix % Ho), b = (int)(pix / Ho);
  float m[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int y = 2 * yo - 1 + dy;
    if (y < 0 || y >= H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int xx = 2 * xo - 1 + dx;
      if (xx < 0 || xx >= W) continue;
      const uint4 v = *reinterpret_cast<const uint4 *>(x + (((size_t)b * H + y) * W + xx) * C + c8 * 8);
      m[0] = fmaxf(m[0], h2f_lo(v.x)); m[1] = fmaxf(m[1], h2f_hi(v.x)); m[2] = fmaxf(m[2], h2f_lo(v.y));
      m[3] = fmaxf(m[3], h2f_hi(v.y)); m[4] = fmaxf(m[4], h2f_lo(v.z)); m[5] = fmaxf(m[5], h2f_hi(v.z));
      m[6] = fmaxf(m[6], h2f_lo(v.w)); m[7] = fmaxf(m[7], h2f_hi(v.w));
    }
  }
  float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (bias) {
    const uint4 bv = *reinterpret_cast<const uint4 *>(bias + c8 * 8);
    bb[0] = h2f_lo(bv.x); bb[1] = h2f_hi(bv.x); bb[2] = h2f_lo(bv.y); bb[3] = h2f_hi(bv.y);
    bb[4] = h2f_lo(bv.z); bb[5] = h2f_hi(bv.z); bb[6] = h2f_lo(bv.w); bb[7] = h2f_hi(bv.w);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k] + bb[k], 0.f);   // max(x) + b = max(x + b): the shift is per channel
  const size_t o_at = (((size_t)b * Ho + yo) * Wo + xo) * C + c8 * 8;
  if constexpr (OUT8) {
    unsigned pk[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < 8; ++k)
      pk[k >> 2] |= ((unsigned)(int)fminf(fmaxf(rintf(m[k] * inv_s_out), -127.f), 127.f) & 0xffu) << (8 * (k & 3));
    *reinterpret_cast<uint2 *>(static_cast<int8_t *>(out_) + o_at) = make_uint2(pk[0], pk[1]);
  } else {
    uint4 o;
    o.x = pack_h2(m[0], m[1]); o.y = pack_h2(m[2], m[3]); o.z = pack_h2(m[4], m[5]); o.w = pack_h2(m[6], m[7]);
    *reinterpret_cast<uint4 *>(static_cast<__half *>(out_) + o_at) = o;
  }
}

// layer_norm over the last dimension of x[rows, C], C = 8 * L with L in {8, 16, 32, 64} lanes per
// row: a lane keeps its 8 channels in registers (one 16-byte load), mean and variance are two
// shuffle reductions over the row's lanes in fp32 (two-pass: sum, then sum of squared deviations),
// so the activation makes one trip through HBM.  The framework's kernel spends 45 us on the
// encoder's [40 000, 256] fp16 rows (0.9 TB/s); this is a straight stream.
template <int L>
__global__ __launch_bounds__(256) void layer_norm_f16_kernel(const __half *__restrict__ x,
                                                             const __half *__restrict__ gamma,
                                                             const __half *__restrict__ beta,
                                                             __half *__restrict__ out, size_t rows, float eps) {
  constexpr int C = 8 * L;
  constexpr int RPB = 256 / L;  // rows per block
  const int lane = threadIdx.x % L;
  const size_t row = (size_t)blockIdx.x * RPB + threadIdx.x / L;
  const bool live = row < rows;  // dead lanes still take part in the shuffles
  float a[8];
  if (live) {
    const uint4 v = *reinterpret_cast<const uint4 *>(x + row * C + lane * 8);
    a[0] = h2f_lo(v.x); a[1] = h2f_hi(v.x); a[2] = h2f_lo(v.y); a[3] = h2f_hi(v.y);
    a[4] = h2f_lo(v.z); a[5] = h2f_hi(v.z); a[6] = h2f_lo(v.w); a[7] = h2f_hi(v.w);
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
  }
  float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
#pragma unroll
  for (int m = L / 2; m >= 1; m >>= 1) s += __shfl_xor(s, m, L);
  const float mean = s * (1.f / C);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] -= mean;
    q = fmaf(a[k], a[k], q);
  }
#pragma unroll
  for (int m = L / 2; m >= 1; m >>= 1) q += __shfl_xor(q, m, L);
  const float rstd = rsqrtf(q * (1.f / C) + eps);
  if (!live) return;
  float g[8], b[8];
  if (gamma) {
    const uint4 gv = *reinterpret_cast<const uint4 *>(gamma + lane * 8);
    g[0] = h2f_lo(gv.x); g[1] = h2f_hi(gv.x); g[2] = h2f_lo(gv.y); g[3] = h2f_hi(gv.y);
    g[4] = h2f_lo(gv.z); g[5] = h2f_hi(gv.z); g[6] = h2f_lo(gv.w); g[7] = h2f_hi(gv.w);
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = 1.f;
  }
  if (beta) {
    const uint4 bv = *reinterpret_cast<const uint4 *>(beta + lane * 8);
    b[0] = h2f_lo(bv.x); b[1] = h2f_hi(bv.x); b[2] = h2f_lo(bv.y); b[3] = h2f_hi(bv.y);
    b[4] = h2f_lo(bv.z); b[5] = h2f_hi(bv.z); b[6] = h2f_lo(bv.w); b[7] = h2f_hi(bv.w);
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = fmaf(a[k] * rstd, g[k], b[k]);
  uint4 o;
  o.x = pack_h2(a[0], a[1]); o.y = pack_h2(a[2], a[3]);
  o.z = pack_h2(a[4], a[5]); o.w = pack_h2(a[6], a[7]);
  *reinterpret_cast<uint4 *>(out + row * C + lane * 8) = o;
}

// FPN top-down step on channels-last activations: a[n, h, w, :] += b[n, sh(h), sw(w), :] with the source
// index of aten's nearest up-sampling (floor(dst * in / out) in float, clamped) -- the reference's
// `laterals[i - 1] += F.interpolate(laterals[i], size=prev_shape, mode="nearest")` (necks/fpn.py:170-176) as
// ONE pass instead of an up-sampled copy, a layout copy and an add.  thread = 8 channels of one pixel.
__global__ __launch_bounds__(256) void upsample_add_f16_kernel(__half *__restrict__ a, const __half *__restrict__ b,
                                                               int N, int H, int W, int Hb, int Wb, int C,
                                                               float sh, float sw) {
  const size_t vpp = (size_t)C / 8;
  const size_t nvec = (size_t)N * H * W * vpp;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const size_t pix = i / vpp;
  const int cv = (int)(i - pix * vpp);
  const int w = (int)(pix % (size_t)W);
  const size_t r = pix / (size_t)W;
  const int h = (int)(r % (size_t)H);
  const size_t n = r / (size_t)H;
  const int hb = min((int)floorf((float)h * sh), Hb - 1), wb = min((int)floorf((float)w * sw), Wb - 1);
  const uint4 x = reinterpret_cast<const uint4 *>(a)[i];
  const uint4 y = *reinterpret_cast<const uint4 *>(b + (((n * Hb + hb) * (size_t)Wb + wb) * C + (size_t)cv * 8));
  uint4 o;
  o.x = pack_h2(h2f_lo(x.x) + h2f_lo(y.x), h2f_hi(x.x) + h2f_hi(y.x));
  o.y = pack_h2(h2f_lo(x.y) + h2f_lo(y.y), h2f_hi(x.y) + h2f_hi(y.y));
  o.z = pack_h2(h2f_lo(x.z) + h2f_lo(y.z), h2f_hi(x.z) + h2f_hi(y.z));
  o.w = pack_h2(h2f_lo(x.w) + h2f_lo(y.w), h2f_hi(x.w) + h2f_hi(y.w));
  reinterpret_cast<uint4 *>(a)[i] = o;
}

// Encoder input assembly (modules/transformer.py:138-152): dst[n, row0 + r, :] = (src[n, r, :] + cam_embed[n, :])
// + level_embed[:], each sum rounded to binary16 like the two framework adds it replaces, written straight
// into the level's rows of the concatenated [cams, sum hw, C] feature tensor (no torch.cat copy).
__global__ __launch_bounds__(256) void feat_embed_f16_kernel(const __half *__restrict__ src,
                                                             const __half *__restrict__ cam,
                                                             const __half *__restrict__ lvl,
                                                             __half *__restrict__ dst, int N, size_t rows, int C,
                                                             size_t dst_stride) {
  const size_t vpr = (size_t)C / 8;
  const size_t per = rows * vpr;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per * (size_t)N) return;
  const size_t n = i / per, k = i - n * per;
  const int cv = (int)(k % vpr);
  const uint4 x = reinterpret_cast<const uint4 *>(src)[i];
  const uint4 c = *reinterpret_cast<const uint4 *>(cam + n * C + (size_t)cv * 8);
  const uint4 l = *reinterpret_cast<const uint4 *>(lvl + (size_t)cv * 8);
  auto two = [](unsigned a, unsigned b, unsigned d) {
    const unsigned t = pack_h2(h2f_lo(a) + h2f_lo(b), h2f_hi(a) + h2f_hi(b));
    return pack_h2(h2f_lo(t) + h2f_lo(d), h2f_hi(t) + h2f_hi(d));
  };
  uint4 o;
  o.x = two(x.x, c.x, l.x); o.y = two(x.y, c.y, l.y); o.z = two(x.z, c.z, l.z); o.w = two(x.w, c.w, l.w);
  *reinterpret_cast<uint4 *>(dst + n * dst_stride + k * 8) = o;
}

// Temporal self-attention glue (temporal_self_attention.py:350-457), two passes that replace three framework copies per
// encoder layer (round 5: 24 + 12 us -> 13 + 6 us at base).
// (1) the stacked projection's output row [heads][queue 2][points][xy] | [heads][queue 2][points] (the reference's
//     view(bs, nq, heads, bev_queue, levels, points, 2) of sampling_offsets / attention_weights) split into the
//     queue-major tensors the MSDA operator takes: off [2, nq, heads, points * 2], w [2, nq, heads, points].  Pure data
//     movement: thread = (query, head, queue), one 16-byte and one 8-byte piece (points == 4).
__global__ __launch_bounds__(256) void tsa_split_f16_kernel(const __half *__restrict__ both, __half *__restrict__ off,
                                                            __half *__restrict__ w, unsigned nq, unsigned heads) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  const unsigned per_q = heads * 2u;
  if (i >= nq * per_q) return;
  const unsigned q = i / per_q, hb = i - q * per_q, h = hb >> 1, b = hb & 1u;
  const unsigned row = per_q * 12u;                     // halves per projection row: heads * 2 * (4 * 2 + 4)
  const uint4 o = *reinterpret_cast<const uint4 *>(both + (size_t)q * row + hb * 8u);
  const uint2 l = *reinterpret_cast<const uint2 *>(both + (size_t)q * row + per_q * 8u + hb * 4u);
  const size_t item = ((size_t)b * nq + q) * heads + h;
  *reinterpret_cast<uint4 *>(off + item * 8u) = o;
  *reinterpret_cast<uint2 *>(w + item * 4u) = l;
}

// (2) the mean over the two BEV-queue entries of the sampled features (torch.mean(dim=0): fp32 sum, one rounding):
//     out[i] = (x[i] + x[n + i]) / 2, 8 halves per thread.
__global__ __launch_bounds__(256) void queue_mean2_f16_kernel(const __half *__restrict__ x, __half *__restrict__ out,
                                                              size_t nvec) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const uint4 a = reinterpret_cast<const uint4 *>(x)[i], b = reinterpret_cast<const uint4 *>(x)[nvec + i];
  uint4 o;
  o.x = pack_h2((h2f_lo(a.x) + h2f_lo(b.x)) * 0.5f, (h2f_hi(a.x) + h2f_hi(b.x)) * 0.5f);
  o.y = pack_h2((h2f_lo(a.y) + h2f_lo(b.y)) * 0.5f, (h2f_hi(a.y) + h2f_hi(b.y)) * 0.5f);
  o.z = pack_h2((h2f_lo(a.z) + h2f_lo(b.z)) * 0.5f, (h2f_hi(a.z) + h2f_hi(b.z)) * 0.5f);
  o.w = pack_h2((h2f_lo(a.w) + h2f_lo(b.w)) * 0.5f, (h2f_hi(a.w) + h2f_hi(b.w)) * 0.5f);
  reinterpret_cast<uint4 *>(out)[i] = o;
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" int bevops_tsa_split(int dtype, const void *both, void *offsets, void *weights, int num_query, int heads,
                                int points, void *stream) {
  if (!both || !offsets || !weights || num_query < 0 || heads <= 0) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16 || points != 4) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(both) || !aligned16(offsets) || (reinterpret_cast<uintptr_t>(weights) & 7u)) return BEVOPS_BAD_PARAM;
  const size_t n = (size_t)num_query * heads * 2;
  if (n == 0) return BEVOPS_SUCCESS;
  if ((n + 255) / 256 > 0x7fffffffull) return BEVOPS_NOT_SUPPORTED;
  hipLaunchKernelGGL(tsa_split_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const __half *)both, (__half *)offsets, (__half *)weights,
                     (unsigned)num_query, (unsigned)heads);
  return launch_status();
}

extern "C" int bevops_queue_mean2(int dtype, const void *x, void *out, size_t count, void *stream) {
  if (!x || !out) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16 || count % 8 != 0) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(x) || !aligned16(out)) return BEVOPS_BAD_PARAM;
  const size_t nvec = count / 8;
  if (nvec == 0) return BEVOPS_SUCCESS;
  if ((nvec + 255) / 256 > 0x7fffffffull) return BEVOPS_NOT_SUPPORTED;
  hipLaunchKernelGGL(queue_mean2_f16_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const __half *)x, (__half *)out, nvec);
  return launch_status();
}

extern "C" int bevops_upsample_add_nhwc(int dtype, void *a, const void *b, int n, int h, int w, int hb, int wb,
                                        int channels, void *stream) {
  if (!a || !b || n <= 0 || h <= 0 || w <= 0 || hb <= 0 || wb <= 0 || channels <= 0) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16 || channels % 8 != 0) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(a) || !aligned16(b)) return BEVOPS_BAD_PARAM;
  const size_t nvec = (size_t)n * h * w * (channels / 8);
  if ((nvec + 255) / 256 > 0x7fffffffull) return BEVOPS_NOT_SUPPORTED;
  hipLaunchKernelGGL(upsample_add_f16_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (__half *)a, (const __half *)b, n, h, w, hb, wb, channels,
                     (float)hb / (float)h, (float)wb / (float)w);
  return launch_status();
}

extern "C" int bevops_feat_embed_nhwc(int dtype, const void *src, const void *cam_embed, const void *level_embed,
                                      void *dst, int n, size_t rows, int channels, size_t dst_batch_stride,
                                      void *stream) {
  if (!src || !cam_embed || !level_embed || !dst || n <= 0 || channels <= 0) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16 || channels % 8 != 0 || dst_batch_stride % 8 != 0) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(src) || !aligned16(cam_embed) || !aligned16(level_embed) || !aligned16(dst)) return BEVOPS_BAD_PARAM;
  if (rows == 0) return BEVOPS_SUCCESS;
  const size_t nvec = (size_t)n * rows * (channels / 8);
  if ((nvec + 255) / 256 > 0x7fffffffull) return BEVOPS_NOT_SUPPORTED;
  hipLaunchKernelGGL(feat_embed_f16_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const __half *)src, (const __half *)cam_embed,
                     (const __half *)level_embed, (__half *)dst, n, rows, channels, dst_batch_stride);
  return launch_status();
}

extern "C" int bevops_layer_norm(int dtype, const void *x, const void *gamma, const void *beta, void *out,
                                 size_t rows, int channels, float eps, void *stream) {
  if (!x || !out || channels <= 0 || !(eps >= 0.f)) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (channels != 64 && channels != 128 && channels != 256 && channels != 512) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(x) || !aligned16(out) || (gamma && !aligned16(gamma)) || (beta && !aligned16(beta)))
    return BEVOPS_BAD_PARAM;
  if (rows == 0) return BEVOPS_SUCCESS;
  const int L = channels / 8;
  const size_t blocks = (rows + (256 / L) - 1) / (256 / L);
  if (blocks > 0x7FFFFFFFull) return BEVOPS_NOT_SUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)blocks), blk(256);
#define BEVOPS_LN(LL)                                                                                  \
  hipLaunchKernelGGL(layer_norm_f16_kernel<LL>, grid, blk, 0, st, (const __half *)x, (const __half *)gamma, \
                     (const __half *)beta, (__half *)out, rows, eps)
  switch (L) {
    case 8: BEVOPS_LN(8); break;
    case 16: BEVOPS_LN(16); break;
    case 32: BEVOPS_LN(32); break;
    default: BEVOPS_LN(64); break;
  }
#undef BEVOPS_LN
  return launch_status();
}

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

extern "C" int bevops_bias_relu_maxpool_nhwc(int dtype, const void *x, const void *bias, void *out, int n, int h, int w,
                                             int channels, void *stream) {
  if (!x || !out || n <= 0 || h <= 0 || w <= 0 || channels <= 0) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16 || channels % 8 != 0 || !aligned16(x) || !aligned16(out) || !aligned16(bias))
    return BEVOPS_NOT_SUPPORTED;
  const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;     // floor((h + 2 - 3) / 2) + 1
  const size_t total = (size_t)n * ho * wo * (channels / 8);
  hipLaunchKernelGGL(bias_relu_maxpool_f16_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const __half *)x, (const __half *)bias, out, n, h, w,
                     channels, ho, wo, 0.f);
  return launch_status();
}

extern "C" int bevops_bias_relu_maxpool_nhwc_int8(int dtype, const void *x, const void *bias, void *out_q,
                                                  float scale_out, int n, int h, int w, int channels, void *stream) {
  if (!x || !out_q || n <= 0 || h <= 0 || w <= 0 || channels <= 0 || !(scale_out > 0.f)) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16 || channels % 8 != 0 || !aligned16(x) || (reinterpret_cast<uintptr_t>(out_q) & 7u) ||
      !aligned16(bias))
    return BEVOPS_NOT_SUPPORTED;
  const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
  const size_t total = (size_t)n * ho * wo * (channels / 8);
  hipLaunchKernelGGL(bias_relu_maxpool_f16_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const __half *)x, (const __half *)bias, out_q, n, h, w,
                     channels, ho, wo, 1.0f / scale_out);
  return launch_status();
}
