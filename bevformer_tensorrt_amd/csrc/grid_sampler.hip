// grid_sampler 2-D (bilinear / nearest / bicubic x zeros / border / reflection x
// align_corners) and 3-D (trilinear / nearest).  Grid is channel-first
// [N,2,Ho,Wo] / [N,3,Do,Ho,Wo] in [-10,10] units, exactly the reference op's
// contract (det2trt/models/functions/grid_sampler.py:140-236).  Replaces
// GridSamplerPlugin::enqueue (TensorRT/plugin/grid_sampler/gridSamplerPlugin.cpp:110-156)
// and grid_sample<T>/grid_sample_int8 (gridSamplerKernel.cu:1933-2043).
//
// MI355X mapping: (output pixel tiles) x (channel chunks) grid -- the footprint
// (source offsets + weights) of an output pixel is resolved once in registers and
// reused for kCPT channel planes; grid reads and output writes are coalesced along
// W_out.  The reference loops all C planes in one thread.
#include "sampler.h"

namespace bevops {
namespace {

constexpr int kBlock = 256;
constexpr int kCPT = 8;

struct Gs2dDims {
  int N, C, H, W, Ho, Wo;
};

template <typename T>
__device__ __forceinline__ float bounded2d(const T *p, float x, float y, int W, int H, int pad,
                                           bool align) {
  x = gs_coord(x, W, pad, align);
  y = gs_coord(y, H, pad, align);
  const int ix = (int)x, iy = (int)y;
  return in2d(iy, ix, H, W) ? ld<T>(p + iy * W + ix) : 0.f;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void grid_sampler_2d_kernel(
    const T *__restrict__ input, const T *__restrict__ grid, T *__restrict__ out, Gs2dDims d,
    int interp, int pad, int align_i, float s_in, float s_grid, float s_out) {
  const long pix = (long)blockIdx.x * kBlock + threadIdx.x;
  const long plane_o = (long)d.Ho * d.Wo;
  if (pix >= plane_o * d.N) return;
  const int n = (int)(pix / plane_o);
  const long s = pix - (long)n * plane_o;
  const bool align = align_i != 0;
  constexpr bool kInt8 = sizeof(T) == 1;
  float gx, gy;
  {
#pragma clang fp contract(off)
    const float gs = kInt8 ? s_grid : 1.f;
    gx = ld<T>(grid + ((long)n * 2 + 0) * plane_o + s) * gs / 10;  // grid_sampler.py:28-29
    gy = ld<T>(grid + ((long)n * 2 + 1) * plane_o + s) * gs / 10;
  }
  const int HW = d.H * d.W;
  const int c0 = blockIdx.y * kCPT, c1 = min(c0 + kCPT, d.C);
  const T *ip = input + ((size_t)n * d.C + c0) * HW;
  T *op = out + ((size_t)n * d.C + c0) * plane_o + s;

  if (interp == BEVOPS_BILINEAR) {
    const float ix = gs_source_index(gx, d.W, pad, align);
    const float iy = gs_source_index(gy, d.H, pad, align);
    Footprint2D<4> f;
    footprint_bilinear(ix, iy, d.H, d.W, f);
    if constexpr (kInt8) {
      int wq[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) wq[k] = q127_rne(f.w[k]);
      const float os = (1.f / 127.f) * s_in / s_out;  // gridSamplerKernel.cu:1137-1139
      for (int c = c0; c < c1; ++c, ip += HW, op += plane_o) {
        int t = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (f.off[k] >= 0) t += (int)ip[f.off[k]] * wq[k];
        *op = t2int8((float)t * os);
      }
    } else {
      for (int c = c0; c < c1; ++c, ip += HW, op += plane_o) {
        float o = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (f.off[k] >= 0) {
#pragma clang fp contract(off)
            o += ld<T>(ip + f.off[k]) * f.w[k];
          }
        st<T>(op, o, 1.f);
      }
    }
  } else if (interp == BEVOPS_NEAREST) {
    const float ix = gs_source_index(gx, d.W, pad, align);
    const float iy = gs_source_index(gy, d.H, pad, align);
    const int o = footprint_nearest(ix, iy, d.H, d.W);
    const float os = kInt8 ? s_in / s_out : 1.f;
    for (int c = c0; c < c1; ++c, ip += HW, op += plane_o) {
      if (o >= 0) {
        if constexpr (kInt8) st<T>(op, ld<T>(ip + o), os);
        else *op = ip[o];
      } else {
        st<T>(op, 0.f, 1.f);
      }
    }
  } else if constexpr (kInt8) {
    // int8 bicubic (gridSamplerKernel.cu:581-613,1205-1262): coefficients quantised by
    // truncation int8(c*127), int32 4-tap dot, temp/127 with C integer division, rows then
    // columns, final T2int8(v * s_in/s_out)
    const float ix = gs_unnormalize(gx, d.W, align), iy = gs_unnormalize(gy, d.H, align);
    const float ix_nw = floorf(ix), iy_nw = floorf(iy);
    float cxf[4], cyf[4];
    cubic_coeffs(cxf, ix - ix_nw);
    cubic_coeffs(cyf, iy - iy_nw);
    int cx[4], cy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      cx[k] = (int)(int8_t)(cxf[k] * 127.f);
      cy[k] = (int)(int8_t)(cyf[k] * 127.f);
    }
    const float os = s_in / s_out;
    for (int c = c0; c < c1; ++c, ip += HW, op += plane_o) {
      int col[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int t = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          t += (int)bounded2d(ip, ix_nw - 1 + k, iy_nw - 1 + i, d.W, d.H, pad, align) * cx[k];
        col[i] = (int)(int8_t)(t / 127);
      }
      const int t = col[0] * cy[0] + col[1] * cy[1] + col[2] * cy[2] + col[3] * cy[3];
      *op = t2int8((float)(int)(int8_t)(t / 127) * os);
    }
  } else {  // bicubic (fp)
    const float ix = gs_unnormalize(gx, d.W, align), iy = gs_unnormalize(gy, d.H, align);
    const float ix_nw = floorf(ix), iy_nw = floorf(iy);
    float cx[4], cy[4];
    cubic_coeffs(cx, ix - ix_nw);
    cubic_coeffs(cy, iy - iy_nw);
    for (int c = c0; c < c1; ++c, ip += HW, op += plane_o) {
      float col[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma clang fp contract(off)
        const float v0 = bounded2d(ip, ix_nw - 1, iy_nw - 1 + i, d.W, d.H, pad, align);
        const float v1 = bounded2d(ip, ix_nw + 0, iy_nw - 1 + i, d.W, d.H, pad, align);
        const float v2 = bounded2d(ip, ix_nw + 1, iy_nw - 1 + i, d.W, d.H, pad, align);
        const float v3 = bounded2d(ip, ix_nw + 2, iy_nw - 1 + i, d.W, d.H, pad, align);
        col[i] = v0 * cx[0] + v1 * cx[1] + v2 * cx[2] + v3 * cx[3];
      }
      float o;
      {
#pragma clang fp contract(off)
        o = col[0] * cy[0] + col[1] * cy[1] + col[2] * cy[2] + col[3] * cy[3];
      }
      st<T>(op, o, 1.f);
    }
  }
}

// ---- channels-last staging for the 2-D bilinear / nearest modes ---------------------------------
// With the NCHW input a thread's 8 channel planes cost 4 two-byte gathers each (32 VMEM instructions
// per output pixel and 8-channel chunk, + 8 stores): the op is bound by the number of memory
// instructions, not by bytes (666 us = 0.8 TB/s at the reference test shape).  When the caller
// lends a workspace, the (small) input is transposed once to [N, H*W, C]; a tap is then ONE
// 16-byte load of 8 fp16 channels (4 loads + 8 stores per pixel and chunk).  Same arithmetic in
// the same order: results are bit-identical to the planar kernel.
template <typename T>
__global__ __launch_bounds__(256) void gs_nchw_to_nhwc_kernel(const T *__restrict__ in, T *__restrict__ out,
                                                              int C, int HW) {
  __shared__ T tile[32][33];
  const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const T *src = in + (size_t)n * C * HW;
  T *dst = out + (size_t)n * C * HW;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, p = p0 + tx;
    if (c < C && p < HW) tile[ty + 8 * i][tx] = src[(size_t)c * HW + p];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = p0 + ty + 8 * i, c = c0 + tx;
    if (c < C && p < HW) dst[(size_t)p * C + c] = tile[tx][ty + 8 * i];
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void grid_sampler_2d_nhwc_kernel(
    const T *__restrict__ xt, const T *__restrict__ grid, T *__restrict__ out, Gs2dDims d, int interp,
    int pad, int align_i) {
  constexpr int V = 16 / sizeof(T);  // channels per 16-byte vector
  constexpr int NV = kCPT / V;       // vectors per 8-channel chunk
  const long pix = (long)blockIdx.x * kBlock + threadIdx.x;
  const long plane_o = (long)d.Ho * d.Wo;
  if (pix >= plane_o * d.N) return;
  const int n = (int)(pix / plane_o);
  const long s = pix - (long)n * plane_o;
  const bool align = align_i != 0;
  float gx, gy;
  {
#pragma clang fp contract(off)
    gx = ld<T>(grid + ((long)n * 2 + 0) * plane_o + s) / 10;  // grid_sampler.py:28-29
    gy = ld<T>(grid + ((long)n * 2 + 1) * plane_o + s) / 10;
  }
  const int c0 = blockIdx.y * kCPT;
  const T *ip = xt + (size_t)n * d.H * d.W * d.C + c0;
  T *op = out + ((size_t)n * d.C + c0) * plane_o + s;
  const float ix = gs_source_index(gx, d.W, pad, align);
  const float iy = gs_source_index(gy, d.H, pad, align);
  float o[kCPT];
#pragma unroll
  for (int j = 0; j < kCPT; ++j) o[j] = 0.f;
  if (interp == BEVOPS_BILINEAR) {
    Footprint2D<4> f;
    footprint_bilinear(ix, iy, d.H, d.W, f);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (f.off[k] >= 0) {
        const uint4 *row = reinterpret_cast<const uint4 *>(ip + (size_t)f.off[k] * d.C);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const uint4 raw = row[v];
          const T *e = reinterpret_cast<const T *>(&raw);
#pragma unroll
          for (int j = 0; j < V; ++j) {
#pragma clang fp contract(off)
            o[v * V + j] += ld<T>(e + j) * f.w[k];
          }
        }
      }
#pragma unroll
    for (int j = 0; j < kCPT; ++j) st<T>(op + (size_t)j * plane_o, o[j], 1.f);
  } else {  // nearest: pure copies
    const int src = footprint_nearest(ix, iy, d.H, d.W);
    if (src >= 0) {
      const uint4 *row = reinterpret_cast<const uint4 *>(ip + (size_t)src * d.C);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const uint4 raw = row[v];
        const T *e = reinterpret_cast<const T *>(&raw);
#pragma unroll
        for (int j = 0; j < V; ++j) op[(size_t)(v * V + j) * plane_o] = e[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < kCPT; ++j) st<T>(op + (size_t)j * plane_o, 0.f, 1.f);
    }
  }
}

struct Gs3dDims {
  int N, C, D, H, W, Do, Ho, Wo;
};

__device__ __forceinline__ bool in3d(int z, int y, int x, int D, int H, int W) {
  return z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void grid_sampler_3d_kernel(
    const T *__restrict__ input, const T *__restrict__ grid, T *__restrict__ out, Gs3dDims d,
    int interp, int pad, int align_i) {
  const long pix = (long)blockIdx.x * kBlock + threadIdx.x;
  const long plane_o = (long)d.Do * d.Ho * d.Wo;
  if (pix >= plane_o * d.N) return;
  const int n = (int)(pix / plane_o);
  const long s = pix - (long)n * plane_o;
  const bool align = align_i != 0;
  const float gx = ld<T>(grid + ((long)n * 3 + 0) * plane_o + s) / 10;
  const float gy = ld<T>(grid + ((long)n * 3 + 1) * plane_o + s) / 10;
  const float gz = ld<T>(grid + ((long)n * 3 + 2) * plane_o + s) / 10;
  const float ix = gs_source_index(gx, d.W, pad, align);
  const float iy = gs_source_index(gy, d.H, pad, align);
  const float iz = gs_source_index(gz, d.D, pad, align);
  const long vol = (long)d.D * d.H * d.W;
  const int c0 = blockIdx.y * kCPT, c1 = min(c0 + kCPT, d.C);
  const T *ip = input + ((size_t)n * d.C + c0) * vol;
  T *op = out + ((size_t)n * d.C + c0) * plane_o + s;
  if (interp == BEVOPS_BILINEAR) {
    int off[8];
    float w[8];
    {
#pragma clang fp contract(off)
      const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), z0 = (int)floorf(iz);
      const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
      const float fx1 = (float)x1 - ix, fx0 = ix - (float)x0;
      const float fy1 = (float)y1 - iy, fy0 = iy - (float)y0;
      const float fz1 = (float)z1 - iz, fz0 = iz - (float)z0;
      w[0] = fx1 * fy1 * fz1; w[1] = fx0 * fy1 * fz1; w[2] = fx1 * fy0 * fz1; w[3] = fx0 * fy0 * fz1;
      w[4] = fx1 * fy1 * fz0; w[5] = fx0 * fy1 * fz0; w[6] = fx1 * fy0 * fz0; w[7] = fx0 * fy0 * fz0;
      const int xs[2] = {x0, x1}, ys[2] = {y0, y1}, zs[2] = {z0, z1};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int x = xs[k & 1], y = ys[(k >> 1) & 1], z = zs[k >> 2];
        off[k] = in3d(z, y, x, d.D, d.H, d.W) ? (z * d.H + y) * d.W + x : -1;
      }
    }
    for (int c = c0; c < c1; ++c, ip += vol, op += plane_o) {
      float o = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (off[k] >= 0) {
#pragma clang fp contract(off)
          o += ld<T>(ip + off[k]) * w[k];
        }
      st<T>(op, o, 1.f);
    }
  } else {
    const int xn = (int)rintf(ix), yn = (int)rintf(iy), zn = (int)rintf(iz);
    const int o = in3d(zn, yn, xn, d.D, d.H, d.W) ? (zn * d.H + yn) * d.W + xn : -1;
    for (int c = c0; c < c1; ++c, ip += vol, op += plane_o) {
      if (o >= 0) *op = ip[o];
      else st<T>(op, 0.f, 1.f);
    }
  }
}

bool enum_ok(int interp, int pad) {
  return interp >= 0 && interp <= 2 && pad >= 0 && pad <= 2;
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" int bevops_grid_sampler_2d_forward(int dtype, const void *input, const void *grid,
                                              void *output, int N, int C, int H_in, int W_in,
                                              int H_out, int W_out, int interpolation,
                                              int padding, int align_corners, float scale_in,
                                              float scale_grid, float scale_out, void *stream) {
  if (!input || !grid || !output) return BEVOPS_BAD_PARAM;
  if (N <= 0 || C <= 0 || H_in <= 0 || W_in <= 0 || H_out <= 0 || W_out <= 0) return BEVOPS_BAD_PARAM;
  if (!enum_ok(interpolation, padding)) return BEVOPS_BAD_PARAM;
  const long pixels = (long)N * H_out * W_out;
  const long blocks = (pixels + kBlock - 1) / kBlock;
  if (blocks > 0x7FFFFFFFL || (long)H_in * W_in > 0x7FFFFFFFL || C > 65535 * kCPT)
    return BEVOPS_NOT_SUPPORTED;
  const Gs2dDims d{N, C, H_in, W_in, H_out, W_out};
  const dim3 g((unsigned)blocks, (unsigned)((C + kCPT - 1) / kCPT));
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case BEVOPS_F32:
      hipLaunchKernelGGL((grid_sampler_2d_kernel<float>), g, dim3(kBlock), 0, st,
                         (const float *)input, (const float *)grid, (float *)output, d,
                         interpolation, padding, align_corners, 1.f, 1.f, 1.f);
      return launch_status();
    case BEVOPS_F16:
      hipLaunchKernelGGL((grid_sampler_2d_kernel<__half>), g, dim3(kBlock), 0, st,
                         (const __half *)input, (const __half *)grid, (__half *)output, d,
                         interpolation, padding, align_corners, 1.f, 1.f, 1.f);
      return launch_status();
    case BEVOPS_I8:
      if (!(scale_in > 0.f) || !(scale_grid > 0.f) || !(scale_out > 0.f)) return BEVOPS_BAD_PARAM;
      hipLaunchKernelGGL((grid_sampler_2d_kernel<int8_t>), g, dim3(kBlock), 0, st,
                         (const int8_t *)input, (const int8_t *)grid, (int8_t *)output, d,
                         interpolation, padding, align_corners, scale_in, scale_grid, scale_out);
      return launch_status();
    default:
      return BEVOPS_NOT_SUPPORTED;
  }
}

extern "C" size_t bevops_grid_sampler_2d_workspace_size(int dtype, int N, int C, int H_in, int W_in) {
  if ((dtype != BEVOPS_F32 && dtype != BEVOPS_F16) || N <= 0 || C <= 0 || H_in <= 0 || W_in <= 0) return 0;
  if (C % kCPT != 0) return 0;  // the channels-last path moves whole 8-channel chunks
  const size_t n = (size_t)N * C * H_in * W_in * (dtype == BEVOPS_F32 ? 4 : 2);
  return (n + 255) & ~size_t(255);
}

extern "C" int bevops_grid_sampler_2d_forward_ws(int dtype, const void *input, const void *grid, void *output,
                                                 int N, int C, int H_in, int W_in, int H_out, int W_out,
                                                 int interpolation, int padding, int align_corners,
                                                 float scale_in, float scale_grid, float scale_out,
                                                 void *workspace, size_t workspace_bytes, void *stream) {
  const size_t need = bevops_grid_sampler_2d_workspace_size(dtype, N, C, H_in, W_in);
  // staging pays when the output is at least a few times the input (the transpose is one extra
  // read + write of the input) and only for the modes whose taps are plain loads
  const bool staged = workspace && need && workspace_bytes >= need && aligned16(workspace) &&
                      (interpolation == BEVOPS_BILINEAR || interpolation == BEVOPS_NEAREST) && H_out > 0 &&
                      W_out > 0 && (size_t)H_out * W_out >= 2 * (size_t)H_in * W_in;
  if (!staged)
    return bevops_grid_sampler_2d_forward(dtype, input, grid, output, N, C, H_in, W_in, H_out, W_out,
                                          interpolation, padding, align_corners, scale_in, scale_grid, scale_out,
                                          stream);
  if (!input || !grid || !output) return BEVOPS_BAD_PARAM;
  if (!enum_ok(interpolation, padding)) return BEVOPS_BAD_PARAM;
  const long pixels = (long)N * H_out * W_out;
  const long blocks = (pixels + kBlock - 1) / kBlock;
  const long HW = (long)H_in * W_in;
  if (blocks > 0x7FFFFFFFL || HW > 0x7FFFFFFFL / (C > 0 ? C : 1) || C > 65535 * kCPT || N > 65535)
    return BEVOPS_NOT_SUPPORTED;
  const Gs2dDims d{N, C, H_in, W_in, H_out, W_out};
  const dim3 gt((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N);
  const dim3 g((unsigned)blocks, (unsigned)(C / kCPT));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == BEVOPS_F32) {
    hipLaunchKernelGGL((gs_nchw_to_nhwc_kernel<float>), gt, dim3(256), 0, st, (const float *)input,
                       (float *)workspace, C, (int)HW);
    hipLaunchKernelGGL((grid_sampler_2d_nhwc_kernel<float>), g, dim3(kBlock), 0, st, (const float *)workspace,
                       (const float *)grid, (float *)output, d, interpolation, padding, align_corners);
  } else {
    hipLaunchKernelGGL((gs_nchw_to_nhwc_kernel<__half>), gt, dim3(256), 0, st, (const __half *)input,
                       (__half *)workspace, C, (int)HW);
    hipLaunchKernelGGL((grid_sampler_2d_nhwc_kernel<__half>), g, dim3(kBlock), 0, st, (const __half *)workspace,
                       (const __half *)grid, (__half *)output, d, interpolation, padding, align_corners);
  }
  return launch_status();
}

extern "C" int bevops_grid_sampler_3d_forward(int dtype, const void *input, const void *grid,
                                              void *output, int N, int C, int D_in, int H_in,
                                              int W_in, int D_out, int H_out, int W_out,
                                              int interpolation, int padding, int align_corners,
                                              void *stream) {
  if (!input || !grid || !output) return BEVOPS_BAD_PARAM;
  if (N <= 0 || C <= 0 || D_in <= 0 || H_in <= 0 || W_in <= 0 || D_out <= 0 || H_out <= 0 ||
      W_out <= 0)
    return BEVOPS_BAD_PARAM;
  if (!enum_ok(interpolation, padding)) return BEVOPS_BAD_PARAM;
  if (interpolation == BEVOPS_BICUBIC) return BEVOPS_NOT_SUPPORTED;  // 4-D only (grid_sampler.py:186)
  const long pixels = (long)N * D_out * H_out * W_out;
  const long blocks = (pixels + kBlock - 1) / kBlock;
  if (blocks > 0x7FFFFFFFL || (long)D_in * H_in * W_in > 0x7FFFFFFFL || C > 65535 * kCPT)
    return BEVOPS_NOT_SUPPORTED;
  const Gs3dDims d{N, C, D_in, H_in, W_in, D_out, H_out, W_out};
  const dim3 g((unsigned)blocks, (unsigned)((C + kCPT - 1) / kCPT));
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case BEVOPS_F32:
      hipLaunchKernelGGL((grid_sampler_3d_kernel<float>), g, dim3(kBlock), 0, st,
                         (const float *)input, (const float *)grid, (float *)output, d,
                         interpolation, padding, align_corners);
      return launch_status();
    case BEVOPS_F16:
      hipLaunchKernelGGL((grid_sampler_3d_kernel<__half>), g, dim3(kBlock), 0, st,
                         (const __half *)input, (const __half *)grid, (__half *)output, d,
                         interpolation, padding, align_corners);
      return launch_status();
    default:
      return BEVOPS_NOT_SUPPORTED;  // int8 is 2-D only (gridSamplerKernel.cu:2040)
  }
}
