// bev_pool_v2: BEVDet "pillar pooling".  For each BEV interval k,
//   out[ranks_bev[s_k], :] = sum_i depth[ranks_depth[s_k+i]] * feat[ranks_feat[s_k+i], :]
// Replaces BEVPoolPlugin::enqueue (TensorRT/plugin/bev_pool_v2/bevPoolPlugin.cpp:68-109)
// and bev_pool_v2 / _h2 / _int8 (bevPoolKernel.cu:19-190).
//
// MI355X mapping: lanes run along the channel axis in 16-byte vectors (8 fp16 /
// 4 fp32 / 16 int8 channels per lane), so each gathered feat row is one contiguous
// segment; fp32 accumulation for fp16 (the reference accumulates in half); the output
// is cleared with hipMemsetAsync ON THE CALLER'S STREAM (the reference's cudaMemset
// runs on the default stream, bevPoolKernel.cu:156 -- an ordering bug not reproduced).
#include "sampler.h"

namespace bevops {
namespace {

constexpr int kBlock = 256;


template <typename T, int V>
__global__ __launch_bounds__(kBlock) void bev_pool_kernel(
    const T *__restrict__ depth, const T *__restrict__ feat, const int *__restrict__ ranks_depth,
    const int *__restrict__ ranks_feat, const int *__restrict__ ranks_bev,
    const int *__restrict__ interval_starts, const int *__restrict__ interval_lengths,
    T *__restrict__ out, int c, int n_intervals, float scale_io) {
  const int vec_per_row = c / V;
  const long idx = (long)blockIdx.x * kBlock + threadIdx.x;
  const long k = idx / vec_per_row;
  const int cv = (int)(idx - k * vec_per_row) * V;
  if (k >= n_intervals) return;
  const int s = interval_starts[k], len = interval_lengths[k];
  T *o = out + (size_t)ranks_bev[s] * c + cv;
  if constexpr (sizeof(T) == 1) {
    int acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0;
    for (int i = 0; i < len; ++i) {
      const int d = (int)depth[ranks_depth[s + i]];
      const int8_t *f = (const int8_t *)feat + (size_t)ranks_feat[s + i] * c + cv;
      int8_t fv[V];
      if constexpr (V == 16) *reinterpret_cast<uint4 *>(fv) = *reinterpret_cast<const uint4 *>(f);
      else if constexpr (V == 4) *reinterpret_cast<unsigned *>(fv) = *reinterpret_cast<const unsigned *>(f);
      else fv[0] = f[0];
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] += (int)fv[j] * d;
    }
    int8_t r[V];
#pragma unroll
    for (int j = 0; j < V; ++j) r[j] = t2int8((float)acc[j] * scale_io);
    if constexpr (V == 16) *reinterpret_cast<uint4 *>(o) = *reinterpret_cast<const uint4 *>(r);
    else if constexpr (V == 4) *reinterpret_cast<unsigned *>(o) = *reinterpret_cast<const unsigned *>(r);
    else o[0] = r[0];
  } else {
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int i = 0; i < len; ++i) {
      const float d = ld<T>(depth + ranks_depth[s + i]);
      const T *f = feat + (size_t)ranks_feat[s + i] * c + cv;
      if constexpr (sizeof(T) == 2 && V == 8) {
        const uint4 r = *reinterpret_cast<const uint4 *>(f);
        acc[0] = fmaf(h2f_lo(r.x), d, acc[0]); acc[1] = fmaf(h2f_hi(r.x), d, acc[1]);
        acc[2] = fmaf(h2f_lo(r.y), d, acc[2]); acc[3] = fmaf(h2f_hi(r.y), d, acc[3]);
        acc[4] = fmaf(h2f_lo(r.z), d, acc[4]); acc[5] = fmaf(h2f_hi(r.z), d, acc[5]);
        acc[6] = fmaf(h2f_lo(r.w), d, acc[6]); acc[7] = fmaf(h2f_hi(r.w), d, acc[7]);
      } else if constexpr (sizeof(T) == 4 && V == 4) {
#pragma clang fp contract(off)
        const float4 r = *reinterpret_cast<const float4 *>(f);
        acc[0] += r.x * d; acc[1] += r.y * d; acc[2] += r.z * d; acc[3] += r.w * d;
      } else {
#pragma clang fp contract(off)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += ld<T>(f + j) * d;
      }
    }
    if constexpr (sizeof(T) == 2 && V == 8) {
      uint4 r;
      r.x = pack_h2(acc[0], acc[1]); r.y = pack_h2(acc[2], acc[3]);
      r.z = pack_h2(acc[4], acc[5]); r.w = pack_h2(acc[6], acc[7]);
      *reinterpret_cast<uint4 *>(o) = r;
    } else if constexpr (sizeof(T) == 4 && V == 4) {
      *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) st<T>(o + j, acc[j], 1.f);
    }
  }
}

// Split-interval variant (fp16 V=8, int8 V=16; channels/V a power of two <= 32): a wave's 64 lanes =
// S point-slices x (channels/V) channel vectors of ONE or more intervals; slice s takes points
// s, s+S, ... of the interval (each step of a wave reads S whole feature rows), the partial sums are
// combined with wave shuffles.  The one-thread-per-(interval, vector) kernel above serialises the
// 100+ point intervals near the ego vehicle (BEVDet-R50: 82 us for 5 MB of traffic).
template <typename T, int V>
__global__ __launch_bounds__(kBlock) void bev_pool_split_kernel(
    const T *__restrict__ depth, const T *__restrict__ feat, const int *__restrict__ ranks_depth,
    const int *__restrict__ ranks_feat, const int *__restrict__ ranks_bev,
    const int *__restrict__ interval_starts, const int *__restrict__ interval_lengths,
    T *__restrict__ out, int c, int n_intervals, float scale_io, int S) {
  const int vpr = c / V;                 // lanes per point slice
  const int lanes_per_k = vpr * S;       // power of two, <= 64
  const long idx = (long)blockIdx.x * kBlock + threadIdx.x;
  const long k = idx / lanes_per_k;
  const int within = (int)(idx - k * lanes_per_k);
  const int sl = within / vpr, cv = (within - sl * vpr) * V;
  const bool live = k < n_intervals;
  const int s = live ? interval_starts[k] : 0, len = live ? interval_lengths[k] : 0;
  if constexpr (sizeof(T) == 1) {
    int acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0;
    for (int i = sl; i < len; i += S) {
      const int d = (int)depth[ranks_depth[s + i]];
      int8_t fv[V];
      *reinterpret_cast<uint4 *>(fv) =
          *reinterpret_cast<const uint4 *>((const int8_t *)feat + (size_t)ranks_feat[s + i] * c + cv);
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] += (int)fv[j] * d;
    }
    for (int m = vpr; m < lanes_per_k; m <<= 1)
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] += __shfl_xor(acc[j], m);
    if (live && sl == 0) {
      int8_t r[V];
#pragma unroll
      for (int j = 0; j < V; ++j) r[j] = t2int8((float)acc[j] * scale_io);
      *reinterpret_cast<uint4 *>(out + (size_t)ranks_bev[s] * c + cv) = *reinterpret_cast<const uint4 *>(r);
    }
  } else {
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int i = sl; i < len; i += S) {
      const float d = ld<T>(depth + ranks_depth[s + i]);
      const uint4 r = *reinterpret_cast<const uint4 *>(feat + (size_t)ranks_feat[s + i] * c + cv);
      acc[0] = fmaf(h2f_lo(r.x), d, acc[0]); acc[1] = fmaf(h2f_hi(r.x), d, acc[1]);
      acc[2] = fmaf(h2f_lo(r.y), d, acc[2]); acc[3] = fmaf(h2f_hi(r.y), d, acc[3]);
      acc[4] = fmaf(h2f_lo(r.z), d, acc[4]); acc[5] = fmaf(h2f_hi(r.z), d, acc[5]);
      acc[6] = fmaf(h2f_lo(r.w), d, acc[6]); acc[7] = fmaf(h2f_hi(r.w), d, acc[7]);
    }
    for (int m = vpr; m < lanes_per_k; m <<= 1)
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] += __shfl_xor(acc[j], m);
    if (live && sl == 0) {
      uint4 r;
      r.x = pack_h2(acc[0], acc[1]); r.y = pack_h2(acc[2], acc[3]);
      r.z = pack_h2(acc[4], acc[5]); r.w = pack_h2(acc[6], acc[7]);
      *reinterpret_cast<uint4 *>(out + (size_t)ranks_bev[s] * c + cv) = r;
    }
  }
}

template <typename T, int V>
int launch_split(const void *depth, const void *feat, const int *rd, const int *rf, const int *rb,
                 const int *is, const int *il, void *out, int c, int n_intervals, float scale_io,
                 hipStream_t st) {
  const int vpr = c / V;
  const int S = 64 / vpr < 8 ? 64 / vpr : 8;
  const long threads = (long)n_intervals * vpr * S;
  const long blocks = (threads + kBlock - 1) / kBlock;
  if (blocks > 0x7FFFFFFFL) return BEVOPS_NOT_SUPPORTED;
  hipLaunchKernelGGL((bev_pool_split_kernel<T, V>), dim3((unsigned)blocks), dim3(kBlock), 0, st,
                     (const T *)depth, (const T *)feat, rd, rf, rb, is, il, (T *)out, c, n_intervals,
                     scale_io, S);
  return launch_status();
}

inline bool split_ok(int channels, int V) {
  if (channels % V) return false;
  const int vpr = channels / V;
  return vpr <= 32 && (vpr & (vpr - 1)) == 0;
}

template <typename T, int V>
int launch(const void *depth, const void *feat, const int *rd, const int *rf, const int *rb,
           const int *is, const int *il, void *out, int c, int n_intervals, float scale_io,
           hipStream_t st) {
  const long threads = (long)n_intervals * (c / V);
  const long blocks = (threads + kBlock - 1) / kBlock;
  if (blocks > 0x7FFFFFFFL) return BEVOPS_NOT_SUPPORTED;
  hipLaunchKernelGGL((bev_pool_kernel<T, V>), dim3((unsigned)blocks), dim3(kBlock), 0, st,
                     (const T *)depth, (const T *)feat, rd, rf, rb, is, il, (T *)out, c,
                     n_intervals, scale_io);
  return launch_status();
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" int bevops_bev_pool_v2_forward(int dtype, const void *depth, const void *feat,
                                          const int32_t *ranks_depth, const int32_t *ranks_feat,
                                          const int32_t *ranks_bev, const int32_t *interval_starts,
                                          const int32_t *interval_lengths, void *output,
                                          int channels, int n_intervals, int out_height,
                                          int out_width, float scale_depth, float scale_feat,
                                          float scale_out, void *stream) {
  if (!output || channels <= 0 || n_intervals < 0 || out_height <= 0 || out_width <= 0)
    return BEVOPS_BAD_PARAM;
  if (n_intervals > 0 && (!depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev ||
                          !interval_starts || !interval_lengths))
    return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F32 && dtype != BEVOPS_F16 && dtype != BEVOPS_I8) return BEVOPS_NOT_SUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t esize = dtype == BEVOPS_F32 ? 4 : dtype == BEVOPS_F16 ? 2 : 1;
  const size_t n_out = (size_t)out_height * out_width * channels;
  if (hipMemsetAsync(output, 0, n_out * esize, st) != hipSuccess) return BEVOPS_FAILURE;
  if (n_intervals == 0) return BEVOPS_SUCCESS;
  const bool al = aligned16(feat) && aligned16(output);
  switch (dtype) {
    case BEVOPS_F32:
      if (channels % 4 == 0 && al)
        return launch<float, 4>(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                                interval_lengths, output, channels, n_intervals, 1.f, st);
      return launch<float, 1>(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                              interval_lengths, output, channels, n_intervals, 1.f, st);
    case BEVOPS_F16:
      if (al && split_ok(channels, 8))
        return launch_split<__half, 8>(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                                       interval_lengths, output, channels, n_intervals, 1.f, st);
      if (channels % 8 == 0 && al)
        return launch<__half, 8>(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                                 interval_lengths, output, channels, n_intervals, 1.f, st);
      return launch<__half, 1>(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                               interval_lengths, output, channels, n_intervals, 1.f, st);
    case BEVOPS_I8: {
      if (!(scale_depth > 0.f) || !(scale_feat > 0.f) || !(scale_out > 0.f)) return BEVOPS_BAD_PARAM;
      const float sio = scale_depth * scale_feat / scale_out;  // bevPoolKernel.cu:188
      if (al && split_ok(channels, 16))
        return launch_split<int8_t, 16>(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                                        interval_lengths, output, channels, n_intervals, sio, st);
      if (channels % 16 == 0 && al)
        return launch<int8_t, 16>(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                                  interval_lengths, output, channels, n_intervals, sio, st);
      if (channels % 4 == 0 && al)
        return launch<int8_t, 4>(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                                 interval_lengths, output, channels, n_intervals, sio, st);
      return launch<int8_t, 1>(depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                               interval_lengths, output, channels, n_intervals, sio, st);
    }
    default:
      return BEVOPS_NOT_SUPPORTED;
  }
}
