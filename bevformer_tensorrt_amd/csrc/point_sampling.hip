// point_sampling_trt (det2trt/models/modules/encoder.py:197-259) as ONE kernel: the BEV pillar anchors of a frame
// projected into every camera, normalised by the image size, with the visibility weights bev_mask.  The reference
// evaluates this inside the engine on every frame (lidar2img is an engine input, tools/bevformer/evaluate_trt.py:
// 131-132; modules/encoder.py:293); here it is the first launch of the frame's HIP graph (round 6: until then the frame
// loop evaluated the ~30 torch passes of the same arithmetic eagerly, once per calibration, outside the graph).
//
// Index / grid generation must be BIT-EXACT, so the arithmetic is the reference's op sequence in
// fp32 with every rounding it makes and no other:
//   cam_i   = ((l2i[i][0] p.x + l2i[i][1] p.y) + l2i[i][2] p.z) + l2i[i][3] p.w     separately rounded products and sums
//                                                                    (encoder.py:223: matmul of a 4x4 with a 4x1; the
//                                                                    host BLAS adds the four products in ascending k)
//   valid   = cam_z > 1e-5                                            (:227)
//   (x, y)  = (cam_x, cam_y) / max(cam_z, 1e-5), then / image (w, h)  (:228-236; IEEE divisions, two of them)
//   valid  &= 0 < y < 1 and 0 < x < 1                                 (:238-243)
//   bev_mask[cam, q] = any_d valid / clamp(sum over cameras of any_d valid, 1e-4)       (:255-258)
// tests/test_geometry_gpu.py holds the results equal to geometry.project_points(projection="fma") on the device and to
// the SHA-256 digests of the reference's own CPU arrays at the base size.
//
// One lane per query: the four anchors of its pillar (4 x 16 bytes, coalesced), a loop over the cameras with the
// matrices read through scalar loads, one 16-byte store of the camera's 4 x (x, y) in binary16 per lane.
#include "common.h"

namespace bevops {
namespace {

constexpr int kPsD = 4;   // anchors per pillar (num_points_in_pillar, configs/bevformer/bevformer_base.py:86)

__device__ __forceinline__ float ps_div(float a, float b) {   // IEEE division, never a reciprocal multiply
#pragma clang fp reciprocal(off) contract(off)
  return __fdiv_rn(a, b);
}

template <typename OutT>
__device__ __forceinline__ void ps_store_ref(OutT *dst, const float (&xy)[2 * kPsD]);
template <>
__device__ __forceinline__ void ps_store_ref<__half>(__half *dst, const float (&xy)[2 * kPsD]) {
  u32x4 v;
  v.x = pack_h2(xy[0], xy[1]);
  v.y = pack_h2(xy[2], xy[3]);
  v.z = pack_h2(xy[4], xy[5]);
  v.w = pack_h2(xy[6], xy[7]);
  *reinterpret_cast<u32x4 *>(dst) = v;
}
template <>
__device__ __forceinline__ void ps_store_ref<float>(float *dst, const float (&xy)[2 * kPsD]) {
  reinterpret_cast<float4 *>(dst)[0] = make_float4(xy[0], xy[1], xy[2], xy[3]);
  reinterpret_cast<float4 *>(dst)[1] = make_float4(xy[4], xy[5], xy[6], xy[7]);
}
__device__ __forceinline__ void ps_store_mask(__half *dst, float v) { *dst = __float2half_rn(v); }
__device__ __forceinline__ void ps_store_mask(float *dst, float v) { *dst = v; }

template <typename OutT>
__global__ __launch_bounds__(256) void point_sampling_kernel(const float4 *__restrict__ pillars,
                                                             const float *__restrict__ l2i, OutT *__restrict__ ref_cam,
                                                             OutT *__restrict__ bev_mask, int ncam, int nq, float img_h,
                                                             float img_w) {
  const unsigned q = blockIdx.x * 256u + threadIdx.x;
  if (q >= (unsigned)nq) return;
  float4 p[kPsD];
#pragma unroll
  for (int d = 0; d < kPsD; ++d) p[d] = pillars[(size_t)d * nq + q];
  constexpr float eps = 1e-5f;
  unsigned seen = 0, count = 0;
  for (int cam = 0; cam < ncam; ++cam) {
    const float *m = l2i + cam * 16;   // wave-uniform: scalar loads
    float xy[2 * kPsD];
    bool any = false;
#pragma unroll
    for (int d = 0; d < kPsD; ++d) {
      float c[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float s = mul_rn(m[4 * i + 0], p[d].x);
        s = add_rn(s, mul_rn(m[4 * i + 1], p[d].y));
        s = add_rn(s, mul_rn(m[4 * i + 2], p[d].z));
        s = add_rn(s, mul_rn(m[4 * i + 3], p[d].w));
        c[i] = s;
      }
      bool valid = c[2] > eps;
      const float den = (c[2] != c[2]) ? c[2] : fmaxf(c[2], eps);   // torch.max hands a NaN on
      const float x = ps_div(ps_div(c[0], den), img_w);
      const float y = ps_div(ps_div(c[1], den), img_h);
      valid = valid && (y > 0.f) && (y < 1.f) && (x < 1.f) && (x > 0.f);
      xy[2 * d] = x;
      xy[2 * d + 1] = y;
      any = any || valid;
    }
    ps_store_ref<OutT>(ref_cam + ((size_t)cam * nq + q) * (2 * kPsD), xy);
    if (any) {
      seen |= 1u << cam;
      ++count;
    }
  }
  // visible / clamp(number of cameras that see the pillar, 1e-4): 1 / count, or 0 / 1e-4 = 0
  const float w = ps_div(1.f, (float)count);
  for (int cam = 0; cam < ncam; ++cam) ps_store_mask(bev_mask + (size_t)cam * nq + q, (seen >> cam) & 1u ? w : 0.f);
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" int bevops_point_sampling(int out_dtype, const float *pillars, const float *lidar2img, void *reference_points_cam,
                                     void *bev_mask, int num_cams, int num_query, int num_points_in_pillar,
                                     float image_h, float image_w, void *stream) {
  if (!pillars || !lidar2img || !reference_points_cam || !bev_mask) return BEVOPS_BAD_PARAM;
  if (num_cams <= 0 || num_query <= 0 || num_points_in_pillar <= 0 || !(image_h > 0.f) || !(image_w > 0.f))
    return BEVOPS_BAD_PARAM;
  if (num_points_in_pillar != kPsD || num_cams > 32) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(pillars) || !aligned16(reference_points_cam)) return BEVOPS_BAD_PARAM;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(((unsigned)num_query + 255u) / 256u);
  if (out_dtype == BEVOPS_F16) {
    hipLaunchKernelGGL(point_sampling_kernel<__half>, grid, dim3(256), 0, st, reinterpret_cast<const float4 *>(pillars),
                       lidar2img, static_cast<__half *>(reference_points_cam), static_cast<__half *>(bev_mask), num_cams,
                       num_query, image_h, image_w);
  } else if (out_dtype == BEVOPS_F32) {
    hipLaunchKernelGGL(point_sampling_kernel<float>, grid, dim3(256), 0, st, reinterpret_cast<const float4 *>(pillars),
                       lidar2img, static_cast<float *>(reference_points_cam), static_cast<float *>(bev_mask), num_cams,
                       num_query, image_h, image_w);
  } else {
    return BEVOPS_NOT_SUPPORTED;
  }
  return launch_status();
}
