/*
 * oracle/bev_pool_ref.c -- TEST INFRASTRUCTURE ONLY.
 * bev_pool_v2 forward ("pillar pooling"), restating
 *   third_party/bev_mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:22-46  (fp32 op the
 *   reference's PyTorch path calls, det2trt/models/functions/bev_pool_v2.py:33-67)
 *   TensorRT/plugin/bev_pool_v2/bevPoolKernel.cu:115-149,188            (int8 flavour)
 * out[ranks_bev[s_k], c] = sum_{i<len_k} depth[ranks_depth[s_k+i]] * feat[ranks_feat[s_k+i], c];
 * every other output cell is 0 (the reference memsets the output, bevPoolKernel.cu:156).
 * Pinned BIT-EXACT (fp32 and int8) against the reference's plugin kernels run on the host
 * (oracle/_ref, tests/test_ref_kernels_cpu.py) and against an independent
 * torch.index_add_ statement of the same sum on the reference test's own index tensors
 * (tests/golden/bev_pool_ref_ranks.npz).
 */
#include <stdint.h>
#include <string.h>

void oracle_bev_pool_v2_f32(const float *depth, const float *feat, const int32_t *ranks_depth,
                            const int32_t *ranks_feat, const int32_t *ranks_bev,
                            const int32_t *interval_starts, const int32_t *interval_lengths,
                            float *out, int c, int n_intervals, long n_out) {
  memset(out, 0, sizeof(float) * n_out);
#pragma omp parallel for schedule(dynamic, 64)
  for (int k = 0; k < n_intervals; ++k) {
    const int s = interval_starts[k], len = interval_lengths[k];
    float *o = out + (long)ranks_bev[s] * c;
    for (int cc = 0; cc < c; ++cc) {
      float psum = 0;
      for (int i = 0; i < len; ++i)
        psum += feat[(long)ranks_feat[s + i] * c + cc] * depth[ranks_depth[s + i]];
      o[cc] = psum;
    }
  }
}

static inline int8_t t2int8_f(float a) {
  a = a > 127 ? 127 : a;
  a = a < -128 ? -128 : a;
  return (int8_t)(a + (a > 0 ? 0.5f : -0.5f));
}

void oracle_bev_pool_v2_s8(const int8_t *depth, const int8_t *feat, const int32_t *ranks_depth,
                           const int32_t *ranks_feat, const int32_t *ranks_bev,
                           const int32_t *interval_starts, const int32_t *interval_lengths,
                           int8_t *out, int c, int n_intervals, long n_out, float scale_io) {
  memset(out, 0, n_out);
#pragma omp parallel for schedule(dynamic, 64)
  for (int k = 0; k < n_intervals; ++k) {
    const int s = interval_starts[k], len = interval_lengths[k];
    int8_t *o = out + (long)ranks_bev[s] * c;
    for (int cc = 0; cc < c; ++cc) {
      int32_t psum = 0;
      for (int i = 0; i < len; ++i)
        psum += (int)feat[(long)ranks_feat[s + i] * c + cc] * (int)depth[ranks_depth[s + i]];
      o[cc] = t2int8_f(psum * scale_io);
    }
  }
}
