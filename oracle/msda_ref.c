/*
 * oracle/msda_ref.c -- TEST INFRASTRUCTURE ONLY (never linked into the product path).
 *
 * CPU restatement of the reference's multi-scale deformable attention (MSDA)
 * plugin arithmetic.  Each function names the reference lines it follows.
 * All paths are relative to the reference tree (DerryHub/BEVFormer_tensorrt).
 *
 *   fp32 :  TensorRT/plugin/multi_scale_deformable_attn/
 *             multiScaleDeformableAttnKernel.cu:611-688  (fused softmax + sampling)
 *             multiScaleDeformableAttnKernel.cu:133-178  (4-tap bilinear, per-corner bounds)
 *           which is arithmetically the eager path
 *             det2trt/models/functions/multi_scale_deformable_attn.py:58-115
 *             det2trt/models/utils/trt_ops.py:44-85
 *   int8 :  multiScaleDeformableAttnKernel.cu:848-955   (<float> flavour, s8 weights x127)
 *           multiScaleDeformableAttnKernel.cu:957-1104  (<__half2> flavour, u8 weights x255;
 *           restated with fp32 intermediate math instead of fp16 -- strictly more
 *           accurate; see DESIGN.md "int8 numerics")
 *
 * Pinning: tests/test_oracle_golden.py checks oracle_msda_f32 against the golden
 * vectors produced by the reference's own Python code (tests/golden/make_golden.py);
 * tests/test_ref_kernels_cpu.py checks every function here against the outputs of the
 * reference's own kernels run on the host (oracle/_ref, tests/golden/refk_msda_*.npz):
 * oracle_msda_f32 and oracle_msda_s8 are BIT-EXACT against them; oracle_msda_s8_u8w
 * shares the integer pipeline of the <__half2> kernel and differs where that kernel
 * sums / requantises in binary16 (>= 80 % identical, >= 99.9 % within 3 LSB).
 *
 * Layouts (all contiguous, row-major):
 *   value  [bs, nk, heads, C]           nk = sum_l H_l*W_l, levels concatenated in order
 *   shapes [L, 2] int32 (h, w)
 *   ref    [bs, nq, 1, 2*ppg]   (x, y) pairs, normalised to [0,1]
 *   off    [bs, nq, heads, L*P*2]  (x, y) in pixels of each level
 *   logit  [bs, nq, heads, L*P]   pre-softmax
 *   out    [bs, nq, heads, C]
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* kernel.cu:133-178 */
static inline float bilinear_f32(const float *base, int H, int W, int step,
                                 float h, float w) {
  const int h_low = (int)floorf(h), w_low = (int)floorf(w);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h - h_low, lw = w - w_low;
  const float hh = 1 - lh, hw = 1 - lw;
  float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = base[(h_low * W + w_low) * step];
  if (h_low >= 0 && w_high <= W - 1) v2 = base[(h_low * W + w_high) * step];
  if (h_high <= H - 1 && w_low >= 0) v3 = base[(h_high * W + w_low) * step];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = base[(h_high * W + w_high) * step];
  const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

/* kernel.cu:611-688.  The reference evaluates exp()/locations once per channel
 * thread; here they are hoisted per (b,q,h) item -- same values, same order of
 * accumulation per output element. */
void oracle_msda_f32(const float *value, const int32_t *shapes, const float *ref,
                     const float *off, const float *logit, float *out, int bs,
                     int nk, int heads, int C, int L, int nq, int P, int ppg) {
  const long n_item = (long)bs * nq * heads;
  const int LP = L * P;
#pragma omp parallel for schedule(static)
  for (long item = 0; item < n_item; ++item) {
    const int h = (int)(item % heads);
    const long bq = item / heads;
    const int b = (int)(bq / nq);
    const float *lg = logit + item * LP;
    const float *of = off + item * LP * 2;
    const float *rp = ref + bq * ppg * 2;
    float mx = -INFINITY;
    for (int j = 0; j < LP; ++j) mx = fmaxf(mx, lg[j]);
    float *wgt = (float *)malloc(sizeof(float) * LP * 3);
    float *him = wgt + LP, *wim = wgt + 2 * LP;
    float sum = 0.f;
    for (int l = 0, j = 0; l < L; ++l) {
      const int H = shapes[2 * l], W = shapes[2 * l + 1];
      for (int p = 0; p < P; ++p, ++j) {
        const int g = p % ppg;
        const float loc_w = rp[2 * g] * W + of[2 * j];
        const float loc_h = rp[2 * g + 1] * H + of[2 * j + 1];
        wgt[j] = expf(lg[j] - mx);
        sum += wgt[j];
        him[j] = loc_h - 0.5f;
        wim[j] = loc_w - 0.5f;
      }
    }
    for (int c = 0; c < C; ++c) {
      const float *vp = value + ((long)b * nk * heads + h) * C + c;
      float acc = 0.f;
      for (int l = 0, j = 0; l < L; ++l) {
        const int H = shapes[2 * l], W = shapes[2 * l + 1];
        for (int p = 0; p < P; ++p, ++j) {
          const float h_im = him[j], w_im = wim[j];
          if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
            acc += bilinear_f32(vp, H, W, heads * C, h_im, w_im) * wgt[j];
        }
        vp += (long)H * W * heads * C;
      }
      out[item * C + c] = acc / sum;
    }
    free(wgt);
  }
}

/* kernel.cu:44-55 : clamp then round half away from zero */
static inline int8_t t2int8_f(float a) {
  a = a > 127 ? 127 : a;
  a = a < -128 ? -128 : a;
  return (int8_t)(a + (a > 0 ? 0.5f : -0.5f));
}
/* kernel.cu:57-62 : RNE then clamp (half flavour; restated on fp32 values) */
static inline int8_t t2int8_rne(float a) {
  float r = nearbyintf(a);
  r = r > 127 ? 127 : r;
  r = r < -128 ? -128 : r;
  return (int8_t)r;
}
static inline int u8_rne(float a) { /* __half2ushort_rn, saturating */
  float r = nearbyintf(a);
  r = r < 0 ? 0 : r;
  r = r > 65535 ? 65535 : r;
  return (int)r;
}

/* kernel.cu:290-358 : int8 bilinear, signed area weights x127, one channel */
static inline int8_t bilinear_s8(const int8_t *base, int H, int W, int step,
                                 float h, float w) {
  const int h_low = (int)floorf(h), w_low = (int)floorf(w);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h - h_low, lw = w - w_low;
  const float hh = 1 - lh, hw = 1 - lw;
  const float scale_area = 1 / 127.f;
  const int a1 = t2int8_f((hh * hw) / scale_area), a2 = t2int8_f((hh * lw) / scale_area),
            a3 = t2int8_f((lh * hw) / scale_area), a4 = t2int8_f((lh * lw) / scale_area);
  int v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = base[(h_low * W + w_low) * step];
  if (h_low >= 0 && w_high <= W - 1) v2 = base[(h_low * W + w_high) * step];
  if (h_high <= H - 1 && w_low >= 0) v3 = base[(h_high * W + w_low) * step];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = base[(h_high * W + w_high) * step];
  const int t = v1 * a1 + v2 * a2 + v3 * a3 + v4 * a4;
  return t2int8_f(t * scale_area);
}

/* kernel.cu:848-955 ; ref points fp32 */
void oracle_msda_s8(const int8_t *value, float s_v, const int32_t *shapes,
                    const float *ref, const int8_t *off, float s_o,
                    const int8_t *logit, float s_w, int8_t *out, float s_out,
                    int bs, int nk, int heads, int C, int L, int nq, int P,
                    int ppg) {
  const long n_item = (long)bs * nq * heads;
  const int LP = L * P;
  const float scale_o = s_v * (1.0f / s_out);
#pragma omp parallel for schedule(static)
  for (long item = 0; item < n_item; ++item) {
    const int h = (int)(item % heads);
    const long bq = item / heads;
    const int b = (int)(bq / nq);
    const int8_t *lg = logit + item * LP;
    const int8_t *of = off + item * LP * 2;
    const float *rp = ref + bq * ppg * 2;
    float mx = -INFINITY;
    for (int j = 0; j < LP; ++j) mx = fmaxf(mx, (float)lg[j] * s_w);
    for (int c = 0; c < C; ++c) {
      const int8_t *vp = value + ((long)b * nk * heads + h) * C + c;
      int32_t acc = 0;
      float sum = 0.f;
      int j = 0;
      for (int l = 0; l < L; ++l) {
        const int H = shapes[2 * l], W = shapes[2 * l + 1];
        for (int p = 0; p < P; ++p, ++j) {
          const int g = p % ppg;
          const float loc_w = rp[2 * g] * W + of[2 * j] * s_o;
          const float loc_h = rp[2 * g + 1] * H + of[2 * j + 1] * s_o;
          const float h_im = loc_h - 0.5f, w_im = loc_w - 0.5f;
          const int8_t wq = t2int8_f(expf(lg[j] * s_w - mx) * 127.f);
          sum += wq;
          if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;
          acc += (int)bilinear_s8(vp, H, W, heads * C, h_im, w_im) * wq;
        }
        vp += (long)H * W * heads * C;
      }
      out[item * C + c] = t2int8_f(acc * (scale_o * (1.0f / sum)));
    }
  }
}

/* kernel.cu:360-460 (bilinear_int8_h2: unsigned area weights x255, RNE) */
static inline int8_t bilinear_u8w(const int8_t *base, int H, int W, int step,
                                  float h, float w) {
  const float h_lowf = floorf(h), w_lowf = floorf(w);
  const int h_low = (int)h_lowf, w_low = (int)w_lowf;
  const float lh = h - h_lowf, lw = w - w_lowf;
  const float hh = 1 - lh, hw = 1 - lw;
  const int a1 = u8_rne(hh * hw * 255.f), a2 = u8_rne(hh * lw * 255.f),
            a3 = u8_rne(lh * hw * 255.f), a4 = u8_rne(lh * lw * 255.f);
  int v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = base[(h_low * W + w_low) * step];
  if (h_low >= 0 && w_low + 1 <= W - 1) v2 = base[(h_low * W + w_low + 1) * step];
  if (h_low + 1 <= H - 1 && w_low >= 0) v3 = base[((h_low + 1) * W + w_low) * step];
  if (h_low + 1 <= H - 1 && w_low + 1 <= W - 1)
    v4 = base[((h_low + 1) * W + w_low + 1) * step];
  const int t = v1 * a1 + v2 * a2 + v3 * a3 + v4 * a4;
  return t2int8_rne((float)t * (1.0f / 255.f));
}

/* kernel.cu:957-1104 ; ref points arrive as fp16 (caller upcasts to fp32 exactly) */
void oracle_msda_s8_u8w(const int8_t *value, float s_v, const int32_t *shapes,
                        const float *ref, const int8_t *off, float s_o,
                        const int8_t *logit, float s_w, int8_t *out,
                        float s_out, int bs, int nk, int heads, int C, int L,
                        int nq, int P, int ppg) {
  const long n_item = (long)bs * nq * heads;
  const int LP = L * P;
  const float scale_o = s_v * (1.0f / s_out);
#pragma omp parallel for schedule(static)
  for (long item = 0; item < n_item; ++item) {
    const int h = (int)(item % heads);
    const long bq = item / heads;
    const int b = (int)(bq / nq);
    const int8_t *lg = logit + item * LP;
    const int8_t *of = off + item * LP * 2;
    const float *rp = ref + bq * ppg * 2;
    float mx = -INFINITY;
    for (int j = 0; j < LP; ++j) mx = fmaxf(mx, (float)lg[j] * s_w);
    for (int c = 0; c < C; ++c) {
      const int8_t *vp = value + ((long)b * nk * heads + h) * C + c;
      int32_t acc = 0;
      float sum = 0.f;
      int j = 0;
      for (int l = 0; l < L; ++l) {
        const int H = shapes[2 * l], W = shapes[2 * l + 1];
        for (int p = 0; p < P; ++p, ++j) {
          const int g = p % ppg;
          const float w_im = rp[2 * g] * W + (of[2 * j] * s_o - 0.5f);
          const float h_im = rp[2 * g + 1] * H + (of[2 * j + 1] * s_o - 0.5f);
          const float w255 = expf(lg[j] * s_w - mx) * 255.f;
          sum += w255; /* un-quantised weight enters the normaliser (:1086) */
          const int cond = (h_im > -1 && w_im > -1 && h_im < H && w_im < W);
          if (!cond) continue;
          const int wq = u8_rne(w255);
          acc += (int)bilinear_u8w(vp, H, W, heads * C, h_im, w_im) * wq;
        }
        vp += (long)H * W * heads * C;
      }
      out[item * C + c] = t2int8_rne((float)acc * (scale_o * (1.0f / sum)));
    }
  }
}
