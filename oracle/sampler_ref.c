/*
 * oracle/sampler_ref.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's grid_sampler and rotate operators as their
 * PyTorch path computes them (BASELINE north_star: "outputs must match the
 * reference PyTorch path"):
 *
 *   grid_sampler : det2trt/models/functions/grid_sampler.py:19-37 (2-D), :70-88 (3-D)
 *                  = aten.grid_sampler(input, grid.permute(..)/10, mode, pad, align)
 *                  whose arithmetic the plugin kernels restate in
 *                  TensorRT/plugin/grid_sampler/gridSamplerKernel.cu:82-155 (unnormalise),
 *                  :157-260 (clip / reflect), :373-437 (source index), :457-559 (bicubic,
 *                  A = -0.75, get_value_bounded), :666-795 (2-D kernel), :1270-1442 (3-D).
 *   rotate       : det2trt/models/functions/rotate.py:12-80 (affine grid built from
 *                  linspace + 3x2 matmul, then torch.grid_sampler(zeros, align=False));
 *                  plugin kernel TensorRT/plugin/rotate/rotateKernel.cu:128-210.
 *
 * Pinned by tests/test_oracle_golden.py against golden vectors produced by the
 * reference's own Python functions (tests/golden/make_golden.py) and by
 * tests/test_ref_kernels_cpu.py against the reference's own kernels run on the host
 * (oracle/_ref, tests/golden/refk_{grid_sampler,rotate}_*.npz): nearest bit-exact,
 * bilinear / bicubic within 2e-5; int8 within 1-4 LSB of kernels that evaluate their
 * coordinates in binary16.
 *
 * Layout: input [N,C,(D,)H,W], grid channel-first [N,2,Ho,Wo] / [N,3,Do,Ho,Wo] in
 * [-10,10] units (x, y[, z]), output [N,C,(Do,)Ho,Wo].
 */
#include <math.h>
#include <stdint.h>

enum { BILINEAR = 0, NEAREST = 1, BICUBIC = 2 };
enum { ZEROS = 0, BORDER = 1, REFLECTION = 2 };

static inline float unnormalize(float c, int size, int align) {
  if (align) return ((c + 1.f) / 2) * (size - 1);
  return ((c + 1.f) * size - 1) / 2;
}
static inline float clip_coord(float in, int limit) {
  return fminf((float)(limit - 1), fmaxf(in, 0.f));
}
static inline float reflect_coord(float in, int twice_low, int twice_high) {
  if (twice_low == twice_high) return 0.f;
  const float mn = (float)twice_low / 2;
  const float span = (float)(twice_high - twice_low) / 2;
  in = fabsf(in - mn);
  const float extra = fmodf(in, span);
  const int flips = (int)floorf(in / span);
  return (flips % 2 == 0) ? extra + mn : span - extra + mn;
}
static inline float safe_int_range(float x) {
  if (x > 2147483646.f || x < -2147483648.f || !isfinite(x)) return -100.f;
  return x;
}
static inline float compute_coord(float c, int size, int pad, int align) {
  if (pad == BORDER) {
    c = clip_coord(c, size);
  } else if (pad == REFLECTION) {
    c = align ? reflect_coord(c, 0, 2 * (size - 1)) : reflect_coord(c, -1, 2 * size - 1);
    c = clip_coord(c, size);
  }
  return safe_int_range(c);
}
static inline float source_index(float c, int size, int pad, int align) {
  return compute_coord(unnormalize(c, size, align), size, pad, align);
}
static inline int in2d(int h, int w, int H, int W) { return h >= 0 && h < H && w >= 0 && w < W; }

static inline float bounded(const float *p, float x, float y, int W, int H, int pad, int align) {
  x = compute_coord(x, W, pad, align);
  y = compute_coord(y, H, pad, align);
  const int ix = (int)x, iy = (int)y;
  return in2d(iy, ix, H, W) ? p[iy * W + ix] : 0.f;
}
static inline float cc1(float x, float A) { return ((A + 2) * x - (A + 3)) * x * x + 1; }
static inline float cc2(float x, float A) { return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A; }
static inline void cubic_coeffs(float c[4], float t) {
  const float A = -0.75f;
  float x1 = t;
  c[0] = cc2(x1 + 1.0f, A);
  c[1] = cc1(x1, A);
  float x2 = 1.0f - t;
  c[2] = cc1(x2, A);
  c[3] = cc2(x2 + 1.0f, A);
}
static inline float cubic_interp1d(float x0, float x1, float x2, float x3, float t) {
  float c[4];
  cubic_coeffs(c, t);
  return x0 * c[0] + x1 * c[1] + x2 * c[2] + x3 * c[3];
}

/* one output pixel of one channel plane, source location already in pixel units
 * for bilinear/nearest (ix, iy after padding), raw unnormalised for bicubic */
static float sample2d(const float *p, int H, int W, float gx, float gy, int interp, int pad,
                      int align) {
  if (interp == BILINEAR) {
    const float ix = source_index(gx, W, pad, align), iy = source_index(gy, H, pad, align);
    const int ix_nw = (int)floorf(ix), iy_nw = (int)floorf(iy);
    const int ix_se = ix_nw + 1, iy_se = iy_nw + 1;
    const float nw = (ix_se - ix) * (iy_se - iy), ne = (ix - ix_nw) * (iy_se - iy);
    const float sw = (ix_se - ix) * (iy - iy_nw), se = (ix - ix_nw) * (iy - iy_nw);
    float o = 0.f;
    if (in2d(iy_nw, ix_nw, H, W)) o += p[iy_nw * W + ix_nw] * nw;
    if (in2d(iy_nw, ix_se, H, W)) o += p[iy_nw * W + ix_se] * ne;
    if (in2d(iy_se, ix_nw, H, W)) o += p[iy_se * W + ix_nw] * sw;
    if (in2d(iy_se, ix_se, H, W)) o += p[iy_se * W + ix_se] * se;
    return o;
  } else if (interp == NEAREST) {
    const float ix = source_index(gx, W, pad, align), iy = source_index(gy, H, pad, align);
    const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
    return in2d(yn, xn, H, W) ? p[yn * W + xn] : 0.f;
  } else {
    float ix = unnormalize(gx, W, align), iy = unnormalize(gy, H, align);
    const float ix_nw = floorf(ix), iy_nw = floorf(iy);
    const float tx = ix - ix_nw, ty = iy - iy_nw;
    float col[4];
    for (int i = 0; i < 4; ++i)
      col[i] = cubic_interp1d(bounded(p, ix_nw - 1, iy_nw - 1 + i, W, H, pad, align),
                              bounded(p, ix_nw + 0, iy_nw - 1 + i, W, H, pad, align),
                              bounded(p, ix_nw + 1, iy_nw - 1 + i, W, H, pad, align),
                              bounded(p, ix_nw + 2, iy_nw - 1 + i, W, H, pad, align), tx);
    return cubic_interp1d(col[0], col[1], col[2], col[3], ty);
  }
}

void oracle_grid_sampler_2d(const float *input, const float *grid, float *out, int N, int C,
                            int H, int W, int Ho, int Wo, int interp, int pad, int align) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int h = 0; h < Ho; ++h)
      for (int w = 0; w < Wo; ++w) {
        /* functions/grid_sampler.py:28-29 : permute + divide by 10 */
        const float gx = grid[((long)(n * 2 + 0) * Ho + h) * Wo + w] / 10;
        const float gy = grid[((long)(n * 2 + 1) * Ho + h) * Wo + w] / 10;
        for (int c = 0; c < C; ++c)
          out[(((long)n * C + c) * Ho + h) * Wo + w] =
              sample2d(input + ((long)n * C + c) * H * W, H, W, gx, gy, interp, pad, align);
      }
}

static inline int in3d(int d, int h, int w, int D, int H, int W) {
  return d >= 0 && d < D && h >= 0 && h < H && w >= 0 && w < W;
}

void oracle_grid_sampler_3d(const float *input, const float *grid, float *out, int N, int C,
                            int D, int H, int W, int Do, int Ho, int Wo, int interp, int pad,
                            int align) {
  const long plane = (long)Do * Ho * Wo;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (long s = 0; s < plane; ++s) {
      const float gx = grid[((long)n * 3 + 0) * plane + s] / 10;
      const float gy = grid[((long)n * 3 + 1) * plane + s] / 10;
      const float gz = grid[((long)n * 3 + 2) * plane + s] / 10;
      const float ix = source_index(gx, W, pad, align);
      const float iy = source_index(gy, H, pad, align);
      const float iz = source_index(gz, D, pad, align);
      for (int c = 0; c < C; ++c) {
        const float *p = input + ((long)n * C + c) * D * H * W;
        float o = 0.f;
        if (interp == BILINEAR) {
          const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), z0 = (int)floorf(iz);
          const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
          /* corner order and weights as aten (tnw, tne, tsw, tse, bnw, bne, bsw, bse) */
          const float tnw = (x1 - ix) * (y1 - iy) * (z1 - iz), tne = (ix - x0) * (y1 - iy) * (z1 - iz);
          const float tsw = (x1 - ix) * (iy - y0) * (z1 - iz), tse = (ix - x0) * (iy - y0) * (z1 - iz);
          const float bnw = (x1 - ix) * (y1 - iy) * (iz - z0), bne = (ix - x0) * (y1 - iy) * (iz - z0);
          const float bsw = (x1 - ix) * (iy - y0) * (iz - z0), bse = (ix - x0) * (iy - y0) * (iz - z0);
          if (in3d(z0, y0, x0, D, H, W)) o += p[((long)z0 * H + y0) * W + x0] * tnw;
          if (in3d(z0, y0, x1, D, H, W)) o += p[((long)z0 * H + y0) * W + x1] * tne;
          if (in3d(z0, y1, x0, D, H, W)) o += p[((long)z0 * H + y1) * W + x0] * tsw;
          if (in3d(z0, y1, x1, D, H, W)) o += p[((long)z0 * H + y1) * W + x1] * tse;
          if (in3d(z1, y0, x0, D, H, W)) o += p[((long)z1 * H + y0) * W + x0] * bnw;
          if (in3d(z1, y0, x1, D, H, W)) o += p[((long)z1 * H + y0) * W + x1] * bne;
          if (in3d(z1, y1, x0, D, H, W)) o += p[((long)z1 * H + y1) * W + x0] * bsw;
          if (in3d(z1, y1, x1, D, H, W)) o += p[((long)z1 * H + y1) * W + x1] * bse;
        } else {
          const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy), zn = (int)nearbyintf(iz);
          if (in3d(zn, yn, xn, D, H, W)) o = p[((long)zn * H + yn) * W + xn];
        }
        out[((long)n * C + c) * plane + s] = o;
      }
    }
}

/* functions/rotate.py:12-66.  angle in degrees, center (x, y) in pixels.
 * The 3x2 "rescaled_theta" product is evaluated as (x*a + y*b) + c in fp32. */
void oracle_rotate(const float *img, float angle_deg, float center_x, float center_y, float *out,
                   int C, int H, int W, int interp) {
  const float cx = center_x - (float)(W * 0.5), cy = center_y - (float)(H * 0.5);
  const float ang = -angle_deg * (float)M_PI / 180.f;
  const float cs = cosf(ang), sn = sinf(ang);
  const float th[6] = {cs, sn, -cx * cs - cy * sn + cx, -sn, cs, cx * sn - cy * cs + cy};
  /* rescaled_theta = 2 * theta^T, column 0 / W, column 1 / H  (rotate.py:44-46) */
  const float ax = 2 * th[0] / W, bx = 2 * th[1] / W, cx2 = 2 * th[2] / W;
  const float ay = 2 * th[3] / H, by = 2 * th[4] / H, cy2 = 2 * th[5] / H;
#pragma omp parallel for schedule(static)
  for (int h = 0; h < H; ++h)
    for (int w = 0; w < W; ++w) {
      const float x = (float)(-W * 0.5 + 0.5) + w, y = (float)(-H * 0.5 + 0.5) + h;
      const float gx = (x * ax + y * bx) + cx2, gy = (x * ay + y * by) + cy2;
      for (int c = 0; c < C; ++c)
        out[((long)c * H + h) * W + w] =
            sample2d(img + (long)c * H * W, H, W, gx, gy, interp, ZEROS, 0);
    }
}

/* ---- INT8 flavours --------------------------------------------------------------
 * gridSamplerKernel.cu:1082-1204 and rotateKernel.cu:415-560: bilinear area weights
 * quantised to int8 (x127, RNE via half2int8), int32 dot over the 4 corners,
 * out = T2int8(t * (1/127) * s_in / s_out); nearest: out = T2int8(v * s_in / s_out).
 * Coordinates are evaluated in fp32 here (the reference uses half2) and out-of-range
 * corners contribute 0 (the reference leaves `inps[]` stale there -- SURVEY.md
 * Appendix B, "reference behaviours not to copy").  Pinned against the reference's int8
 * kernels run on the host (tests/test_ref_kernels_cpu.py): nearest grid_sampler identical,
 * the rest >= 97 % within 1 LSB, max 3-4.  Dense [C,H,W] int8 layout. */
static inline int8_t t2int8_f(float a) {
  a = a > 127 ? 127 : a;
  a = a < -128 ? -128 : a;
  return (int8_t)(a + (a > 0 ? 0.5f : -0.5f));
}
static inline int q127_rne(float area) {
  float r = nearbyintf(area * 127.f);
  r = r > 127 ? 127 : r;
  r = r < -128 ? -128 : r;
  return (int)r;
}
static inline int bounded_s8(const int8_t *p, float x, float y, int W, int H, int pad, int align) {
  x = compute_coord(x, W, pad, align);
  y = compute_coord(y, H, pad, align);
  const int ix = (int)x, iy = (int)y;
  return in2d(iy, ix, H, W) ? p[iy * W + ix] : 0;
}

static int8_t sample2d_s8(const int8_t *p, int H, int W, float gx, float gy, int interp, int pad,
                          int align, float s_in, float s_out) {
  if (interp == BICUBIC) { /* gridSamplerKernel.cu:581-613,1205-1262 */
    const float ux = unnormalize(gx, W, align), uy = unnormalize(gy, H, align);
    const float x_nw = floorf(ux), y_nw = floorf(uy);
    float cxf[4], cyf[4];
    cubic_coeffs(cxf, ux - x_nw);
    cubic_coeffs(cyf, uy - y_nw);
    int col[4];
    for (int i = 0; i < 4; ++i) {
      int t = 0;
      for (int k = 0; k < 4; ++k)
        t += bounded_s8(p, x_nw - 1 + k, y_nw - 1 + i, W, H, pad, align) * (int)(int8_t)(cxf[k] * 127);
      col[i] = (int8_t)(t / 127);
    }
    int t = 0;
    for (int i = 0; i < 4; ++i) t += col[i] * (int)(int8_t)(cyf[i] * 127);
    return t2int8_f((float)(int8_t)(t / 127) * (s_in / s_out));
  }
  const float ix = source_index(gx, W, pad, align), iy = source_index(gy, H, pad, align);
  if (interp == NEAREST) {
    const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
    const float v = in2d(yn, xn, H, W) ? (float)p[yn * W + xn] : 0.f;
    return in2d(yn, xn, H, W) ? t2int8_f(v * (s_in / s_out)) : 0;
  }
  const int x0 = (int)floorf(ix), y0 = (int)floorf(iy), x1 = x0 + 1, y1 = y0 + 1;
  const int wq[4] = {q127_rne((x1 - ix) * (y1 - iy)), q127_rne((ix - x0) * (y1 - iy)),
                     q127_rne((x1 - ix) * (iy - y0)), q127_rne((ix - x0) * (iy - y0))};
  int t = 0;
  if (in2d(y0, x0, H, W)) t += p[y0 * W + x0] * wq[0];
  if (in2d(y0, x1, H, W)) t += p[y0 * W + x1] * wq[1];
  if (in2d(y1, x0, H, W)) t += p[y1 * W + x0] * wq[2];
  if (in2d(y1, x1, H, W)) t += p[y1 * W + x1] * wq[3];
  return t2int8_f((float)t * ((1.f / 127.f) * s_in / s_out));
}

void oracle_grid_sampler_2d_s8(const int8_t *input, const int8_t *grid, int8_t *out, int N, int C,
                               int H, int W, int Ho, int Wo, int interp, int pad, int align,
                               float s_in, float s_grid, float s_out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int h = 0; h < Ho; ++h)
      for (int w = 0; w < Wo; ++w) {
        const float gx = grid[((long)(n * 2 + 0) * Ho + h) * Wo + w] * s_grid / 10;
        const float gy = grid[((long)(n * 2 + 1) * Ho + h) * Wo + w] * s_grid / 10;
        for (int c = 0; c < C; ++c)
          out[(((long)n * C + c) * Ho + h) * Wo + w] = sample2d_s8(
              input + ((long)n * C + c) * H * W, H, W, gx, gy, interp, pad, align, s_in, s_out);
      }
}

void oracle_rotate_s8(const int8_t *img, float angle_deg, float center_x, float center_y,
                      int8_t *out, int C, int H, int W, int interp, float s_in, float s_out) {
  const float cx = center_x - (float)(W * 0.5), cy = center_y - (float)(H * 0.5);
  const float ang = -angle_deg * (float)M_PI / 180.f;
  const float cs = cosf(ang), sn = sinf(ang);
  const float th[6] = {cs, sn, -cx * cs - cy * sn + cx, -sn, cs, cx * sn - cy * cs + cy};
  const float ax = 2 * th[0] / W, bx = 2 * th[1] / W, cx2 = 2 * th[2] / W;
  const float ay = 2 * th[3] / H, by = 2 * th[4] / H, cy2 = 2 * th[5] / H;
#pragma omp parallel for schedule(static)
  for (int h = 0; h < H; ++h)
    for (int w = 0; w < W; ++w) {
      const float x = (float)(-W * 0.5 + 0.5) + w, y = (float)(-H * 0.5 + 0.5) + h;
      const float gx = (x * ax + y * bx) + cx2, gy = (x * ay + y * by) + cy2;
      for (int c = 0; c < C; ++c)
        out[((long)c * H + h) * W + w] =
            sample2d_s8(img + (long)c * H * W, H, W, gx, gy, interp, ZEROS, 0, s_in, s_out);
    }
}
