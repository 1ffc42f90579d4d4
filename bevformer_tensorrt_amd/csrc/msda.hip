// Multi-scale deformable attention forward for MI355X (gfx950).
//
// Replaces ms_deformable_im2col_cuda{,_h2,_int8} + MultiScaleDeformableAttnPlugin::enqueue
// of the reference (TensorRT/plugin/multi_scale_deformable_attn/
// multiScaleDeformableAttnKernel.cu:611-1218, ...Plugin.cpp:71-140).  Numerical
// contract: SURVEY.md Appendix A.1/A.2 (fused softmax over L*P logits, location
// = ref*(W,H) + offset - 0.5, 4-tap bilinear with per-corner bounds, sum/S).
//
// Design (not the reference's one-thread-per-output-scalar):
//   * one QUAD of lanes (4 lanes) owns one (batch, query, head) item; a lane holds
//     C/4 = 8 channels, so one tap is 4 x 16 B (fp16) contiguous = one 64 B segment
//     of the [bs, nk, heads, C] value map; a wave covers 16 items.
//   * the L*P points of an item are SPLIT over the quad: each lane loads 1/4 of the
//     logits/offsets (perfectly coalesced 16 B/lane across the wave), does the
//     softmax-exp / location / bounds / corner-weight math for its own points only,
//     and hands (4 corner weights, 4 corner byte-offsets) to its three neighbours
//     with DPP quad broadcasts -- no LDS, no redundant exp.  softmax max/sum are
//     quad DPP reductions.
//   * taps go through a buffer descriptor (32-bit offsets, wave-uniform SRD from
//     kernargs), fp32 accumulate (v_fma_mix_f32 on packed halves).
//   * items whose every point of the whole wave is out of range leave early
//     (cameras that do not see a BEV pillar).
//   * blockIdx is remapped so each XCD walks a contiguous item range (L2 locality
//     for neighbouring BEV queries).
#include "msda_common.h"

namespace bevops {
namespace {

// ---------------------------------------------------------------------------
// quad kernel: C == 32, (L*P) % 4 == 0.  PPL = points owned per lane = L*P/4.
// CH = how many of its own points a lane prepares per pass (register pressure
// knob: state is 8 VGPR per prepared point).
// ---------------------------------------------------------------------------
template <typename T, int PPL, int CH>
__global__ __launch_bounds__(kBlock) void msda_quad_kernel(
    const T *__restrict__ value, unsigned value_bytes, const int32_t *__restrict__ shapes,
    const T *__restrict__ ref, const T *__restrict__ off, const T *__restrict__ logit,
    T *__restrict__ out, MsdaDims d, unsigned n_item) {
  static_assert(PPL % CH == 0, "CH must divide PPL");
  constexpr int V = 8;  // channels per lane
  __shared__ int4 lvl[kMaxLevels];  // {H, W, first pixel of the level, -}
  if (threadIdx.x == 0) {
    int start = 0;
    for (int l = 0; l < d.L; ++l) {
      const int H = shapes[2 * l], W = shapes[2 * l + 1];
      lvl[l] = make_int4(H, W, start, 0);
      start += H * W;
    }
  }
  __syncthreads();

  const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned item = vb * (kBlock / 4) + (threadIdx.x >> 2);
  if (item >= n_item) return;  // whole quads leave together
  const unsigned sub = threadIdx.x & 3u;
  const unsigned bq = item / (unsigned)d.heads;
  const unsigned h = item - bq * (unsigned)d.heads;
  const unsigned b = bq / (unsigned)d.nq;
  constexpr int LP = 4 * PPL;
  const unsigned row_bytes = (unsigned)d.heads * 32u * (unsigned)sizeof(T);  // one pixel, all heads
  const unsigned lane_base =
      ((b * (unsigned)d.nk * (unsigned)d.heads + h) * 32u + sub * V) * (unsigned)sizeof(T);

  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(value), 0, value_bytes, 0x00020000);

  // ---- own logits -> softmax numerators ----
  // row of this item in sampling_offsets / attention_weights (camera-shared when d.shared)
  const size_t in_item = d.shared ? (size_t)(item - b * (unsigned)d.nq * (unsigned)d.heads) : (size_t)item;
  float e[PPL];
  load_f<PPL>(logit + in_item * LP + sub * PPL, e);
  float m = e[0];
#pragma unroll
  for (int k = 1; k < PPL; ++k) m = fmaxf(m, e[k]);
  m = quad_max(m);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    e[k] = __expf(e[k] - m);
    s += e[k];
  }
  s = quad_sum(s);

  float offs[2 * PPL];
  load_f<2 * PPL>(off + (in_item * LP + sub * PPL) * 2, offs);
  const T *refp = ref + (size_t)bq * (unsigned)d.ppg * 2u;

  // running (level, point-in-level, point-in-group) of this lane's next own point
  int j0 = (int)sub * PPL;
  int l = j0 / d.P;
  int p = j0 - l * d.P;
  int g = p % d.ppg;

  float acc[V];
#pragma unroll
  for (int c = 0; c < V; ++c) acc[c] = 0.f;

#pragma unroll
  for (int pass = 0; pass < PPL / CH; ++pass) {
    float ow[CH][4];
    unsigned oo[CH][4];
    bool any_valid = false;
#pragma unroll
    for (int kk = 0; kk < CH; ++kk) {
      const int k = pass * CH + kk;
      const int4 t = lvl[l];
      const int H = t.x, W = t.y;
      const float2 r = load_ref(refp + 2 * g);
      const float x = loc_im(r.x, (float)W, offs[2 * k]);
      const float y = loc_im(r.y, (float)H, offs[2 * k + 1]);
      const bool valid = (y > -1.f) && (x > -1.f) && (y < (float)H) && (x < (float)W);
      any_valid |= valid;
      const float xf = floorf(x), yf = floorf(y);
      const float lx = x - xf, ly = y - yf;
      const float hx = 1.f - lx, hy = 1.f - ly;
      const int x0 = (int)xf, y0 = (int)yf;
      const bool x0ok = valid && x0 >= 0, x1ok = valid && x0 + 1 <= W - 1;
      const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= H - 1;
      const float wt = e[k];
      ow[kk][0] = (y0ok && x0ok) ? wt * (hy * hx) : 0.f;
      ow[kk][1] = (y0ok && x1ok) ? wt * (hy * lx) : 0.f;
      ow[kk][2] = (y1ok && x0ok) ? wt * (ly * hx) : 0.f;
      ow[kk][3] = (y1ok && x1ok) ? wt * (ly * lx) : 0.f;
      const int x0c = min(max(x0, 0), W - 1), x1c = min(max(x0 + 1, 0), W - 1);
      const int y0c = min(max(y0, 0), H - 1), y1c = min(max(y0 + 1, 0), H - 1);
      const unsigned r0 = (unsigned)(t.z + y0c * W), r1 = (unsigned)(t.z + y1c * W);
      oo[kk][0] = (r0 + (unsigned)x0c) * row_bytes;
      oo[kk][1] = (r0 + (unsigned)x1c) * row_bytes;
      oo[kk][2] = (r1 + (unsigned)x0c) * row_bytes;
      oo[kk][3] = (r1 + (unsigned)x1c) * row_bytes;
      // advance (l, p, g)
      ++p; ++g;
      if (g == d.ppg) g = 0;
      if (p == d.P) { p = 0; g = 0; ++l; }
    }
    // cameras that do not see this pillar: nothing in the whole wave to gather
    if (!__any(any_valid)) continue;

#define BEVOPS_MSDA_SRC(S)                                                      \
    _Pragma("unroll") for (int kk = 0; kk < CH; ++kk) {                         \
      const float w0 = quad_bcast<S>(ow[kk][0]), w1 = quad_bcast<S>(ow[kk][1]); \
      const float w2 = quad_bcast<S>(ow[kk][2]), w3 = quad_bcast<S>(ow[kk][3]); \
      const unsigned o0 = quad_bcast<S>(oo[kk][0]) + lane_base;                 \
      const unsigned o1 = quad_bcast<S>(oo[kk][1]) + lane_base;                 \
      const unsigned o2 = quad_bcast<S>(oo[kk][2]) + lane_base;                 \
      const unsigned o3 = quad_bcast<S>(oo[kk][3]) + lane_base;                 \
      tap8(value, rs, o0, w0, acc);                                             \
      tap8(value, rs, o1, w1, acc);                                             \
      tap8(value, rs, o2, w2, acc);                                             \
      tap8(value, rs, o3, w3, acc);                                             \
    }
    BEVOPS_MSDA_SRC(0)
    BEVOPS_MSDA_SRC(1)
    BEVOPS_MSDA_SRC(2)
    BEVOPS_MSDA_SRC(3)
#undef BEVOPS_MSDA_SRC
  }

  const float inv = 1.0f / s;
#pragma unroll
  for (int c = 0; c < V; ++c) acc[c] *= inv;
  store8(out + (size_t)item * 32u + sub * V, acc);
}

// ---------------------------------------------------------------------------
// generic kernel: any heads / C / L / P / ppg, 64-bit addressing.  One thread per
// output element (item, channel); the slow, always-correct path for shapes the
// quad kernel does not cover.
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<__half>(__half v) { return __half2float(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f<__half>(float v) { return __float2half_rn(v); }

template <typename T>
__global__ __launch_bounds__(kBlock) void msda_generic_kernel(
    const T *__restrict__ value, const int32_t *__restrict__ shapes, const T *__restrict__ ref,
    const T *__restrict__ off, const T *__restrict__ logit, T *__restrict__ out, MsdaDims d,
    size_t n_out) {
  const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n_out) return;
  const int c = (int)(idx % d.C);
  const size_t item = idx / d.C;
  const int h = (int)(item % d.heads);
  const size_t bq = item / d.heads;
  const size_t b = bq / d.nq;
  const int LP = d.L * d.P;
  const size_t in_item = d.shared ? item - b * (size_t)d.nq * d.heads : item;
  const T *lg = logit + in_item * LP;
  const T *of = off + in_item * LP * 2;
  const T *rp = ref + bq * d.ppg * 2;
  const size_t step = (size_t)d.heads * d.C;
  const T *vp = value + (b * d.nk * d.heads + h) * (size_t)d.C + c;
  float m = -INFINITY;
  for (int j = 0; j < LP; ++j) m = fmaxf(m, to_f(lg[j]));
  float acc = 0.f, s = 0.f;
  int j = 0;
  for (int l = 0; l < d.L; ++l) {
    const int H = shapes[2 * l], W = shapes[2 * l + 1];
    for (int p = 0; p < d.P; ++p, ++j) {
      const int g = p % d.ppg;
      const float x = loc_im(to_f(rp[2 * g]), (float)W, to_f(of[2 * j]));
      const float y = loc_im(to_f(rp[2 * g + 1]), (float)H, to_f(of[2 * j + 1]));
      const float wt = __expf(to_f(lg[j]) - m);
      s += wt;
      if (y > -1.f && x > -1.f && y < (float)H && x < (float)W) {
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = (int)xf, y0 = (int)yf;
        const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
        float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
        if (y0 >= 0 && x0 >= 0) v1 = to_f(vp[((size_t)y0 * W + x0) * step]);
        if (y0 >= 0 && x0 + 1 <= W - 1) v2 = to_f(vp[((size_t)y0 * W + x0 + 1) * step]);
        if (y0 + 1 <= H - 1 && x0 >= 0) v3 = to_f(vp[((size_t)(y0 + 1) * W + x0) * step]);
        if (y0 + 1 <= H - 1 && x0 + 1 <= W - 1)
          v4 = to_f(vp[((size_t)(y0 + 1) * W + x0 + 1) * step]);
        acc += wt * (hy * hx * v1 + hy * lx * v2 + ly * hx * v3 + ly * lx * v4);
      }
    }
    vp += (size_t)H * W * step;
  }
  out[idx] = from_f<T>(acc / s);
}


// ---------------------------------------------------------------------------
// INT8 flavours.  Same quad structure; a lane holds 8 int8
// channels (8-byte taps).  The owner lane quantises the 4 bilinear area weights of a
// point into one packed dword and the softmax weight into an int, so a point costs 6
// DPP broadcasts; the consumer transposes the 4 corner dwords with v_perm_b32 and
// reduces them with v_dot4_i32_i8.
//   U8W = false : reference <float> flavour  (kernel.cu:848-955, 290-358): signed x127
//                 weights, T2int8 = clamp + round-half-away, S = sum of QUANTISED weights
//   U8W = true  : reference <__half2> flavour (kernel.cu:957-1104, 360-460): unsigned x255
//                 weights, RNE rounding, S = sum of UN-quantised weights; gfx950 has no
//                 mixed-sign dot4, so v*a (a in 0..255) = dot4(v, a^0x80) + 128*dot4(v, 1).
//                 Intermediate math is fp32 here (the reference uses half2).
// ---------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void load_i8(const int8_t *p, float (&d)[N]) {
  int8_t raw[N];
  if constexpr (N == 16) *reinterpret_cast<uint4 *>(raw) = *reinterpret_cast<const uint4 *>(p);
  else if constexpr (N == 8) *reinterpret_cast<uint2 *>(raw) = *reinterpret_cast<const uint2 *>(p);
  else if constexpr (N == 4) *reinterpret_cast<unsigned *>(raw) = *reinterpret_cast<const unsigned *>(p);
  else if constexpr (N == 2) *reinterpret_cast<unsigned short *>(raw) = *reinterpret_cast<const unsigned short *>(p);
  else if constexpr (N == 1) raw[0] = p[0];
  else {
    static_assert(N % 16 == 0, "N");
#pragma unroll
    for (int i = 0; i < N / 16; ++i)
      reinterpret_cast<uint4 *>(raw)[i] = reinterpret_cast<const uint4 *>(p)[i];
  }
#pragma unroll
  for (int i = 0; i < N; ++i) d[i] = (float)raw[i];
}
template <typename RefT, int PPL, int CH, bool U8W>
__global__ __launch_bounds__(kBlock) void msda_quad_int8_kernel(
    const int8_t *__restrict__ value, unsigned value_bytes, const int32_t *__restrict__ shapes,
    const RefT *__restrict__ ref, const int8_t *__restrict__ off, const int8_t *__restrict__ logit,
    int8_t *__restrict__ out, MsdaDims d, unsigned n_item, float s_v, float s_o, float s_w,
    float s_out) {
  static_assert(PPL % CH == 0, "CH must divide PPL");
  constexpr int V = 8;
  __shared__ int4 lvl[kMaxLevels];
  if (threadIdx.x == 0) {
    int start = 0;
    for (int l = 0; l < d.L; ++l) {
      const int H = shapes[2 * l], W = shapes[2 * l + 1];
      lvl[l] = make_int4(H, W, start, 0);
      start += H * W;
    }
  }
  __syncthreads();
  const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned slot = vb * (kBlock / 4) + (threadIdx.x >> 2);
  if (slot >= n_item) return;
  const unsigned sub = threadIdx.x & 3u;
  const unsigned item = slot;
  const unsigned bq = item / (unsigned)d.heads;
  const unsigned h = item - bq * (unsigned)d.heads;
  const unsigned b = bq / (unsigned)d.nq;
  const unsigned row_bytes = (unsigned)d.heads * 32u;
  const unsigned lane_base = (b * (unsigned)d.nk * (unsigned)d.heads + h) * 32u + sub * V;
  constexpr int LP = 4 * PPL;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(value), 0, value_bytes, 0x00020000);

  const size_t in_item = d.shared ? (size_t)(item - b * (unsigned)d.nq * (unsigned)d.heads) : (size_t)item;
  float e[PPL];
  load_i8<PPL>(logit + in_item * LP + sub * PPL, e);
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    e[k] = mul_rn(e[k], s_w);
    m = fmaxf(m, e[k]);
  }
  m = quad_max(m);
  float s = 0.f;
  int wq[PPL];
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    if constexpr (U8W) {
      const float w255 = mul_rn(__expf(sub_rn(e[k], m)), 255.f);
      s = add_rn(s, w255);
      wq[k] = (int)u16_rne(w255);
    } else {
      wq[k] = t2i8_away(mul_rn(__expf(sub_rn(e[k], m)), 127.f));
      s += (float)wq[k];
    }
  }
  s = quad_sum(s);

  float offs[2 * PPL];
  load_i8<2 * PPL>(off + (in_item * LP + sub * PPL) * 2, offs);
  const RefT *refp = ref + (size_t)bq * (unsigned)d.ppg * 2u;
  int j0 = (int)sub * PPL;
  int l = j0 / d.P;
  int p = j0 - l * d.P;
  int g = p % d.ppg;

  int acc[V];
#pragma unroll
  for (int c = 0; c < V; ++c) acc[c] = 0;

#pragma unroll
  for (int pass = 0; pass < PPL / CH; ++pass) {
    unsigned oaw[CH], oo[CH][4];
    int owq[CH];
    bool any_valid = false;
#pragma unroll
    for (int kk = 0; kk < CH; ++kk) {
      const int k = pass * CH + kk;
      const int4 t = lvl[l];
      const int H = t.x, W = t.y;
      const float2 r = load_ref(refp + 2 * g);
      float x, y;
      {
#pragma clang fp contract(off)
        if constexpr (U8W) {  // kernel.cu:1040-1056: ref*size + (off*scale - 0.5)
          x = r.x * (float)W + (offs[2 * k] * s_o - 0.5f);
          y = r.y * (float)H + (offs[2 * k + 1] * s_o - 0.5f);
        } else {  // kernel.cu:905-917
          x = (r.x * (float)W + offs[2 * k] * s_o) - 0.5f;
          y = (r.y * (float)H + offs[2 * k + 1] * s_o) - 0.5f;
        }
      }
      const bool valid = (y > -1.f) && (x > -1.f) && (y < (float)H) && (x < (float)W);
      any_valid |= valid;
      const float xf = floorf(x), yf = floorf(y);
      const float lx = x - xf, ly = y - yf;
      const float hx = 1.f - lx, hy = 1.f - ly;
      const int x0 = (int)xf, y0 = (int)yf;
      const bool x0ok = x0 >= 0, x1ok = x0 + 1 <= W - 1;
      const bool y0ok = y0 >= 0, y1ok = y0 + 1 <= H - 1;
      unsigned a0, a1, a2, a3;
      if constexpr (U8W) {
        a0 = u16_rne(hy * hx * 255.f); a1 = u16_rne(hy * lx * 255.f);
        a2 = u16_rne(ly * hx * 255.f); a3 = u16_rne(ly * lx * 255.f);
      } else {
#pragma clang fp contract(off)
        const float sa = 1 / 127.f;
        a0 = (unsigned)t2i8_away((hy * hx) / sa); a1 = (unsigned)t2i8_away((hy * lx) / sa);
        a2 = (unsigned)t2i8_away((ly * hx) / sa); a3 = (unsigned)t2i8_away((ly * lx) / sa);
      }
      a0 = (y0ok && x0ok) ? a0 : 0u; a1 = (y0ok && x1ok) ? a1 : 0u;
      a2 = (y1ok && x0ok) ? a2 : 0u; a3 = (y1ok && x1ok) ? a3 : 0u;
      oaw[kk] = (a0 & 255u) | ((a1 & 255u) << 8) | ((a2 & 255u) << 16) | ((a3 & 255u) << 24);
      owq[kk] = valid ? wq[k] : 0;
      const int x0c = min(max(x0, 0), W - 1), x1c = min(max(x0 + 1, 0), W - 1);
      const int y0c = min(max(y0, 0), H - 1), y1c = min(max(y0 + 1, 0), H - 1);
      const unsigned r0 = (unsigned)(t.z + y0c * W), r1 = (unsigned)(t.z + y1c * W);
      oo[kk][0] = (r0 + (unsigned)x0c) * row_bytes;
      oo[kk][1] = (r0 + (unsigned)x1c) * row_bytes;
      oo[kk][2] = (r1 + (unsigned)x0c) * row_bytes;
      oo[kk][3] = (r1 + (unsigned)x1c) * row_bytes;
      ++p; ++g;
      if (g == d.ppg) g = 0;
      if (p == d.P) { p = 0; g = 0; ++l; }
    }
    if (!__any(any_valid)) continue;

#define BEVOPS_MSDA_I8_SRC(S)                                                                  \
    _Pragma("unroll") for (int kk = 0; kk < CH; ++kk) {                                        \
      const unsigned aw = quad_bcast<S>(oaw[kk]);                                              \
      const int wgt = (int)quad_bcast<S>((unsigned)owq[kk]);                                   \
      u32x2 tp[4];                                                                             \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) tp[q] = __builtin_amdgcn_raw_buffer_load_b64( \
          rs, (int)(quad_bcast<S>(oo[kk][q]) + lane_base), 0, 0);                              \
      _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                       \
        unsigned tr[4];                                                                        \
        transpose4x4(hf ? tp[0].y : tp[0].x, hf ? tp[1].y : tp[1].x, hf ? tp[2].y : tp[2].x,   \
                     hf ? tp[3].y : tp[3].x, tr);                                              \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                        \
          int tsum, smp;                                                                       \
          if constexpr (U8W) {                                                                 \
            tsum = __builtin_amdgcn_sdot4((int)tr[c], (int)(aw ^ 0x80808080u), 0, false) +     \
                   (__builtin_amdgcn_sdot4((int)tr[c], 0x01010101, 0, false) << 7);            \
            smp = t2i8_rne((float)tsum * (1.0f / 255.f));                                      \
          } else {                                                                             \
            tsum = __builtin_amdgcn_sdot4((int)tr[c], (int)aw, 0, false);                      \
            smp = t2i8_away((float)tsum * (1 / 127.f));                                        \
          }                                                                                    \
          acc[hf * 4 + c] += smp * wgt;                                                        \
        }                                                                                      \
      }                                                                                        \
    }
    BEVOPS_MSDA_I8_SRC(0)
    BEVOPS_MSDA_I8_SRC(1)
    BEVOPS_MSDA_I8_SRC(2)
    BEVOPS_MSDA_I8_SRC(3)
#undef BEVOPS_MSDA_I8_SRC
  }

  int8_t res[V];
  {
#pragma clang fp contract(off)
    const float scale_o = s_v * (1.0f / s_out);
    const float f = scale_o * (1.0f / s);
#pragma unroll
    for (int c = 0; c < V; ++c)
      res[c] = (int8_t)(U8W ? t2i8_rne((float)acc[c] * f) : t2i8_away((float)acc[c] * f));
  }
  *reinterpret_cast<uint2 *>(out + (size_t)item * 32u + sub * V) = *reinterpret_cast<const uint2 *>(res);
}

// generic int8: one thread per output element, any shape (reference needs C%4==0, P%4==0)
template <typename RefT, bool U8W>
__global__ __launch_bounds__(kBlock) void msda_generic_int8_kernel(
    const int8_t *__restrict__ value, const int32_t *__restrict__ shapes,
    const RefT *__restrict__ ref, const int8_t *__restrict__ off, const int8_t *__restrict__ logit,
    int8_t *__restrict__ out, MsdaDims d, size_t n_out, float s_v, float s_o, float s_w,
    float s_out) {
  const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n_out) return;
  const int c = (int)(idx % d.C);
  const size_t item = idx / d.C;
  const int h = (int)(item % d.heads);
  const size_t bq = item / d.heads;
  const size_t b = bq / d.nq;
  const int LP = d.L * d.P;
  const size_t in_item = d.shared ? item - b * (size_t)d.nq * d.heads : item;
  const int8_t *lg = logit + in_item * LP;
  const int8_t *of = off + in_item * LP * 2;
  const RefT *rp = ref + bq * d.ppg * 2;
  const size_t step = (size_t)d.heads * d.C;
  const int8_t *vp = value + (b * d.nk * d.heads + h) * (size_t)d.C + c;
  float m = -INFINITY;
  for (int j = 0; j < LP; ++j) m = fmaxf(m, (float)lg[j] * s_w);
  int acc = 0;
  float s = 0.f;
  int j = 0;
  for (int l = 0; l < d.L; ++l) {
    const int H = shapes[2 * l], W = shapes[2 * l + 1];
    for (int p = 0; p < d.P; ++p, ++j) {
      const int g = p % d.ppg;
      const float2 r = load_ref(rp + 2 * g);
      float x, y;
      int wq;
      {
#pragma clang fp contract(off)
        if constexpr (U8W) {
          x = r.x * (float)W + ((float)of[2 * j] * s_o - 0.5f);
          y = r.y * (float)H + ((float)of[2 * j + 1] * s_o - 0.5f);
          const float w255 = __expf((float)lg[j] * s_w - m) * 255.f;
          s += w255;
          wq = (int)u16_rne(w255);
        } else {
          x = (r.x * (float)W + (float)of[2 * j] * s_o) - 0.5f;
          y = (r.y * (float)H + (float)of[2 * j + 1] * s_o) - 0.5f;
          wq = t2i8_away(__expf((float)lg[j] * s_w - m) * 127.f);
          s += (float)wq;
        }
      }
      if (!(y > -1.f && x > -1.f && y < (float)H && x < (float)W)) continue;
      const float xf = floorf(x), yf = floorf(y);
      const int x0 = (int)xf, y0 = (int)yf;
      const float lx = x - xf, ly = y - yf, hx = 1.f - lx, hy = 1.f - ly;
      int v1 = 0, v2 = 0, v3 = 0, v4 = 0;
      if (y0 >= 0 && x0 >= 0) v1 = vp[((size_t)y0 * W + x0) * step];
      if (y0 >= 0 && x0 + 1 <= W - 1) v2 = vp[((size_t)y0 * W + x0 + 1) * step];
      if (y0 + 1 <= H - 1 && x0 >= 0) v3 = vp[((size_t)(y0 + 1) * W + x0) * step];
      if (y0 + 1 <= H - 1 && x0 + 1 <= W - 1) v4 = vp[((size_t)(y0 + 1) * W + x0 + 1) * step];
      int smp;
      if constexpr (U8W) {
        const int t = v1 * (int)u16_rne(hy * hx * 255.f) + v2 * (int)u16_rne(hy * lx * 255.f) +
                      v3 * (int)u16_rne(ly * hx * 255.f) + v4 * (int)u16_rne(ly * lx * 255.f);
        smp = t2i8_rne((float)t * (1.0f / 255.f));
      } else {
#pragma clang fp contract(off)
        const float sa = 1 / 127.f;
        const int t = v1 * t2i8_away((hy * hx) / sa) + v2 * t2i8_away((hy * lx) / sa) +
                      v3 * t2i8_away((ly * hx) / sa) + v4 * t2i8_away((ly * lx) / sa);
        smp = t2i8_away((float)t * sa);
      }
      acc += smp * wq;
    }
    vp += (size_t)H * W * step;
  }
  {
#pragma clang fp contract(off)
    const float f = (s_v * (1.0f / s_out)) * (1.0f / s);
    out[idx] = (int8_t)(U8W ? t2i8_rne((float)acc * f) : t2i8_away((float)acc * f));
  }
}

thread_local int g_variant = 0;

template <typename T, int PPL, int CH>
int launch_quad(const T *value, const int32_t *shapes, const T *ref, const T *off, const T *logit,
                T *out, const MsdaDims &d, hipStream_t st) {
  const size_t n_item = (size_t)d.bs * d.nq * d.heads;
  const size_t vbytes = (size_t)d.bs * d.nk * d.heads * d.C * sizeof(T);
  const unsigned grid = (unsigned)((n_item + kBlock / 4 - 1) / (kBlock / 4));
  hipLaunchKernelGGL((msda_quad_kernel<T, PPL, CH>), dim3(grid), dim3(kBlock), 0, st, value,
                     (unsigned)vbytes, shapes, ref, off, logit, out, d, (unsigned)n_item);
  return launch_status();
}

template <typename T>
int msda_float(const T *value, const int32_t *shapes, const T *ref, const T *off, const T *logit,
               T *out, const MsdaDims &d, hipStream_t st) {
  const size_t n_item = (size_t)d.bs * d.nq * d.heads;
  const size_t vbytes = (size_t)d.bs * d.nk * d.heads * d.C * sizeof(T);
  const int LP = d.L * d.P;
  const bool quad_ok = d.C == 32 && LP % 4 == 0 && d.L <= kMaxLevels &&
                       vbytes < 0xFFFFFF00ull && n_item < 0x7FFFFFFFull &&
                       n_item * 32 * sizeof(T) < 0xFFFFFFFFFFull && aligned16(value) &&
                       aligned16(off) && aligned16(logit) && aligned16(out) && aligned16(ref) &&
                       g_variant != 99;
  if (quad_ok) {
    const int v = g_variant;
    switch (LP / 4) {
      case 1: return launch_quad<T, 1, 1>(value, shapes, ref, off, logit, out, d, st);
      case 2: return launch_quad<T, 2, 2>(value, shapes, ref, off, logit, out, d, st);
      case 4:
        if (v == 2) return launch_quad<T, 4, 2>(value, shapes, ref, off, logit, out, d, st);
        return launch_quad<T, 4, 4>(value, shapes, ref, off, logit, out, d, st);
      case 8:
        if (v == 1) return launch_quad<T, 8, 8>(value, shapes, ref, off, logit, out, d, st);
        if (v == 2) return launch_quad<T, 8, 2>(value, shapes, ref, off, logit, out, d, st);
        return launch_quad<T, 8, 4>(value, shapes, ref, off, logit, out, d, st);
      case 16:
        if (v == 2) return launch_quad<T, 16, 2>(value, shapes, ref, off, logit, out, d, st);
        return launch_quad<T, 16, 4>(value, shapes, ref, off, logit, out, d, st);
      default: break;
    }
  }
  const size_t n_out = n_item * d.C;
  const size_t grid = (n_out + kBlock - 1) / kBlock;
  if (grid > 0x7FFFFFFFull) return BEVOPS_NOT_SUPPORTED;
  hipLaunchKernelGGL((msda_generic_kernel<T>), dim3((unsigned)grid), dim3(kBlock), 0, st, value,
                     shapes, ref, off, logit, out, d, n_out);
  return launch_status();
}

template <typename RefT, bool U8W, int PPL, int CH>
int launch_quad_i8(const int8_t *value, const int32_t *shapes, const RefT *ref, const int8_t *off,
                   const int8_t *logit, int8_t *out, const MsdaDims &d, float s_v, float s_o,
                   float s_w, float s_out, hipStream_t st) {
  const size_t n_item = (size_t)d.bs * d.nq * d.heads;
  const size_t vbytes = (size_t)d.bs * d.nk * d.heads * d.C;
  const unsigned grid = (unsigned)((n_item + kBlock / 4 - 1) / (kBlock / 4));
  hipLaunchKernelGGL((msda_quad_int8_kernel<RefT, PPL, CH, U8W>), dim3(grid), dim3(kBlock), 0, st,
                     value, (unsigned)vbytes, shapes, ref, off, logit, out, d, (unsigned)n_item, s_v,
                     s_o, s_w, s_out);
  return launch_status();
}

template <typename RefT, bool U8W>
int msda_int8(const int8_t *value, const int32_t *shapes, const RefT *ref, const int8_t *off,
              const int8_t *logit, int8_t *out, const MsdaDims &d, float s_v, float s_o, float s_w,
              float s_out, hipStream_t st) {
  const size_t n_item = (size_t)d.bs * d.nq * d.heads;
  const size_t vbytes = (size_t)d.bs * d.nk * d.heads * d.C;
  const int LP = d.L * d.P;
  const bool quad_ok = d.C == 32 && LP % 4 == 0 && d.L <= kMaxLevels && vbytes < 0xFFFFFF00ull &&
                       n_item < 0x7FFFFFFFull && aligned16(value) && aligned16(off) &&
                       aligned16(logit) && aligned16(out) && aligned16(ref) && g_variant != 99;
  if (quad_ok) {
    switch (LP / 4) {
      case 1: return launch_quad_i8<RefT, U8W, 1, 1>(value, shapes, ref, off, logit, out, d, s_v, s_o, s_w, s_out, st);
      case 2: return launch_quad_i8<RefT, U8W, 2, 2>(value, shapes, ref, off, logit, out, d, s_v, s_o, s_w, s_out, st);
      case 4: return launch_quad_i8<RefT, U8W, 4, 4>(value, shapes, ref, off, logit, out, d, s_v, s_o, s_w, s_out, st);
      case 8: return launch_quad_i8<RefT, U8W, 8, 4>(value, shapes, ref, off, logit, out, d, s_v, s_o, s_w, s_out, st);
      case 16: return launch_quad_i8<RefT, U8W, 16, 4>(value, shapes, ref, off, logit, out, d, s_v, s_o, s_w, s_out, st);
      default: break;
    }
  }
  const size_t n_out = n_item * d.C;
  const size_t grid = (n_out + kBlock - 1) / kBlock;
  if (grid > 0x7FFFFFFFull) return BEVOPS_NOT_SUPPORTED;
  hipLaunchKernelGGL((msda_generic_int8_kernel<RefT, U8W>), dim3((unsigned)grid), dim3(kBlock), 0,
                     st, value, shapes, ref, off, logit, out, d, n_out, s_v, s_o, s_w, s_out);
  return launch_status();
}

}  // namespace
}  // namespace bevops

using namespace bevops;

static thread_local bool g_sca_direct = true;   // bevops_sca_forward_planned: see set_variant 3012 / 3013
static thread_local int g_variant_raw = 0;   // the value last REQUESTED (19 maps to 17 + a flag below)
extern "C" int bevops_msda_set_variant(int variant) {
  const int prev = g_variant_raw;   // handing this back to set_variant restores the flags too
  // The 30xx values are independent knobs of the fused SCA op, NOT kernel-family selectors: they leave the family
  // selection (g_variant / g_variant_raw) alone, so a save / restore pair around a family switch -- prev =
  // set_variant(10); ...; set_variant(prev) -- still restores the family after any 30xx call in between.
  if (variant >= 3001 && variant <= 3008) {   // A/B: slices per CU of the planned fused SCA sampling (default 2)
    msda_hm5_set_plan_blocks(variant - 3000);
    return prev;
  }
  if (variant == 3010 || variant == 3011) {   // A/B: camera reduce of the fused SCA op unrolled (default) / rolled
    msda_sca_set_reduce_rolled(variant == 3011);
    return prev;
  }
  if (variant == 3012 || variant == 3013) {   // A/B: planned SCA stores single-camera pairs into the output (default) / not
    g_sca_direct = variant == 3012;
    return prev;
  }
  if (variant == 3014 || variant == 3015) {   // A/B: planned SCA sampler with the record broadcasts folded into their
    msda_hm5_set_fold(variant == 3014);       // consumers + fused LDS row taps (3014, default) / the round-5 build (3015)
    return prev;
  }
  g_variant_raw = variant;
  // 19 (A/B): int8 hm4 on the one-block-per-CU plan (the partner of the default two-blocks plan); g_variant then
  // reads 17 = "hm4 wherever it is instantiated"
  msda_hm4_set_no_occ(variant == 19);
  g_variant = variant == 19 ? 17 : variant;
  return prev;
}

// A head-major re-layout pays when a batch's maps overflow an XCD's 4 MiB L2 and there are enough
// samples per pixel to amortise it (profiles/r01: base SCA 1.75x, base TSA 1.1x; small / tiny
// maps are L2-resident already and stay on the layout-preserving kernels)
static bool hm_pays(int esize, int bs, int nk, int heads, int channels, int num_levels, int num_query,
                    int num_point) {
  const double samples = (double)bs * num_query * heads * num_levels * num_point;
  const double pixels = (double)bs * nk * heads;
  const double plane_mb = (double)nk * heads * channels * esize / 1048576.0;
  return (samples >= 16.0 * pixels && plane_mb >= 4.0) || (samples >= 4.0 * pixels && plane_mb >= 16.0);
}

extern "C" size_t bevops_msda_workspace_size(int dtype, int bs, int nk, int heads, int channels,
                                             int num_levels, int num_query, int num_point) {
  if (bs <= 0 || nk <= 0 || heads <= 0 || num_levels <= 0 || num_query <= 0 || num_point <= 0) return 0;
  if ((num_levels * num_point) % 4 != 0) return 0;
  if (dtype == BEVOPS_I8) {
    // padded head-major int8 planes (msda_hm4.hip): 128-byte entries, at most (H + 2)(W + 1) <= 3 H W
    // + 2 of them per level -- an upper bound; bevops_msda_workspace_size_shapes gives the exact size
    if (channels != 32 || g_variant == 10 || g_variant == 99) return 0;
    if (g_variant != 17 && !hm_pays(1, bs, nk, heads, channels, num_levels, num_query, num_point)) return 0;
    return (size_t)bs * heads * ((size_t)3 * nk + 2 * num_levels + 4) * 128 + 4096;
  }
  if (dtype != BEVOPS_F16) return 0;
  return msda_hm_workspace_bytes(bs, nk, heads, channels, num_levels);
}

extern "C" size_t bevops_msda_workspace_size_shapes(int dtype, const int32_t *spatial_shapes_host,
                                                    int bs, int nk, int heads, int channels,
                                                    int num_levels, int num_query, int num_point) {
  const size_t a = bevops_msda_workspace_size(dtype, bs, nk, heads, channels, num_levels, num_query,
                                              num_point);
  if (a == 0 || !spatial_shapes_host) return a;
  const size_t c = msda_hm4_workspace_bytes(spatial_shapes_host, bs, heads, channels, num_levels,
                                            num_query, num_point, dtype == BEVOPS_I8);
  if (dtype == BEVOPS_I8) return c;   // exact (0: shape outside the head-major domain)
  if (dtype != BEVOPS_F16) return a;
  size_t b = msda_hm3_workspace_bytes(spatial_shapes_host, bs, heads, channels, num_levels,
                                      num_query, num_point);
  const size_t e = msda_hm5_workspace_bytes(spatial_shapes_host, bs, heads, channels, num_levels, num_query,
                                            num_point);
  if (e > b) b = e;
  return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

extern "C" size_t bevops_msda_packed_size(int dtype, const int32_t *spatial_shapes_host, int bs, int nk,
                                          int heads, int channels, int num_levels, int num_query,
                                          int num_point) {
  if ((dtype != BEVOPS_F16 && dtype != BEVOPS_I8) || !spatial_shapes_host || bs <= 0 || nk <= 0 || heads <= 0 ||
      num_levels <= 0 || num_query <= 0 || num_point <= 0)
    return 0;
  return msda_hm4_workspace_bytes(spatial_shapes_host, bs, heads, channels, num_levels, num_query, num_point,
                                  dtype == BEVOPS_I8);
}

extern "C" int bevops_msda_pack_value(int dtype, int ref_dtype, const void *value,
                                      const int32_t *spatial_shapes_host, void *packed, size_t packed_bytes,
                                      int bs, int nk, int heads, int channels, int num_levels, int num_query,
                                      int num_point, void *stream) {
  if (!value || !spatial_shapes_host || !packed) return BEVOPS_BAD_PARAM;
  if (bs <= 0 || nk <= 0 || heads <= 0 || channels <= 0 || num_levels <= 0 || num_query <= 0 || num_point <= 0)
    return BEVOPS_BAD_PARAM;
  long total = 0;
  for (int l = 0; l < num_levels; ++l) {
    const long H = spatial_shapes_host[2 * l], W = spatial_shapes_host[2 * l + 1];
    if (H <= 0 || W <= 0) return BEVOPS_BAD_PARAM;
    total += H * W;
  }
  if (total != nk) return BEVOPS_BAD_PARAM;
  return msda_hm4_pack(dtype, ref_dtype, value, spatial_shapes_host, bs, nk, heads, channels, num_levels,
                       num_query, num_point, packed, packed_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int bevops_msda_forward_prepacked(int dtype, const void *packed, size_t packed_bytes,
                                             const int32_t *spatial_shapes_host, const void *reference_points,
                                             int ref_dtype, const void *sampling_offsets,
                                             const void *attention_weights, void *output, int bs, int nk,
                                             int heads, int channels, int num_levels, int num_query,
                                             int num_point, int points_per_group, float scale_value,
                                             float scale_offset, float scale_weight, float scale_out,
                                             int shared_offsets, void *stream) {
  if (!packed || !spatial_shapes_host || !reference_points || !sampling_offsets || !attention_weights || !output)
    return BEVOPS_BAD_PARAM;
  if (bs <= 0 || nk <= 0 || heads <= 0 || channels <= 0 || num_levels <= 0 || num_query <= 0 ||
      num_point <= 0 || points_per_group <= 0)
    return BEVOPS_BAD_PARAM;
  if (dtype == BEVOPS_F16 && ref_dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (dtype == BEVOPS_I8 && (!(scale_value > 0.f) || !(scale_offset > 0.f) || !(scale_weight > 0.f) ||
                             !(scale_out > 0.f)))
    return BEVOPS_BAD_PARAM;
  return msda_hm4_forward_prepacked(dtype, ref_dtype, packed, packed_bytes, spatial_shapes_host, reference_points,
                                    sampling_offsets, attention_weights, output, bs, nk, heads, channels,
                                    num_levels, num_query, num_point, points_per_group, shared_offsets ? 1 : 0,
                                    scale_value, scale_offset, scale_weight, scale_out, 0, 0,
                                    static_cast<hipStream_t>(stream));
}

extern "C" size_t bevops_sca_workspace_size(int dtype, const int32_t *spatial_shapes_host, int num_cams,
                                            int nk, int heads, int channels, int num_levels,
                                            int num_query, int num_point) {
  if (dtype != BEVOPS_F16 || !spatial_shapes_host || num_cams <= 0 || nk <= 0 || heads <= 0 ||
      num_levels <= 0 || num_query <= 0 || num_point <= 0)
    return 0;
  return msda_hm3_sca_workspace_bytes(spatial_shapes_host, num_cams, heads, channels, num_levels,
                                      num_query, num_point);
}

extern "C" int bevops_sca_forward(int dtype, const void *value, const int32_t *spatial_shapes_host,
                                  const void *reference_points_cam, const void *sampling_offsets,
                                  const void *attention_weights, const void *bev_mask, void *output,
                                  int num_cams, int nk, int heads, int channels, int num_levels,
                                  int num_query, int num_point, int points_per_group,
                                  void *workspace, size_t workspace_bytes, void *stream) {
  if (!value || !spatial_shapes_host || !reference_points_cam || !sampling_offsets ||
      !attention_weights || !bev_mask || !output)
    return BEVOPS_BAD_PARAM;
  if (num_cams <= 0 || nk <= 0 || heads <= 0 || channels <= 0 || num_levels <= 0 || num_query <= 0 ||
      num_point <= 0 || points_per_group <= 0)
    return BEVOPS_BAD_PARAM;
  long total = 0;
  for (int l = 0; l < num_levels; ++l) {
    const long H = spatial_shapes_host[2 * l], W = spatial_shapes_host[2 * l + 1];
    if (H <= 0 || W <= 0) return BEVOPS_BAD_PARAM;
    total += H * W;
  }
  if (total != nk) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  return msda_hm3_sca_forward_f16((const __half *)value, spatial_shapes_host,
                                  (const __half *)reference_points_cam, (const __half *)sampling_offsets,
                                  (const __half *)attention_weights, (const __half *)bev_mask,
                                  (__half *)output, num_cams, nk, heads, channels, num_levels, num_query,
                                  num_point, points_per_group, workspace, workspace_bytes,
                                  static_cast<hipStream_t>(stream));
}

extern "C" size_t bevops_sca_prepacked_workspace_size(int num_cams, int heads, int channels, int num_query) {
  if (num_cams <= 0 || heads <= 0 || channels <= 0 || num_query <= 0) return 0;
  return (size_t)num_cams * num_query * heads * channels * sizeof(__half);
}

extern "C" int bevops_sca_forward_prepacked(int dtype, const void *packed, size_t packed_bytes,
                                            const int32_t *spatial_shapes_host, const void *reference_points_cam,
                                            const void *sampling_offsets, const void *attention_weights,
                                            const void *bev_mask, void *output, int num_cams, int nk, int heads,
                                            int channels, int num_levels, int num_query, int num_point,
                                            int points_per_group, void *workspace, size_t workspace_bytes,
                                            void *stream) {
  if (!packed || !spatial_shapes_host || !reference_points_cam || !sampling_offsets || !attention_weights || !bev_mask ||
      !output || !workspace)
    return BEVOPS_BAD_PARAM;
  if (num_cams <= 0 || nk <= 0 || heads <= 0 || channels <= 0 || num_levels <= 0 || num_query <= 0 || num_point <= 0 ||
      points_per_group <= 0)
    return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (workspace_bytes < bevops_sca_prepacked_workspace_size(num_cams, heads, channels, num_query) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15u))
    return BEVOPS_BAD_PARAM;
  hipStream_t st = static_cast<hipStream_t>(stream);
  __half *sampled = static_cast<__half *>(workspace);
  const int rc = msda_hm5_sca_sample_f16(packed, packed_bytes, spatial_shapes_host, (const __half *)reference_points_cam,
                                         (const __half *)sampling_offsets, (const __half *)attention_weights,
                                         (const __half *)bev_mask, sampled, num_cams, nk, heads, channels, num_levels,
                                         num_query, num_point, points_per_group, st);
  if (rc != BEVOPS_SUCCESS) return rc;
  msda_sca_reduce_launch(sampled, (const __half *)bev_mask, (__half *)output, num_cams, num_query, heads * channels, false, st);
  return launch_status();
}

extern "C" size_t bevops_sca_plan_size(int num_cams, int num_query) { return msda_hm5_plan_bytes(num_cams, num_query); }

extern "C" int bevops_sca_plan_build(int dtype, const void *bev_mask, int num_cams, int num_query, void *plan,
                                     size_t plan_bytes, void *stream) {
  if (!bev_mask || !plan || num_cams <= 0 || num_query <= 0) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  return msda_hm5_plan_build((const __half *)bev_mask, num_cams, num_query, plan, plan_bytes,
                             static_cast<hipStream_t>(stream));
}

extern "C" int bevops_sca_forward_planned(int dtype, const void *packed, size_t packed_bytes,
                                          const int32_t *spatial_shapes_host, const void *reference_points_cam,
                                          const void *sampling_offsets, const void *attention_weights,
                                          const void *bev_mask, const void *plan, size_t plan_bytes, void *output,
                                          int num_cams, int nk, int heads, int channels, int num_levels, int num_query,
                                          int num_point, int points_per_group, void *workspace, size_t workspace_bytes,
                                          void *stream) {
  if (!packed || !spatial_shapes_host || !reference_points_cam || !sampling_offsets || !attention_weights || !bev_mask ||
      !plan || !output || !workspace)
    return BEVOPS_BAD_PARAM;
  if (num_cams <= 0 || nk <= 0 || heads <= 0 || channels <= 0 || num_levels <= 0 || num_query <= 0 || num_point <= 0 ||
      points_per_group <= 0)
    return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (workspace_bytes < bevops_sca_prepacked_workspace_size(num_cams, heads, channels, num_query) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15u))
    return BEVOPS_BAD_PARAM;
  hipStream_t st = static_cast<hipStream_t>(stream);
  __half *sampled = static_cast<__half *>(workspace);
  const int rc = msda_hm5_sca_sample_planned_f16(packed, packed_bytes, spatial_shapes_host,
                                                 (const __half *)reference_points_cam, (const __half *)sampling_offsets,
                                                 (const __half *)attention_weights, plan, plan_bytes, sampled,
                                                 g_sca_direct ? (__half *)output : nullptr, num_cams, nk, heads, channels,
                                                 num_levels, num_query, num_point, points_per_group, st);
  if (rc != BEVOPS_SUCCESS) return rc;
  msda_sca_reduce_launch(sampled, (const __half *)bev_mask, (__half *)output, num_cams, num_query, heads * channels,
                         g_sca_direct, st);
  return launch_status();
}

extern "C" int bevops_msda_forward(int dtype, const void *value, const int32_t *spatial_shapes,
                                   const int32_t *spatial_shapes_host,
                                   const void *reference_points, int ref_dtype,
                                   const void *sampling_offsets, const void *attention_weights,
                                   void *output, int bs, int nk, int heads, int channels,
                                   int num_levels, int num_query, int num_point,
                                   int points_per_group, float scale_value, float scale_offset,
                                   float scale_weight, float scale_out, void *stream) {
  return bevops_msda_forward_ws(dtype, value, spatial_shapes, spatial_shapes_host,
                                reference_points, ref_dtype, sampling_offsets, attention_weights,
                                output, bs, nk, heads, channels, num_levels, num_query, num_point,
                                points_per_group, scale_value, scale_offset, scale_weight,
                                scale_out, 0, nullptr, 0, stream);
}

extern "C" int bevops_msda_forward_ws(int dtype, const void *value, const int32_t *spatial_shapes,
                                      const int32_t *spatial_shapes_host,
                                      const void *reference_points, int ref_dtype,
                                      const void *sampling_offsets, const void *attention_weights,
                                      void *output, int bs, int nk, int heads, int channels,
                                      int num_levels, int num_query, int num_point,
                                      int points_per_group, float scale_value, float scale_offset,
                                      float scale_weight, float scale_out, int shared_offsets,
                                      void *workspace, size_t workspace_bytes, void *stream) {
  if (!value || !spatial_shapes || !reference_points || !sampling_offsets || !attention_weights ||
      !output)
    return BEVOPS_BAD_PARAM;
  if (bs <= 0 || nk <= 0 || heads <= 0 || channels <= 0 || num_levels <= 0 || num_query <= 0 ||
      num_point <= 0 || points_per_group <= 0)
    return BEVOPS_BAD_PARAM;
  if (spatial_shapes_host) {
    long total = 0;
    for (int l = 0; l < num_levels; ++l) {
      const long H = spatial_shapes_host[2 * l], W = spatial_shapes_host[2 * l + 1];
      if (H <= 0 || W <= 0) return BEVOPS_BAD_PARAM;
      total += H * W;
    }
    if (total != nk) return BEVOPS_BAD_PARAM;
  }
  const MsdaDims d{bs, nk, heads, channels, num_levels, num_query, num_point, points_per_group,
                   shared_offsets ? 1 : 0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case BEVOPS_F32:
      if (ref_dtype != BEVOPS_F32) return BEVOPS_NOT_SUPPORTED;
      return msda_float<float>((const float *)value, spatial_shapes, (const float *)reference_points,
                               (const float *)sampling_offsets, (const float *)attention_weights,
                               (float *)output, d, st);
    case BEVOPS_F16:
      if (ref_dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
      // head-major path (msda_hm.hip) when the caller lends a workspace and the call is big
      // enough to amortise the re-layout; variants 10 (never) / 11 (hm forced) / 15 (hm2 forced)
      if (workspace && g_variant != 10 && g_variant != 99 && g_variant != 1 && g_variant != 2) {
        const bool pays = hm_pays(2, bs, nk, heads, channels, num_levels, num_query, num_point);
        const int LP = num_levels * num_point;
        // hm4 (software-pipelined, msda_hm4.hip): fp16 default where every pyramid level is
        // LDS-resident (tiny / small SCA: 106 vs 142 us at small SCA); for the base SCA call hm3 and
        // hm4 are level (571 vs 579 us kernel, profiles/r02) and hm3 stays.  Variant 17 forces hm4 for
        // every shape it supports, 16 hm3
        const bool staged_all = spatial_shapes_host && g_variant == 0 && num_query >= 8192 &&
                                msda_hm4_all_staged(spatial_shapes_host, bs, heads, channels, num_levels,
                                                    num_query, num_point);
        const bool h4 = g_variant == 17 || staged_all;
        if (spatial_shapes_host && h4) {
          const int rc = msda_hm4_forward(
              BEVOPS_F16, BEVOPS_F16, value, spatial_shapes_host, reference_points, sampling_offsets,
              attention_weights, output, bs, nk, heads, channels, num_levels, num_query, num_point,
              points_per_group, shared_offsets ? 1 : 0, 1.f, 1.f, 1.f, 1.f, workspace, workspace_bytes, 0, 0, st);
          if (rc != BEVOPS_NOT_SUPPORTED || g_variant != 0) return rc;
        }
        // hm5 (msda_hm5.hip): hm3's planes, re-scheduled, plus the exact visibility pre-pass; default
        // for the 4-level x 8-point SCA shape.  Variants 1000 + flags select its A/B builds
        if (spatial_shapes_host && (g_variant == 1000 || g_variant == 1001 || (g_variant == 0 && pays))) {
          const int rc = msda_hm5_forward_f16(
              (const __half *)value, spatial_shapes_host, (const __half *)reference_points,
              (const __half *)sampling_offsets, (const __half *)attention_weights, (__half *)output,
              bs, nk, heads, channels, num_levels, num_query, num_point, points_per_group,
              shared_offsets ? 1 : 0, workspace, workspace_bytes, g_variant >= 1000 ? g_variant - 1000 : 0, false, st);
          if (rc != BEVOPS_NOT_SUPPORTED || g_variant != 0) return rc;
        }
        if (spatial_shapes_host && (g_variant == 16 || (g_variant == 0 && pays && LP >= 16))) {
          const int rc = msda_hm3_forward_f16(
              (const __half *)value, spatial_shapes_host, (const __half *)reference_points,
              (const __half *)sampling_offsets, (const __half *)attention_weights, (__half *)output,
              bs, nk, heads, channels, num_levels, num_query, num_point, points_per_group,
              shared_offsets ? 1 : 0, workspace, workspace_bytes, st);
          if (rc != BEVOPS_NOT_SUPPORTED || g_variant == 16) return rc;
        }
        if (g_variant == 11 || g_variant == 15 || pays) {
          const int rc = msda_hm_forward_f16(
              (const __half *)value, spatial_shapes, spatial_shapes_host,
              (const __half *)reference_points, (const __half *)sampling_offsets,
              (const __half *)attention_weights, (__half *)output, bs, nk, heads, channels,
              num_levels, num_query, num_point, points_per_group, shared_offsets ? 1 : 0, workspace,
              workspace_bytes, g_variant, st);
          if (rc != BEVOPS_NOT_SUPPORTED) return rc;
        }
      }
      return msda_float<__half>((const __half *)value, spatial_shapes,
                                (const __half *)reference_points, (const __half *)sampling_offsets,
                                (const __half *)attention_weights, (__half *)output, d, st);
    case BEVOPS_I8:
      // supportsFormatCombination (multiScaleDeformableAttnPlugin.cpp:151-156)
      if (channels % 4 != 0 || num_point % 4 != 0) return BEVOPS_NOT_SUPPORTED;
      if (!(scale_value > 0.f) || !(scale_offset > 0.f) || !(scale_weight > 0.f) ||
          !(scale_out > 0.f))
        return BEVOPS_BAD_PARAM;
      // head-major int8 path (msda_hm4.hip) when the caller lends a workspace and the call is big
      // enough (variant 17 forces it, 10 / 99 keep the layout-preserving kernels)
      if (workspace && spatial_shapes_host && g_variant != 10 && g_variant != 99 &&
          (ref_dtype == BEVOPS_F32 || ref_dtype == BEVOPS_F16) &&
          (g_variant == 17 || hm_pays(1, bs, nk, heads, channels, num_levels, num_query, num_point))) {
        const int rc = msda_hm4_forward(BEVOPS_I8, ref_dtype, value, spatial_shapes_host, reference_points,
                                        sampling_offsets, attention_weights, output, bs, nk, heads, channels,
                                        num_levels, num_query, num_point, points_per_group,
                                        shared_offsets ? 1 : 0, scale_value, scale_offset, scale_weight,
                                        scale_out, workspace, workspace_bytes, 0, 0, st);
        if (rc != BEVOPS_NOT_SUPPORTED) return rc;
      }
      if (ref_dtype == BEVOPS_F32)
        return msda_int8<float, false>((const int8_t *)value, spatial_shapes,
                                       (const float *)reference_points,
                                       (const int8_t *)sampling_offsets,
                                       (const int8_t *)attention_weights, (int8_t *)output, d,
                                       scale_value, scale_offset, scale_weight, scale_out, st);
      if (ref_dtype == BEVOPS_F16)
        return msda_int8<__half, true>((const int8_t *)value, spatial_shapes,
                                       (const __half *)reference_points,
                                       (const int8_t *)sampling_offsets,
                                       (const int8_t *)attention_weights, (int8_t *)output, d,
                                       scale_value, scale_offset, scale_weight, scale_out, st);
      return BEVOPS_NOT_SUPPORTED;
    default:
      return BEVOPS_NOT_SUPPORTED;
  }
}
