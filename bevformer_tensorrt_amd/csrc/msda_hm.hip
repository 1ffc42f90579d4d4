// MSDA fp16, head-major path ("hm"): built on what profiles/r01 showed --
//   * a bilinear tap costs one 128-byte cache LINE whatever it uses (~2.4 clk/line/CU from
//     L2, 4-5x more beyond the XCD's 4 MiB L2), and on the reference layout
//     [bs, nk, heads, 32] the two x-corners of a sample are 512 B apart (2 lines) while a
//     camera's maps (15.8 MB at base) thrash the L2: 10 GB of fabric traffic per SCA call
//     for 0.59 GB of algorithmic bytes.
// So:
//   1. repack kernel: value -> VH [bs][heads][nkp][32] (head-major, level starts padded to
//      even pixels): the two x-corners of a sample become 128 contiguous bytes -- ONE line
//      when the pixel index is even, two half-lines otherwise (3 lines per sample on
//      average instead of 4);
//   2. work is ordered (batch, head)-major and blocks are XCD-remapped, so at any time an
//      XCD gathers from one or two (camera, head) planes (1.97 MB at base) -> L2-resident;
//   3. an OCTET of lanes owns one (b, q, h) item: lanes 0-3 take the x0 corner, lanes 4-7
//      the x1 corner (8 channels each), so one buffer_load_dwordx4 per bilinear ROW fetches
//      the 128-byte pair; the two quads of an octet run the quad algorithm of msda.hip
//      (points split 4 ways, DPP broadcasts) on their own corner and are summed at the end
//      with one DPP row_shl:4 per channel;
//   (LDS staging of the trailing levels with 1024- / 512-thread blocks and a two-copy layout were built in round 1,
//   measured slower, and are no longer in this file: profiles/r01c, design/msda.md.)
#include "msda_common.h"

namespace bevops {
namespace {

constexpr int kPixBytes = 64;       // 32 ch x fp16
constexpr int kTabBytes = 256;      // level table at the front of dynamic LDS

__host__ __device__ inline int hm_nkp(int nk, int L) { return (nk + L + 2) & ~1; }

// level table entry: {H, W, first pixel in VH (even), first pixel in the source}
__device__ __forceinline__ void build_levels(const int32_t *shapes, int L, int4 *tab) {
  int src = 0, dst = 0;
  for (int l = 0; l < L; ++l) {
    const int H = shapes[2 * l], W = shapes[2 * l + 1];
    tab[l] = make_int4(H, W, dst, src);
    src += H * W;
    dst += (H * W + 1) & ~1;
  }
  if (L < kMaxLevels) tab[L] = make_int4(0, 0, dst, src);  // end sentinel
}

// ---- 1. repack: [bs, nk, heads, 32] -> [bs, heads, nkp, 32]; pads written as zero -------
__global__ __launch_bounds__(256) void msda_hm_repack_kernel(const __half *__restrict__ value,
                                                             const int32_t *__restrict__ shapes,
                                                             __half *__restrict__ vh, int bs,
                                                             int nk, int heads, int L, int nkp) {
  __shared__ int4 tab[kMaxLevels + 1];
  if (threadIdx.x == 0) build_levels(shapes, L, tab);
  __syncthreads();
  // thread = (b, dst pixel p, head h, 16-byte chunk c): h, c fastest -> 512 B source rows
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(idx & 3);
  const int h = (int)((idx >> 2) % heads);
  const size_t r = (idx >> 2) / heads;
  const int p = (int)(r % nkp);
  const size_t b = r / nkp;
  if (b >= (size_t)bs) return;
  int src = -1;
  for (int l = 0; l < L; ++l) {
    const int4 t = tab[l];
    const int rel = p - t.z;
    if (rel >= 0 && rel < t.x * t.y) src = t.w + rel;
  }
  uint4 v = make_uint4(0, 0, 0, 0);
  if (src >= 0)
    v = *reinterpret_cast<const uint4 *>(value + (((size_t)b * nk + src) * heads + h) * 32 + c * 8);
  const size_t o = (((size_t)b * heads + h) * nkp + p) * 32 + c * 8;
  *reinterpret_cast<uint4 *>(vh + o) = v;
}

__device__ __forceinline__ void fma8(const u32x4 r, float w, float (&acc)[8]) {
  acc[0] = fmaf(w, h2f_lo(r.x), acc[0]); acc[1] = fmaf(w, h2f_hi(r.x), acc[1]);
  acc[2] = fmaf(w, h2f_lo(r.y), acc[2]); acc[3] = fmaf(w, h2f_hi(r.y), acc[3]);
  acc[4] = fmaf(w, h2f_lo(r.z), acc[4]); acc[5] = fmaf(w, h2f_hi(r.z), acc[5]);
  acc[6] = fmaf(w, h2f_lo(r.w), acc[6]); acc[7] = fmaf(w, h2f_hi(r.w), acc[7]);
}
__device__ __forceinline__ float row_shl4(float v) {  // lane i <- lane i+4 (within a 16-lane row)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0xf, true));
}

// ---- 2. main kernel ------------------------------------------------------------------
// PPL = points per quad lane (L*P/4), CH = own points prepared per pass.  Block = (batch*head, query chunk).
constexpr int kHmThreads = 256;
template <int PPL, int CH>
__global__ __launch_bounds__(kHmThreads) void msda_hm_kernel(
    const __half *__restrict__ vh, unsigned vh_bytes, const int32_t *__restrict__ shapes,
    const __half *__restrict__ ref, const __half *__restrict__ off,
    const __half *__restrict__ logit, __half *__restrict__ out, MsdaDims d, int nkp, int chunk,
    int nchunk) {
  constexpr int THREADS = kHmThreads;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int4 *lvl = reinterpret_cast<int4 *>(smem);
  if (threadIdx.x == 0) build_levels(shapes, d.L, lvl);
  __syncthreads();

  const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned bh = vb / (unsigned)nchunk, ck = vb - bh * (unsigned)nchunk;
  const unsigned b = bh / (unsigned)d.heads, h = bh - b * (unsigned)d.heads;
  const unsigned plane = bh * (unsigned)nkp * kPixBytes;  // byte offset of this (b, h) plane
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__half *>(vh), 0, vh_bytes, 0x00020000);

  constexpr int LP = 4 * PPL;
  constexpr int IPP = THREADS / 8;  // items per pass
  const unsigned lane8 = threadIdx.x & 7u;
  const unsigned sub = lane8 & 3u;   // channel group AND owner index inside the quad
  const unsigned side = lane8 >> 2;  // 0: x0 corner, 1: x1 corner
  const unsigned q_end = min((ck + 1u) * (unsigned)chunk, (unsigned)d.nq);

  for (unsigned q = ck * (unsigned)chunk + (threadIdx.x >> 3); q < q_end; q += IPP) {
    const size_t item = ((size_t)b * d.nq + q) * d.heads + h;
    const size_t in_item = ((size_t)(d.shared ? 0u : b) * d.nq + q) * d.heads + h;
    // all three operand streams of the item are requested before any math: round 2 found the reference
    // points loaded behind the softmax (a second exposed round trip per item) and, for the few-point calls
    // (TSA / decoder: 4 point steps), the taps issued and waited for one step at a time
    unsigned offraw[PPL], refraw[PPL];
    load_raw<PPL>(off + (in_item * LP + sub * PPL) * 2, offraw);
    const __half *refp = ref + ((size_t)b * d.nq + q) * (unsigned)d.ppg * 2u;
    {
      const int jj = (int)sub * PPL;
      int pp = jj - (jj / d.P) * d.P;
      int gg = pp % d.ppg;
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        refraw[k] = *reinterpret_cast<const unsigned *>(refp + 2 * gg);
        ++pp; ++gg;
        if (gg == d.ppg) gg = 0;
        if (pp == d.P) { pp = 0; gg = 0; }
      }
    }
    unsigned lgraw[(PPL + 1) / 2];
    {
      const __half *lp = logit + in_item * LP + sub * PPL;
      if constexpr (PPL == 1) lgraw[0] = *reinterpret_cast<const unsigned short *>(lp);
      else load_raw<PPL / 2>(lp, lgraw);
    }
    __builtin_amdgcn_sched_barrier(0);   // the three requests go out together, the math follows
    float e[PPL];
#pragma unroll
    for (int k = 0; k < PPL; ++k) e[k] = (k & 1) ? h2f_hi(lgraw[k / 2]) : h2f_lo(lgraw[k / 2]);
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
    // (this lane's 2*PPL offsets stay packed: one dword = the (x, y) pair of a point)

    int j0 = (int)sub * PPL;
    int l = j0 / d.P;
    int p = j0 - l * d.P;
    int g = p % d.ppg;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;

#pragma unroll
    for (int pass = 0; pass < PPL / CH; ++pass) {
      __builtin_amdgcn_sched_barrier(0);  // do not hoist later passes' point math up here
      float ow[CH][2];     // attention x bilinear weight of (row0, row1) on MY x-corner
      unsigned oo[CH][2];  // pixel index (row0, row1) of MY x-corner in the (b,h) plane
      bool any_valid = false;
#pragma unroll
      for (int kk = 0; kk < CH; ++kk) {
        const int k = pass * CH + kk;
        const int4 t = lvl[l];
        const int H = t.x, W = t.y;
        const float2 r = make_float2(h2f_lo(refraw[k]), h2f_hi(refraw[k]));
        float x = loc_im(r.x, (float)W, h2f_lo(offraw[k]));
        float y = loc_im(r.y, (float)H, h2f_hi(offraw[k]));
        const bool valid = (y > -1.f) && (x > -1.f) && (y < (float)H) && (x < (float)W);
        any_valid |= valid;
        if (!valid) { x = 0.f; y = 0.f; }   // non-finite locations must give 0, not NaN * 0
        const float xf = floorf(x), yf = floorf(y);
        const float lx = x - xf, ly = y - yf;
        const int x0 = (int)xf, y0 = (int)yf;
        // The octet reads pixels (xb, xb+1) of a row as one 128-byte pair: xb = x0 clamped
        // into [0, W-2], so the pair stays inside the row at the borders.  Each side takes
        // the weight of whichever true corner (x0 or x0+1) its column c coincides with.
        const int xb = min(max(x0, 0), max(W - 2, 0));
        const int c = xb + (int)side;
        const bool c_ok = valid && c <= W - 1;  // only false for side 1 of a 1-pixel-wide map
        const float wx = (c_ok && c == x0) ? (1.f - lx) : ((c_ok && c == x0 + 1) ? lx : 0.f);
        const float wxe = wx * e[k];
        ow[kk][0] = (y0 >= 0) ? (1.f - ly) * wxe : 0.f;
        ow[kk][1] = (y0 + 1 <= H - 1) ? ly * wxe : 0.f;
        const int cc = min(c, W - 1);
        const int y0c = min(max(y0, 0), H - 1), y1c = min(max(y0 + 1, 0), H - 1);
        oo[kk][0] = (unsigned)(t.z + y0c * W + cc);
        oo[kk][1] = (unsigned)(t.z + y1c * W + cc);
        ++p; ++g;
        if (g == d.ppg) g = 0;
        if (p == d.P) { p = 0; g = 0; ++l; }
      }
      if (!__any(any_valid)) continue;

#define BEVOPS_HM_SRC(S)                                                                     \
      _Pragma("unroll") for (int kk = 0; kk < CH; ++kk) {                                    \
        const float w0 = quad_bcast<S>(ow[kk][0]), w1 = quad_bcast<S>(ow[kk][1]);            \
        const unsigned p0 = quad_bcast<S>(oo[kk][0]), p1 = quad_bcast<S>(oo[kk][1]);         \
        const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(plane + p0 * kPixBytes + sub * 16u), 0, 0); \
        const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(plane + p1 * kPixBytes + sub * 16u), 0, 0); \
        fma8(r0, w0, acc);                                                                   \
        fma8(r1, w1, acc);                                                                   \
      }                                                                                      \
      /* keep at most CH points (2*CH loads) in flight per lane: the scheduler otherwise */  \
      /* hoists all 8*CH loads of the pass and spills under the 128-VGPR budget -- except */ \
      /* for the one-point-per-lane calls, whose 8 loads fit and should fly together */      \
      if constexpr (PPL > 1) __builtin_amdgcn_sched_barrier(0);
      BEVOPS_HM_SRC(0)
      BEVOPS_HM_SRC(1)
      BEVOPS_HM_SRC(2)
      BEVOPS_HM_SRC(3)
#undef BEVOPS_HM_SRC
    }
    const float inv = 1.0f / s;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = (acc[c] + row_shl4(acc[c])) * inv;
    if (side == 0) {
      uint4 v;
      v.x = pack_h2(acc[0], acc[1]); v.y = pack_h2(acc[2], acc[3]);
      v.z = pack_h2(acc[4], acc[5]); v.w = pack_h2(acc[6], acc[7]);
      *reinterpret_cast<uint4 *>(out + item * 32u + sub * 8u) = v;
    }
  }
}


// =======================================================================================
// hm2 -- second generation of the head-major path.  PMC on hm (profiles/r01) showed the
// kernel ~75 % VALU-bound (3.2e8 wave-instructions per base SCA call: 16 v_fma_mix + 4 DPP
// movs + 4 address ops + ~14 redundant owner-math ops per point step), so hm2 removes
// instructions as well as cache lines:
//   * layout: pixel PAIRS, channel-interleaved -- 128 B = 32 x half2(v[p][c], v[p+1][c]) --
//     in two copies (pairs starting on even / odd pixels), so every x-pair of a bilinear row
//     is ONE aligned line and one v_dot2c_f32_f16 does both x-corners of a channel
//     (8 dot2 per point instead of 16 fma_mix; weights ride as half2, accumulate fp32);
//   * all 8 lanes of an octet are equal (4 channels each); the L*P points are split 8 ways
//     (no redundant owner math) and handed over through a wave-private LDS mailbox
//     (one ds_write_b128 per point, one broadcast ds_read_b128 per point step) instead of
//     4 DPP movs.
// =======================================================================================
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
constexpr int kPairBytes = 128;

__host__ __device__ inline int hm2_npairs(int nk, int L) { return (hm_nkp(nk, L) >> 1) + 1; }

__device__ __forceinline__ int hm_lookup(const int4 *tab, int L, int p) {
  int src = -1;
  for (int l = 0; l < L; ++l) {
    const int4 t = tab[l];
    const int rel = p - t.z;
    if (rel >= 0 && rel < t.x * t.y) src = t.w + rel;
  }
  return src;
}

__global__ __launch_bounds__(256) void msda_hm2_repack_kernel(const __half *__restrict__ value,
                                                              const int32_t *__restrict__ shapes,
                                                              char *__restrict__ vh2, int bs, int nk,
                                                              int heads, int L, int npairs,
                                                              unsigned copy_b) {
  __shared__ int4 tab[kMaxLevels + 1];
  if (threadIdx.x == 0) build_levels(shapes, L, tab);
  __syncthreads();
  // thread = (b, pair k, copy, head h, 4-channel group c8): c8, h fastest
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c8 = (int)(idx & 7);
  const int h = (int)((idx >> 3) % heads);
  size_t r = (idx >> 3) / heads;
  const int copy = (int)(r & 1);
  r >>= 1;
  const int k = (int)(r % npairs);
  const size_t b = r / npairs;
  if (b >= (size_t)bs) return;
  const int f = 2 * k + copy;
  const int s0 = hm_lookup(tab, L, f), s1 = hm_lookup(tab, L, f + 1);
  uint2 a = make_uint2(0, 0), c = make_uint2(0, 0);
  if (s0 >= 0) a = *reinterpret_cast<const uint2 *>(value + (((size_t)b * nk + s0) * heads + h) * 32 + c8 * 4);
  if (s1 >= 0) c = *reinterpret_cast<const uint2 *>(value + (((size_t)b * nk + s1) * heads + h) * 32 + c8 * 4);
  uint4 o;
  o.x = (a.x & 0xffffu) | (c.x << 16);
  o.y = (a.x >> 16) | (c.x & 0xffff0000u);
  o.z = (a.y & 0xffffu) | (c.y << 16);
  o.w = (a.y >> 16) | (c.y & 0xffff0000u);
  *reinterpret_cast<uint4 *>(vh2 + (copy ? copy_b : 0u) +
                             (((size_t)b * heads + h) * npairs + k) * kPairBytes + c8 * 16) = o;
}

__device__ __forceinline__ float octet_max(float v) {
  v = quad_max(v);
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true)));
}
__device__ __forceinline__ float octet_sum(float v) {
  v = quad_sum(v);
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
}
__device__ __forceinline__ float dot2(unsigned pair, unsigned w, float acc) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, pair), __builtin_bit_cast(h2_t, w), acc, false);
}

template <int LP>
__global__ __launch_bounds__(256) void msda_hm2_kernel(
    const char *__restrict__ vh2, unsigned vh_bytes, const int32_t *__restrict__ shapes,
    const __half *__restrict__ ref, const __half *__restrict__ off,
    const __half *__restrict__ logit, __half *__restrict__ out, MsdaDims d, int npairs, int chunk,
    int nchunk, unsigned copy_b) {
  constexpr int NOWN = LP >= 8 ? 8 : LP;  // owner lanes per octet
  constexpr int PP = LP / NOWN;           // points per owner
  constexpr int kBox = LP * 16 + 16;      // mailbox bytes per octet (+16: bank spread)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int4 *lvl = reinterpret_cast<int4 *>(smem);
  if (threadIdx.x == 0) build_levels(shapes, d.L, lvl);
  __syncthreads();

  const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned bh = vb / (unsigned)nchunk, ck = vb - bh * (unsigned)nchunk;
  const unsigned b = bh / (unsigned)d.heads, h = bh - b * (unsigned)d.heads;
  const unsigned plane = bh * (unsigned)npairs * kPairBytes;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(vh2), 0, vh_bytes, 0x00020000);
  const unsigned lane8 = threadIdx.x & 7u;
  char *box = smem + kTabBytes + (threadIdx.x >> 3) * kBox;
  const unsigned q_end = min((ck + 1u) * (unsigned)chunk, (unsigned)d.nq);

  for (unsigned q = ck * (unsigned)chunk + (threadIdx.x >> 3); q < q_end; q += 32) {
    const size_t item = ((size_t)b * d.nq + q) * d.heads + h;
    const size_t in_item = ((size_t)(d.shared ? 0u : b) * d.nq + q) * d.heads + h;
    const bool owner = lane8 < (unsigned)NOWN;
    float e[PP];
    unsigned offraw[PP];
#pragma unroll
    for (int k = 0; k < PP; ++k) { e[k] = -INFINITY; offraw[k] = 0; }
    if (owner) {
      // offsets are read once; when an item's offsets fill whole 128-byte lines (LP >= 32)
      // stream them non-temporally so they do not evict the value maps from L2.  Narrower
      // rows share lines with the neighbouring heads' items -> keep those cacheable
      // (measured: nt on half-lines costs +50 % on the small-model SCA call).
      if (LP >= 32 && !d.shared) {  // camera-shared rows are re-read by the other cameras
        load_f_nt<PP>(logit + in_item * LP + lane8 * PP, e);
        load_raw_nt<PP>(off + (in_item * LP + lane8 * PP) * 2, offraw);
      } else {
        load_f<PP>(logit + in_item * LP + lane8 * PP, e);
        load_raw<PP>(off + (in_item * LP + lane8 * PP) * 2, offraw);
      }
    }
    float m = e[0];
#pragma unroll
    for (int k = 1; k < PP; ++k) m = fmaxf(m, e[k]);
    m = octet_max(m);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      e[k] = owner ? __expf(e[k] - m) : 0.f;
      s += e[k];
    }
    s = octet_sum(s);

    bool any_valid = false;
    if (owner) {
      const __half *refp = ref + ((size_t)b * d.nq + q) * (unsigned)d.ppg * 2u;
      const int j0 = (int)lane8 * PP;
      int l = j0 / d.P;
      int p = j0 - l * d.P;
      int g = p % d.ppg;
#pragma unroll
      for (int k = 0; k < PP; ++k) {
        const int4 t = lvl[l];
        const int H = t.x, W = t.y;
        const float2 r = load_ref(refp + 2 * g);
        float x = loc_im(r.x, (float)W, h2f_lo(offraw[k]));
        float y = loc_im(r.y, (float)H, h2f_hi(offraw[k]));
        const bool valid = (y > -1.f) && (x > -1.f) && (y < (float)H) && (x < (float)W);
        any_valid |= valid;
        if (!valid) { x = 0.f; y = 0.f; }   // non-finite locations must give 0, not NaN * 0
        const float xf = floorf(x), yf = floorf(y);
        const float lx = x - xf, ly = y - yf;
        const int x0 = (int)xf, y0 = (int)yf;
        // the row pair read is pixels (xb, xb+1), xb = x0 clamped into [0, W-2]
        const int xb = min(max(x0, 0), max(W - 2, 0));
        const float wa = (xb == x0) ? (1.f - lx) : ((xb == x0 + 1) ? lx : 0.f);
        const float wb = (xb + 1 > W - 1) ? 0.f : ((xb + 1 == x0) ? (1.f - lx) : ((xb + 1 == x0 + 1) ? lx : 0.f));
        const float ev = valid ? e[k] : 0.f;
        const float wr0 = (y0 >= 0) ? (1.f - ly) * ev : 0.f;
        const float wr1 = (y0 + 1 <= H - 1) ? ly * ev : 0.f;
        const int y0c = min(max(y0, 0), H - 1), y1c = min(max(y0 + 1, 0), H - 1);
        const unsigned f0 = (unsigned)(t.z + y0c * W + xb), f1 = (unsigned)(t.z + y1c * W + xb);
        uint4 pl;
        pl.x = pack_h2(wr0 * wa, wr0 * wb);
        pl.y = pack_h2(wr1 * wa, wr1 * wb);
        pl.z = plane + (f0 >> 1) * kPairBytes + ((f0 & 1u) ? copy_b : 0u);
        pl.w = plane + (f1 >> 1) * kPairBytes + ((f1 & 1u) ? copy_b : 0u);
        *reinterpret_cast<uint4 *>(box + (j0 + k) * 16) = pl;
        ++p; ++g;
        if (g == d.ppg) g = 0;
        if (p == d.P) { p = 0; g = 0; ++l; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // mailbox: same wave writes & reads
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (__any(any_valid)) {
#pragma unroll 8
      for (int j = 0; j < LP; ++j) {
        const uint4 pl = *reinterpret_cast<const uint4 *>(box + j * 16);
        const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(pl.z + lane8 * 16u), 0, 0);
        const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(pl.w + lane8 * 16u), 0, 0);
        acc[0] = dot2(r0.x, pl.x, acc[0]); acc[1] = dot2(r0.y, pl.x, acc[1]);
        acc[2] = dot2(r0.z, pl.x, acc[2]); acc[3] = dot2(r0.w, pl.x, acc[3]);
        acc[0] = dot2(r1.x, pl.y, acc[0]); acc[1] = dot2(r1.y, pl.y, acc[1]);
        acc[2] = dot2(r1.z, pl.y, acc[2]); acc[3] = dot2(r1.w, pl.y, acc[3]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float inv = 1.0f / s;
    uint2 v;
    v.x = pack_h2(acc[0] * inv, acc[1] * inv);
    v.y = pack_h2(acc[2] * inv, acc[3] * inv);
    if constexpr (LP >= 32)
      __builtin_nontemporal_store(((unsigned long long)v.y << 32) | v.x,
                                  reinterpret_cast<unsigned long long *>(out + item * 32u + lane8 * 4u));
    else
      *reinterpret_cast<uint2 *>(out + item * 32u + lane8 * 4u) = v;
  }
}

template <int LP>
int launch_hm2(const char *vh2, size_t vh_bytes, const int32_t *shapes, const __half *ref,
               const __half *off, const __half *logit, __half *out, const MsdaDims &d, int npairs,
               unsigned copy_b, hipStream_t st) {
  const int chunk = 128;
  const int nchunk = (d.nq + chunk - 1) / chunk;
  const size_t lds = kTabBytes + 32 * (LP * 16 + 16);
  hipLaunchKernelGGL((msda_hm2_kernel<LP>), dim3((unsigned)(d.bs * d.heads * nchunk)), dim3(256), lds,
                     st, vh2, (unsigned)vh_bytes, shapes, ref, off, logit, out, d, npairs, chunk, nchunk,
                     copy_b);
  return launch_status();
}

size_t hm2_bytes(int bs, int nk, int heads, int L) {
  return 2 * (size_t)bs * heads * hm2_npairs(nk, L) * kPairBytes;
}

template <int PPL, int CH>
int launch_hm(const __half *vh, size_t vh_bytes, const int32_t *shapes, const __half *ref,
              const __half *off, const __half *logit, __half *out, const MsdaDims &d, int nkp, hipStream_t st) {
  const int chunk = 128;
  const int nchunk = (d.nq + chunk - 1) / chunk;
  hipLaunchKernelGGL((msda_hm_kernel<PPL, CH>), dim3((unsigned)(d.bs * d.heads * nchunk)), dim3(kHmThreads), kTabBytes,
                     st, vh, (unsigned)vh_bytes, shapes, ref, off, logit, out, d, nkp, chunk, nchunk);
  return launch_status();
}

}  // namespace

size_t hm_copy_bytes(int bs, int nk, int heads, int L) {
  return (((size_t)bs * heads * hm_nkp(nk, L) * kPixBytes) + 127) & ~size_t(127);
}

size_t msda_hm_workspace_bytes(int bs, int nk, int heads, int C, int L) {
  if (C != 32 || L > kMaxLevels - 1) return 0;
  const size_t one = hm_copy_bytes(bs, nk, heads, L);
  const size_t v2 = hm2_bytes(bs, nk, heads, L);   // (hm2's two alignment copies: the larger of the two layouts)
  const size_t want = one > v2 ? one : v2;
  if (want < 0xFFFFFF00ull) return want;
  return one < 0xFFFFFF00ull ? one : 0;
}

int msda_hm_forward_f16(const __half *value, const int32_t *shapes, const int32_t *shapes_host,
                        const __half *ref, const __half *off, const __half *logit, __half *out,
                        int bs, int nk, int heads, int C, int L, int nq, int P, int ppg, int shared,
                        void *workspace, size_t workspace_bytes, int variant, hipStream_t st) {
  if (C != 32 || L > kMaxLevels - 1) return BEVOPS_NOT_SUPPORTED;
  const size_t one = hm_copy_bytes(bs, nk, heads, L);
  const int LP = L * P;
  // default: hm2 for the many-point calls (SCA: L*P >= 16), hm for the few-point ones (TSA:
  // only half of an hm2 octet would own a point, and its two-copy re-layout costs more than
  // it saves); variant 11 forces hm, 15 forces hm2
  if ((variant == 15 || LP >= 16) && variant != 11) {
    const size_t need2 = hm2_bytes(bs, nk, heads, L);
    const bool lp_ok = LP == 4 || LP == 8 || LP == 16 || LP == 32 || LP == 64;
    if (workspace && workspace_bytes >= need2 && need2 < 0xFFFFFF00ull && lp_ok &&
        !(reinterpret_cast<uintptr_t>(workspace) & 127u)) {
      const int npairs = hm2_npairs(nk, L);
      char *vh2 = static_cast<char *>(workspace);
      const unsigned copy_b = (unsigned)(need2 / 2);
      const size_t threads = (size_t)bs * npairs * 2 * heads * 8;
      hipLaunchKernelGGL(msda_hm2_repack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                         st, value, shapes, vh2, bs, nk, heads, L, npairs, copy_b);
      const MsdaDims d2{bs, nk, heads, C, L, nq, P, ppg, shared};
      switch (LP) {
        case 4: return launch_hm2<4>(vh2, need2, shapes, ref, off, logit, out, d2, npairs, copy_b, st);
        case 8: return launch_hm2<8>(vh2, need2, shapes, ref, off, logit, out, d2, npairs, copy_b, st);
        case 16: return launch_hm2<16>(vh2, need2, shapes, ref, off, logit, out, d2, npairs, copy_b, st);
        case 32: return launch_hm2<32>(vh2, need2, shapes, ref, off, logit, out, d2, npairs, copy_b, st);
        default: return launch_hm2<64>(vh2, need2, shapes, ref, off, logit, out, d2, npairs, copy_b, st);
      }
    }
    if (variant == 15) return BEVOPS_NOT_SUPPORTED;
  }
  if (!workspace || workspace_bytes < one || one >= 0xFFFFFF00ull || LP % 4 != 0 ||
      (reinterpret_cast<uintptr_t>(workspace) & 127u))
    return BEVOPS_NOT_SUPPORTED;
  const int nkp = hm_nkp(nk, L);
  __half *vh = static_cast<__half *>(workspace);
  {
    const size_t threads = (size_t)bs * nkp * heads * 4;
    hipLaunchKernelGGL(msda_hm_repack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,
                       value, shapes, vh, bs, nk, heads, L, nkp);
  }
  (void)shapes_host;
  const MsdaDims d{bs, nk, heads, C, L, nq, P, ppg, shared};
  switch (LP / 4) {
    case 1: return launch_hm<1, 1>(vh, one, shapes, ref, off, logit, out, d, nkp, st);
    case 2: return launch_hm<2, 2>(vh, one, shapes, ref, off, logit, out, d, nkp, st);
    case 4: return launch_hm<4, 2>(vh, one, shapes, ref, off, logit, out, d, nkp, st);
    case 8: return launch_hm<8, 2>(vh, one, shapes, ref, off, logit, out, d, nkp, st);
    case 16: return launch_hm<16, 2>(vh, one, shapes, ref, off, logit, out, d, nkp, st);
    default: return BEVOPS_NOT_SUPPORTED;
  }
}

}  // namespace bevops
