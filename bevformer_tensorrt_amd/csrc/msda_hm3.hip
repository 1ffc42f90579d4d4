// MSDA fp16, third head-major generation ("hm3").  PMC on hm2 (profiles/r01c + the level-split
// probe tools/msda_split_probe.py) showed two things: the kernel issues ~2.8 VALU
// wave-instructions per sample of which only 1.0 are the multiply-adds, and half of the base SCA
// call's samples land on the two small pyramid levels (29x50, 15x25) that are paid at the L2
// line rate although a (camera, head) plane of them is only 117 KB.  hm3 therefore
//   * pads every level with zeros (one zero row above and below, one zero pixel after each row
//     -- the pixel "before" a row is the pad of the previous row), so the per-corner border
//     logic of the bilinear footprint disappears: a sample is valid iff floor(x)+1 in [0, W] and
//     floor(y)+1 in [0, H], its four corners always exist, weights are (1-lx, lx) x (1-ly, ly);
//   * stores the big levels as one 128-byte entry PER PIXEL f = 32 x half2(v[f][c], v[f+1][c])
//     (the two alignment copies of hm2 interleaved), so the tap address is plane + f*128 with
//     no parity arithmetic and one v_dot2c_f32_f16 blends both x-corners of a channel;
//   * keeps the trailing levels that fit next to the mailboxes in LDS (row-major, 64 B per
//     pixel), staged once per block with a straight copy of a pre-padded plane; their taps are
//     ds_read2_b64 (x-pair = 128 contiguous bytes), blended in packed fp16 and accumulated in
//     fp32 -- the LDS pipe works in parallel with the L1/L2 path of the big levels;
//   * the level table is computed on the host and travels as a kernel argument (no per-block
//     serial global loads), blocks are long (one (camera, head) plane x a chunk of queries).
#include "msda_common.h"
#include "msda_pad.h"

namespace bevops {
namespace {

// ---- re-layout: [bs, nk, heads, 32] -> big set [bs][heads][g_entries][128 B] and staged set
// [bs][heads][s_entries][64 B].  thread = (b, entry, head, 16-byte chunk)
__global__ __launch_bounds__(256) void msda_hm3_repack_kernel(const __half *__restrict__ value,
                                                              char *__restrict__ gset,
                                                              char *__restrict__ sset, Hm3Tab t,
                                                              int bs, int nk, int heads) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n_big = (size_t)bs * t.g_entries * heads * 8;
  if (idx < n_big) {
    const int c8 = (int)(idx & 7);
    const int h = (int)((idx >> 3) % heads);
    const size_t r = (idx >> 3) / heads;
    const int f = (int)(r % t.g_entries);
    const size_t b = r / t.g_entries;
    const int s0 = hm3_source(t, 0, t.ls, f), s1 = hm3_source(t, 0, t.ls, f + 1);
    uint2 a = make_uint2(0, 0), c = make_uint2(0, 0);
    if (s0 >= 0) a = *reinterpret_cast<const uint2 *>(value + (((size_t)b * nk + s0) * heads + h) * 32 + c8 * 4);
    if (s1 >= 0) c = *reinterpret_cast<const uint2 *>(value + (((size_t)b * nk + s1) * heads + h) * 32 + c8 * 4);
    uint4 o;
    o.x = (a.x & 0xffffu) | (c.x << 16);
    o.y = (a.x >> 16) | (c.x & 0xffff0000u);
    o.z = (a.y & 0xffffu) | (c.y << 16);
    o.w = (a.y >> 16) | (c.y & 0xffff0000u);
    *reinterpret_cast<uint4 *>(gset + (((size_t)b * heads + h) * t.g_entries + f) * kEntBytes + c8 * 16) = o;
    return;
  }
  const size_t j = idx - n_big;
  const int c4 = (int)(j & 3);
  const int h = (int)((j >> 2) % heads);
  const size_t r = (j >> 2) / heads;
  if (t.s_entries == 0) return;
  const int f = (int)(r % t.s_entries);
  const size_t b = r / t.s_entries;
  if (b >= (size_t)bs) return;
  const int s0 = hm3_source(t, t.ls, t.L, f);
  uint4 v = make_uint4(0, 0, 0, 0);
  if (s0 >= 0) v = *reinterpret_cast<const uint4 *>(value + (((size_t)b * nk + s0) * heads + h) * 32 + c4 * 8);
  *reinterpret_cast<uint4 *>(sset + (((size_t)b * heads + h) * t.s_entries + f) * kLdsPixBytes + c4 * 16) = v;
}

// N points served by the L1/L2 path: per point two 128-byte entries (bilinear rows), 8 dot2.
// All 2N loads are issued before the first use (straight-line code on purpose: with a
// per-point branch the compiler serialises mailbox read -> load -> math for every point).
template <int N>
__device__ __forceinline__ void taps_big(const __amdgpu_buffer_rsrc_t rs, const char *box, unsigned lane16,
                                         float (&acc)[4]) {
  uint4 pl[N];
#pragma unroll
  for (int i = 0; i < N; ++i) pl[i] = *reinterpret_cast<const uint4 *>(box + i * 16);
  u32x4 r0[N], r1[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    r0[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(pl[i].z + lane16), 0, 0);
    r1[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(pl[i].w + lane16), 0, 0);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    acc[0] = dot2f(r0[i].x, pl[i].x, acc[0]); acc[1] = dot2f(r0[i].y, pl[i].x, acc[1]);
    acc[2] = dot2f(r0[i].z, pl[i].x, acc[2]); acc[3] = dot2f(r0[i].w, pl[i].x, acc[3]);
    acc[0] = dot2f(r1[i].x, pl[i].y, acc[0]); acc[1] = dot2f(r1[i].y, pl[i].y, acc[1]);
    acc[2] = dot2f(r1[i].z, pl[i].y, acc[2]); acc[3] = dot2f(r1[i].w, pl[i].y, acc[3]);
  }
}

// N points served from LDS: per row the x-pair is 128 contiguous bytes; this lane takes its 4
// channels of the left and of the right pixel (ds_read2_b64), blends the four corners in
// packed fp16 and adds the result to the fp32 accumulators
template <int N>
__device__ __forceinline__ void taps_lds(const char *smem, const char *box, unsigned lane8b,
                                         float (&acc)[4]) {
  uint4 pl[N];
#pragma unroll
  for (int i = 0; i < N; ++i) pl[i] = *reinterpret_cast<const uint4 *>(box + i * 16);
  uint2 l0[N], r0[N], l1[N], r1[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const char *p0 = smem + pl[i].z + lane8b;
    const char *p1 = smem + pl[i].w + lane8b;
    l0[i] = *reinterpret_cast<const uint2 *>(p0);
    r0[i] = *reinterpret_cast<const uint2 *>(p0 + kLdsPixBytes);
    l1[i] = *reinterpret_cast<const uint2 *>(p1);
    r1[i] = *reinterpret_cast<const uint2 *>(p1 + kLdsPixBytes);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const h2_t w0 = as_h2(pl[i].x), w1 = as_h2(pl[i].y);
    const h2_t w00 = {w0[0], w0[0]}, w01 = {w0[1], w0[1]}, w10 = {w1[0], w1[0]}, w11 = {w1[1], w1[1]};
    h2_t a = as_h2(l0[i].x) * w00, b = as_h2(l0[i].y) * w00;
    a = as_h2(r0[i].x) * w01 + a; b = as_h2(r0[i].y) * w01 + b;
    a = as_h2(l1[i].x) * w10 + a; b = as_h2(l1[i].y) * w10 + b;
    a = as_h2(r1[i].x) * w11 + a; b = as_h2(r1[i].y) * w11 + b;
    add_h2(acc[0], acc[1], a);
    add_h2(acc[2], acc[3], b);
  }
}

// MASKED (fused SCA, bevops_sca_forward): `qmask` [bs, nq] holds the camera-visibility weight of
// every (batch = camera, query) pair; pairs with weight 0 are skipped altogether (no loads, no
// math, no store) -- the block first compacts its chunk of queries to the visible ones, the
// way the original PyTorch SCA rebatches (third_party/.../spatial_cross_attention.py:143-191).
template <int LP, int THREADS, bool MASKED>
__global__ __launch_bounds__(THREADS) void msda_hm3_kernel(
    const char *__restrict__ gset, unsigned g_bytes, const char *__restrict__ sset,
    const __half *__restrict__ ref, const __half *__restrict__ off,
    const __half *__restrict__ logit, __half *__restrict__ out, MsdaDims d, Hm3Tab t, int chunk,
    int nchunk, int stage_bytes, const __half *__restrict__ qmask) {
  constexpr int NOWN = LP >= 8 ? 8 : LP;  // owner lanes per octet
  constexpr int PP = LP / NOWN;           // points per owner
  constexpr int MB = LP >= 8 ? 8 : LP;    // points per mailbox phase
  constexpr int NPH = LP / MB;            // phases
  constexpr int OPP = MB / PP;            // owner lanes per phase
  constexpr int kBox = MB * 16 + 16;      // mailbox bytes per octet (+16: bank spread)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // smem: [level table] [staged planes] [mailboxes]
  // block -> ((batch, head) plane, chunk of queries).  With 8 heads the dispatcher's round robin
  // (block i -> XCD i % 8) is used as is: XCD x keeps head x, and all 8 XCDs walk the same
  // (batch, chunk) sequence together -- the 8 heads' 128-byte pieces of one 1 KB offsets row (and
  // the two halves of a logits line) are then requested at about the same time instead of by
  // XCDs that are whole planes apart.  Otherwise: contiguous plane ranges per XCD.
  unsigned bh, ck;
  if (d.heads == 8) {
    const unsigned rest = blockIdx.x >> 3;
    bh = (rest / (unsigned)nchunk) * 8u + (blockIdx.x & 7u);
    ck = rest % (unsigned)nchunk;
  } else {
    const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
    bh = vb / (unsigned)nchunk;
    ck = vb - bh * (unsigned)nchunk;
  }
  const unsigned b = bh / (unsigned)d.heads, h = bh - b * (unsigned)d.heads;
  if (threadIdx.x < (unsigned)t.L) {
    const int l = threadIdx.x;
    const bool staged = l >= t.ls;
    const unsigned sh = staged ? 6u : 7u;
    const unsigned base = staged ? (unsigned)kTab : bh * (unsigned)t.g_entries * kEntBytes;
    float4 f;
    f.x = (float)t.W[l];
    f.y = (float)t.H[l];
    f.z = __uint_as_float(base + ((unsigned)t.ent0[l] << sh));
    f.w = __uint_as_float((unsigned)(t.W[l] + 1) << sh);
    *reinterpret_cast<float4 *>(smem + l * kTabEnt) = f;
    *reinterpret_cast<int2 *>(smem + l * kTabEnt + 16) = make_int2(t.W[l] + 1, (int)sh);
  }
  if (stage_bytes) {
    const uint4 *src = reinterpret_cast<const uint4 *>(sset + (size_t)bh * stage_bytes);
    uint4 *dst = reinterpret_cast<uint4 *>(smem + kTab);
    for (int i = threadIdx.x; i < stage_bytes / 16; i += THREADS) dst[i] = src[i];
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(gset), 0, g_bytes, 0x00020000);
  // streamed operands through buffer descriptors: 32-bit byte offsets that advance by a
  // constant per item instead of 64-bit pointer arithmetic (the host checks they fit)
  const unsigned n_in = (unsigned)(d.shared ? 1 : d.bs) * (unsigned)d.nq * (unsigned)d.heads * LP;
  const __amdgpu_buffer_rsrc_t rs_lg = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__half *>(logit), 0, n_in * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_of = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__half *>(off), 0, n_in * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_rf = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__half *>(ref), 0, (unsigned)d.bs * (unsigned)d.nq * (unsigned)d.ppg * 4u, 0x00020000);
  const unsigned lane8 = threadIdx.x & 7u;
  const unsigned lane16 = lane8 * 16u, lane8b = lane8 * 8u;
  char *box = smem + kTab + stage_bytes + (threadIdx.x >> 3) * kBox;
  const unsigned q_end = min((ck + 1u) * (unsigned)chunk, (unsigned)d.nq);
  const int js = t.ls * d.P;  // first point index served from LDS

  // per-lane constants of the owner's PP points: (level, reference-point group) do not depend
  // on the query
  const bool owner = lane8 < (unsigned)NOWN;
  unsigned lvo[PP], gof[PP];
  {
    const int j0 = (int)lane8 * PP;
    int l = j0 / d.P;
    int p = j0 - l * d.P;
    int g = p % d.ppg;
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      lvo[k] = owner ? (unsigned)l * kTabEnt : 0u;
      gof[k] = 4u * (unsigned)g;
      ++p; ++g;
      if (g == d.ppg) g = 0;
      if (p == d.P) { p = 0; g = 0; ++l; }
    }
  }
  // an owner's PP reference points are one contiguous run (BEVFormer: 4 pillar anchors, 4
  // points per owner) -> one load instead of PP
  const bool ref_run = PP > 1 && d.ppg == PP && d.P % PP == 0;
  // software pipeline: the logits / offsets / reference points of the NEXT item are requested
  // before the current one is processed
  constexpr int NLG = (PP + 1) / 2;
  struct Pre { unsigned lg[NLG], of[PP], rf[PP]; };
  const int aux = (LP >= 32 && !d.shared) ? 2 : 0;  // read-once full lines: non-temporal
  constexpr unsigned kStride = THREADS / 8;
  const unsigned q0 = ck * (unsigned)chunk;
  // items of this block: ordinal i -> query q0 + i, or q0 + list[i] after compaction
  unsigned n_items = q_end - q0;
  const unsigned short *qlist = reinterpret_cast<const unsigned short *>(
      smem + kTab + stage_bytes + (THREADS / 8) * kBox);
  if constexpr (MASKED) {
    unsigned short *wl = const_cast<unsigned short *>(qlist);
    unsigned *wtot = reinterpret_cast<unsigned *>(smem + kTab + stage_bytes + (THREADS / 8) * kBox + chunk * 2);
    unsigned base_count = 0;
    for (unsigned t0 = 0; t0 < n_items; t0 += THREADS) {
      const unsigned i = t0 + threadIdx.x;
      const bool vis = i < n_items && __half2float(qmask[(size_t)b * d.nq + q0 + i]) != 0.f;
      const unsigned long long bal = __ballot(vis);
      const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
      if (lane == 0) wtot[wv] = (unsigned)__popcll(bal);
      __syncthreads();
      unsigned before = base_count, all = 0;
      for (unsigned w2 = 0; w2 < THREADS / 64; ++w2) {
        const unsigned c = wtot[w2];
        if (w2 < wv) before += c;
        all += c;
      }
      if (vis) wl[before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)i;
      base_count += all;
      __syncthreads();
    }
    n_items = base_count;
  }
  auto query_of = [&](unsigned i) { return MASKED ? q0 + (unsigned)qlist[i] : q0 + i; };
  const unsigned b_in = d.shared ? 0u : b;
  const unsigned lg_base = ((b_in * (unsigned)d.nq * (unsigned)d.heads + h) * LP + lane8 * PP) * 2u;
  const unsigned lg_q = (unsigned)d.heads * LP * 2u;
  const unsigned rf_base = b * (unsigned)d.nq * (unsigned)d.ppg * 4u, rf_q = (unsigned)d.ppg * 4u;
  unsigned o_lg = 0, o_of = 0, o_rf = 0;
  auto locate = [&](unsigned q) {
    o_lg = lg_base + q * lg_q;
    o_of = 2u * o_lg;
    o_rf = rf_base + q * rf_q;
  };
  auto request = [&](Pre &r) {
#pragma unroll
    for (int k = 0; k < NLG; ++k) r.lg[k] = 0xfc00fc00u;  // -inf, -inf
#pragma unroll
    for (int k = 0; k < PP; ++k) { r.of[k] = 0; r.rf[k] = 0; }
    if (!owner) return;
    if constexpr (PP == 1) {
      r.lg[0] = 0xfc000000u | (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs_lg, (int)o_lg, 0, 0);
      r.of[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_of, (int)o_of, 0, 0);
    } else if constexpr (PP == 2) {
      r.lg[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_lg, (int)o_lg, 0, 0);
      const u32x2 v = aux ? __builtin_amdgcn_raw_buffer_load_b64(rs_of, (int)o_of, 0, 2)
                          : __builtin_amdgcn_raw_buffer_load_b64(rs_of, (int)o_of, 0, 0);
      r.of[0] = v.x; r.of[1] = v.y;
    } else if constexpr (PP == 4) {
      const u32x2 g = aux ? __builtin_amdgcn_raw_buffer_load_b64(rs_lg, (int)o_lg, 0, 2)
                          : __builtin_amdgcn_raw_buffer_load_b64(rs_lg, (int)o_lg, 0, 0);
      r.lg[0] = g.x; r.lg[1] = g.y;
      const u32x4 v = aux ? __builtin_amdgcn_raw_buffer_load_b128(rs_of, (int)o_of, 0, 2)
                          : __builtin_amdgcn_raw_buffer_load_b128(rs_of, (int)o_of, 0, 0);
      r.of[0] = v.x; r.of[1] = v.y; r.of[2] = v.z; r.of[3] = v.w;
    } else {
      static_assert(PP == 8, "PP");
      const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(rs_lg, (int)o_lg, 0, 2);
      r.lg[0] = g.x; r.lg[1] = g.y; r.lg[2] = g.z; r.lg[3] = g.w;
      const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs_of, (int)o_of, 0, 2);
      const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs_of, (int)o_of + 16, 0, 2);
      r.of[0] = v0.x; r.of[1] = v0.y; r.of[2] = v0.z; r.of[3] = v0.w;
      r.of[4] = v1.x; r.of[5] = v1.y; r.of[6] = v1.z; r.of[7] = v1.w;
    }
    bool done = false;
    if constexpr (PP == 4) {
      if (ref_run) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_rf, (int)o_rf, 0, 0);
        r.rf[0] = v.x; r.rf[1] = v.y; r.rf[2] = v.z; r.rf[3] = v.w;
        done = true;
      }
    }
    if (!done) {
#pragma unroll
      for (int k = 0; k < PP; ++k)
        r.rf[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_rf, (int)(o_rf + gof[k]), 0, 0);
    }
  };
  const int rot = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) % NPH;
  Pre cur;
  unsigned i = threadIdx.x >> 3;
  if (i < n_items) { locate(query_of(i)); request(cur); }
  for (; i < n_items; i += kStride) {
    const unsigned q = query_of(i);
    Pre nxt;
    if (i + kStride < n_items) { locate(query_of(i + kStride)); request(nxt); }
    __half *outp = out + (((size_t)b * d.nq + q) * d.heads + h) * 32u + lane8 * 4u;
    float e[PP];
#pragma unroll
    for (int k = 0; k < PP; ++k) e[k] = (k & 1) ? h2f_hi(cur.lg[k / 2]) : h2f_lo(cur.lg[k / 2]);
    float m = e[0];
#pragma unroll
    for (int k = 1; k < PP; ++k) m = fmaxf(m, e[k]);
    m = oct_max(m);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      e[k] = owner ? __expf(e[k] - m) : 0.f;
      s += e[k];
    }
    s = oct_sum(s);

    uint4 pl[PP];
    bool any_valid = false;
    if (owner) {
#pragma unroll
      for (int k = 0; k < PP; ++k) {
        const float4 tf = *reinterpret_cast<const float4 *>(smem + lvo[k]);
        const int2 ti = *reinterpret_cast<const int2 *>(smem + lvo[k] + 16);
        float x = fmaf(h2f_lo(cur.rf[k]), tf.x, h2f_lo(cur.of[k])) - 0.5f;
        float y = fmaf(h2f_hi(cur.rf[k]), tf.y, h2f_hi(cur.of[k])) - 0.5f;
        const bool valid = (y > -1.f) && (x > -1.f) && (y < tf.y) && (x < tf.x);
        // outside the range gate (incl. non-finite locations: reference points of pillars behind a
        // camera overflow binary16) -> exactly 0, never NaN * 0
        if (!valid) { x = 0.f; y = 0.f; }
        const float xf = floorf(x), yf = floorf(y);
        const float lx = x - xf, ly = y - yf;
        any_valid |= valid;
        const float ev = valid ? e[k] : 0.f;
        const float wr1 = ly * ev, wr0 = ev - wr1;
        const float b0 = wr0 * lx, b1 = wr1 * lx;
        pl[k].x = pack_h2(wr0 - b0, b0);
        pl[k].y = pack_h2(wr1 - b1, b1);
        // entry (yp, x0), yp = floor(y) + 1 in [0, H], x0 = floor(x) in [-1, W-1]
        const int rel = __mul24((int)yf + 1, ti.x) + (int)xf;
        pl[k].z = __float_as_uint(tf.z) + ((valid ? (unsigned)rel : 0u) << ti.y);
        pl[k].w = pl[k].z + __float_as_uint(tf.w);
      }
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pi = 0; pi < NPH; ++pi) {
      // waves of a block start together: rotate the phase order per wave so that they do not
      // all queue on the same pipe (L2 path / LDS) at the same time
      const int ph = (pi + rot) % NPH;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // mailbox: same wave writes & reads
      const bool mine = owner && (int)(lane8 / OPP) == ph;
      if (mine) {
#pragma unroll
        for (int k = 0; k < PP; ++k)
          *reinterpret_cast<uint4 *>(box + ((lane8 % OPP) * PP + k) * 16) = pl[k];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (!__any(mine && any_valid)) continue;
      const int jlo = ph * MB;
      constexpr int BT = MB < 4 ? MB : 4;  // points whose loads are in flight together
      if (jlo + MB <= js) {
#pragma unroll
        for (int j = 0; j < MB; j += BT) taps_big<BT>(rs, box + j * 16, lane16, acc);
      } else if (jlo >= js) {
#pragma unroll
        for (int j = 0; j < MB; j += BT) taps_lds<BT>(smem, box + j * 16, lane8b, acc);
      } else {
        for (int j = 0; j < MB; ++j) {
          if (jlo + j < js) taps_big<1>(rs, box + j * 16, lane16, acc);
          else taps_lds<1>(smem, box + j * 16, lane8b, acc);
        }
      }
    }
    const float inv = __builtin_amdgcn_rcpf(s);
    uint2 v;
    v.x = pack_h2(acc[0] * inv, acc[1] * inv);
    v.y = pack_h2(acc[2] * inv, acc[3] * inv);
    if constexpr (LP >= 32)
      __builtin_nontemporal_store(((unsigned long long)v.y << 32) | v.x,
                                  reinterpret_cast<unsigned long long *>(outp));
    else
      *reinterpret_cast<uint2 *>(outp) = v;
    cur = nxt;
  }
}

template <int LP>
int launch_hm3(const Hm3Plan &pl, const char *gset, const char *sset, const __half *ref,
               const __half *off, const __half *logit, __half *out, const MsdaDims &d,
               const __half *qmask, hipStream_t st) {
  constexpr int MB = LP >= 8 ? 8 : LP;
  const int octets = pl.threads / 8;
  // staged: long blocks (the plane copy is amortised over 10 items per octet); else hm2's
  const int chunk = pl.threads == 1024 ? 1280 : 128;
  const int nchunk = (d.nq + chunk - 1) / chunk;
  const size_t lds = kTab + pl.stage_bytes + (size_t)octets * (MB * 16 + 16) + (qmask ? chunk * 2 + 64 : 0);
  const dim3 grid((unsigned)(d.bs * d.heads * nchunk));
  const unsigned gb = (unsigned)pl.g_bytes;
  const int stage = pl.threads == 1024 ? pl.stage_bytes : 0;
#define BEVOPS_HM3_GO(THREADS, MASKED)                                                                   \
  do {                                                                                                   \
    if (!ensure_dynamic_lds<msda_hm3_kernel<LP, THREADS, MASKED>>(lds)) return (int)BEVOPS_FAILURE;      \
    hipLaunchKernelGGL((msda_hm3_kernel<LP, THREADS, MASKED>), grid, dim3(THREADS), lds, st, gset, gb,   \
                       sset, ref, off, logit, out, d, pl.t, chunk, nchunk, stage, qmask);                \
    return launch_status();                                                                              \
  } while (0)
  if (pl.threads == 1024) {
    if (qmask) BEVOPS_HM3_GO(1024, true);
    BEVOPS_HM3_GO(1024, false);
  }
  if (qmask) BEVOPS_HM3_GO(256, true);
  BEVOPS_HM3_GO(256, false);
#undef BEVOPS_HM3_GO
}

// fused SCA, second step: slots[q, :] = sum over cameras of mask[b, q] * sampled[b, q, :], reading
// only the visible (camera, query) pairs (the scratch rows of the others are never written).  thread = (query,
// 8-channel vector).  BS > 0: the camera loop is unrolled -- all masks are requested first, then all visible rows, and
// only then does the ascending-camera fma chain start (same order, same bits as the rolled loop, one memory round trip
// instead of one per visible camera); BS == 0: any camera count, rolled.  SKIP: a query that exactly one camera sees
// with weight exactly 1 is left alone -- the planned sampler has stored that row itself (msda_hm5.hip: kSoleBit; the
// same rule on the same mask).
template <int BS, bool SKIP>
__global__ __launch_bounds__(256) void sca_camera_reduce_kernel(const __half *__restrict__ sampled,
                                                                const __half *__restrict__ qmask,
                                                                __half *__restrict__ out, int bs, int nq,
                                                                int width) {
  const int vecs = width / 8;
  size_t q;
  int c;
  if constexpr (BS > 0) {   // the launcher sends only grids below 2^31 threads here: 32-bit index arithmetic
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    const unsigned q32 = idx / (unsigned)vecs;
    q = q32;
    c = (int)(idx - q32 * (unsigned)vecs);
  } else {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    q = idx / vecs;
    c = (int)(idx - q * vecs);
  }
  if (q >= (size_t)nq) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto add = [&](float m, const uint4 &v) {
    acc[0] = fmaf(m, h2f_lo(v.x), acc[0]); acc[1] = fmaf(m, h2f_hi(v.x), acc[1]);
    acc[2] = fmaf(m, h2f_lo(v.y), acc[2]); acc[3] = fmaf(m, h2f_hi(v.y), acc[3]);
    acc[4] = fmaf(m, h2f_lo(v.z), acc[4]); acc[5] = fmaf(m, h2f_hi(v.z), acc[5]);
    acc[6] = fmaf(m, h2f_lo(v.w), acc[6]); acc[7] = fmaf(m, h2f_hi(v.w), acc[7]);
  };
  if constexpr (BS > 0) {
    float m[BS];
    uint4 v[BS];
#pragma unroll
    for (int b = 0; b < BS; ++b) m[b] = __half2float(qmask[(size_t)b * nq + q]);
    if constexpr (SKIP) {
      int seen = 0;
      bool unit = false;
#pragma unroll
      for (int b = 0; b < BS; ++b)
        if (m[b] != 0.f) { ++seen; unit = m[b] == 1.f; }
      if (seen == 1 && unit) return;
    }
#pragma unroll
    for (int b = 0; b < BS; ++b) {
      v[b] = make_uint4(0u, 0u, 0u, 0u);
      if (m[b] != 0.f) v[b] = *reinterpret_cast<const uint4 *>(sampled + ((size_t)b * nq + q) * width + c * 8);
    }
#pragma unroll
    for (int b = 0; b < BS; ++b)
      if (m[b] != 0.f) add(m[b], v[b]);
  } else {
    if constexpr (SKIP) {
      int seen = 0;
      bool unit = false;
      for (int b = 0; b < bs; ++b) {
        const float m = __half2float(qmask[(size_t)b * nq + q]);
        if (m != 0.f) { ++seen; unit = m == 1.f; }
      }
      if (seen == 1 && unit) return;
    }
    for (int b = 0; b < bs; ++b) {
      const float m = __half2float(qmask[(size_t)b * nq + q]);
      if (m == 0.f) continue;
      add(m, *reinterpret_cast<const uint4 *>(sampled + ((size_t)b * nq + q) * width + c * 8));
    }
  }
  uint4 o;
  o.x = pack_h2(acc[0], acc[1]); o.y = pack_h2(acc[2], acc[3]);
  o.z = pack_h2(acc[4], acc[5]); o.w = pack_h2(acc[6], acc[7]);
  *reinterpret_cast<uint4 *>(out + q * width + c * 8) = o;
}

}  // namespace

// fp16 re-layout into the padded sets (shared with msda_hm4.hip, whose fp16 planes have the same
// layout; `tab` points at an Hm3Tab -- passed untyped because the struct lives in each
// translation unit's unnamed namespace)
void msda_hm3_repack_launch(const void *value, char *gset, char *sset, const void *tab, int bs, int nk, int heads,
                            hipStream_t st) {
  const Hm3Tab &t = *static_cast<const Hm3Tab *>(tab);
  const size_t threads = (size_t)bs * t.g_entries * heads * 8 + (size_t)bs * t.s_entries * heads * 4;
  hipLaunchKernelGGL(msda_hm3_repack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,
                     static_cast<const __half *>(value), gset, sset, t, bs, nk, heads);
}

static thread_local bool g_reduce_rolled = false;
void msda_sca_set_reduce_rolled(bool on) { g_reduce_rolled = on; }

void msda_sca_reduce_launch(const __half *sampled, const __half *qmask, __half *out, int bs, int nq, int width,
                            bool skip_sole, hipStream_t st) {
  const size_t threads = (size_t)nq * (width / 8);
  const dim3 grid((unsigned)((threads + 255) / 256));
  const int unrolled = (threads < (1ull << 31) && !g_reduce_rolled && (bs <= 3 || bs == 6)) ? bs : 0;
#define BEVOPS_REDUCE_GO(BS)                                                                                         \
  do {                                                                                                               \
    if (skip_sole)                                                                                                   \
      hipLaunchKernelGGL((sca_camera_reduce_kernel<BS, true>), grid, dim3(256), 0, st, sampled, qmask, out, bs, nq,  \
                         width);                                                                                     \
    else                                                                                                             \
      hipLaunchKernelGGL((sca_camera_reduce_kernel<BS, false>), grid, dim3(256), 0, st, sampled, qmask, out, bs, nq, \
                         width);                                                                                     \
  } while (0)
  switch (unrolled) {   // 6 = the rig; 1 .. 3 = a rank's cameras of the camera-sharded frame
    case 1: BEVOPS_REDUCE_GO(1); break;
    case 2: BEVOPS_REDUCE_GO(2); break;
    case 3: BEVOPS_REDUCE_GO(3); break;
    case 6: BEVOPS_REDUCE_GO(6); break;
    default: BEVOPS_REDUCE_GO(0); break;
  }
#undef BEVOPS_REDUCE_GO
}

size_t msda_hm3_workspace_bytes(const int32_t *shapes_host, int bs, int heads, int C, int L, int nq,
                                int P) {
  Hm3Plan pl;
  if (C != 32 || !hm3_plan(shapes_host, bs, heads, L, nq, hm3_box_bytes(L * P), pl)) return 0;
  return pl.g_bytes + 128 + pl.s_bytes;
}

int msda_hm3_forward_f16(const __half *value, const int32_t *shapes_host, const __half *ref,
                         const __half *off, const __half *logit, __half *out, int bs, int nk,
                         int heads, int C, int L, int nq, int P, int ppg, int shared,
                         void *workspace, size_t workspace_bytes, hipStream_t st) {
  const int LP = L * P;
  const bool lp_ok = LP == 4 || LP == 8 || LP == 16 || LP == 32 || LP == 64;
  Hm3Plan pl;
  if (C != 32 || !lp_ok || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 127u) ||
      !hm3_plan(shapes_host, bs, heads, L, nq, hm3_box_bytes(LP), pl))
    return BEVOPS_NOT_SUPPORTED;
  if ((double)bs * nq * heads * LP * 4.0 >= 4294967040.0) return BEVOPS_NOT_SUPPORTED;  // 32-bit offsets
  const size_t g_room = (pl.g_bytes + 127) & ~size_t(127);
  if (workspace_bytes < g_room + pl.s_bytes) return BEVOPS_NOT_SUPPORTED;
  char *gset = static_cast<char *>(workspace);
  char *sset = gset + g_room;
  {
    const size_t threads = (size_t)bs * pl.t.g_entries * heads * 8 + (size_t)bs * pl.t.s_entries * heads * 4;
    hipLaunchKernelGGL(msda_hm3_repack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,
                       value, gset, sset, pl.t, bs, nk, heads);
  }
  const MsdaDims d{bs, nk, heads, C, L, nq, P, ppg, shared};
  switch (LP) {
    case 4: return launch_hm3<4>(pl, gset, sset, ref, off, logit, out, d, nullptr, st);
    case 8: return launch_hm3<8>(pl, gset, sset, ref, off, logit, out, d, nullptr, st);
    case 16: return launch_hm3<16>(pl, gset, sset, ref, off, logit, out, d, nullptr, st);
    case 32: return launch_hm3<32>(pl, gset, sset, ref, off, logit, out, d, nullptr, st);
    default: return launch_hm3<64>(pl, gset, sset, ref, off, logit, out, d, nullptr, st);
  }
}

// ---- fused SCA: camera-shared offsets / logits, visibility-masked sampling, masked
// camera sum.  workspace = [big set][staged set][sampled: bs * nq * heads * 32 fp16]
size_t msda_hm3_sca_workspace_bytes(const int32_t *shapes_host, int bs, int heads, int C, int L, int nq,
                                    int P) {
  const size_t a = msda_hm3_workspace_bytes(shapes_host, bs, heads, C, L, nq, P);
  if (a == 0) return 0;
  return ((a + 255) & ~size_t(255)) + (size_t)bs * nq * heads * C * sizeof(__half);
}

int msda_hm3_sca_forward_f16(const __half *value, const int32_t *shapes_host, const __half *ref,
                             const __half *off, const __half *logit, const __half *qmask,
                             __half *out, int bs, int nk, int heads, int C, int L, int nq, int P,
                             int ppg, void *workspace, size_t workspace_bytes, hipStream_t st) {
  const int LP = L * P;
  const bool lp_ok = LP == 4 || LP == 8 || LP == 16 || LP == 32 || LP == 64;
  Hm3Plan pl;
  if (C != 32 || !lp_ok || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 127u) ||
      !hm3_plan(shapes_host, bs, heads, L, nq, hm3_box_bytes(LP), pl))
    return BEVOPS_NOT_SUPPORTED;
  if ((double)bs * nq * heads * LP * 4.0 >= 4294967040.0) return BEVOPS_NOT_SUPPORTED;
  const size_t need = msda_hm3_sca_workspace_bytes(shapes_host, bs, heads, C, L, nq, P);
  if (workspace_bytes < need) return BEVOPS_BAD_PARAM;
  const size_t g_room = (pl.g_bytes + 127) & ~size_t(127);
  char *gset = static_cast<char *>(workspace);
  char *sset = gset + g_room;
  __half *sampled = reinterpret_cast<__half *>(
      gset + ((msda_hm3_workspace_bytes(shapes_host, bs, heads, C, L, nq, P) + 255) & ~size_t(255)));
  {
    const size_t threads = (size_t)bs * pl.t.g_entries * heads * 8 + (size_t)bs * pl.t.s_entries * heads * 4;
    hipLaunchKernelGGL(msda_hm3_repack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,
                       value, gset, sset, pl.t, bs, nk, heads);
  }
  const MsdaDims d{bs, nk, heads, C, L, nq, P, ppg, 1};
  int rc;
  switch (LP) {
    case 4: rc = launch_hm3<4>(pl, gset, sset, ref, off, logit, sampled, d, qmask, st); break;
    case 8: rc = launch_hm3<8>(pl, gset, sset, ref, off, logit, sampled, d, qmask, st); break;
    case 16: rc = launch_hm3<16>(pl, gset, sset, ref, off, logit, sampled, d, qmask, st); break;
    case 32: rc = launch_hm3<32>(pl, gset, sset, ref, off, logit, sampled, d, qmask, st); break;
    default: rc = launch_hm3<64>(pl, gset, sset, ref, off, logit, sampled, d, qmask, st); break;
  }
  if (rc != BEVOPS_SUCCESS) return rc;
  msda_sca_reduce_launch(sampled, qmask, out, bs, nq, heads * C, false, st);
  return launch_status();
}

}  // namespace bevops
