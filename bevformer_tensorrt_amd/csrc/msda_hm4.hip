// MSDA, fourth head-major generation ("hm4"): fp16 and int8 on ONE software-pipelined skeleton.
//
// What hm3 left on the table (profiles/r01d-f): its three pipes -- the per-item front
// end (VALU), the LDS taps of the staged levels and the L1/L2 taps of the big levels -- ran one after
// the other inside every wave (front end -> 2 phases of L2 taps -> 2 phases of LDS taps, each
// waiting for its own loads), 16 waves per CU all in the same order, so the call took about the SUM
// of the pipes (550 us) where the slowest one alone (the L2 line rate, 2 lines per big-level sample)
// is 240 us.  hm4 keeps hm3's data layout idea (msda_pad.h) and changes the schedule:
//   * 512-thread blocks = 8 waves per CU with a 256-register budget each (the staged planes still
//     allow one block per CU): all L*P point records of an item fit one mailbox, and a wave keeps
//     TWO batches of big-level taps (16 x 16-byte loads per lane) in flight;
//   * one straight-line loop body per group of 8 items: issue big batch 0 and 1 -> front end of the
//     NEXT group (softmax, locations, corner weights) -> LDS batch -> consume big 0 -> issue big 2
//     -> LDS batch -> ... : the L2 round trips hide behind the wave's own VALU / LDS work instead
//     of behind other waves that are in the same phase;
//   * streamed operands (logits, offsets, reference points) are requested TWO groups ahead.
// int8 (the reference's INT8 plugin flavours, multiScaleDeformableAttnKernel.cu:848-1104) rides the
// same skeleton with its own entries:
//   * big levels: one 128-byte entry per pixel f = the whole 2x2 footprint (f, f+1, f+W', f+W'+1)
//     of all 32 channels, bytes pre-transposed so that a lane's 16-byte load IS four v_dot4
//     operands (channel c: v00, v01, v10, v11) -- ONE cache line per sample instead of two, no
//     v_perm transposes;
//   * staged levels: 64-byte pixel-pair entries (channel-interleaved bytes of pixels f and f+1);
//   * per channel and sample the reference computes T2int8(tsum / 127) (round half away; ties
//     cannot occur because 127 and 255 are odd) -- evaluated exactly in integers as the top byte
//     of a saturating v_mad_i32_i24(tsum, round(2^24 / 127), 2^23): 1 instruction instead of 6;
//     the four samples of a batch are gathered with v_perm_b32 and accumulated with ONE
//     v_dot4_i32_i8 against their packed softmax weights, as the reference's own dp4a does.
// Results: fp16 within 1e-2 of the oracle (same arithmetic as hm3); int8 bit-identical to the
// layout-preserving int8 kernels of msda.hip (tests/test_msda_hm4_gpu.py).
#include "msda_common.h"
#include "msda_pad.h"

namespace bevops {
namespace {

// ------------------------------------------------------------------------------------------------
// re-layouts
// ------------------------------------------------------------------------------------------------
// Re-layouts.  fp16 planes have hm3's layout and use its kernel (msda_hm3_repack_launch).  int8:
// grid x = slabs of 32 entries (big set first, then staged set), y = (batch, head) plane; 256
// threads = 32 entries x 8 chunks of 4 channels; one division per thread for the padded row
// (floating point, exact -- `row_of`).  (Measured alternatives, profiles/r02: one thread per
// (entry, head, chunk) with the head fastest 97 us, a 1-D grid with integer divisions 108 us, this
// mapping 78 us for the base SCA planes.)
__device__ __forceinline__ int row_of(int rel, int wp) {
  // floor(rel / wp) for 0 <= rel < 2^22: (rel + 0.5) / wp is never closer than 0.5 / wp to an integer
  return (int)(((float)rel + 0.5f) / (float)wp);
}
struct RepackPos {
  bool big;
  int f, lv, yp, x;   // entry, level (-1: none), padded row, column
};
__device__ __forceinline__ RepackPos repack_pos(const Hm3Tab &t, int slab, int per_slab, int lane_entry) {
  RepackPos p;
  const int big_slabs = (t.g_entries + per_slab - 1) / per_slab;
  p.big = slab < big_slabs;
  p.f = (p.big ? slab : slab - big_slabs) * per_slab + lane_entry;
  const int entries = p.big ? t.g_entries : t.s_entries;
  p.lv = -1;
  p.yp = p.x = 0;
  if (p.f >= entries) { p.f = -1; return p; }
  const int l0 = p.big ? 0 : t.ls, l1 = p.big ? t.ls : t.L;
  for (int l = l0; l < l1; ++l)
    if (p.f >= t.ent0[l] - 1) p.lv = l;   // the entry before a level's first one is a base too
  if (p.lv >= 0) {
    const int wp = t.W[p.lv] + 1, rel = p.f - t.ent0[p.lv];
    p.yp = rel < 0 ? -1 : row_of(rel, wp);
    p.x = rel - p.yp * wp;
    if (p.yp > t.H[p.lv] + 1) p.lv = -1;  // trailing entry of the set
  }
  return p;
}

// int8: big set = 2x2-footprint entries (dword c of 16-byte chunk k = channel 4k+c of pixels
// f, f+1, f+W', f+W'+1); staged set = pixel-pair entries (bytes c(x0), c(x1) interleaved)
__global__ __launch_bounds__(256) void msda_hm4_repack_i8_kernel(const int8_t *__restrict__ value,
                                                                 char *__restrict__ gset,
                                                                 char *__restrict__ sset, Hm3Tab t,
                                                                 int nk, int heads, unsigned bias) {
  // `bias` = 0x80808080 for the x255 flavour: its planes hold v + 128 as u8 (pads included: they
  // stand for the value 0), see i8_sample_u.
  const int c8 = threadIdx.x & 7;
  const RepackPos p = repack_pos(t, blockIdx.x, 32, (int)(threadIdx.x >> 3));
  if (p.f < 0) return;
  const unsigned bh = blockIdx.y, b = bh / (unsigned)heads, h = bh - b * (unsigned)heads;
  unsigned px[4] = {0u, 0u, 0u, 0u};  // pixels f, f+1, f+W', f+W'+1 (4 channels each)
  if (p.lv >= 0) {
    const int W = t.W[p.lv], H = t.H[p.lv], wp = W + 1;
    const int8_t *base = value + (((size_t)b * nk + t.src0[p.lv]) * heads + h) * 32 + c8 * 4;
    auto at = [&](int yy, int xx) -> unsigned {   // padded (row, column) -> 4 channels or zeros
      if (xx >= wp) { xx -= wp; ++yy; }
      if (yy < 1 || yy > H || xx >= W) return 0u;
      return *reinterpret_cast<const unsigned *>(base + ((size_t)(yy - 1) * W + xx) * heads * 32);
    };
    px[0] = at(p.yp, p.x);
    px[1] = at(p.yp, p.x + 1);
    if (p.big) { px[2] = at(p.yp + 1, p.x); px[3] = at(p.yp + 1, p.x + 1); }
  }
  if (p.big) {
    unsigned o[4];
    transpose4x4(px[0], px[1], px[2], px[3], o);
    *reinterpret_cast<uint4 *>(gset + ((size_t)bh * t.g_entries + p.f) * kEntBytes + c8 * 16) =
        make_uint4(o[0] ^ bias, o[1] ^ bias, o[2] ^ bias, o[3] ^ bias);
  } else {
    uint2 o;
    o.x = __builtin_amdgcn_perm(px[1], px[0], 0x05010400u) ^ bias;  // c0(x0), c0(x1), c1(x0), c1(x1)
    o.y = __builtin_amdgcn_perm(px[1], px[0], 0x07030602u) ^ bias;  // c2(x0), c2(x1), c3(x0), c3(x1)
    *reinterpret_cast<uint2 *>(sset + ((size_t)bh * t.s_entries + p.f) * kLdsPixBytes + c8 * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
struct H4Args {
  const char *gset;
  unsigned g_bytes;
  const char *sset;
  const void *ref, *off, *logit;
  void *out;
  MsdaDims d;
  Hm3Tab t;
  int chunk, nchunk, stage_bytes;
  const __half *qmask;
  float s_v, s_o, s_w, s_out;
};

// One sample, four channels: tsum[c] = dot4(corners of channel c, area weights), then the exact
// T2int8(tsum / DIV) for DIV in {127, 255}: no ties (DIV odd), so round-half-away == round-half-even
// == floor(tsum / DIV + 0.5); with M = round(2^24 / DIV) the product error stays 60x below the
// 1 / (2 DIV) margin for |tsum| <= 2^15, and the mad SATURATES exactly where the reference clamps
// (|tsum| / DIV >= 127.5 resp. <= -128.5 lands beyond int32): the result sits in bits 31..24.
// Hand-placed because gfx950 needs 3 wait states between a DOT write and a different VALU opcode
// reading it, and the compiler's hazard recogniser does not look inside an asm statement: the four
// dots run first, then the four mads (a mad directly behind its dot read a stale register).
__device__ __forceinline__ void i8_sample_s(const unsigned (&v)[4], unsigned aw, int magic, int half,
                                            int (&x)[4]) {
  asm("v_dot4_i32_i8 %0, %4, %8, 0\n\t"
      "v_dot4_i32_i8 %1, %5, %8, 0\n\t"
      "v_dot4_i32_i8 %2, %6, %8, 0\n\t"
      "v_dot4_i32_i8 %3, %7, %8, 0\n\t"
      "v_mad_i32_i24 %0, %0, %9, %10 clamp\n\t"
      "v_mad_i32_i24 %1, %1, %9, %10 clamp\n\t"
      "v_mad_i32_i24 %2, %2, %9, %10 clamp\n\t"
      "v_mad_i32_i24 %3, %3, %9, %10 clamp"
      : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(aw), "v"(magic), "v"(half));
}
// unsigned x255 area weights (gfx950 has no mixed-sign dot4): the x255 flavour's planes hold the
// values BIASED by +128 (u8), so sum (v + 128) a - 128 sum(a) is ONE unsigned dot4 whose addend
// `neg` = -128 (a0 + a1 + a2 + a3) rides in the record
__device__ __forceinline__ void i8_sample_u(const unsigned (&v)[4], unsigned aw, int neg, int magic, int half,
                                            int (&x)[4]) {
  asm("v_dot4_u32_u8 %0, %4, %8, %9\n\t"
      "v_dot4_u32_u8 %1, %5, %8, %9\n\t"
      "v_dot4_u32_u8 %2, %6, %8, %9\n\t"
      "v_dot4_u32_u8 %3, %7, %8, %9\n\t"
      "v_mad_i32_i24 %0, %0, %10, %11 clamp\n\t"
      "v_mad_i32_i24 %1, %1, %10, %11 clamp\n\t"
      "v_mad_i32_i24 %2, %2, %10, %11 clamp\n\t"
      "v_mad_i32_i24 %3, %3, %10, %11 clamp"
      : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(aw), "v"(neg), "v"(magic), "v"(half));
}
// the four samples' top bytes -> one dword (sample 0 in byte 0)
__device__ __forceinline__ int gather_hi(int x0, int x1, int x2, int x3) {
  const unsigned a = __builtin_amdgcn_perm((unsigned)x1, (unsigned)x0, 0x00000703u);
  const unsigned b = __builtin_amdgcn_perm((unsigned)x3, (unsigned)x2, 0x00000703u);
  return (int)__builtin_amdgcn_perm(b, a, 0x05040100u);
}

// I8: int8 tensors (U8W: the <__half2> flavour with unsigned x255 weights and RNE rounding, else the
// <float> flavour with signed x127 weights); RefT: reference point type (fp16 path: __half).
// NBIG: batches (of BT points) served by the L1/L2 path, the remaining ones come from LDS.
// RR: the PP reference points of an owner lane are one contiguous run (BEVFormer SCA: 4 anchors).
// SCHED: schedule bits of the production builds (results identical whatever they are): 32 operand request at the END
// of the loop body (fp16), 64 one big batch in flight instead of two, 128 default cache policy for the streamed
// operands / output (int8), 1024 compiled for 4 waves per SIMD (<= 128 VGPRs) so that TWO 512-thread blocks share a
// CU when the staged planes are small (the int8 two-blocks plan).  The timing ablations, the "request behind the first
// batches" schedule and the int8 pixel-pair entry format of rounds 2 / 4 are no longer in this file (history;
// profiles/r02/hm4_ablation.jsonl, hm4_schedule_variants.jsonl, profiles/r04/msda_i8_pair_ab.jsonl).
template <int LP, int NBIG, int THREADS, bool I8, bool U8W, typename RefT, bool MASKED, bool RR, int SCHED = 0>
__global__ __launch_bounds__(THREADS, (SCHED & 1024) ? 4 : 1) void msda_hm4_kernel(const H4Args a) {
  constexpr int NOWN = LP >= 8 ? 8 : LP;  // owner lanes per octet
  constexpr int PP = LP / NOWN;           // points per owner
  constexpr int BT = LP >= 4 ? 4 : LP;    // points per tap batch
  constexpr int NB = LP / BT;
  constexpr int NLDS = NB - NBIG;
  constexpr int D = (NBIG >= 2 && !(SCHED & 64)) ? 2 : 1;    // big batches in flight
  constexpr int kBox = LP * 16 + 16;      // mailbox bytes per octet (+16: bank spread)
  constexpr int ESZ = I8 ? 1 : 2;         // bytes per logit / offset component
  constexpr unsigned kBigEnt = (unsigned)kEntBytes;
  static_assert(NBIG >= 0 && NBIG <= NB, "NBIG");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const MsdaDims &d = a.d;
  const Hm3Tab &t = a.t;
  // smem: [level table] [staged planes] [mailboxes] [compaction list]
  unsigned bh, ck;
  if (d.heads == 8) {  // XCD x keeps head x; all XCDs walk the same (batch, chunk) sequence
    const unsigned rest = blockIdx.x >> 3;
    bh = (rest / (unsigned)a.nchunk) * 8u + (blockIdx.x & 7u);
    ck = rest % (unsigned)a.nchunk;
  } else {
    const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
    bh = vb / (unsigned)a.nchunk;
    ck = vb - bh * (unsigned)a.nchunk;
  }
  const unsigned b = bh / (unsigned)d.heads, h = bh - b * (unsigned)d.heads;
  if (threadIdx.x < (unsigned)t.L) {
    const int l = threadIdx.x;
    const bool staged = l >= t.ls;
    const unsigned sh = staged ? 6u : 7u;
    const unsigned base = staged ? (unsigned)kTab : bh * (unsigned)t.g_entries * kBigEnt;
    float4 f;
    f.x = (float)t.W[l];
    f.y = (float)t.H[l];
    f.z = __uint_as_float(base + ((unsigned)t.ent0[l] << sh));
    f.w = __uint_as_float((unsigned)(t.W[l] + 1) << sh);
    *reinterpret_cast<float4 *>(smem + l * kTabEnt) = f;
    *reinterpret_cast<int2 *>(smem + l * kTabEnt + 16) = make_int2(t.W[l] + 1, (int)sh);
  }
  if (a.stage_bytes) {
    const uint4 *src = reinterpret_cast<const uint4 *>(a.sset + (size_t)bh * a.stage_bytes);
    uint4 *dst = reinterpret_cast<uint4 *>(smem + kTab);
    for (int i = threadIdx.x; i < a.stage_bytes / 16; i += THREADS) dst[i] = src[i];
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.gset), 0, a.g_bytes, 0x00020000);
  const unsigned n_in = (unsigned)(d.shared ? 1 : d.bs) * (unsigned)d.nq * (unsigned)d.heads * LP;
  const __amdgpu_buffer_rsrc_t rs_lg = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(a.logit), 0, n_in * (unsigned)ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_of = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(a.off), 0, n_in * 2u * (unsigned)ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_rf = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(a.ref), 0, (unsigned)d.bs * (unsigned)d.nq * (unsigned)d.ppg * 2u * (unsigned)sizeof(RefT),
      0x00020000);
  const unsigned lane8 = threadIdx.x & 7u;
  const unsigned lane16 = lane8 * 16u, lane8b = lane8 * 8u;
  char *box = smem + kTab + a.stage_bytes + (threadIdx.x >> 3) * kBox;
  const unsigned q_end = min((ck + 1u) * (unsigned)a.chunk, (unsigned)d.nq);

  // per-lane constants of the owner's PP points: level and reference-point group
  const bool owner = NOWN == 8 || lane8 < (unsigned)NOWN;
  unsigned lvo[PP], gof[PP];
  {
    const int j0 = (int)lane8 * PP;
    int l = j0 / d.P;
    int p = j0 - l * d.P;
    int g = p % d.ppg;
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      lvo[k] = owner ? (unsigned)l * kTabEnt : 0u;
      gof[k] = 2u * (unsigned)sizeof(RefT) * (unsigned)g;
      ++p; ++g;
      if (g == d.ppg) g = 0;
      if (p == d.P) { p = 0; g = 0; ++l; }
    }
  }
  constexpr unsigned kStride = THREADS / 8;
  const unsigned q0 = ck * (unsigned)a.chunk;
  unsigned n_items = q_end - q0;
  const unsigned short *qlist = reinterpret_cast<const unsigned short *>(
      smem + kTab + a.stage_bytes + (THREADS / 8) * kBox);
  if constexpr (MASKED) {  // compact the chunk to the (camera, query) pairs with a non-zero weight
    unsigned short *wl = const_cast<unsigned short *>(qlist);
    unsigned *wtot = reinterpret_cast<unsigned *>(smem + kTab + a.stage_bytes + (THREADS / 8) * kBox + a.chunk * 2);
    unsigned base_count = 0;
    for (unsigned t0 = 0; t0 < n_items; t0 += THREADS) {
      const unsigned i = t0 + threadIdx.x;
      const bool vis = i < n_items && __half2float(a.qmask[(size_t)b * d.nq + q0 + i]) != 0.f;
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

  // ---- streamed operands: byte offsets into the three descriptors
  constexpr int NLG = I8 ? 1 + (PP > 4) : (PP + 1) / 2;           // logits dwords per lane
  constexpr int NOF = I8 ? (PP + 1) / 2 : PP;                      // offsets dwords per lane
  constexpr int NRF = (int)sizeof(RefT) / 2 * PP;                  // reference dwords per lane
  struct Pre { unsigned lg[NLG], of[NOF], rf[NRF]; };
  const unsigned b_in = d.shared ? 0u : b;
  const unsigned lg_base = ((b_in * (unsigned)d.nq * (unsigned)d.heads + h) * LP + lane8 * PP) * (unsigned)ESZ;
  const unsigned lg_q = (unsigned)d.heads * LP * (unsigned)ESZ;
  const unsigned rf_base = b * (unsigned)d.nq * (unsigned)d.ppg * 2u * (unsigned)sizeof(RefT);
  const unsigned rf_q = (unsigned)d.ppg * 2u * (unsigned)sizeof(RefT);
  static_assert(!RR || PP == 4, "RR");
  constexpr int aux = (LP >= 32 && !(SCHED & 128)) ? 2 : 0;   // long read-once rows: non-temporal, the maps keep the L2
  auto request = [&](Pre &r, unsigned q) {
    const unsigned o_lg = lg_base + q * lg_q, o_of = 2u * o_lg, o_rf = rf_base + q * rf_q;
#pragma unroll
    for (int k = 0; k < NLG; ++k) r.lg[k] = I8 ? 0x80808080u : 0xfc00fc00u;  // most negative logits
#pragma unroll
    for (int k = 0; k < NOF; ++k) r.of[k] = 0;
#pragma unroll
    for (int k = 0; k < NRF; ++k) r.rf[k] = 0;
    if (!owner) return;
    constexpr int LGB = PP * ESZ;  // logits bytes of this lane; offsets: twice that
    if constexpr (LGB == 1) {
      r.lg[0] = 0x80808000u | (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_lg, (int)o_lg, 0, 0);
      r.of[0] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs_of, (int)o_of, 0, 0);
    } else if constexpr (LGB == 2) {
      r.lg[0] = (I8 ? 0x80800000u : 0xfc000000u) | (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs_lg, (int)o_lg, 0, 0);
      r.of[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_of, (int)o_of, 0, 0);
    } else if constexpr (LGB == 4) {
      r.lg[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_lg, (int)o_lg, 0, 0);
      const u32x2 v = aux ? __builtin_amdgcn_raw_buffer_load_b64(rs_of, (int)o_of, 0, 2)
                          : __builtin_amdgcn_raw_buffer_load_b64(rs_of, (int)o_of, 0, 0);
      r.of[0] = v.x; r.of[1] = v.y;
    } else if constexpr (LGB == 8) {
      const u32x2 g = aux ? __builtin_amdgcn_raw_buffer_load_b64(rs_lg, (int)o_lg, 0, 2)
                          : __builtin_amdgcn_raw_buffer_load_b64(rs_lg, (int)o_lg, 0, 0);
      r.lg[0] = g.x; r.lg[1] = g.y;
      const u32x4 v = aux ? __builtin_amdgcn_raw_buffer_load_b128(rs_of, (int)o_of, 0, 2)
                          : __builtin_amdgcn_raw_buffer_load_b128(rs_of, (int)o_of, 0, 0);
      r.of[0] = v.x; r.of[1] = v.y; r.of[2] = v.z; r.of[3] = v.w;
    } else {
      static_assert(LGB == 16, "points per owner");
      const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(rs_lg, (int)o_lg, 0, 2);
      r.lg[0] = g.x; r.lg[1] = g.y; r.lg[2] = g.z; r.lg[3] = g.w;
      const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs_of, (int)o_of, 0, 2);
      const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs_of, (int)o_of + 16, 0, 2);
      r.of[0] = v0.x; r.of[1] = v0.y; r.of[2] = v0.z; r.of[3] = v0.w;
      r.of[4] = v1.x; r.of[5] = v1.y; r.of[6] = v1.z; r.of[7] = v1.w;
    }
    if constexpr (RR) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_rf, (int)o_rf, 0, 0);
      r.rf[0] = v.x; r.rf[1] = v.y; r.rf[2] = v.z; r.rf[3] = v.w;
      if constexpr (sizeof(RefT) == 4) {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs_rf, (int)o_rf + 16, 0, 0);
        r.rf[4] = w.x; r.rf[5] = w.y; r.rf[6] = w.z; r.rf[7] = w.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < PP; ++k) {
        if constexpr (sizeof(RefT) == 4) {
          const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_rf, (int)(o_rf + gof[k]), 0, 0);
          r.rf[2 * k] = v.x; r.rf[2 * k + 1] = v.y;
        } else {
          r.rf[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_rf, (int)(o_rf + gof[k]), 0, 0);
        }
      }
    }
  };

  // ---- front end of one item on its owner lanes: softmax, locations, corner weights, addresses.
  // Records: fp16 {half2 row-0 weights, half2 row-1 weights, row-0 address, row-1 address};
  //          int8 {4 packed area weights, softmax weight (0 when out of view), address, row-1 address}
  auto front = [&](const Pre &r, uint4 (&pl)[PP], float &s_out, bool &any_out) {
    float e[PP];
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      if constexpr (I8) e[k] = mul_rn((float)(int)(signed char)((r.lg[k / 4] >> (8 * (k & 3))) & 0xffu), a.s_w);
      else e[k] = (k & 1) ? h2f_hi(r.lg[k / 2]) : h2f_lo(r.lg[k / 2]);
    }
    float m = e[0];
#pragma unroll
    for (int k = 1; k < PP; ++k) m = fmaxf(m, e[k]);
    m = oct_max(m);
    float s = 0.f;
    int wq[PP];
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      const float ex = owner ? __expf(I8 ? sub_rn(e[k], m) : e[k] - m) : 0.f;
      if constexpr (!I8) {
        e[k] = ex;
        s += ex;
        wq[k] = 0;
      } else if constexpr (U8W) {  // kernel.cu:1028-1037: S sums the UN-quantised x255 weights
        const float w255 = mul_rn(ex, 255.f);
        s = add_rn(s, w255);
        wq[k] = (int)rintf(w255);   // in [0, 255]: u16_rne's clamps cannot bite
      } else {                     // kernel.cu:926-930: S sums the quantised x127 weights
        wq[k] = owner ? t2i8_away_nonneg(mul_rn(ex, 127.f)) : 0;
        s += (float)wq[k];
      }
    }
    s_out = oct_sum(s);
    bool any_valid = false;
#pragma unroll
    for (int k = 0; k < PP; ++k) {
      const float4 tf = *reinterpret_cast<const float4 *>(smem + lvo[k]);
      const int2 ti = *reinterpret_cast<const int2 *>(smem + lvo[k] + 16);
      float rx, ry, ox, oy;
      if constexpr (sizeof(RefT) == 4) {
        rx = __uint_as_float(r.rf[2 * k]); ry = __uint_as_float(r.rf[2 * k + 1]);
      } else {
        rx = h2f_lo(r.rf[k]); ry = h2f_hi(r.rf[k]);
      }
      float x, y;
      if constexpr (I8) {
#pragma clang fp contract(off)
        const unsigned pr = (r.of[k / 2] >> (16 * (k & 1))) & 0xffffu;
        ox = (float)(int)(signed char)(pr & 0xffu);
        oy = (float)(int)(signed char)(pr >> 8);
        if constexpr (U8W) {  // kernel.cu:1040-1056: ref * size + (off * scale - 0.5)
          x = rx * tf.x + (ox * a.s_o - 0.5f);
          y = ry * tf.y + (oy * a.s_o - 0.5f);
        } else {              // kernel.cu:905-917
          x = (rx * tf.x + ox * a.s_o) - 0.5f;
          y = (ry * tf.y + oy * a.s_o) - 0.5f;
        }
      } else {
        ox = h2f_lo(r.of[k]); oy = h2f_hi(r.of[k]);
        x = fmaf(rx, tf.x, ox) - 0.5f;
        y = fmaf(ry, tf.y, oy) - 0.5f;
      }
      const bool valid = owner && (y > -1.f) && (x > -1.f) && (y < tf.y) && (x < tf.x);
      // a sample outside the range gate contributes exactly 0, as in the reference -- also when its
      // location is not finite (reference points of pillars behind a camera overflow binary16):
      // without this its fractions would be NaN and NaN * 0 would reach the output
      if (!valid) { x = 0.f; y = 0.f; }
      const float xf = floorf(x), yf = floorf(y);
      const float lx = x - xf, ly = y - yf;
      any_valid |= valid;
      if constexpr (I8) {
        const float hx = 1.f - lx, hy = 1.f - ly;
        unsigned a0, a1, a2, a3;
        if constexpr (U8W) {
          // (products of two numbers in [0, 1] times 255: u16_rne's clamps cannot bite)
          a0 = (unsigned)rintf(mul_rn(mul_rn(hy, hx), 255.f)); a1 = (unsigned)rintf(mul_rn(mul_rn(hy, lx), 255.f));
          a2 = (unsigned)rintf(mul_rn(mul_rn(ly, hx), 255.f)); a3 = (unsigned)rintf(mul_rn(mul_rn(ly, lx), 255.f));
        } else {   // kernel.cu:298-358 divides by the rounded 1/127: div_by_inv127 is that quotient
          a0 = (unsigned)t2i8_away_nonneg(div_by_inv127(mul_rn(hy, hx)));
          a1 = (unsigned)t2i8_away_nonneg(div_by_inv127(mul_rn(hy, lx)));
          a2 = (unsigned)t2i8_away_nonneg(div_by_inv127(mul_rn(ly, hx)));
          a3 = (unsigned)t2i8_away_nonneg(div_by_inv127(mul_rn(ly, lx)));
        }
        pl[k].x = (a0 & 255u) | ((a1 & 255u) << 8) | ((a2 & 255u) << 16) | ((a3 & 255u) << 24);
        pl[k].y = valid ? (unsigned)wq[k] : 0u;
        // x255 flavour: the dot's addend -128 (a0 + a1 + a2 + a3) rides above the weight byte (>= -130 560: fits;
        // the consumer's arithmetic shift by 8 returns it exactly because the weight byte is non-negative)
        if constexpr (U8W) pl[k].y |= (unsigned)(-(int)((a0 + a1 + a2 + a3) << 7)) << 8;
      } else {
        const float ev = valid ? e[k] : 0.f;
        const float wr1 = ly * ev, wr0 = ev - wr1;
        const float b0 = wr0 * lx, b1 = wr1 * lx;
        pl[k].x = pack_h2(wr0 - b0, b0);
        pl[k].y = pack_h2(wr1 - b1, b1);
      }
      // entry (yp, x0), yp = floor(y) + 1 in [0, H], x0 = floor(x) in [-1, W-1]
      const int rel = __mul24((int)yf + 1, ti.x) + (int)xf;
      pl[k].z = __float_as_uint(tf.z) + ((valid ? (unsigned)rel : 0u) << ti.y);
      pl[k].w = pl[k].z + __float_as_uint(tf.w);
    }
    any_out = any_valid;
  };
  auto post = [&](const uint4 (&pl)[PP]) {  // owner -> mailbox
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (owner) {
#pragma unroll
      for (int k = 0; k < PP; ++k) *reinterpret_cast<uint4 *>(box + (lane8 * PP + k) * 16) = pl[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };

  const int magic = U8W ? 65793 : 132104;  // round(2^24 / 255), round(2^24 / 127)
  const int half = 1 << 23;

  Pre pre1, pre2;  // operands of the next group and of the one after it
  unsigned i = threadIdx.x >> 3;
  uint4 pl[PP];
  float s_cur = 1.f;
  bool any_cur = false;
  {
    Pre pre0;
    if (i < n_items) request(pre0, query_of(i));
    else request(pre0, q0);  // harmless, in range
    request(pre1, i + kStride < n_items ? query_of(i + kStride) : q0);
    front(pre0, pl, s_cur, any_cur);
    post(pl);
  }
  for (; i < n_items; i += kStride) {
    const unsigned q = query_of(i);
    const unsigned q_pre = i + 2 * kStride < n_items ? query_of(i + 2 * kStride) : q0;
    if constexpr (!(SCHED & 32)) request(pre2, q_pre);
    float s_nxt;
    bool any_nxt;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int ia[4] = {0, 0, 0, 0}, wsum = 0;
    if (__any(any_cur)) {
      uint4 bp[D][BT];
      u32x4 r0[D][BT], r1[D][BT];
      auto issue = [&](int tb) {
        const int sl = tb % D;
#pragma unroll
        for (int j = 0; j < BT; ++j) bp[sl][j] = *reinterpret_cast<const uint4 *>(box + (tb * BT + j) * 16);
#pragma unroll
        for (int j = 0; j < BT; ++j) {
          r0[sl][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(bp[sl][j].z + lane16), 0, 0);
          if constexpr (!I8) r1[sl][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(bp[sl][j].w + lane16), 0, 0);
        }
      };
      // int8: 4 samples x 4 channels -> requantised samples, gathered per channel, one dot4 each
      auto i8_batch = [&](const unsigned (&v)[BT][4], const uint4 (&rec)[BT]) {
        int x[BT][4];
#pragma unroll
        for (int j = 0; j < BT; ++j) {
          if constexpr (U8W) i8_sample_u(v[j], rec[j].x, (int)rec[j].y >> 8, magic, half, x[j]);
          else i8_sample_s(v[j], rec[j].x, magic, half, x[j]);
        }
        unsigned w4;   // the four samples' softmax weight bytes (byte 0 of rec.y), sample 0 in byte 0
        if constexpr (BT == 4) {
          const unsigned lo = __builtin_amdgcn_perm(rec[1].y, rec[0].y, 0x0c0c0400u);
          const unsigned hi = __builtin_amdgcn_perm(rec[3].y, rec[2].y, 0x04000c0cu);
          w4 = lo | hi;
        } else {
          w4 = rec[0].y & 0xffu;
          if constexpr (BT > 1) w4 |= (rec[1].y & 0xffu) << 8;
          if constexpr (BT > 2) w4 |= (rec[2].y & 0xffu) << 16;
        }
        if constexpr (U8W) {
          // unsigned softmax weights: s w = (s + 128) w - 128 w; the second term once per batch
          wsum += (int)__builtin_amdgcn_udot4(w4, 0x01010101u, 0u, false);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int g4 = gather_hi(x[0][c], BT > 1 ? x[1][c] : 0, BT > 2 ? x[2][c] : 0, BT > 3 ? x[3][c] : 0);
          if constexpr (U8W)
            ia[c] = (int)__builtin_amdgcn_udot4((unsigned)g4 ^ 0x80808080u, w4, (unsigned)ia[c], false);
          else
            ia[c] = __builtin_amdgcn_sdot4(g4, (int)w4, ia[c], false);
        }
      };
      auto consume = [&](int tb) {
        const int sl = tb % D;
        if constexpr (I8) {
          unsigned v[BT][4];
#pragma unroll
          for (int j = 0; j < BT; ++j) { v[j][0] = r0[sl][j].x; v[j][1] = r0[sl][j].y; v[j][2] = r0[sl][j].z; v[j][3] = r0[sl][j].w; }
          i8_batch(v, bp[sl]);
        } else {
#pragma unroll
          for (int j = 0; j < BT; ++j) {
            const unsigned w0 = bp[sl][j].x, w1 = bp[sl][j].y;
            acc[0] = dot2f(r0[sl][j].x, w0, acc[0]); acc[1] = dot2f(r0[sl][j].y, w0, acc[1]);
            acc[2] = dot2f(r0[sl][j].z, w0, acc[2]); acc[3] = dot2f(r0[sl][j].w, w0, acc[3]);
            acc[0] = dot2f(r1[sl][j].x, w1, acc[0]); acc[1] = dot2f(r1[sl][j].y, w1, acc[1]);
            acc[2] = dot2f(r1[sl][j].z, w1, acc[2]); acc[3] = dot2f(r1[sl][j].w, w1, acc[3]);
          }
        }
      };
      auto lds_batch = [&](int u) {
        uint4 rec[BT];
#pragma unroll
        for (int j = 0; j < BT; ++j) rec[j] = *reinterpret_cast<const uint4 *>(box + ((NBIG + u) * BT + j) * 16);
        if constexpr (I8) {
          uint2 t0[BT], t1[BT];
#pragma unroll
          for (int j = 0; j < BT; ++j) {
            t0[j] = *reinterpret_cast<const uint2 *>(smem + rec[j].z + lane8b);
            t1[j] = *reinterpret_cast<const uint2 *>(smem + rec[j].w + lane8b);
          }
          unsigned v[BT][4];
#pragma unroll
          for (int j = 0; j < BT; ++j) {
            v[j][0] = __builtin_amdgcn_perm(t1[j].x, t0[j].x, 0x05040100u);
            v[j][1] = __builtin_amdgcn_perm(t1[j].x, t0[j].x, 0x07060302u);
            v[j][2] = __builtin_amdgcn_perm(t1[j].y, t0[j].y, 0x05040100u);
            v[j][3] = __builtin_amdgcn_perm(t1[j].y, t0[j].y, 0x07060302u);
          }
          i8_batch(v, rec);
        } else {
          uint2 l0[BT], rr0[BT], l1[BT], rr1[BT];
#pragma unroll
          for (int j = 0; j < BT; ++j) {
            const char *p0 = smem + rec[j].z + lane8b;
            const char *p1 = smem + rec[j].w + lane8b;
            l0[j] = *reinterpret_cast<const uint2 *>(p0);
            rr0[j] = *reinterpret_cast<const uint2 *>(p0 + kLdsPixBytes);
            l1[j] = *reinterpret_cast<const uint2 *>(p1);
            rr1[j] = *reinterpret_cast<const uint2 *>(p1 + kLdsPixBytes);
          }
#pragma unroll
          for (int j = 0; j < BT; ++j) {
            const h2_t w0 = as_h2(rec[j].x), w1 = as_h2(rec[j].y);
            const h2_t w00 = {w0[0], w0[0]}, w01 = {w0[1], w0[1]}, w10 = {w1[0], w1[0]}, w11 = {w1[1], w1[1]};
            h2_t va = as_h2(l0[j].x) * w00, vb = as_h2(l0[j].y) * w00;
            va = as_h2(rr0[j].x) * w01 + va; vb = as_h2(rr0[j].y) * w01 + vb;
            va = as_h2(l1[j].x) * w10 + va; vb = as_h2(l1[j].y) * w10 + vb;
            va = as_h2(rr1[j].x) * w11 + va; vb = as_h2(rr1[j].y) * w11 + vb;
            add_h2(acc[0], acc[1], va);
            add_h2(acc[2], acc[3], vb);
          }
        }
      };
      // The schedule, pinned step by step (the compiler is free inside a step only):
      //   issue big 0 .. D-1 | front end of the next group | HEAD LDS batches |
      //   { consume big t | issue big t+D | one LDS batch } ... | remaining LDS batches
      // Buffer loads retire in order, so the first consume also waits for the (HBM) operand
      // request made just before this block: the front end and the LDS head start cover it.
      constexpr int HEAD = NBIG == 0 ? NLDS : (NLDS > NBIG ? NLDS - NBIG + 1 : (NLDS >= 2 ? 2 : NLDS));
      uint4 npl[PP];
#pragma unroll
      for (int tb = 0; tb < D && tb < NBIG; ++tb) issue(tb);
      __builtin_amdgcn_sched_barrier(0);
      front(pre1, npl, s_nxt, any_nxt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < HEAD; ++u) {
        lds_batch(u);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int tb = 0; tb < NBIG; ++tb) {
        consume(tb);
        __builtin_amdgcn_sched_barrier(0);
        if (tb + D < NBIG) issue(tb + D);
        __builtin_amdgcn_sched_barrier(0);
        if (HEAD + tb < NLDS) {
          lds_batch(HEAD + tb);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int u = HEAD + NBIG; u < NLDS; ++u) lds_batch(u);
#pragma unroll
      for (int k = 0; k < PP; ++k) pl[k] = npl[k];
    } else {
      front(pre1, pl, s_nxt, any_nxt);
    }
    // ---- normalise, store
    if constexpr (I8) {
      int8_t *outp = reinterpret_cast<int8_t *>(a.out) + (((size_t)b * d.nq + q) * d.heads + h) * 32u + lane8 * 4u;
      unsigned res = 0;
      {
#pragma clang fp contract(off)
        const float scale_o = a.s_v * (1.0f / a.s_out);
        const float f = scale_o * (1.0f / s_cur);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int av = U8W ? ia[c] - (wsum << 7) : ia[c];
          const int rq = U8W ? t2i8_rne((float)av * f) : t2i8_away((float)av * f);
          res |= ((unsigned)rq & 0xffu) << (8 * c);
        }
      }
      *reinterpret_cast<unsigned *>(outp) = res;
    } else {
      __half *outp = reinterpret_cast<__half *>(a.out) + (((size_t)b * d.nq + q) * d.heads + h) * 32u + lane8 * 4u;
      const float inv = __builtin_amdgcn_rcpf(s_cur);
      uint2 v;
      v.x = pack_h2(acc[0] * inv, acc[1] * inv);
      v.y = pack_h2(acc[2] * inv, acc[3] * inv);
      if constexpr (LP >= 32 && !(SCHED & 128)) {
        __builtin_nontemporal_store(((unsigned long long)v.y << 32) | v.x,
                                    reinterpret_cast<unsigned long long *>(outp));
      } else {
        *reinterpret_cast<uint2 *>(outp) = v;
      }
    }
    if constexpr (SCHED & 32) request(pre2, q_pre);
    post(pl);
    s_cur = s_nxt;
    any_cur = any_nxt;
    pre1 = pre2;
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
constexpr int kH4Threads = 512;
inline int h4_box_bytes(int LP) { return (kH4Threads / 8) * (LP * 16 + 16); }

// INT8 calls with L*P = 32 prefer the "two blocks per CU" plan: only the planes that fit kOccStageCap bytes stay
// LDS-resident (base SCA: the 15 x 25 level, 28 KB; the 29 x 50 level joins the big set, whose int8 taps cost one
// cache line per sample), the kernel is compiled for 4 waves per SIMD and keeps one big batch in flight, so 16
// waves share a CU instead of 8: 553 vs 582 us (x127 flavour), 570 vs 609 us (x255) per base SCA call, bit-identical
// (profiles/r02/hm4_int8_occupancy_ab.jsonl).  The same plan in fp16 (two lines per big-level sample, spills
// under the 128-register cap) takes 950 us instead of 565: fp16 keeps the one-block plan.
constexpr int kOccStageCap = 40 * 1024;
thread_local bool g_h4_no_occ = false;   // variant 19 / the ablation variants: the one-block plan for int8 too
struct H4Plan {
  Hm3Plan p;
  int nbig;   // tap batches served by L1/L2
  bool occ2;  // the two-blocks-per-CU plan
};

bool h4_plan(const int32_t *shapes_host, int bs, int heads, int L, int P, int nq, H4Plan &pl, bool i8) {
  const int LP = L * P;
  const int bt = LP >= 4 ? 4 : LP;
  pl.occ2 = false;
  if (i8 && !g_h4_no_occ && LP == 32) {
    // (hm3_plan budgets the staged planes as kLdsLimit - kTab - box bytes: a cap is a larger pretended box)
    if (hm3_plan(shapes_host, bs, heads, L, nq, kLdsLimit - kTab - kOccStageCap, pl.p) && pl.p.t.ls < L &&
        (pl.p.t.ls * P) % bt == 0 && pl.p.t.ls * P / bt == 6) {
      pl.nbig = 6;
      pl.occ2 = true;
      return true;
    }
  }
  if (!hm3_plan(shapes_host, bs, heads, L, nq, h4_box_bytes(LP), pl.p)) return false;
  if ((pl.p.t.ls * P) % bt) return false;  // a batch never straddles the big / staged boundary
  pl.nbig = pl.p.t.ls * P / bt;
  return true;
}

template <int LP, int NBIG, bool I8, bool U8W, typename RefT, bool MASKED, bool RR, int SCHED = 0>
int h4_go(const H4Args &a, hipStream_t st) {
  const size_t lds = kTab + a.stage_bytes + (size_t)h4_box_bytes(LP) + (a.qmask ? a.chunk * 2 + 64 : 0);
  if (lds > (size_t)kLdsLimit) return BEVOPS_NOT_SUPPORTED;
  if (!ensure_dynamic_lds<msda_hm4_kernel<LP, NBIG, kH4Threads, I8, U8W, RefT, MASKED, RR, SCHED>>(lds))
    return BEVOPS_FAILURE;
  const dim3 grid((unsigned)(a.d.bs * a.d.heads * a.nchunk));
  hipLaunchKernelGGL((msda_hm4_kernel<LP, NBIG, kH4Threads, I8, U8W, RefT, MASKED, RR, SCHED>), grid,
                     dim3(kH4Threads), lds, st, a);
  return launch_status();
}

// instantiated (L*P, big batches) combinations: the model's calls.  Anything else -> NOT_SUPPORTED
// (the caller keeps its older kernels for those).
template <bool I8, bool U8W, typename RefT, bool MASKED>
int h4_dispatch(int LP, int nbig, bool occ2, const H4Args &a, hipStream_t st) {
  // points of an owner lane (4 of them when L*P = 32) share ONE run of reference points
  const bool rr = LP == 32 && a.d.ppg == 4 && a.d.P % 4 == 0;
  // production schedule (profiles/r02/hm4_variants.jsonl): fp16 requests the next operands at
  // the END of the loop body (32); int8 streams them with the default cache policy (128)
  constexpr int PROD = I8 ? 128 : 32;
#define BEVOPS_H4_CASE(LP_, NBIG_)                                                        \
  if (LP == LP_ && nbig == NBIG_) {                                                       \
    if constexpr (LP_ == 32) {                                                            \
      if (rr) return h4_go<LP_, NBIG_, I8, U8W, RefT, MASKED, true, PROD>(a, st);         \
    }                                                                                     \
    return h4_go<LP_, NBIG_, I8, U8W, RefT, MASKED, false, PROD>(a, st);                  \
  }
  if constexpr (I8 && !MASKED) {
    // the two-blocks-per-CU plan (h4_plan): <= 128 VGPRs, one big batch in flight
    if (occ2) {
      if (LP != 32 || nbig != 6) return BEVOPS_NOT_SUPPORTED;
      if (rr) return h4_go<32, 6, I8, U8W, RefT, MASKED, true, PROD | 1024 | 64>(a, st);
      return h4_go<32, 6, I8, U8W, RefT, MASKED, false, PROD | 1024 | 64>(a, st);
    }
  }
  BEVOPS_H4_CASE(32, 4)   // base SCA: 4 levels x 8 points, two levels staged
  BEVOPS_H4_CASE(32, 8)   //   ... nothing staged (few queries)
  BEVOPS_H4_CASE(32, 6)   //   ... one level staged
  BEVOPS_H4_CASE(8, 0)    // tiny / small SCA: 1 level x 8 points, staged
  BEVOPS_H4_CASE(8, 2)    //   ... not staged
  BEVOPS_H4_CASE(4, 1)    // TSA / decoder: 1 level x 4 points
  BEVOPS_H4_CASE(4, 0)
#undef BEVOPS_H4_CASE
  return BEVOPS_NOT_SUPPORTED;
}

bool h4_instantiated(int LP, int nbig) {
  return (LP == 32 && (nbig == 4 || nbig == 8 || nbig == 6)) || (LP == 8 && (nbig == 0 || nbig == 2)) ||
         (LP == 4 && (nbig == 1 || nbig == 0));
}

int h4_chunk(const Hm3Plan &p, int nq, int variant_chunk) {
  if (variant_chunk > 0) return variant_chunk;
  return p.stage_bytes ? 1280 : 512;
}

}  // namespace

void msda_hm4_set_no_occ(bool v) { g_h4_no_occ = v; }

size_t msda_hm4_workspace_bytes(const int32_t *shapes_host, int bs, int heads, int C, int L, int nq, int P, bool i8) {
  H4Plan pl;
  if (C != 32 || !shapes_host || !h4_plan(shapes_host, bs, heads, L, P, nq, pl, i8)) return 0;
  return ((pl.p.g_bytes + 127) & ~size_t(127)) + 128 + pl.p.s_bytes;
}

// every level LDS-resident and an instantiated kernel: the shapes where hm4 is the fp16 default
bool msda_hm4_all_staged(const int32_t *shapes_host, int bs, int heads, int C, int L, int nq, int P) {
  H4Plan pl;
  return C == 32 && shapes_host && h4_plan(shapes_host, bs, heads, L, P, nq, pl, false) && pl.nbig == 0 &&
         h4_instantiated(L * P, 0);
}

// The call in two halves (bevops_msda_pack_value / bevops_msda_forward_prepacked): the re-layout
// of `value` into the padded head-major sets, and the sampling kernel on those sets -- for callers
// that sample one value tensor several times, or produce the packed form themselves.
// dtype: BEVOPS_F16 (ref fp16) or BEVOPS_I8 (ref fp32 -> x127 flavour, ref fp16 -> x255 flavour:
// its planes are biased, so the packed form is flavour-specific).
int msda_hm4_pack(int dtype, int ref_dtype, const void *value, const int32_t *shapes_host, int bs, int nk,
                  int heads, int C, int L, int nq, int P, void *packed, size_t packed_bytes, hipStream_t st) {
  H4Plan pl;
  if (C != 32 || !packed || (reinterpret_cast<uintptr_t>(packed) & 127u) || !shapes_host ||
      !h4_plan(shapes_host, bs, heads, L, P, nq, pl, dtype == BEVOPS_I8))
    return BEVOPS_NOT_SUPPORTED;
  if (dtype != BEVOPS_F16 && dtype != BEVOPS_I8) return BEVOPS_NOT_SUPPORTED;
  const size_t g_room = (pl.p.g_bytes + 127) & ~size_t(127);
  if (packed_bytes < g_room + pl.p.s_bytes) return BEVOPS_BAD_PARAM;
  char *gset = static_cast<char *>(packed);
  char *sset = gset + g_room;
  const Hm3Tab &t = pl.p.t;
  if (dtype == BEVOPS_I8) {
    if (bs * heads > 65535) return BEVOPS_NOT_SUPPORTED;
    const dim3 grid((unsigned)(((t.g_entries + 31) >> 5) + ((t.s_entries + 31) >> 5)), (unsigned)(bs * heads));
    hipLaunchKernelGGL(msda_hm4_repack_i8_kernel, grid, dim3(256), 0, st, (const int8_t *)value, gset, sset, t, nk,
                       heads, ref_dtype == BEVOPS_F16 ? 0x80808080u : 0u);
  } else {
    msda_hm3_repack_launch(value, gset, sset, &t, bs, nk, heads, st);
  }
  return launch_status();
}

int msda_hm4_forward_prepacked(int dtype, int ref_dtype, const void *packed, size_t packed_bytes,
                               const int32_t *shapes_host, const void *ref, const void *off, const void *logit,
                               void *out, int bs, int nk, int heads, int C, int L, int nq, int P, int ppg,
                               int shared, float s_v, float s_o, float s_w, float s_out, int chunk_override,
                               int ablate, hipStream_t st) {
  const int LP = L * P;
  H4Plan pl;
  if (C != 32 || !packed || (reinterpret_cast<uintptr_t>(packed) & 127u) || !shapes_host ||
      !h4_plan(shapes_host, bs, heads, L, P, nq, pl, dtype == BEVOPS_I8))
    return BEVOPS_NOT_SUPPORTED;
  if ((double)bs * nq * heads * LP * 4.0 >= 4294967040.0) return BEVOPS_NOT_SUPPORTED;  // 32-bit offsets
  const size_t g_room = (pl.p.g_bytes + 127) & ~size_t(127);
  if (packed_bytes < g_room + pl.p.s_bytes) return BEVOPS_NOT_SUPPORTED;
  const char *gset = static_cast<const char *>(packed);
  H4Args a;
  a.gset = gset; a.g_bytes = (unsigned)pl.p.g_bytes; a.sset = gset + g_room;
  a.ref = ref; a.off = off; a.logit = logit; a.out = out;
  a.d = MsdaDims{bs, nk, heads, C, L, nq, P, ppg, shared};
  a.t = pl.p.t;
  a.chunk = h4_chunk(pl.p, nq, chunk_override);
  if (a.chunk > 0xffff) a.chunk = 0xff00;
  a.nchunk = (nq + a.chunk - 1) / a.chunk;
  a.stage_bytes = pl.p.stage_bytes;
  a.qmask = nullptr;
  a.s_v = s_v; a.s_o = s_o; a.s_w = s_w; a.s_out = s_out;
  const bool i8 = dtype == BEVOPS_I8;
  if (ablate) return BEVOPS_NOT_SUPPORTED;   // (the timing builds of round 2 are gone)
  if (dtype == BEVOPS_F16) return h4_dispatch<false, false, __half, false>(LP, pl.nbig, pl.occ2, a, st);
  if (i8 && ref_dtype == BEVOPS_F32) return h4_dispatch<true, false, float, false>(LP, pl.nbig, pl.occ2, a, st);
  if (i8 && ref_dtype == BEVOPS_F16) return h4_dispatch<true, true, __half, false>(LP, pl.nbig, pl.occ2, a, st);
  return BEVOPS_NOT_SUPPORTED;
}

int msda_hm4_forward(int dtype, int ref_dtype, const void *value, const int32_t *shapes_host, const void *ref,
                     const void *off, const void *logit, void *out, int bs, int nk, int heads, int C, int L,
                     int nq, int P, int ppg, int shared, float s_v, float s_o, float s_w, float s_out,
                     void *workspace, size_t workspace_bytes, int chunk_override, int ablate, hipStream_t st) {
  H4Plan pl;   // (checked first so that an unsupported shape costs no launch)
  if (C != 32 || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 127u) || !shapes_host ||
      !h4_plan(shapes_host, bs, heads, L, P, nq, pl, dtype == BEVOPS_I8) || !h4_instantiated(L * P, pl.nbig))
    return BEVOPS_NOT_SUPPORTED;
  const int rc = msda_hm4_pack(dtype, ref_dtype, value, shapes_host, bs, nk, heads, C, L, nq, P, workspace,
                               workspace_bytes, st);
  if (rc == BEVOPS_BAD_PARAM) return BEVOPS_NOT_SUPPORTED;   // workspace too small: the caller's other kernels
  if (rc != BEVOPS_SUCCESS) return rc;
  return msda_hm4_forward_prepacked(dtype, ref_dtype, workspace, workspace_bytes, shapes_host, ref, off, logit, out,
                                    bs, nk, heads, C, L, nq, P, ppg, shared, s_v, s_o, s_w, s_out, chunk_override,
                                    ablate, st);
}

}  // namespace bevops
