// MSDA fp16, fifth head-major generation ("hm5") for the BEVFormer-base SCA call shape
// (4 levels x 8 points, 4 reference anchors, 32 channels per head).  Same padded head-major value
// layout as hm3 / hm4 (msda_pad.h: 128-byte pixel-pair entries for the big levels, LDS-resident
// 64-byte pixels for the staged tail), same arithmetic per sample as hm3 (fp32 locations / softmax /
// area weights, combined weight rounded to binary16 once, v_dot2_f32_f16 on the big levels, packed
// fp16 row blend + fp32 accumulation on the staged ones).  What changes is the schedule, after the
// round-2 finding that hm3's and hm4's pipes do not overlap:
//   * vmcnt retires loads IN ORDER.  hm3 / hm4 request the next item's logits / offsets (HBM, long
//     latency) ahead of the current item's L2 taps, so every tap wait also waits for that request:
//     one HBM round trip per 8 items, on every wave.  Here the operand request is the LAST vector
//     memory instruction of an iteration and targets the item three iterations ahead (three
//     register sets, loop unrolled by three): no tap wait ever has a younger HBM request in front
//     of it, and a request has two full iterations to land.
//   * interleaved phases.  Lane k of an octet owns points 4k..4k+3, i.e. ONE level (k / 2) and all
//     four anchors.  Phase j takes point j of every lane: 8 samples = 2 per level = NB big + NS
//     staged.  Every phase therefore has front-end work for all 64 lanes (1 record per lane, no
//     idle owner lanes, no 16-register record store), big-level loads, LDS taps and multiply-adds;
//     the loads of phase j fly across the LDS taps of phase j and the front end of phase j+1.
//   * the staged taps are read with two ds_read_b64 (256 B/clk) instead of one ds_read2_b64
//     (128 B/clk, MI355X_MICROARCH.md LDS table).
//   * exact visibility pre-pass (msda_hm5_vis_kernel): an item (batch, query, head) all of whose
//     L*P samples fail the reference's range gate (multiScaleDeformableAttnKernel.cu:673) contributes
//     exactly 0 -- the pre-pass decides that from reference points + offsets alone (the gate does
//     not involve the logits), stores the zeros and a visibility byte; the sampling kernel compacts
//     its chunk of queries to the visible items (ballot + prefix), never reads the logits /
//     offsets of the others and never gives them a tap slot.  Items with an anchor inside
//     [0, 1]^2 are classified visible without reading their offsets (a classification only: the
//     sampling kernel evaluates every sample of a listed item exactly).  On the 6-camera rig
//     geometry 81 % of the (camera, pillar) pairs are invisible.
#include "msda_common.h"
#include "msda_pad.h"

namespace bevops {
namespace {

template <int J>
struct IC { static constexpr int v = J; };

// LDS accesses through address_space(3) pointers built from 32-bit byte addresses: the natural
// alignment of the pointee lets the compiler pick ds_read_b128 / ds_write_b128 / ds_read_b64
typedef __attribute__((address_space(3))) char lds_c;
typedef __attribute__((address_space(3))) u32x4 lds_u4;
typedef __attribute__((address_space(3))) u32x2 lds_u2;
typedef __attribute__((address_space(3))) unsigned short lds_u16;
typedef __attribute__((address_space(3))) unsigned lds_u32;

struct H5Set {
  unsigned lg[2];  // 4 logits of this lane's points
  u32x4 of;        // 4 (x, y) offsets
  unsigned rf;     // anchor (lane & 3) of the item
};

struct H5Lane {    // per-lane level constants: lane k of an octet serves level k / 2
  float W, H;
  unsigned base;   // byte offset of padded (row 0, col 0): big -> in the plane set, staged -> in LDS
  unsigned row;    // padded row bytes
  int wp, sh;
};

__device__ __forceinline__ H5Lane h5_lane_consts(const Hm3Tab &t, unsigned lane8, unsigned bh, unsigned stage_off) {
  const int myl = (int)(lane8 >> 1);
  H5Lane c{1.f, 1.f, 0u, 0u, 1, 6};
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    if (l == myl) {
      const bool staged = l >= t.ls;
      c.sh = staged ? 6 : 7;
      c.W = (float)t.W[l];
      c.H = (float)t.H[l];
      c.wp = t.W[l] + 1;
      c.base = (staged ? stage_off : bh * (unsigned)t.g_entries * kEntBytes) + ((unsigned)t.ent0[l] << c.sh);
      c.row = (unsigned)(t.W[l] + 1) << c.sh;
    }
  }
  return c;
}

// the reference's range gate and hm3's record arithmetic for one sample
struct H5Loc { float x, y; bool valid; };
__device__ __forceinline__ H5Loc h5_locate(unsigned rf, unsigned of, const H5Lane &c) {
  H5Loc r;
  r.x = fmaf(h2f_lo(rf), c.W, h2f_lo(of)) - 0.5f;
  r.y = fmaf(h2f_hi(rf), c.H, h2f_hi(of)) - 0.5f;
  r.valid = (r.y > -1.f) && (r.x > -1.f) && (r.y < c.H) && (r.x < c.W);
  return r;
}

// ---- visibility pre-pass.  First a lane per (batch, query): any anchor inside [0, 1]^2 -> all its heads
// are listed without reading their offsets.  The others are checked exactly, 8 items (heads) of a pair
// per wave step (their offsets are one contiguous run), four pairs in flight: vis[item] = 1 when any
// sample passes the reference's range gate, otherwise the item's 32 outputs are stored as zeros here.
__global__ __launch_bounds__(256) void msda_hm5_vis_kernel(const __half *__restrict__ ref,
                                                           const __half *__restrict__ off,
                                                           __half *__restrict__ out,
                                                           unsigned char *__restrict__ vis, MsdaDims d,
                                                           Hm3Tab t, unsigned n_pair) {
  const unsigned lane = threadIdx.x & 63u, lane8 = lane & 7u, oiw = lane >> 3;
  const unsigned wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
  const unsigned pair = wave * 64u + lane;
  const bool ok = pair < n_pair;
  const H5Lane c = h5_lane_consts(t, lane8, 0u, 0u);
  const uint4 r4 = *reinterpret_cast<const uint4 *>(ref + (size_t)(ok ? pair : n_pair - 1u) * 8u);
  auto in01 = [](unsigned r) {
    const float x = h2f_lo(r), y = h2f_hi(r);
    return x >= 0.f && x <= 1.f && y >= 0.f && y <= 1.f;
  };
  const bool anyin = in01(r4.x) || in01(r4.y) || in01(r4.z) || in01(r4.w);
  if (ok && anyin) {
    unsigned char *v = vis + (size_t)pair * d.heads;
    if (d.heads == 8) {
      *reinterpret_cast<unsigned long long *>(v) = 0x0101010101010101ull;
    } else {
      for (int hh = 0; hh < d.heads; ++hh) v[hh] = 1;
    }
  }
  unsigned long long todo = __ballot(ok && !anyin);
  const unsigned groups = ((unsigned)d.heads + 7u) >> 3;
  while (todo) {
    constexpr int U = 4;
    unsigned pr[U];
    bool on[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      on[u] = todo != 0ull;
      const unsigned bit = on[u] ? (unsigned)__builtin_ctzll(todo) : 0u;
      pr[u] = wave * 64u + bit;
      if (on[u]) todo &= todo - 1ull;
    }
    for (unsigned g = 0; g < groups; ++g) {
      const unsigned hd = g * 8u + oiw;
      const bool live = hd < (unsigned)d.heads;
      unsigned rf[U];
      u32x4 of[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        rf[u] = 0u;
        of[u] = u32x4{0u, 0u, 0u, 0u};
        if (on[u] && live) {
          rf[u] = *reinterpret_cast<const unsigned *>(ref + ((size_t)pr[u] * 4u + (lane8 & 3u)) * 2u);
          of[u] = __builtin_nontemporal_load(
              reinterpret_cast<const u32x4 *>(off + ((size_t)pr[u] * d.heads + hd) * 64u + lane8 * 8u));
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        bool v = false;
        v |= h5_locate(quad_bcast<0>(rf[u]), of[u].x, c).valid;
        v |= h5_locate(quad_bcast<1>(rf[u]), of[u].y, c).valid;
        v |= h5_locate(quad_bcast<2>(rf[u]), of[u].z, c).valid;
        v |= h5_locate(quad_bcast<3>(rf[u]), of[u].w, c).valid;
        const unsigned long long bal = __ballot(v && on[u] && live);
        const bool visible = ((bal >> (oiw * 8u)) & 0xffull) != 0ull;
        if (on[u] && live) {
          const size_t item = (size_t)pr[u] * d.heads + hd;
          if (lane8 == 0u) vis[item] = visible ? 1 : 0;
          if (!visible) *reinterpret_cast<uint2 *>(out + item * 32u + lane8 * 4u) = make_uint2(0u, 0u);
        }
      }
    }
  }
}

// ---- visibility plan of the fused SCA op (LISTED == 3 below): per camera the ascending list of the queries whose
// bev_mask weight is non-zero.  bev_mask depends on the calibration matrices only (encoder.py:255-258); nuScenes matrices
// change every frame, so the frame's graph rebuilds the plan behind bevops_point_sampling on every replay and every layer
// of the frame samples with it.  Layout: int32 counts[kPlanCams] (64 bytes), then per camera a list of `nq_pad` 32-bit
// entries: the query index in bits 0-15 and, in bit 16, "this camera is the only one that sees the query, with weight
// exactly 1" -- the sampler stores such a pair's result straight into the op's output row (1 * v + 0 is v) and the
// camera reduce leaves those rows alone (kSoleBit); then the builder's scratch: kPlanBlocks block counts per camera.
// Round 6: two launches of (256-query block, camera) grids -- count, then place behind the sum of the blocks in front
// -- instead of one 1 024-thread block per camera walking its 40 000 queries (63 us at the base size: it was built
// once per rig then; now it is on every frame's critical path).
constexpr int kPlanCams = 16;
constexpr int kPlanBlocks = 256;   // 256-query blocks per camera: num_query <= 65 535
inline size_t h5_plan_pad(int nq) { return ((size_t)nq + 63) & ~size_t(63); }
__device__ __forceinline__ unsigned h5_plan_pad_dev(int nq) { return ((unsigned)nq + 63u) & ~63u; }
constexpr unsigned kSoleBit = 0x10000u;

__device__ __forceinline__ bool h5_plan_visible(const unsigned short *__restrict__ qmask, unsigned cam, unsigned q, int nq) {
  return q < (unsigned)nq && (qmask[(size_t)cam * nq + q] & 0x7fffu) != 0;   // not +-0
}

__global__ __launch_bounds__(256) void msda_hm5_plan_count_kernel(const unsigned short *__restrict__ qmask, int nq,
                                                                  unsigned *__restrict__ partial) {
  __shared__ unsigned wcnt[4];
  const unsigned cam = blockIdx.y, q = blockIdx.x * 256u + threadIdx.x;
  const unsigned long long bal = __ballot(h5_plan_visible(qmask, cam, q, nq));
  if ((threadIdx.x & 63u) == 0u) wcnt[threadIdx.x >> 6] = (unsigned)__popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) partial[cam * kPlanBlocks + blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

__global__ __launch_bounds__(256) void msda_hm5_plan_place_kernel(const unsigned short *__restrict__ qmask, int bs, int nq,
                                                                  unsigned nq_pad, int *__restrict__ counts,
                                                                  unsigned *__restrict__ lists,
                                                                  const unsigned *__restrict__ partial) {
  __shared__ unsigned wsum[4], wcnt[4];
  const unsigned cam = blockIdx.y, blk = blockIdx.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const unsigned q = blk * 256u + threadIdx.x;
  // entries of this camera in front of this block: the sum of the earlier blocks' counts (gridDim.x <= 256: one each)
  unsigned before = threadIdx.x < blk ? partial[cam * kPlanBlocks + threadIdx.x] : 0u;
#pragma unroll
  for (unsigned d = 32; d >= 1; d >>= 1) before += (unsigned)__shfl_xor((int)before, (int)d, 64);
  const bool v = h5_plan_visible(qmask, cam, q, nq);
  const unsigned long long bal = __ballot(v);
  if (lane == 0u) {
    wsum[wv] = before;
    wcnt[wv] = (unsigned)__popcll(bal);
  }
  __syncthreads();
  unsigned o = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  unsigned all = 0;
#pragma unroll
  for (unsigned w2 = 0; w2 < 4; ++w2) {
    if (w2 < wv) o += wcnt[w2];
    all += wcnt[w2];
  }
  if (blk == gridDim.x - 1u && threadIdx.x == 0) counts[cam] = (int)(o + all);   // (o = the blocks in front: wv == 0)
  if (v) {
    bool sole = qmask[(size_t)cam * nq + q] == 0x3c00u;   // binary16 1.0
    for (int c2 = 0; c2 < bs; ++c2)
      if (c2 != (int)cam && (qmask[(size_t)c2 * nq + q] & 0x7fffu) != 0) sole = false;
    lists[(size_t)cam * nq_pad + o + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = q | (sole ? kSoleBit : 0u);
  }
}

// ---- sampling kernel.  LISTED: 0 every query of the chunk, 1 the items whose
// visibility byte is set (pre-pass), 2 the (batch, query) pairs with a non-zero `qmask` weight (fused SCA:
// `vis` then points at the [bs, nq] fp16 bev_mask and the offsets / logits are shared by all batches), 3 the fused SCA
// op on a visibility PLAN (`vis` points at it, msda_hm5_plan_kernel): the visible (camera, query) pairs of ALL cameras
// form one global sequence, and block j of the gridDim.x / heads blocks of a head takes the j-th equal slice of it --
// every block carries the same number of items whatever the visibility pattern (one block per 1 280-query chunk: 0 ..
// 830 items per block on the 6-camera rig, 39 % of the blocks empty, the average non-empty block 3 rounds of 128
// octets behind a 130 KB plane copy), stages the plane of each camera its slice touches (at most two at one block per
// CU) and needs no in-kernel compaction.
// The 8 records of a phase reach the octet's lanes through DPP: two row shifts give every lane the record of slot
// (lane % 4) and of slot 4 + (lane % 4), a quad_perm broadcast per slot and dword does the rest (26 VALU instructions
// per phase, no LDS traffic; an LDS mailbox -- one ds_write_b128 per lane, one broadcast ds_read_b128 per record --
// measured 8 us slower per base SCA call in round 3).  Builds measured in rounds 3-5 and no longer in this file (the
// history has them: 768-thread blocks, two phases of loads in flight, the mailbox, persistent blocks on strided
// sub-chunks, specialised waves, the level-class split, ablation / timing builds): design/msda.md.
constexpr int kH5Threads = 1024;   // one block per CU (the staged planes fill the LDS): 16 waves at <= 128 registers
// FOLD (round 6, the planned kernel's default): the DPP broadcast of a record dword is folded INTO the instruction that
// consumes it -- the two weight pairs of a big-level sample ride as the DPP source of its sixteen v_dot2c_f32_f16
// (v_dot2c is VOP2: it takes a quad_perm source; hand-placed asm blocks, the compiler does not combine a broadcast
// with eight uses), a slot's entry address is ONE v_add_u32 with a DPP source (the compiler's own combine: the
// broadcast has a single use) and its second row one more add -- instead of 3 broadcast moves + 3 address adds per
// sample and register copies of all eight records: same arithmetic in the same order, the same bits.  The two 8-byte
// LDS taps of a staged row are also left to fuse into one ds_read2_b64 (two address adds less per slot; the LDS pipe has
// the room on the rig geometry).  724 -> 660 VALU instructions per wave and item, 96-97 -> 92 us per call
// (profiles/r06/sca_fold_ab*.jsonl, sca_fold_pmc.txt).  Built, measured and removed in the same round: the last staged
// level as pair entries in LDS so that its samples run on v_dot2c too (612 instructions, 93.5-94 us: slower).
template <int LISTED, bool FOLD = false>
__global__ __launch_bounds__(kH5Threads, 1) void msda_hm5_kernel(
    const char *__restrict__ gset, unsigned g_bytes, const char *__restrict__ sset,
    const __half *__restrict__ ref, const __half *__restrict__ off, const __half *__restrict__ logit,
    __half *__restrict__ out, MsdaDims d, Hm3Tab t, int chunk, int nchunk, int stage_bytes,
    const unsigned char *__restrict__ vis, __half *__restrict__ direct) {
  constexpr int THREADS = kH5Threads;
  constexpr int NB = 4;         // big-level samples per phase (two big levels x two lanes)
  constexpr int NS = 8 - NB;    // staged samples per phase
  constexpr unsigned OCT = THREADS / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // smem: [staged planes][query list (u16; LISTED == 3: the plan's 32-bit entries)][wave totals]
  unsigned bh, ck = 0;
  const unsigned per_plane = (unsigned)nchunk;
  if (d.heads == 8) {   // XCD x keeps head x; all XCDs walk the same (batch, chunk) sequence
    const unsigned rest = blockIdx.x >> 3;
    bh = (rest / per_plane) * 8u + (blockIdx.x & 7u);
    ck = rest % per_plane;
  } else {
    const unsigned vb = xcd_remap(blockIdx.x, gridDim.x);
    bh = vb / per_plane;
    ck = vb - bh * per_plane;
  }
  unsigned b = bh / (unsigned)d.heads, h = bh - b * (unsigned)d.heads;
  if constexpr (LISTED == 3) {   // head = XCD (d.heads == 8: blockIdx.x & 7); the camera comes with each plan segment
    h = blockIdx.x % (unsigned)d.heads;
    b = 0;
    bh = h;
  }
  unsigned short *wl = reinterpret_cast<unsigned short *>(smem + stage_bytes);
  unsigned *wtot = reinterpret_cast<unsigned *>(smem + stage_bytes + chunk * 2);
  auto visible = [&](unsigned q) -> bool {
    if constexpr (LISTED == 3) return true;
    else if constexpr (LISTED == 2) return (reinterpret_cast<const unsigned short *>(vis)[(size_t)b * d.nq + q] & 0x7fffu) != 0;   // not +-0
    else if constexpr (LISTED == 1) return vis[((size_t)b * d.nq + q) * d.heads + h] != 0;
    else return true;
  };
  auto stage_plane = [&]() {
    if (stage_bytes) {
      const uint4 *src = reinterpret_cast<const uint4 *>(sset + (size_t)bh * stage_bytes);
      uint4 *dst = reinterpret_cast<uint4 *>(smem);
      for (int i = threadIdx.x; i < stage_bytes / 16; i += THREADS) dst[i] = src[i];
    }
  };
  unsigned q0 = 0, n_items = 0;
  if constexpr (LISTED != 3) {
    q0 = ck * (unsigned)chunk;
    const unsigned q_end = min(q0 + (unsigned)chunk, (unsigned)d.nq);
    n_items = q_end - q0;
    if constexpr (LISTED != 0) {
      unsigned base_count = 0;
      for (unsigned t0 = 0; t0 < n_items; t0 += THREADS) {
        const unsigned i = t0 + threadIdx.x;
        const bool v = i < n_items && visible(q0 + i);
        const unsigned long long bal = __ballot(v);
        const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
        if (lane == 0) wtot[wv] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = base_count, all = 0;
        for (unsigned w2 = 0; w2 < THREADS / 64; ++w2) {
          const unsigned cnt = wtot[w2];
          if (w2 < wv) before += cnt;
          all += cnt;
        }
        if (v) wl[before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)i;
        base_count += all;
        __syncthreads();
      }
      n_items = base_count;
    }
    if (n_items == 0) return;   // nothing of this chunk is visible from this camera: no plane copy either
    stage_plane();
    __syncthreads();
  }
  const unsigned wave_first = (threadIdx.x >> 6) * 8u;

  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(gset), 0, g_bytes, 0x00020000);
  const unsigned n_in = (unsigned)(d.shared ? 1 : d.bs) * (unsigned)d.nq * (unsigned)d.heads * 32u;
  const __amdgpu_buffer_rsrc_t rs_lg =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__half *>(logit), 0, n_in * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_of =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__half *>(off), 0, n_in * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_rf = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__half *>(ref), 0, (unsigned)d.bs * (unsigned)d.nq * 16u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      out, 0, (unsigned)d.bs * (unsigned)d.nq * (unsigned)d.heads * 64u, 0x00020000);
  // LISTED == 3: the op's own output rows [nq, heads, 32], for the pairs the plan marks kSoleBit (no `direct`: none)
  const __amdgpu_buffer_rsrc_t rs_dir = __builtin_amdgcn_make_buffer_rsrc(
      direct, 0, direct ? (unsigned)d.nq * (unsigned)d.heads * 64u : 0u, 0x00020000);
  const unsigned lane8 = threadIdx.x & 7u;
  const unsigned lane16 = lane8 * 16u, lane8b = lane8 * 8u;
  unsigned out_base = (b * (unsigned)d.nq * (unsigned)d.heads + h) * 64u + lane8b;
  const unsigned out_q = (unsigned)d.heads * 64u;
  const unsigned sbase = (unsigned)(uintptr_t)(lds_c *)smem;
  const unsigned qlist_a = sbase + (unsigned)stage_bytes;
  H5Lane c = h5_lane_consts(t, lane8, bh, sbase);

  const unsigned lg_base = (((d.shared ? 0u : b) * (unsigned)d.nq * (unsigned)d.heads + h) * 32u + lane8 * 4u) * 2u;
  const unsigned lg_q = (unsigned)d.heads * 64u;
  unsigned rf_base = b * (unsigned)d.nq * 16u + (lane8 & 3u) * 4u;
  auto query_of = [&](unsigned i) -> unsigned {
    const unsigned ii = min(i, n_items - 1u);   // octets past the end repeat the last item, unstored
    if constexpr (LISTED == 3) return (unsigned)*(const lds_u16 *)(size_t)(qlist_a + ii * 4u);   // low half of the entry
    else return LISTED != 0 ? q0 + (unsigned)*(const lds_u16 *)(size_t)(qlist_a + ii * 2u) : q0 + ii;
  };
  auto request = [&](H5Set &s, unsigned i) __attribute__((always_inline)) {
    const unsigned q = query_of(i);
    const unsigned o_lg = lg_base + q * lg_q;
    // read-once full lines: non-temporal (shared offsets / logits are re-read by every camera: default policy)
    u32x2 g;
    if constexpr (LISTED >= 2) {
      g = __builtin_amdgcn_raw_buffer_load_b64(rs_lg, (int)o_lg, 0, 0);
      s.of = __builtin_amdgcn_raw_buffer_load_b128(rs_of, (int)(2u * o_lg), 0, 0);
    } else {
      g = __builtin_amdgcn_raw_buffer_load_b64(rs_lg, (int)o_lg, 0, 2);
      s.of = __builtin_amdgcn_raw_buffer_load_b128(rs_of, (int)(2u * o_lg), 0, 2);
    }
    s.lg[0] = g.x; s.lg[1] = g.y;
    s.rf = __builtin_amdgcn_raw_buffer_load_b32(rs_rf, (int)(rf_base + q * 16u), 0, 0);
  };

  // state carried from one phase / iteration to the next
  float m = 0.f, ssum = 0.f;
  u32x4 rec;
  auto fe_begin = [&](const H5Set &s) __attribute__((always_inline)) {
    const float a = fmaxf(fmaxf(h2f_lo(s.lg[0]), h2f_hi(s.lg[0])), fmaxf(h2f_lo(s.lg[1]), h2f_hi(s.lg[1])));
    m = oct_max(a);
    ssum = 0.f;
  };
  auto fe = [&](const H5Set &s, auto jc) __attribute__((always_inline)) {
    constexpr int J = decltype(jc)::v;
    const float lgv = (J & 1) ? h2f_hi(s.lg[J >> 1]) : h2f_lo(s.lg[J >> 1]);
    const float e = __expf(lgv - m);
    ssum += e;
    const unsigned ofj = J == 0 ? s.of.x : J == 1 ? s.of.y : J == 2 ? s.of.z : s.of.w;
    H5Loc p = h5_locate(quad_bcast<J>(s.rf), ofj, c);
    // outside the range gate (incl. non-finite locations: reference points of pillars behind a camera
    // overflow binary16) -> exactly 0, never NaN * 0
    if (!p.valid) { p.x = 0.f; p.y = 0.f; }
    const float xf = floorf(p.x), yf = floorf(p.y);
    const float lx = p.x - xf, ly = p.y - yf;
    const float ev = p.valid ? e : 0.f;
    const float wr1 = ly * ev, wr0 = ev - wr1;
    const float b0 = wr0 * lx, b1 = wr1 * lx;
    rec.x = pack_h2(wr0 - b0, b0);
    rec.y = pack_h2(wr1 - b1, b1);
    const int rel = __mul24((int)yf + 1, c.wp) + (int)xf;
    rec.z = c.base + ((p.valid ? (unsigned)rel : 0u) << c.sh);
    rec.w = rec.z + c.row;
  };

  float acc[4];
  // record of slot S for every lane of the octet, by DPP broadcast
  u32x4 rlo, rhi;   // records of slot (lane & 3) and of slot 4 + (lane & 3)
  auto spread = [&]() __attribute__((always_inline)) {
    // lanes 4..7 of every octet take lane - 4's record (row_shr:4 into banks 1, 3), the others keep
    // their own; and the other way round (row_shl:4 into banks 0, 2)
    rlo.x = (unsigned)__builtin_amdgcn_update_dpp((int)rec.x, (int)rec.x, 0x114, 0xf, 0xa, false);
    rlo.y = (unsigned)__builtin_amdgcn_update_dpp((int)rec.y, (int)rec.y, 0x114, 0xf, 0xa, false);
    rlo.z = (unsigned)__builtin_amdgcn_update_dpp((int)rec.z, (int)rec.z, 0x114, 0xf, 0xa, false);
    rhi.x = (unsigned)__builtin_amdgcn_update_dpp((int)rec.x, (int)rec.x, 0x104, 0xf, 0x5, false);
    rhi.y = (unsigned)__builtin_amdgcn_update_dpp((int)rec.y, (int)rec.y, 0x104, 0xf, 0x5, false);
    rhi.z = (unsigned)__builtin_amdgcn_update_dpp((int)rec.z, (int)rec.z, 0x104, 0xf, 0x5, false);
  };
  auto record = [&](auto sc) __attribute__((always_inline)) -> u32x4 {
    constexpr int S = decltype(sc)::v;
    const u32x4 &src = S < 4 ? rlo : rhi;
    u32x4 r;
    r.x = quad_bcast<S & 3>(src.x);
    r.y = quad_bcast<S & 3>(src.y);
    r.z = quad_bcast<S & 3>(src.z);
    // second row: the slot's level is lane-independent (slot S serves level S / 2)
    r.w = r.z + (((unsigned)t.W[S >> 1] + 1u) << ((S >> 1) >= t.ls ? 6 : 7));
    return r;
  };
  // one item per octet: 4 phases; `cur` = this item's operands, `nxt` = the next item's (landed)
  auto body = [&](H5Set &cur, const H5Set &nxt, unsigned i, H5Set &far) __attribute__((always_inline)) {
    acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
    float s_cur = 0.f;
    auto phase = [&](auto jc) __attribute__((always_inline)) {
      constexpr int J = decltype(jc)::v;
      // the segment that ends with the big-level loads runs at raised priority: a wave that is about to feed
      // the L2 path goes before waves that are in their multiply-add segments (518 vs 524 us)
      __builtin_amdgcn_s_setprio(3);
      spread();
      if constexpr (FOLD) {
        static_assert(!FOLD || (NB == 4 && NS == 4), "the folded phase is written for 4 big + 4 staged slots");
        const unsigned row_b0 = ((unsigned)t.W[0] + 1u) << 7, row_b1 = ((unsigned)t.W[1] + 1u) << 7;
        const unsigned row_s2 = ((unsigned)t.W[2] + 1u) << 6, row_s3 = ((unsigned)t.W[3] + 1u) << 6;
        // big levels: slot S = lane S of every quad (rlo); its address is one add with a DPP source
        u32x4 r0[4], r1[4];
        {
          const unsigned a0 = quad_bcast<0>(rlo.z) + lane16, a1 = quad_bcast<1>(rlo.z) + lane16;
          const unsigned a2 = quad_bcast<2>(rlo.z) + lane16, a3 = quad_bcast<3>(rlo.z) + lane16;
          r0[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a0, 0, 0);
          r1[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(a0 + row_b0), 0, 0);
          r0[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a1, 0, 0);
          r1[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(a1 + row_b0), 0, 0);
          r0[2] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a2, 0, 0);
          r1[2] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(a2 + row_b1), 0, 0);
          r0[3] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)a3, 0, 0);
          r1[3] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(a3 + row_b1), 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // staged levels, two slots at a time (slots 4, 5 = level 2; 6, 7 = level 3; records in rhi)
        unsigned wx[2], wy[2];
        u32x2 l0[2], q0r[2], l1[2], q1r[2];
        auto lds_pair = [&](auto hc) __attribute__((always_inline)) {
          constexpr int Hh = decltype(hc)::v;
          const unsigned row = Hh == 0 ? row_s2 : row_s3;
          constexpr int SA = 2 * Hh, SB = 2 * Hh + 1;     // lanes of the quad (rhi holds slot 4 + (lane & 3))
          wx[0] = quad_bcast<SA>(rhi.x); wy[0] = quad_bcast<SA>(rhi.y);
          wx[1] = quad_bcast<SB>(rhi.x); wy[1] = quad_bcast<SB>(rhi.y);
          const unsigned aA = quad_bcast<SA>(rhi.z) + lane8b, aB = quad_bcast<SB>(rhi.z) + lane8b;
          unsigned aAr = aA + (unsigned)kLdsPixBytes, aA1 = aA + row, aA1r = aA + row + (unsigned)kLdsPixBytes;
          unsigned aBr = aB + (unsigned)kLdsPixBytes, aB1 = aB + row, aB1r = aB + row + (unsigned)kLdsPixBytes;
          // (not laundered: the two ds_read_b64 of a row fuse into one ds_read2_b64 -- see the kernel's header)
          l0[0] = *(const lds_u2 *)(size_t)aA;   q0r[0] = *(const lds_u2 *)(size_t)aAr;
          l1[0] = *(const lds_u2 *)(size_t)aA1;  q1r[0] = *(const lds_u2 *)(size_t)aA1r;
          l0[1] = *(const lds_u2 *)(size_t)aB;   q0r[1] = *(const lds_u2 *)(size_t)aBr;
          l1[1] = *(const lds_u2 *)(size_t)aB1;  q1r[1] = *(const lds_u2 *)(size_t)aB1r;
        };
        auto lds_math2 = [&]() __attribute__((always_inline)) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const h2_t w0 = as_h2(wx[s]), w1 = as_h2(wy[s]);
            const h2_t w00 = {w0[0], w0[0]}, w01 = {w0[1], w0[1]}, w10 = {w1[0], w1[0]}, w11 = {w1[1], w1[1]};
            h2_t a = as_h2(l0[s].x) * w00, bb = as_h2(l0[s].y) * w00;
            a = as_h2(q0r[s].x) * w01 + a; bb = as_h2(q0r[s].y) * w01 + bb;
            a = as_h2(l1[s].x) * w10 + a; bb = as_h2(l1[s].y) * w10 + bb;
            a = as_h2(q1r[s].x) * w11 + a; bb = as_h2(q1r[s].y) * w11 + bb;
            add_h2(acc[0], acc[1], a);
            add_h2(acc[2], acc[3], bb);
          }
        };
        // sixteen dots of two big slots: weights as the DPP source (lane SA / SB of the quad), hazards padded by hand
        // (a VALU write needs 2 states before a DPP read of it; a DOT result 3 states before another VALU reads it --
        // invisible to the compiler inside an asm statement)
// one slot = 8 dots (one asm statement per slot: the compiler then waits for THAT slot's two loads only).  The DPP
// source registers (rlo / rhi) are written by spread() at the head of the phase, far more than the 2 wait states a DPP
// read needs; TAIL = "\n\ts_nop 2" on the last slot of a run (the next VALU that reads an accumulator may follow within
// the 3 states a DOT result needs; between dots of one opcode chained through the accumulator there is no hazard)
#define H5_DOT8(WX, WY, S, A0, A1, TAIL)                                                                              \
        asm("v_dot2c_f32_f16_dpp %0, %4, %6 quad_perm:[" #S "," #S "," #S "," #S "] row_mask:0xf bank_mask:0xf\n\t"     \
            "v_dot2c_f32_f16_dpp %1, %4, %7 quad_perm:[" #S "," #S "," #S "," #S "] row_mask:0xf bank_mask:0xf\n\t"     \
            "v_dot2c_f32_f16_dpp %2, %4, %8 quad_perm:[" #S "," #S "," #S "," #S "] row_mask:0xf bank_mask:0xf\n\t"     \
            "v_dot2c_f32_f16_dpp %3, %4, %9 quad_perm:[" #S "," #S "," #S "," #S "] row_mask:0xf bank_mask:0xf\n\t"     \
            "v_dot2c_f32_f16_dpp %0, %5, %10 quad_perm:[" #S "," #S "," #S "," #S "] row_mask:0xf bank_mask:0xf\n\t"    \
            "v_dot2c_f32_f16_dpp %1, %5, %11 quad_perm:[" #S "," #S "," #S "," #S "] row_mask:0xf bank_mask:0xf\n\t"    \
            "v_dot2c_f32_f16_dpp %2, %5, %12 quad_perm:[" #S "," #S "," #S "," #S "] row_mask:0xf bank_mask:0xf\n\t"    \
            "v_dot2c_f32_f16_dpp %3, %5, %13 quad_perm:[" #S "," #S "," #S "," #S "] row_mask:0xf bank_mask:0xf" TAIL   \
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                                  \
            : "v"(WX), "v"(WY), "v"(A0.x), "v"(A0.y), "v"(A0.z), "v"(A0.w), "v"(A1.x), "v"(A1.y), "v"(A1.z), "v"(A1.w))
#define H5_DOTS(WX, WY, SA, SB, A0, A1, B0, B1) \
        H5_DOT8(WX, WY, SA, A0, A1, "");         \
        H5_DOT8(WX, WY, SB, B0, B1, "\n\ts_nop 2")
        lds_pair(IC<0>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (J < 3) {
          fe(cur, IC<J + 1>{});
        } else {
          s_cur = ssum;
          fe_begin(nxt);
          fe(nxt, IC<0>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        lds_math2();
        __builtin_amdgcn_sched_barrier(0);
        lds_pair(IC<1>{});
        __builtin_amdgcn_sched_barrier(0);
        H5_DOTS(rlo.x, rlo.y, 0, 1, r0[0], r1[0], r0[1], r1[1]);
        __builtin_amdgcn_sched_barrier(0);
        lds_math2();
        __builtin_amdgcn_sched_barrier(0);
        H5_DOTS(rlo.x, rlo.y, 2, 3, r0[2], r1[2], r0[3], r1[3]);
        __builtin_amdgcn_sched_barrier(0);
#undef H5_DOTS
#undef H5_DOT8
        return;
      }
      // big levels: records, then all 2 * NB loads
      u32x4 rb[NB > 0 ? NB : 1];
      u32x4 r0[NB > 0 ? NB : 1], r1[NB > 0 ? NB : 1];
      if constexpr (NB > 0) rb[0] = record(IC<0>{});
      if constexpr (NB > 1) rb[1] = record(IC<1>{});
      if constexpr (NB > 2) rb[2] = record(IC<2>{});
      if constexpr (NB > 3) rb[3] = record(IC<3>{});
      if constexpr (NB > 4) rb[4] = record(IC<4>{});
      if constexpr (NB > 5) rb[5] = record(IC<5>{});
      if constexpr (NB > 6) rb[6] = record(IC<6>{});
      if constexpr (NB > 7) rb[7] = record(IC<7>{});
#pragma unroll
      for (int s = 0; s < NB; ++s) {
        r0[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(rb[s].z + lane16), 0, 0);
        r1[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(rb[s].w + lane16), 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      // staged levels in two halves: records, LDS taps (two ds_read_b64 per row: the laundered second
      // address keeps the compiler from fusing them into the half-rate ds_read2_b64), packed-fp16 row
      // blend, fp32 accumulation.  The front end of the next phase and the big-level multiply-adds
      // sit between the LDS reads and their use
      constexpr int HS = (NS + 1) / 2, HB = (NB + 1) / 2;
      u32x4 rl[HS > 0 ? HS : 1];
      u32x2 l0[HS > 0 ? HS : 1], q0r[HS > 0 ? HS : 1], l1[HS > 0 ? HS : 1], q1r[HS > 0 ? HS : 1];
      auto lds_one = [&](auto sc, auto ic) __attribute__((always_inline)) {
        constexpr int s = decltype(ic)::v;
        if constexpr (decltype(sc)::v < 8 && s < HS) {
            rl[s] = record(sc);
            const unsigned a0 = rl[s].z + lane8b, a1 = rl[s].w + lane8b;
            unsigned a0r = a0 + (unsigned)kLdsPixBytes, a1r = a1 + (unsigned)kLdsPixBytes;
            asm("" : "+v"(a0r));
            asm("" : "+v"(a1r));
            l0[s] = *(const lds_u2 *)(size_t)a0;
            q0r[s] = *(const lds_u2 *)(size_t)a0r;
            l1[s] = *(const lds_u2 *)(size_t)a1;
            q1r[s] = *(const lds_u2 *)(size_t)a1r;
        }
      };
      // half `H` (0 / 1) of the staged slots
      auto lds_issue = [&](auto hc) __attribute__((always_inline)) {
        constexpr int H0 = decltype(hc)::v * HS;
        if constexpr (H0 + 0 < NS) lds_one(IC<NB + H0 + 0>{}, IC<0>{});
        if constexpr (H0 + 1 < NS && HS > 1) lds_one(IC<NB + H0 + 1>{}, IC<1>{});
        if constexpr (H0 + 2 < NS && HS > 2) lds_one(IC<NB + H0 + 2>{}, IC<2>{});
        if constexpr (H0 + 3 < NS && HS > 3) lds_one(IC<NB + H0 + 3>{}, IC<3>{});
      };
      auto lds_math = [&](int n) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < HS; ++s) {
          if (s >= n) break;
          const h2_t w0 = as_h2(rl[s].x), w1 = as_h2(rl[s].y);
          const h2_t w00 = {w0[0], w0[0]}, w01 = {w0[1], w0[1]}, w10 = {w1[0], w1[0]}, w11 = {w1[1], w1[1]};
          h2_t a = as_h2(l0[s].x) * w00, bb = as_h2(l0[s].y) * w00;
          a = as_h2(q0r[s].x) * w01 + a; bb = as_h2(q0r[s].y) * w01 + bb;
          a = as_h2(l1[s].x) * w10 + a; bb = as_h2(l1[s].y) * w10 + bb;
          a = as_h2(q1r[s].x) * w11 + a; bb = as_h2(q1r[s].y) * w11 + bb;
          add_h2(acc[0], acc[1], a);
          add_h2(acc[2], acc[3], bb);
        }
      };
      auto big_math = [&](int s0, int n) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < HB; ++k) {
          if (k >= n) break;
          const int s = s0 + k;
          acc[0] = dot2f(r0[s].x, rb[s].x, acc[0]); acc[1] = dot2f(r0[s].y, rb[s].x, acc[1]);
          acc[2] = dot2f(r0[s].z, rb[s].x, acc[2]); acc[3] = dot2f(r0[s].w, rb[s].x, acc[3]);
          acc[0] = dot2f(r1[s].x, rb[s].y, acc[0]); acc[1] = dot2f(r1[s].y, rb[s].y, acc[1]);
          acc[2] = dot2f(r1[s].z, rb[s].y, acc[2]); acc[3] = dot2f(r1[s].w, rb[s].y, acc[3]);
        }
      };
      lds_issue(IC<0>{});
      __builtin_amdgcn_sched_barrier(0);
      // front end of the next phase (of the next item after the last one)
      if constexpr (J < 3) {
        fe(cur, IC<J + 1>{});
      } else {
        s_cur = ssum;
        fe_begin(nxt);
        fe(nxt, IC<0>{});
      }
      __builtin_amdgcn_sched_barrier(0);
      lds_math(HS);
      __builtin_amdgcn_sched_barrier(0);
      lds_issue(IC<1>{});
      __builtin_amdgcn_sched_barrier(0);
      big_math(0, HB);
      __builtin_amdgcn_sched_barrier(0);
      lds_math(NS - HS);
      __builtin_amdgcn_sched_barrier(0);
      big_math(HB, NB - HB);
      __builtin_amdgcn_sched_barrier(0);
    };
    phase(IC<0>{});
    phase(IC<1>{});
    phase(IC<2>{});
    phase(IC<3>{});
    const float s = oct_sum(s_cur);
    const float inv = __builtin_amdgcn_rcpf(s);
    {
      // unconditional: an octet past the end of the list recomputes the last item and stores the
      // same bytes again (a conditional store lets the compiler sink the last phase's loads into
      // the branch, behind the LDS taps they are meant to overlap)
      u32x2 v;
      v.x = pack_h2(acc[0] * inv, acc[1] * inv);
      v.y = pack_h2(acc[2] * inv, acc[3] * inv);
      if constexpr (LISTED == 3) {
        // two unconditional stores, one of them past the end of its buffer (dropped by the range check): the
        // per-camera scratch row, or -- a pair only this camera sees -- the op's output row itself
        const unsigned e = *(const lds_u32 *)(size_t)(qlist_a + min(i, n_items - 1u) * 4u);
        const unsigned q = e & 0xffffu;
        const bool sole = (e & kSoleBit) != 0u;
        const unsigned none = 0xfffffff0u;
        __builtin_amdgcn_raw_buffer_store_b64(v, rs_out, (int)(sole ? none : out_base + q * out_q), 0, 2);
        // (the reduce computes fma(1, v, +0): a -0 of the binary16 rounding becomes +0 there, so it does here)
        const h2_t zero = {(_Float16)0.f, (_Float16)0.f};
        u32x2 vd;
        vd.x = __builtin_bit_cast(unsigned, as_h2(v.x) + zero);
        vd.y = __builtin_bit_cast(unsigned, as_h2(v.y) + zero);
        __builtin_amdgcn_raw_buffer_store_b64(vd, rs_dir, (int)(sole ? h * 64u + lane8b + q * out_q : none), 0, 2);
      } else {
        const unsigned q = query_of(i);
        __builtin_amdgcn_raw_buffer_store_b64(v, rs_out, (int)(out_base + q * out_q), 0, 2);
      }
    }
    // the operand request is the youngest vector memory instruction of the iteration
    cur = nxt;
    const_cast<H5Set &>(nxt) = far;
    request(far, i + 3u * OCT);
  };

  // S0 = this item's operands, S1 = the next item's (landed), S2 = the one after (in flight).  One body
  // per loop trip; the sets move down by register copies: S2 was requested BEFORE this trip's taps,
  // which have all been waited for, so copying it never stalls
  auto run_items = [&]() __attribute__((always_inline)) {
    if (n_items <= wave_first) return;   // this wave's octets are all past the end of the list
    const unsigned nrounds = (n_items - wave_first + OCT - 1u) / OCT;
    H5Set S0, S1, S2;
    unsigned i = threadIdx.x >> 3;
    request(S0, i);
    request(S1, i + OCT);
    request(S2, i + 2u * OCT);
    fe_begin(S0);
    fe(S0, IC<0>{});
    for (unsigned r = 0; r < nrounds; ++r) {
      body(S0, S1, i, S2);
      i += OCT;
    }
  };
  if constexpr (LISTED == 3) {
    const int *counts = reinterpret_cast<const int *>(vis);
    const unsigned nq_pad = (unsigned)h5_plan_pad_dev(d.nq);
    const unsigned *lists = reinterpret_cast<const unsigned *>(vis + kPlanCams * 4);
    unsigned *wl32 = reinterpret_cast<unsigned *>(smem + stage_bytes);
    const unsigned keep = direct ? ~0u : 0xffffu;   // without an output to store into, no pair is "sole"
    const unsigned nb = gridDim.x / (unsigned)d.heads, j = blockIdx.x / (unsigned)d.heads;
    unsigned total = 0;
    for (int cam = 0; cam < d.bs; ++cam) total += (unsigned)__builtin_amdgcn_readfirstlane(counts[cam]);
    // (total <= 16 x 65 535 and nb <= 2 048: the products fit 32 bits; the quotients are wave-uniform -- say so, or the
    // division's VALU expansion drags every loop bound below into vector registers)
    const unsigned p0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(total * j / nb));
    const unsigned p1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(total * (j + 1u) / nb));
    unsigned pre = 0;
    for (int cam = 0; cam < d.bs; ++cam) {
      const unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane(counts[cam]);
      const unsigned lo = max(p0, pre), hi = min(p1, pre + cnt);
      const unsigned first = lo - pre;   // (only meaningful when lo < hi)
      pre += cnt;
      if (lo >= hi) continue;
      b = (unsigned)cam;
      bh = b * (unsigned)d.heads + h;
      c = h5_lane_consts(t, lane8, bh, sbase);
      out_base = (b * (unsigned)d.nq * (unsigned)d.heads + h) * 64u + lane8b;
      rf_base = b * (unsigned)d.nq * 16u + (lane8 & 3u) * 4u;
      const unsigned *src = lists + (size_t)cam * nq_pad + first;
      bool staged = false;
      for (unsigned done = 0; done < hi - lo; done += (unsigned)chunk) {   // pieces of at most `chunk` list entries
        __syncthreads();   // every wave is through with the previous piece's list (and the previous camera's planes)
        n_items = min((unsigned)chunk, hi - lo - done);
        for (unsigned i = threadIdx.x; i < n_items; i += THREADS) wl32[i] = src[done + i] & keep;
        if (!staged) { stage_plane(); staged = true; }
        __syncthreads();
        run_items();
      }
    }
  } else {
    run_items();
  }
}

inline int h5_lds_extra(int threads, int chunk) { (void)threads; return chunk * 2 + 128; }   // query list + wave totals
inline int h5_plan_lds_extra(int chunk) { return chunk * 4; }                                  // the plan's 32-bit entries
constexpr int kH5Chunk = 1280;

template <int LISTED>
int h5_go(const Hm3Plan &pl, const char *gset, const char *sset, const __half *ref, const __half *off,
          const __half *logit, __half *out, const MsdaDims &d, const unsigned char *vis, int chunk, hipStream_t st) {
  const int nchunk = (d.nq + chunk - 1) / chunk;
  const size_t lds = (size_t)pl.stage_bytes + h5_lds_extra(kH5Threads, chunk);
  if (lds > (size_t)kLdsLimit) return BEVOPS_NOT_SUPPORTED;
  if (!ensure_dynamic_lds<msda_hm5_kernel<LISTED>>(lds)) return (int)BEVOPS_FAILURE;
  const unsigned planes = (unsigned)(d.bs * d.heads);
  hipLaunchKernelGGL((msda_hm5_kernel<LISTED>), dim3(planes * (unsigned)nchunk), dim3(kH5Threads), lds, st, gset,
                     (unsigned)pl.g_bytes, sset, ref, off, logit, out, d, pl.t, chunk, nchunk, pl.stage_bytes, vis,
                     (__half *)nullptr);
  return launch_status();
}

}  // namespace

static bool h5_shape_ok(int C, int L, int P, int ppg) { return C == 32 && L == 4 && P == 8 && ppg == 4; }

// The padded-plane layout hm5 reads, for producers that write it directly (tsgemm.hip: the value projection's
// epilogue).  `tab` receives the Hm3Tab (untyped: the struct lives in each translation unit's unnamed
// namespace), `g_room` the bytes reserved for the big set (the staged set follows).
bool msda_hm5_layout(const int32_t *shapes_host, int bs, int heads, int C, int L, int nq, int P, void *tab,
                     size_t *g_room, size_t *s_bytes) {
  Hm3Plan pl;
  if (!h5_shape_ok(C, L, P, 4) || !hm3_plan(shapes_host, bs, heads, L, nq, h5_lds_extra(1024, kH5Chunk), pl) ||
      pl.t.ls != 2)
    return false;
  *static_cast<Hm3Tab *>(tab) = pl.t;
  *g_room = (pl.g_bytes + 127) & ~size_t(127);
  *s_bytes = pl.s_bytes;
  return true;
}

size_t msda_hm5_workspace_bytes(const int32_t *shapes_host, int bs, int heads, int C, int L, int nq, int P) {
  Hm3Plan pl;
  if (!h5_shape_ok(C, L, P, 4) || !hm3_plan(shapes_host, bs, heads, L, nq, h5_lds_extra(1024, kH5Chunk), pl)) return 0;
  const size_t planes = ((pl.g_bytes + 127) & ~size_t(127)) + pl.s_bytes;
  return ((planes + 255) & ~size_t(255)) + (((size_t)bs * nq * heads + 255) & ~size_t(255));   // [planes][visibility bytes]
}

// Fused SCA sampling on the planes `packed` already holds (written by the value projection's GEMM
// epilogue, bevops_value_proj_packed, or by msda_hm3_repack_launch): camera-shared offsets / logits, the
// (camera, query) pairs with bev_mask weight 0 skipped, `sampled` [cams, nq, heads, 32] written for the others.
int msda_hm5_sca_sample_f16(const void *packed, size_t packed_bytes, const int32_t *shapes_host, const __half *ref,
                            const __half *off, const __half *logit, const __half *qmask, __half *sampled, int bs,
                            int nk, int heads, int C, int L, int nq, int P, int ppg, hipStream_t st) {
  Hm3Plan pl;
  if (!h5_shape_ok(C, L, P, ppg) || !packed || (reinterpret_cast<uintptr_t>(packed) & 127u) ||
      !hm3_plan(shapes_host, bs, heads, L, nq, h5_lds_extra(1024, kH5Chunk), pl) || pl.t.ls != 2)
    return BEVOPS_NOT_SUPPORTED;
  if ((double)nq * heads * 32 * 4.0 >= 4294967040.0 || (double)bs * nq * heads * 64.0 >= 4294967040.0) return BEVOPS_NOT_SUPPORTED;
  const size_t g_room = (pl.g_bytes + 127) & ~size_t(127);
  if (packed_bytes < g_room + pl.s_bytes) return BEVOPS_BAD_PARAM;
  const char *gset = static_cast<const char *>(packed);
  const MsdaDims d{bs, nk, heads, C, L, nq, P, ppg, 1};
  return h5_go<2>(pl, gset, gset + g_room, ref, off, logit, sampled, d, reinterpret_cast<const unsigned char *>(qmask),
                  kH5Chunk, st);
}

// ---- the same sampling on a visibility plan (msda_hm5_plan_kernel): balanced slices of the visible (camera, query)
// pairs, `blocks_per_cu` blocks per CU's worth of slices (1: one slice per CU).
// (two slices per CU measured best on the 6-camera rig: 108 us per call against 124 for one block per 1 280-query chunk,
// 112 / 115 with three / four, profiles/r05/sca_plan_ab.jsonl; touching a slice's offset / logit rows ahead of the loop
// so that the per-item requests hit the L2 was built and measured SLOWER, 117 us, and removed)
static thread_local int g_h5_plan_k = 2;
void msda_hm5_set_plan_blocks(int k) { g_h5_plan_k = k < 1 ? 1 : (k > 8 ? 8 : k); }
static thread_local bool g_h5_fold = true;      // the FOLD build of the planned kernel (default) / the round-5 build
void msda_hm5_set_fold(bool on) { g_h5_fold = on; }

size_t msda_hm5_plan_bytes(int bs, int nq) {
  if (bs <= 0 || bs > kPlanCams || nq <= 0 || nq > 65535) return 0;
  return (size_t)kPlanCams * 4 + (size_t)bs * h5_plan_pad(nq) * 4 + (size_t)bs * kPlanBlocks * 4;   // counts, lists, scratch
}

int msda_hm5_plan_build(const __half *qmask, int bs, int nq, void *plan, size_t plan_bytes, hipStream_t st) {
  const size_t need = msda_hm5_plan_bytes(bs, nq);
  if (need == 0) return BEVOPS_NOT_SUPPORTED;
  if (!qmask || !plan || plan_bytes < need || (reinterpret_cast<uintptr_t>(plan) & 15u)) return BEVOPS_BAD_PARAM;
  const unsigned short *mk = reinterpret_cast<const unsigned short *>(qmask);
  unsigned *lists = reinterpret_cast<unsigned *>(static_cast<char *>(plan) + kPlanCams * 4);
  unsigned *partial = lists + (size_t)bs * h5_plan_pad(nq);
  const dim3 grid(((unsigned)nq + 255u) / 256u, (unsigned)bs);
  hipLaunchKernelGGL(msda_hm5_plan_count_kernel, grid, dim3(256), 0, st, mk, nq, partial);
  hipLaunchKernelGGL(msda_hm5_plan_place_kernel, grid, dim3(256), 0, st, mk, bs, nq, (unsigned)h5_plan_pad(nq),
                     static_cast<int *>(plan), lists, partial);
  return launch_status();
}

constexpr int kH5PlanChunk = 2048;   // list entries of a slice kept in LDS at a time
int msda_hm5_sca_sample_planned_f16(const void *packed, size_t packed_bytes, const int32_t *shapes_host,
                                    const __half *ref, const __half *off, const __half *logit, const void *plan,
                                    size_t plan_bytes, __half *sampled, __half *direct, int bs, int nk, int heads, int C,
                                    int L, int nq, int P, int ppg, hipStream_t st) {
  Hm3Plan pl;
  if (!h5_shape_ok(C, L, P, ppg) || !packed || (reinterpret_cast<uintptr_t>(packed) & 127u) ||
      !hm3_plan(shapes_host, bs, heads, L, nq, h5_lds_extra(1024, kH5Chunk), pl) || pl.t.ls != 2)
    return BEVOPS_NOT_SUPPORTED;
  if ((double)nq * heads * 32 * 4.0 >= 4294967040.0 || (double)bs * nq * heads * 64.0 >= 4294967040.0) return BEVOPS_NOT_SUPPORTED;
  const size_t need = msda_hm5_plan_bytes(bs, nq);
  if (need == 0) return BEVOPS_NOT_SUPPORTED;
  // the plan of ANOTHER camera set or query count (a 6-camera plan handed to a camera-sharded rank's subset) has
  // another size: rejected here instead of sampling from the wrong lists
  if (!plan || plan_bytes != need || (reinterpret_cast<uintptr_t>(plan) & 15u)) return BEVOPS_BAD_PARAM;
  const size_t g_room = (pl.g_bytes + 127) & ~size_t(127);
  if (packed_bytes < g_room + pl.s_bytes) return BEVOPS_BAD_PARAM;
  const char *gset = static_cast<const char *>(packed);
  const MsdaDims d{bs, nk, heads, C, L, nq, P, ppg, 1};
  constexpr int THREADS = kH5Threads;
  const size_t lds = (size_t)pl.stage_bytes + h5_plan_lds_extra(kH5PlanChunk);
  if (lds > (size_t)kLdsLimit) return BEVOPS_NOT_SUPPORTED;
  const bool fold = g_h5_fold;
  auto kern = fold ? msda_hm5_kernel<3, true> : msda_hm5_kernel<3, false>;
  if (!(fold ? ensure_dynamic_lds<msda_hm5_kernel<3, true>>(lds) : ensure_dynamic_lds<msda_hm5_kernel<3, false>>(lds)))
    return (int)BEVOPS_FAILURE;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  // slices per head: the CUs an XCD's share of the grid lands on (block i runs on XCD i % 8, head = i % heads)
  const unsigned per_head = (unsigned)((cus > 0 ? cus : 256) * g_h5_plan_k + heads - 1) / (unsigned)heads;
  hipLaunchKernelGGL(kern, dim3(per_head * (unsigned)heads), dim3(THREADS), lds, st, gset, (unsigned)pl.g_bytes,
                     gset + g_room, ref, off, logit, sampled, d, pl.t, kH5PlanChunk, 1, pl.stage_bytes,
                     static_cast<const unsigned char *>(plan), direct);
  return launch_status();
}

// flags (A/B switch of the tests, bevops_msda_set_variant(1000 + flags)): 1 = no visibility pre-pass (every item is
// sampled; the partner of the default).  The other builds of rounds 3 / 4 -- 768-thread blocks, 2 560-query chunks,
// records through an LDS mailbox, persistent blocks on strided sub-chunks, no raised priority, the level-class split
// probe and the ablation (timing) builds -- were measured (design/msda.md, profiles/r03, profiles/r04) and removed from
// the library in round 5, their template parameters with them.
int msda_hm5_forward_f16(const __half *value, const int32_t *shapes_host, const __half *ref, const __half *off,
                         const __half *logit, __half *out, int bs, int nk, int heads, int C, int L, int nq, int P,
                         int ppg, int shared, void *workspace, size_t workspace_bytes, int flags, bool prepacked,
                         hipStream_t st) {
  Hm3Plan pl;
  if (!h5_shape_ok(C, L, P, ppg) || shared || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 127u) ||
      !hm3_plan(shapes_host, bs, heads, L, nq, h5_lds_extra(1024, kH5Chunk), pl))
    return BEVOPS_NOT_SUPPORTED;
  if (pl.t.ls != 2) return BEVOPS_NOT_SUPPORTED;   // two big + two staged levels (the base SCA pyramid)
  if ((double)bs * nq * heads * 32 * 4.0 >= 4294967040.0) return BEVOPS_NOT_SUPPORTED;  // 32-bit offsets
  if (workspace_bytes < msda_hm5_workspace_bytes(shapes_host, bs, heads, C, L, nq, P)) return BEVOPS_NOT_SUPPORTED;
  if (flags & ~1) return BEVOPS_NOT_SUPPORTED;
  const size_t g_room = (pl.g_bytes + 127) & ~size_t(127);
  char *gset = static_cast<char *>(workspace);
  char *sset = gset + g_room;
  unsigned char *vis = reinterpret_cast<unsigned char *>(gset + ((g_room + pl.s_bytes + 255) & ~size_t(255)));
  if (!prepacked) msda_hm3_repack_launch(value, gset, sset, &pl.t, bs, nk, heads, st);
  const MsdaDims d{bs, nk, heads, C, L, nq, P, ppg, shared};
  if (flags & 1) return h5_go<0>(pl, gset, sset, ref, off, logit, out, d, vis, kH5Chunk, st);
  const unsigned n_pair = (unsigned)bs * (unsigned)nq;
  const unsigned waves = (n_pair + 63u) / 64u;
  hipLaunchKernelGGL(msda_hm5_vis_kernel, dim3((waves + 3) / 4), dim3(256), 0, st, ref, off, out, vis, d, pl.t, n_pair);
  return h5_go<1>(pl, gset, sset, ref, off, logit, out, d, vis, kH5Chunk, st);
}

}  // namespace bevops
