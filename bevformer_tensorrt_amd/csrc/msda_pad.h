// Padded head-major value layout shared by msda_hm3.hip (fp16) and msda_hm4.hip (fp16 / int8).
// Every pyramid level is stored per (batch, head) plane with a zero row above and below and one
// zero pixel after each row (the pixel "before" a row is the pad of the previous row), so a
// bilinear footprint never needs per-corner border logic: a sample is valid iff floor(x) + 1 in
// [0, W] and floor(y) + 1 in [0, H], and its four corners always exist.  Levels are split into a
// "big" set read through L1/L2 (128-byte entries) and a "staged" tail that a block keeps in LDS
// (64-byte entries); what an entry holds is the kernel family's business.
#pragma once
#include "msda_common.h"

namespace bevops {
namespace {

constexpr int kHm3MaxLevels = 8;
constexpr int kEntBytes = 128;   // big levels (fp16: pixel pair; int8: 2x2 footprint)
constexpr int kLdsPixBytes = 64; // staged levels (fp16: one pixel; int8: pixel pair)
constexpr int kLdsLimit = 160 * 1024;

struct Hm3Tab {
  int L, ls;                    // levels, first LDS-staged level (== L: none)
  int H[kHm3MaxLevels], W[kHm3MaxLevels];
  int ent0[kHm3MaxLevels];      // entry index of padded (row 0, col 0) in its set (big / staged)
  int src0[kHm3MaxLevels];      // first source pixel of the level
  int g_entries, s_entries;     // entries per (batch, head) plane of each set
};

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

// padded-set entry -> source pixel of the level it falls in, or -1 for a pad
__device__ __forceinline__ int hm3_source(const Hm3Tab &t, int l0, int l1, int f) {
  int src = -1;
  for (int l = l0; l < l1; ++l) {
    const int Wp = t.W[l] + 1;
    const int rel = f - t.ent0[l];
    if (rel >= 0 && rel < (t.H[l] + 2) * Wp) {
      const int yp = rel / Wp, x = rel - yp * Wp;
      if (yp >= 1 && yp <= t.H[l] && x < t.W[l]) src = t.src0[l] + (yp - 1) * t.W[l] + x;
    }
  }
  return src;
}

// level table in LDS, 32 B per level: {float W, float H, u32 byte offset of entry (row 0, col 0),
// u32 row bytes} {i32 W + 1, u32 log2(bytes per entry), -, -}
constexpr int kTabEnt = 32;
constexpr int kTab = kHm3MaxLevels * kTabEnt;

// host: padded-set layout.  Set = one leading zero entry, then per level (H+2) rows of (W+1)
// entries, then one trailing zero entry (the pair partner of the last one).
struct Hm3Plan {
  Hm3Tab t;
  size_t g_bytes, s_bytes;  // whole sets (all batches and heads)
  int stage_bytes;          // per (batch, head) staged plane
  int threads;
};

// mailbox bytes of msda_hm3_kernel's 1024-thread block (8-point phases)
inline int hm3_box_bytes(int LP) { return 128 * ((LP >= 8 ? 8 : LP) * 16 + 16); }

// `box_bytes`: LDS the kernel needs next to the level table and the staged planes
inline bool hm3_plan(const int32_t *shapes_host, int bs, int heads, int L, int nq, int box_bytes, Hm3Plan &pl) {
  if (!shapes_host || L > kHm3MaxLevels) return false;
  Hm3Tab &t = pl.t;
  t.L = L;
  int src = 0;
  for (int l = 0; l < L; ++l) {
    t.H[l] = shapes_host[2 * l];
    t.W[l] = shapes_host[2 * l + 1];
    if (t.H[l] > 0x7fff || t.W[l] > 0x7fff) return false;
    t.src0[l] = src;
    src += t.H[l] * t.W[l];
  }
  auto padded = [&](int l) { return (t.H[l] + 2) * (t.W[l] + 1); };
  // longest tail of levels whose padded planes fit in LDS next to the block's mailboxes;
  // staging a plane per block only pays with enough queries per plane
  const int budget = kLdsLimit - kTab - box_bytes;
  int ls = L;
  if (nq >= 2048) {
    long tail = 2;
    for (int l = L - 1; l >= 0; --l) {
      tail += padded(l);
      if (tail * kLdsPixBytes > budget) break;
      ls = l;
    }
  }
  t.ls = ls;
  int e = 1;
  for (int l = 0; l < ls; ++l) { t.ent0[l] = e; e += padded(l); }
  t.g_entries = e + 1;
  e = 1;
  for (int l = ls; l < L; ++l) { t.ent0[l] = e; e += padded(l); }
  t.s_entries = ls < L ? e + 1 : 0;
  pl.stage_bytes = t.s_entries * kLdsPixBytes;
  pl.g_bytes = (size_t)bs * heads * t.g_entries * kEntBytes;
  pl.s_bytes = (((size_t)bs * heads * pl.stage_bytes) + 127) & ~size_t(127);
  pl.threads = ls < L ? 1024 : 256;
  return pl.g_bytes + 128 < 0xFFFFFF00ull;
}

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float oct_max(float v) {
  v = quad_max(v);
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true)));
}
__device__ __forceinline__ float oct_sum(float v) {
  v = quad_sum(v);
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
}
__device__ __forceinline__ float dot2f(unsigned pair, unsigned w, float acc) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, pair), __builtin_bit_cast(h2_t, w), acc, false);
}
__device__ __forceinline__ h2_t as_h2(unsigned u) { return __builtin_bit_cast(h2_t, u); }

// acc += (float)lo/hi half of a packed fp16 pair -- v_fma_mix_f32 with the constant 1.0 (the
// compiler emits v_cvt_f32_f16 + v_add_f32 for the plain C++ form)
__device__ __forceinline__ void add_h2(float &a0, float &a1, h2_t v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(a0) : "v"(u));
  asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a1) : "v"(u));
}

}  // namespace
}  // namespace bevops
