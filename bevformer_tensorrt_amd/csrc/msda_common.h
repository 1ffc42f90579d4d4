// Shared pieces of the MSDA kernels (msda.hip: quad / generic / int8 kernels on the
// reference layout; msda_hm.hip: head-major re-layout path).
#pragma once
#include "common.h"

namespace bevops {
namespace {

constexpr int kMaxLevels = 16;
constexpr int kBlock = 256;

struct MsdaDims {
  int bs, nk, heads, C, L, nq, P, ppg;
  int shared;  // 1: sampling_offsets / attention_weights are [1, nq, heads, .] shared by all bs
};

// ---------------------------------------------------------------------------
// element loaders: N consecutive T -> float
// ---------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void load_f(const float *p, float (&d)[N]) {
  if constexpr (N == 1) {
    d[0] = p[0];
  } else if constexpr (N == 2) {
    const float2 v = *reinterpret_cast<const float2 *>(p);
    d[0] = v.x; d[1] = v.y;
  } else {
    static_assert(N % 4 == 0, "N");
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      const float4 v = reinterpret_cast<const float4 *>(p)[i];
      d[4 * i] = v.x; d[4 * i + 1] = v.y; d[4 * i + 2] = v.z; d[4 * i + 3] = v.w;
    }
  }
}
template <int N>
__device__ __forceinline__ void load_f(const __half *p, float (&d)[N]) {
  if constexpr (N == 1) {
    d[0] = __half2float(p[0]);
  } else if constexpr (N == 2) {
    const unsigned v = *reinterpret_cast<const unsigned *>(p);
    d[0] = h2f_lo(v); d[1] = h2f_hi(v);
  } else if constexpr (N == 4) {
    const uint2 v = *reinterpret_cast<const uint2 *>(p);
    d[0] = h2f_lo(v.x); d[1] = h2f_hi(v.x); d[2] = h2f_lo(v.y); d[3] = h2f_hi(v.y);
  } else {
    static_assert(N % 8 == 0, "N");
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
      const uint4 v = reinterpret_cast<const uint4 *>(p)[i];
      d[8 * i] = h2f_lo(v.x); d[8 * i + 1] = h2f_hi(v.x);
      d[8 * i + 2] = h2f_lo(v.y); d[8 * i + 3] = h2f_hi(v.y);
      d[8 * i + 4] = h2f_lo(v.z); d[8 * i + 5] = h2f_hi(v.z);
      d[8 * i + 6] = h2f_lo(v.w); d[8 * i + 7] = h2f_hi(v.w);
    }
  }
}
// streaming (read-once) operands: non-temporal so they do not evict the gathered value maps
// from the XCD's L2
template <typename V>
__device__ __forceinline__ V ld_nt(const void *p) {
  return __builtin_nontemporal_load(reinterpret_cast<const V *>(p));
}
template <int N>
__device__ __forceinline__ void load_f_nt(const __half *p, float (&d)[N]) {
  if constexpr (N == 1) {
    d[0] = __half2float(__ushort_as_half(ld_nt<unsigned short>(p)));
  } else if constexpr (N == 2) {
    const unsigned v = ld_nt<unsigned>(p);
    d[0] = h2f_lo(v); d[1] = h2f_hi(v);
  } else if constexpr (N == 4) {
    const unsigned long long v = ld_nt<unsigned long long>(p);
    const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    d[0] = h2f_lo(lo); d[1] = h2f_hi(lo); d[2] = h2f_lo(hi); d[3] = h2f_hi(hi);
  } else {
    static_assert(N % 8 == 0, "N");
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
      const u32x4 v = ld_nt<u32x4>(p + 8 * i);
      d[8 * i] = h2f_lo(v.x); d[8 * i + 1] = h2f_hi(v.x);
      d[8 * i + 2] = h2f_lo(v.y); d[8 * i + 3] = h2f_hi(v.y);
      d[8 * i + 4] = h2f_lo(v.z); d[8 * i + 5] = h2f_hi(v.z);
      d[8 * i + 6] = h2f_lo(v.w); d[8 * i + 7] = h2f_hi(v.w);
    }
  }
}
template <int N>
__device__ __forceinline__ void load_raw_nt(const __half *p, unsigned (&d)[N]) {
  if constexpr (N == 1) {
    d[0] = ld_nt<unsigned>(p);
  } else if constexpr (N == 2) {
    const unsigned long long v = ld_nt<unsigned long long>(p);
    d[0] = (unsigned)v; d[1] = (unsigned)(v >> 32);
  } else {
    static_assert(N % 4 == 0, "N");
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      const u32x4 v = ld_nt<u32x4>(p + 8 * i);
      d[4 * i] = v.x; d[4 * i + 1] = v.y; d[4 * i + 2] = v.z; d[4 * i + 3] = v.w;
    }
  }
}

// N dwords (= N (x, y) half pairs) starting at p
template <int N>
__device__ __forceinline__ void load_raw(const __half *p, unsigned (&d)[N]) {
  if constexpr (N == 1) {
    d[0] = *reinterpret_cast<const unsigned *>(p);
  } else if constexpr (N == 2) {
    const uint2 v = *reinterpret_cast<const uint2 *>(p);
    d[0] = v.x; d[1] = v.y;
  } else {
    static_assert(N % 4 == 0, "N");
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      const uint4 v = reinterpret_cast<const uint4 *>(p)[i];
      d[4 * i] = v.x; d[4 * i + 1] = v.y; d[4 * i + 2] = v.z; d[4 * i + 3] = v.w;
    }
  }
}

__device__ __forceinline__ float2 load_ref(const float *p) {
  return *reinterpret_cast<const float2 *>(p);
}
__device__ __forceinline__ float2 load_ref(const __half *p) {
  const unsigned v = *reinterpret_cast<const unsigned *>(p);
  return make_float2(h2f_lo(v), h2f_hi(v));
}

// one 8-channel tap: acc[c] += w * value[c]
__device__ __forceinline__ void tap8(const __half *, __amdgpu_buffer_rsrc_t rs, unsigned voff,
                                     float w, float (&acc)[8]) {
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0);
  acc[0] = fmaf(w, h2f_lo(r.x), acc[0]); acc[1] = fmaf(w, h2f_hi(r.x), acc[1]);
  acc[2] = fmaf(w, h2f_lo(r.y), acc[2]); acc[3] = fmaf(w, h2f_hi(r.y), acc[3]);
  acc[4] = fmaf(w, h2f_lo(r.z), acc[4]); acc[5] = fmaf(w, h2f_hi(r.z), acc[5]);
  acc[6] = fmaf(w, h2f_lo(r.w), acc[6]); acc[7] = fmaf(w, h2f_hi(r.w), acc[7]);
}
__device__ __forceinline__ void tap8(const float *, __amdgpu_buffer_rsrc_t rs, unsigned voff,
                                     float w, float (&acc)[8]) {
  const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0);
  const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(voff + 16u), 0, 0);
  acc[0] = fmaf(w, __uint_as_float(a.x), acc[0]); acc[1] = fmaf(w, __uint_as_float(a.y), acc[1]);
  acc[2] = fmaf(w, __uint_as_float(a.z), acc[2]); acc[3] = fmaf(w, __uint_as_float(a.w), acc[3]);
  acc[4] = fmaf(w, __uint_as_float(b.x), acc[4]); acc[5] = fmaf(w, __uint_as_float(b.y), acc[5]);
  acc[6] = fmaf(w, __uint_as_float(b.z), acc[6]); acc[7] = fmaf(w, __uint_as_float(b.w), acc[7]);
}
__device__ __forceinline__ void store8(__half *p, const float (&a)[8]) {
  uint4 v;
  v.x = pack_h2(a[0], a[1]); v.y = pack_h2(a[2], a[3]);
  v.z = pack_h2(a[4], a[5]); v.w = pack_h2(a[6], a[7]);
  *reinterpret_cast<uint4 *>(p) = v;
}
__device__ __forceinline__ void store8(float *p, const float (&a)[8]) {
  reinterpret_cast<float4 *>(p)[0] = make_float4(a[0], a[1], a[2], a[3]);
  reinterpret_cast<float4 *>(p)[1] = make_float4(a[4], a[5], a[6], a[7]);
}

// int8 rounding rules of the reference and the byte transpose of the
// dot4 operands
__device__ __forceinline__ int t2i8_away(float a) {  // kernel.cu:44-55
  a = fminf(fmaxf(a, -128.f), 127.f);
  return (int)(a + (a > 0.f ? 0.5f : -0.5f));
}
__device__ __forceinline__ int t2i8_rne(float a) {  // kernel.cu:57-62
  return (int)fminf(fmaxf(rintf(a), -128.f), 127.f);
}
__device__ __forceinline__ unsigned u16_rne(float a) {  // __half2ushort_rn
  return (unsigned)fminf(fmaxf(rintf(a), 0.f), 65535.f);
}
// 4x4 byte transpose: r[k] = 4 channels of corner k  ->  o[c] = channel c of corners 0..3
__device__ __forceinline__ void transpose4x4(unsigned r0, unsigned r1, unsigned r2, unsigned r3,
                                             unsigned (&o)[4]) {
  const unsigned a = __builtin_amdgcn_perm(r1, r0, 0x05010400u);
  const unsigned b = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
  const unsigned c = __builtin_amdgcn_perm(r3, r2, 0x05010400u);
  const unsigned d = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
  o[0] = __builtin_amdgcn_perm(c, a, 0x05040100u);
  o[1] = __builtin_amdgcn_perm(c, a, 0x07060302u);
  o[2] = __builtin_amdgcn_perm(d, b, 0x05040100u);
  o[3] = __builtin_amdgcn_perm(d, b, 0x07060302u);
}

// t2i8_away for a >= 0 (area weights, softmax weights): the clamp from below and the sign select
// drop out ((int)(0 + 0.5) == (int)(0 - 0.5) == 0)
__device__ __forceinline__ int t2i8_away_nonneg(float a) { return (int)add_rn(fminf(a, 127.f), 0.5f); }
// RN(p / sa) for sa = fl(1/127), p >= 0, WITHOUT the 10-instruction IEEE division: 127.0f is the
// correctly rounded reciprocal of sa, so one residual correction of q0 = RN(127 p) gives the
// correctly rounded quotient (Markstein; checked against the division over 40 M products on the
// host and on every tested input against the dividing kernels, tests/test_msda_hm4_gpu.py)
__device__ __forceinline__ float div_by_inv127(float p) {
  const float sa = 1 / 127.f;
  const float q0 = mul_rn(p, 127.f);
  const float r = __builtin_fmaf(-q0, sa, p);
  return __builtin_fmaf(r, 127.f, q0);
}

// location arithmetic kept un-fused so that it rounds exactly like the
// reference's fp32 kernel (mul, add, sub as separate roundings).
__device__ __forceinline__ float loc_im(float ref, float size, float off) {
#pragma clang fp contract(off)
  const float t = ref * size;
  const float u = t + off;
  return u - 0.5f;
}


}  // namespace

// msda_hm.hip -- fp16 head-major path.  Returns BEVOPS_NOT_SUPPORTED when the shape is
// outside its domain (caller falls back to the quad kernel).
size_t msda_hm_workspace_bytes(int bs, int nk, int heads, int C, int L);
int msda_hm_forward_f16(const __half *value, const int32_t *shapes, const int32_t *shapes_host,
                        const __half *ref, const __half *off, const __half *logit, __half *out,
                        int bs, int nk, int heads, int C, int L, int nq, int P, int ppg, int shared,
                        void *workspace, size_t workspace_bytes, int variant, hipStream_t st);
// msda_hm3.hip -- padded head-major path with LDS-resident small levels; needs the shapes on
// the host.  workspace_bytes == 0 / NOT_SUPPORTED when the shape is outside its domain.
size_t msda_hm3_workspace_bytes(const int32_t *shapes_host, int bs, int heads, int C, int L, int nq,
                                int P);
int msda_hm3_forward_f16(const __half *value, const int32_t *shapes_host, const __half *ref,
                         const __half *off, const __half *logit, __half *out, int bs, int nk,
                         int heads, int C, int L, int nq, int P, int ppg, int shared,
                         void *workspace, size_t workspace_bytes, hipStream_t st);
// msda_hm4.hip -- software-pipelined successor of hm3 on the same padded layout idea; fp16 and
// both int8 flavours (dtype BEVOPS_F16 / BEVOPS_I8, ref_dtype selects the int8 flavour).
size_t msda_hm4_workspace_bytes(const int32_t *shapes_host, int bs, int heads, int C, int L, int nq, int P, bool i8);
int msda_hm4_forward(int dtype, int ref_dtype, const void *value, const int32_t *shapes_host, const void *ref,
                     const void *off, const void *logit, void *out, int bs, int nk, int heads, int C, int L,
                     int nq, int P, int ppg, int shared, float s_v, float s_o, float s_w, float s_out,
                     void *workspace, size_t workspace_bytes, int chunk_override, int ablate, hipStream_t st);
bool msda_hm4_all_staged(const int32_t *shapes_host, int bs, int heads, int C, int L, int nq, int P);
void msda_hm4_set_no_occ(bool v);
int msda_hm4_pack(int dtype, int ref_dtype, const void *value, const int32_t *shapes_host, int bs, int nk,
                  int heads, int C, int L, int nq, int P, void *packed, size_t packed_bytes, hipStream_t st);
int msda_hm4_forward_prepacked(int dtype, int ref_dtype, const void *packed, size_t packed_bytes,
                               const int32_t *shapes_host, const void *ref, const void *off, const void *logit,
                               void *out, int bs, int nk, int heads, int C, int L, int nq, int P, int ppg,
                               int shared, float s_v, float s_o, float s_w, float s_out, int chunk_override,
                               int ablate, hipStream_t st);
// msda_hm5.hip -- re-scheduled successor of hm3 for the 4-level x 8-point SCA shape (same planes) with an
// exact visibility pre-pass; `flags`: see msda_hm5_forward_f16
size_t msda_hm5_workspace_bytes(const int32_t *shapes_host, int bs, int heads, int C, int L, int nq, int P);
int msda_hm5_forward_f16(const __half *value, const int32_t *shapes_host, const __half *ref, const __half *off,
                         const __half *logit, __half *out, int bs, int nk, int heads, int C, int L, int nq, int P,
                         int ppg, int shared, void *workspace, size_t workspace_bytes, int flags, bool prepacked,
                         hipStream_t st);
int msda_hm5_sca_sample_f16(const void *packed, size_t packed_bytes, const int32_t *shapes_host, const __half *ref,
                            const __half *off, const __half *logit, const __half *qmask, __half *sampled, int bs,
                            int nk, int heads, int C, int L, int nq, int P, int ppg, hipStream_t st);
// visibility plan of the fused SCA op (per camera the ascending list of its visible queries) and the sampling on it
size_t msda_hm5_plan_bytes(int bs, int nq);
int msda_hm5_plan_build(const __half *qmask, int bs, int nq, void *plan, size_t plan_bytes, hipStream_t st);
int msda_hm5_sca_sample_planned_f16(const void *packed, size_t packed_bytes, const int32_t *shapes_host,
                                    const __half *ref, const __half *off, const __half *logit, const void *plan,
                                    size_t plan_bytes, __half *sampled, __half *direct, int bs, int nk, int heads, int C,
                                    int L, int nq, int P, int ppg, hipStream_t st);
void msda_hm5_set_plan_blocks(int k);
void msda_hm5_set_fold(bool on);
void msda_sca_set_reduce_rolled(bool on);   // A/B partner of the unrolled camera reduce (set_variant 3010 / 3011)
// skip_sole: rows of queries that exactly one camera sees with weight 1 are left alone (the planned sampler has
// stored them already, msda_hm5.hip: kSoleBit)
void msda_sca_reduce_launch(const __half *sampled, const __half *qmask, __half *out, int bs, int nq, int width,
                            bool skip_sole, hipStream_t st);
bool msda_hm5_layout(const int32_t *shapes_host, int bs, int heads, int C, int L, int nq, int P, void *tab,
                     size_t *g_room, size_t *s_bytes);
void msda_hm3_repack_launch(const void *value, char *gset, char *sset, const void *tab, int bs, int nk, int heads,
                            hipStream_t st);
size_t msda_hm3_sca_workspace_bytes(const int32_t *shapes_host, int bs, int heads, int C, int L, int nq,
                                    int P);
int msda_hm3_sca_forward_f16(const __half *value, const int32_t *shapes_host, const __half *ref,
                             const __half *off, const __half *logit, const __half *qmask,
                             __half *out, int bs, int nk, int heads, int C, int L, int nq, int P,
                             int ppg, void *workspace, size_t workspace_bytes, hipStream_t st);
}  // namespace bevops
