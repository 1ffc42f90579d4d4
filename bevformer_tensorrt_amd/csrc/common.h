// Shared device/host vocabulary for libbevops_hip.so (gfx950 only).
// Restates, MI355X-first, what the reference keeps in TensorRT/common/
// {cuda_helper.h:15-23 (launch geometry), cuda_int8.h:11-53 (packed int8),
// helper.h:19-25 (status codes)}.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/bevops.h"

namespace bevops {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;  // CDNA wavefront

// single roundings that must not be contracted into an fma with a neighbouring multiply (the
// int8 softmax: logit * scale - max; the oracle and the reference's host build round each step)
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}

// XCD-aware block remap: the dispatcher places block b on XCD b % 8
// (MI355X_MICROARCH.md "Workgroup dispatch"); give every XCD one contiguous
// chunk of the logical grid so neighbouring work shares that XCD's 4 MiB L2.
// Bijective for any grid size.  Speed only -- never relied on for correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  constexpr unsigned kXcd = 8;
  const unsigned q = nblk / kXcd, r = nblk % kXcd;
  const unsigned x = bid % kXcd, i = bid / kXcd;
  return x * q + (x < r ? x : r) + i;
}

// quad (4-lane) broadcast / reduce via DPP quad_perm: no LDS, one VALU op.
template <int S>
__device__ __forceinline__ float quad_bcast(float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(0, __float_as_int(v), S * 0x55, 0xf, 0xf, true));
}
template <int S>
__device__ __forceinline__ unsigned quad_bcast(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, S * 0x55, 0xf, 0xf, true);
}
__device__ __forceinline__ float quad_xor1(float v) {  // lanes {1,0,3,2}
  return __int_as_float(
      __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_xor2(float v) {  // lanes {2,3,0,1}
  return __int_as_float(
      __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, quad_xor1(v));
  return fmaxf(v, quad_xor2(v));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += quad_xor1(v);
  return v + quad_xor2(v);
}

__device__ __forceinline__ float h2f_lo(unsigned u) {
  return __half2float(__ushort_as_half((unsigned short)(u & 0xffffu)));
}
__device__ __forceinline__ float h2f_hi(unsigned u) {
  return __half2float(__ushort_as_half((unsigned short)(u >> 16)));
}
__device__ __forceinline__ unsigned pack_h2(float a, float b) {  // one v_cvt_pk_f16_f32 (RNE)
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  typedef float f2v __attribute__((ext_vector_type(2)));
  const f2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2v));
}

inline int launch_status() {
  return hipGetLastError() == hipSuccess ? BEVOPS_SUCCESS : BEVOPS_FAILURE;
}
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// hipFuncAttributeMaxDynamicSharedMemorySize, set once per (kernel instance, device) and raised
// only when a launch needs more than what was set before -- not on every launch
template <auto Kern>
inline bool ensure_dynamic_lds(size_t lds) {
  static std::atomic<int> have[16];
  if (lds <= 64 * 1024) return true;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::atomic<int> &slot = have[dev & 15];
  if ((int)lds <= slot.load(std::memory_order_acquire)) return true;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return false;
  slot.store((int)lds, std::memory_order_release);
  return true;
}

}  // namespace bevops
