/* cuda_on_cpu/cuda_fp16.h -- TEST INFRASTRUCTURE ONLY.  IEEE binary16 emulation with the
 * interface of CUDA's <cuda_fp16.h> as far as the reference's hot-path kernels use it.
 * Every arithmetic intrinsic evaluates in double and rounds ONCE to binary16
 * (round-to-nearest-even): exact for + - * / (53 >= 2*11+2 bits), and for fma up to a
 * double-rounding event of probability ~2^-40.  Transcendentals (hexp, hsin, hcos, hrcp)
 * are the correctly rounded values; the GPU's are approximations within 1-2 ulp of them.
 */
#ifndef CUDA_ON_CPU_FP16_H
#define CUDA_ON_CPU_FP16_H

#include "cuda_runtime.h"

namespace cuda_cpu {
inline float half_bits_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal: value = man * 2^-24
      float f = (float)man * 5.9604644775390625e-08f;
      memcpy(&bits, &f, 4);
      bits |= sign;
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 112u) << 23) | (man << 13);
  }
  float out;
  memcpy(&out, &bits, 4);
  return out;
}

/* double -> binary16, round to nearest even, from the exact double value */
inline uint16_t double_to_half_bits(double d) {
  const uint16_t sign = std::signbit(d) ? 0x8000u : 0;
  if (d != d) return (uint16_t)(sign | 0x7e00u);
  const double a = fabs(d);
  if (a >= 65520.0) return (uint16_t)(sign | 0x7c00u);  // 65504 + half an ulp ties to even = inf
  if (a <= 2.98023223876953125e-08) return sign;         // <= 2^-25: ties to even = 0
  int e = ilogb(a);
  if (e < -14) e = -14;                                  // subnormal quantum 2^-24
  const double q = ldexp(1.0, e - 10);
  const double v = nearbyint(a / q) * q;                 // a/q is exact (power-of-two scale); RNE
  if (v < 6.103515625e-05) return (uint16_t)(sign | (uint16_t)(v * 16777216.0));
  const int e2 = ilogb(v);
  const unsigned man = (unsigned)((ldexp(v, -e2) - 1.0) * 1024.0);
  return (uint16_t)(sign | ((unsigned)(e2 + 15) << 10) | man);
}
}  // namespace cuda_cpu

struct __half_raw {
  unsigned short x;
};

struct __half {
  unsigned short x;
  __half() = default;
  __half(const __half_raw &r) : x(r.x) {}
  __half(float f) : x(cuda_cpu::double_to_half_bits((double)f)) {}
  __half(double f) : x(cuda_cpu::double_to_half_bits(f)) {}
  __half(short v) : x(cuda_cpu::double_to_half_bits((double)v)) {}
  __half(unsigned short v) : x(cuda_cpu::double_to_half_bits((double)v)) {}
  __half(int v) : x(cuda_cpu::double_to_half_bits((double)v)) {}
  __half(unsigned int v) : x(cuda_cpu::double_to_half_bits((double)v)) {}
  __half(long long v) : x(cuda_cpu::double_to_half_bits((double)v)) {}
  __half(unsigned long long v) : x(cuda_cpu::double_to_half_bits((double)v)) {}
  operator float() const { return cuda_cpu::half_bits_to_float(x); }
  operator short() const { return (short)truncf(cuda_cpu::half_bits_to_float(x)); }
  operator unsigned short() const { return (unsigned short)truncf(cuda_cpu::half_bits_to_float(x)); }
  operator int() const { return (int)truncf(cuda_cpu::half_bits_to_float(x)); }
  operator unsigned int() const { return (unsigned int)truncf(cuda_cpu::half_bits_to_float(x)); }
  operator long long() const { return (long long)truncf(cuda_cpu::half_bits_to_float(x)); }
  operator unsigned long long() const { return (unsigned long long)truncf(cuda_cpu::half_bits_to_float(x)); }
  operator bool() const { return (x & 0x7fffu) != 0; }
};

struct __half2 {
  __half x, y;
  __half2() = default;
  __half2(const __half &a, const __half &b) : x(a), y(b) {}
};
typedef __half half;
typedef __half2 half2;

namespace cuda_cpu {
inline double hd(const __half &h) { return (double)half_bits_to_float(h.x); }
inline __half dh(double d) {
  __half r;
  r.x = double_to_half_bits(d);
  return r;
}
inline bool hnan(const __half &h) { return (h.x & 0x7fffu) > 0x7c00u; }
}  // namespace cuda_cpu

/* conversions */
inline __half __float2half(float f) { return __half(f); }
inline __half __float2half_rn(float f) { return __half(f); }
inline __half __double2half(double f) { return __half(f); }
inline float __half2float(const __half h) { return cuda_cpu::half_bits_to_float(h.x); }
inline __half __int2half_rn(int v) { return cuda_cpu::dh((double)v); }
inline __half __short2half_rn(short v) { return cuda_cpu::dh((double)v); }
inline __half __uint2half_rn(unsigned v) { return cuda_cpu::dh((double)v); }
inline __half __ushort2half_rn(unsigned short v) { return cuda_cpu::dh((double)v); }
namespace cuda_cpu {
template <class I>
inline I sat_rn(const __half h, double lo, double hi) {  // cvt.rni.<int>.f16: NaN -> 0, saturating
  if (hnan(h)) return 0;
  double r = nearbyint(hd(h));
  r = r < lo ? lo : (r > hi ? hi : r);
  return (I)r;
}
}  // namespace cuda_cpu
inline int __half2int_rn(const __half h) { return cuda_cpu::sat_rn<int>(h, -2147483648.0, 2147483647.0); }
inline short __half2short_rn(const __half h) { return cuda_cpu::sat_rn<short>(h, -32768.0, 32767.0); }
inline unsigned short __half2ushort_rn(const __half h) { return cuda_cpu::sat_rn<unsigned short>(h, 0.0, 65535.0); }
inline unsigned int __half2uint_rn(const __half h) { return cuda_cpu::sat_rn<unsigned int>(h, 0.0, 4294967295.0); }
inline int __half2int_rz(const __half h) { return cuda_cpu::hnan(h) ? 0 : (int)trunc(cuda_cpu::hd(h)); }
inline int __half2int_rd(const __half h) { return cuda_cpu::hnan(h) ? 0 : (int)floor(cuda_cpu::hd(h)); }

inline __half2 __float2half2_rn(float f) { return __half2(__half(f), __half(f)); }
inline __half2 __floats2half2_rn(float a, float b) { return __half2(__half(a), __half(b)); }
inline __half2 __halves2half2(const __half a, const __half b) { return __half2(a, b); }
inline __half2 __half2half2(const __half a) { return __half2(a, a); }
inline __half __low2half(const __half2 a) { return a.x; }
inline __half __high2half(const __half2 a) { return a.y; }
inline float __low2float(const __half2 a) { return __half2float(a.x); }
inline float __high2float(const __half2 a) { return __half2float(a.y); }
inline __half2 __lowhigh2highlow(const __half2 a) { return __half2(a.y, a.x); }
inline __half2 __low2half2(const __half2 a) { return __half2(a.x, a.x); }
inline __half2 __high2half2(const __half2 a) { return __half2(a.y, a.y); }

/* scalar arithmetic */
inline __half __hadd(const __half a, const __half b) { return cuda_cpu::dh(cuda_cpu::hd(a) + cuda_cpu::hd(b)); }
inline __half __hsub(const __half a, const __half b) { return cuda_cpu::dh(cuda_cpu::hd(a) - cuda_cpu::hd(b)); }
inline __half __hmul(const __half a, const __half b) { return cuda_cpu::dh(cuda_cpu::hd(a) * cuda_cpu::hd(b)); }
inline __half __hdiv(const __half a, const __half b) { return cuda_cpu::dh(cuda_cpu::hd(a) / cuda_cpu::hd(b)); }
inline __half __hfma(const __half a, const __half b, const __half c) {
  return cuda_cpu::dh(fma(cuda_cpu::hd(a), cuda_cpu::hd(b), cuda_cpu::hd(c)));
}
inline __half __hneg(const __half a) {
  __half r = a;
  r.x ^= 0x8000u;
  return r;
}
inline __half __habs(const __half a) {
  __half r = a;
  r.x &= 0x7fffu;
  return r;
}
inline bool __heq(const __half a, const __half b) { return cuda_cpu::hd(a) == cuda_cpu::hd(b); }
inline bool __hne(const __half a, const __half b) { return cuda_cpu::hd(a) != cuda_cpu::hd(b); }
inline bool __hlt(const __half a, const __half b) { return cuda_cpu::hd(a) < cuda_cpu::hd(b); }
inline bool __hle(const __half a, const __half b) { return cuda_cpu::hd(a) <= cuda_cpu::hd(b); }
inline bool __hgt(const __half a, const __half b) { return cuda_cpu::hd(a) > cuda_cpu::hd(b); }
inline bool __hge(const __half a, const __half b) { return cuda_cpu::hd(a) >= cuda_cpu::hd(b); }
inline bool __hisnan(const __half a) { return cuda_cpu::hnan(a); }
inline int __hisinf(const __half a) { return a.x == 0x7c00u ? 1 : (a.x == 0xfc00u ? -1 : 0); }
inline __half __hmax(const __half a, const __half b) {
  if (cuda_cpu::hnan(a)) return b;
  if (cuda_cpu::hnan(b)) return a;
  return cuda_cpu::hd(a) > cuda_cpu::hd(b) ? a : b;
}
inline __half __hmin(const __half a, const __half b) {
  if (cuda_cpu::hnan(a)) return b;
  if (cuda_cpu::hnan(b)) return a;
  return cuda_cpu::hd(a) < cuda_cpu::hd(b) ? a : b;
}
inline __half hfloor(const __half a) { return cuda_cpu::dh(floor(cuda_cpu::hd(a))); }
inline __half hceil(const __half a) { return cuda_cpu::dh(ceil(cuda_cpu::hd(a))); }
inline __half htrunc(const __half a) { return cuda_cpu::dh(trunc(cuda_cpu::hd(a))); }
inline __half hrint(const __half a) { return cuda_cpu::dh(nearbyint(cuda_cpu::hd(a))); }
inline __half hexp(const __half a) { return cuda_cpu::dh(exp(cuda_cpu::hd(a))); }
inline __half hsin(const __half a) { return cuda_cpu::dh(sin(cuda_cpu::hd(a))); }
inline __half hcos(const __half a) { return cuda_cpu::dh(cos(cuda_cpu::hd(a))); }
inline __half hrcp(const __half a) { return cuda_cpu::dh(1.0 / cuda_cpu::hd(a)); }
inline __half hsqrt(const __half a) { return cuda_cpu::dh(sqrt(cuda_cpu::hd(a))); }

/* operators (cuda_fp16.hpp defines exactly these for __half / __half2) */
inline __half operator+(const __half &a, const __half &b) { return __hadd(a, b); }
inline __half operator-(const __half &a, const __half &b) { return __hsub(a, b); }
inline __half operator*(const __half &a, const __half &b) { return __hmul(a, b); }
inline __half operator/(const __half &a, const __half &b) { return __hdiv(a, b); }
inline __half &operator+=(__half &a, const __half &b) { return a = __hadd(a, b); }
inline __half &operator-=(__half &a, const __half &b) { return a = __hsub(a, b); }
inline __half &operator*=(__half &a, const __half &b) { return a = __hmul(a, b); }
inline __half &operator/=(__half &a, const __half &b) { return a = __hdiv(a, b); }
inline __half operator+(const __half &a) { return a; }
inline __half operator-(const __half &a) { return __hneg(a); }
inline bool operator==(const __half &a, const __half &b) { return __heq(a, b); }
inline bool operator!=(const __half &a, const __half &b) { return __hne(a, b); }
inline bool operator<(const __half &a, const __half &b) { return __hlt(a, b); }
inline bool operator>(const __half &a, const __half &b) { return __hgt(a, b); }
inline bool operator<=(const __half &a, const __half &b) { return __hle(a, b); }
inline bool operator>=(const __half &a, const __half &b) { return __hge(a, b); }

/* packed arithmetic */
#define CUDA_CPU_H2_BIN(name, scalar) \
  inline __half2 name(const __half2 a, const __half2 b) { return __half2(scalar(a.x, b.x), scalar(a.y, b.y)); }
CUDA_CPU_H2_BIN(__hadd2, __hadd)
CUDA_CPU_H2_BIN(__hsub2, __hsub)
CUDA_CPU_H2_BIN(__hmul2, __hmul)
CUDA_CPU_H2_BIN(__h2div, __hdiv)
CUDA_CPU_H2_BIN(__hmax2, __hmax)
CUDA_CPU_H2_BIN(__hmin2, __hmin)
#undef CUDA_CPU_H2_BIN
inline __half2 __hfma2(const __half2 a, const __half2 b, const __half2 c) {
  return __half2(__hfma(a.x, b.x, c.x), __hfma(a.y, b.y, c.y));
}
inline __half2 __hneg2(const __half2 a) { return __half2(__hneg(a.x), __hneg(a.y)); }
inline __half2 __habs2(const __half2 a) { return __half2(__habs(a.x), __habs(a.y)); }
#define CUDA_CPU_H2_CMP(name, scalar)                                                     \
  inline __half2 name(const __half2 a, const __half2 b) {                                 \
    return __half2(__half(scalar(a.x, b.x) ? 1.0f : 0.0f), __half(scalar(a.y, b.y) ? 1.0f : 0.0f)); \
  }
CUDA_CPU_H2_CMP(__heq2, __heq)
CUDA_CPU_H2_CMP(__hne2, __hne)
CUDA_CPU_H2_CMP(__hlt2, __hlt)
CUDA_CPU_H2_CMP(__hle2, __hle)
CUDA_CPU_H2_CMP(__hgt2, __hgt)
CUDA_CPU_H2_CMP(__hge2, __hge)
#undef CUDA_CPU_H2_CMP
inline bool __hbeq2(const __half2 a, const __half2 b) { return __heq(a.x, b.x) && __heq(a.y, b.y); }
#define CUDA_CPU_H2_UN(name, scalar) \
  inline __half2 name(const __half2 a) { return __half2(scalar(a.x), scalar(a.y)); }
CUDA_CPU_H2_UN(h2floor, hfloor)
CUDA_CPU_H2_UN(h2ceil, hceil)
CUDA_CPU_H2_UN(h2trunc, htrunc)
CUDA_CPU_H2_UN(h2rint, hrint)
CUDA_CPU_H2_UN(h2exp, hexp)
CUDA_CPU_H2_UN(h2sin, hsin)
CUDA_CPU_H2_UN(h2cos, hcos)
CUDA_CPU_H2_UN(h2rcp, hrcp)
#undef CUDA_CPU_H2_UN

inline __half2 operator+(const __half2 &a, const __half2 &b) { return __hadd2(a, b); }
inline __half2 operator-(const __half2 &a, const __half2 &b) { return __hsub2(a, b); }
inline __half2 operator*(const __half2 &a, const __half2 &b) { return __hmul2(a, b); }
inline __half2 operator/(const __half2 &a, const __half2 &b) { return __h2div(a, b); }
inline __half2 &operator+=(__half2 &a, const __half2 &b) { return a = __hadd2(a, b); }
inline __half2 &operator-=(__half2 &a, const __half2 &b) { return a = __hsub2(a, b); }
inline __half2 &operator*=(__half2 &a, const __half2 &b) { return a = __hmul2(a, b); }
inline __half2 &operator/=(__half2 &a, const __half2 &b) { return a = __h2div(a, b); }
inline __half2 operator-(const __half2 &a) { return __hneg2(a); }

#endif  // CUDA_ON_CPU_FP16_H
