/* cuda_on_cpu/cuda_runtime.h -- TEST INFRASTRUCTURE ONLY (oracle/, never the product path).
 *
 * A minimal stand-in for the CUDA runtime that lets g++ compile the REFERENCE's own
 * plugin kernel sources (.cu under /root/reference/TensorRT/plugin, read where they lie,
 * never copied) as ordinary C++ and run them on the host, so that oracle/_ref/ holds the
 * reference's real arithmetic to pin the C restatements in oracle/ against.
 *
 * Execution model: a launch `k<<<grid, block, shmem, stream>>>(args)` (rewritten to
 * cuda_cpu::launch(grid, block, shmem, stream, callable)(args) by chevron.py on the fly)
 * runs every block of the grid (OpenMP over blocks) and, inside a block, every thread
 * one after the other.  That is only valid for kernels without __shared__ memory,
 * __syncthreads or warp shuffles -- true for every kernel on the hot path
 * (multiScaleDeformableAttnKernel.cu, gridSamplerKernel.cu, rotateKernel.cu,
 * modulatedDeformableConv2dKernel.cu, bevPoolKernel.cu); __syncthreads/__shfl are
 * deliberately left undeclared so anything else fails to compile.
 * __CUDA_ARCH__ is left undefined: the sources' portable (#else) branches are the ones
 * compiled (dp4a as four multiply-adds, hmax through __hgt).
 */
#ifndef CUDA_ON_CPU_RUNTIME_H
#define CUDA_ON_CPU_RUNTIME_H

#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <limits>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __restrict__ __restrict
#define __launch_bounds__(...)

struct uint3 {
  unsigned x, y, z;
};
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace cuda_cpu {
struct ThreadCtx {
  uint3 tid, bid;
  dim3 bdim, gdim;
};
inline ThreadCtx &ctx() {
  static thread_local ThreadCtx c;
  return c;
}
}  // namespace cuda_cpu
#define threadIdx (cuda_cpu::ctx().tid)
#define blockIdx (cuda_cpu::ctx().bid)
#define blockDim (cuda_cpu::ctx().bdim)
#define gridDim (cuda_cpu::ctx().gdim)

typedef struct CUstream_st *cudaStream_t;
typedef int cudaError_t;
enum { cudaSuccess = 0 };
inline cudaError_t cudaGetLastError() { return cudaSuccess; }
inline const char *cudaGetErrorString(cudaError_t) { return "cuda_on_cpu"; }
inline cudaError_t cudaMemset(void *p, int v, size_t n) {
  memset(p, v, n);
  return cudaSuccess;
}
inline cudaError_t cudaMemsetAsync(void *p, int v, size_t n, cudaStream_t = 0) {
  memset(p, v, n);
  return cudaSuccess;
}

namespace cuda_cpu {
template <class F>
struct Launch {
  dim3 grid, block;
  F fn;
  template <class... A>
  void operator()(A... args) const {  // kernel parameters are passed by value, as on the device
    const long nblocks = (long)grid.x * grid.y * grid.z;
#pragma omp parallel for schedule(dynamic, 4)
    for (long b = 0; b < nblocks; ++b) {
      ThreadCtx &c = ctx();
      c.gdim = grid;
      c.bdim = block;
      c.bid.x = (unsigned)(b % grid.x);
      c.bid.y = (unsigned)((b / grid.x) % grid.y);
      c.bid.z = (unsigned)(b / ((long)grid.x * grid.y));
      for (unsigned tz = 0; tz < block.z; ++tz)
        for (unsigned ty = 0; ty < block.y; ++ty)
          for (unsigned tx = 0; tx < block.x; ++tx) {
            c.tid.x = tx;
            c.tid.y = ty;
            c.tid.z = tz;
            fn(args...);
          }
    }
  }
};
template <class G, class B, class F>
Launch<F> launch(G grid, B block, size_t /*shmem*/, cudaStream_t /*stream*/, F fn) {
  return Launch<F>{dim3(grid), dim3(block), fn};
}
}  // namespace cuda_cpu

/* device math that nvcc puts in the global namespace */
template <class T>
inline T max(const T a, const T b) {
  return a < b ? b : a;
}
template <class T>
inline T min(const T a, const T b) {
  return b < a ? b : a;
}
inline float max(float a, int b) { return fmaxf(a, (float)b); }
inline float max(int a, float b) { return fmaxf((float)a, b); }
inline float min(float a, int b) { return fminf(a, (float)b); }
inline float min(int a, float b) { return fminf((float)a, b); }
inline double max(double a, float b) { return fmax(a, (double)b); }
inline double max(float a, double b) { return fmax((double)a, b); }
inline double min(double a, float b) { return fmin(a, (double)b); }
inline double min(float a, double b) { return fmin((double)a, b); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __fdividef(float a, float b) { return a / b; }
inline float __expf(float x) { return expf(x); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline int __float2int_rn(float x) { return (int)nearbyintf(x); }
inline int __float2int_rd(float x) { return (int)floorf(x); }

#endif  // CUDA_ON_CPU_RUNTIME_H
