// Microbenchmark: what does a random "tap" gather cost on MI355X?
// Measures, per configuration, taps/ns for
//   G : global/buffer loads, GROUP lanes x 16 B = one contiguous tap, random tap index in a
//       footprint of F bytes, tap stride S bytes (S=64: half-line taps, S=128: full-line taps)
//   L : LDS gather, 4 lanes x ds_read_b128 = one 64 B tap from a 128 KiB LDS image
// Build: hipcc --offload-arch=gfx950 -O3 gather_bench.hip -o gather_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// GROUP lanes cooperate on one tap of GROUP*16 bytes.  UNROLL independent taps in flight.
template <int GROUP, int UNROLL>
__global__ __launch_bounds__(256) void gather_global(const char *base, unsigned n_taps,
                                                     unsigned stride, int iters, unsigned *sink) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned grp = tid / GROUP, sub = tid % GROUP;
  u32x4 acc = {0, 0, 0, 0};
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, n_taps * stride, 0x00020000);
  for (int it = 0; it < iters; ++it) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const unsigned idx = __umulhi(hash32(grp * 9781u + (unsigned)(it * UNROLL + u) * 0x9E3779B9u), n_taps);
      v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(idx * stride + sub * 16u), 0, 0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[tid] = acc.x;
}

// XCD-affine: block b (on XCD b%8) gathers only from region (b%8) of `region` bytes.
template <int GROUP, int UNROLL>
__global__ __launch_bounds__(256) void gather_xcd(const char *base, unsigned region, unsigned stride,
                                                  int iters, unsigned *sink) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned grp = tid / GROUP, sub = tid % GROUP;
  const unsigned n_taps = region / stride;
  const unsigned rbase = (blockIdx.x % 8u) * region;
  u32x4 acc = {0, 0, 0, 0};
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 8u * region, 0x00020000);
  for (int it = 0; it < iters; ++it) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const unsigned idx = __umulhi(hash32(grp * 9781u + (unsigned)(it * UNROLL + u) * 0x9E3779B9u), n_taps);
      v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(rbase + idx * stride + sub * 16u), 0, 0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[tid] = acc.x;
}

// 8-byte-per-lane variant (int8 taps: 4 lanes x 8 B = 32 B, or 8 lanes x 8 B = 64 B)
template <int GROUP, int UNROLL>
__global__ __launch_bounds__(256) void gather_global8(const char *base, unsigned n_taps,
                                                      unsigned stride, int iters, unsigned *sink) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned grp = tid / GROUP, sub = tid % GROUP;
  u32x2 acc = {0, 0};
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, n_taps * stride, 0x00020000);
  for (int it = 0; it < iters; ++it) {
    u32x2 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const unsigned idx = __umulhi(hash32(grp * 9781u + (unsigned)(it * UNROLL + u) * 0x9E3779B9u), n_taps);
      v[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(idx * stride + sub * 8u), 0, 0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y) == 0x12345678u) sink[tid] = acc.x;
}

template <int UNROLL>
__global__ __launch_bounds__(1024) void gather_lds(const char *base, unsigned lds_bytes, int iters,
                                                  unsigned *sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (unsigned i = threadIdx.x * 16; i < lds_bytes; i += blockDim.x * 16)
    *(u32x4 *)(lds + i) = *(const u32x4 *)(base + i);
  __syncthreads();
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned grp = tid / 4, sub = tid % 4;
  const unsigned n_taps = lds_bytes / 64;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const unsigned idx = __umulhi(hash32(grp * 9781u + (unsigned)(it * UNROLL + u) * 0x9E3779B9u), n_taps);
      v[u] = *(const u32x4 *)(lds + idx * 64u + sub * 16u);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[tid] = acc.x;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename F>
float time_ms(F launch, int reps = 5) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a));
    launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t max_bytes = 512ull << 20;
  char *buf; unsigned *sink;
  CK(hipMalloc(&buf, max_bytes));
  CK(hipMalloc(&sink, 64 << 20));
  CK(hipMemset(buf, 1, max_bytes));
  const int blocks = 256 * 8, threads = 256;
  const double n_cu = 256.0;
  printf("kind,group,bytes_per_tap,stride,footprint_MB,unroll,iters,ms,taps_per_ns,taps_per_clk_per_cu@2.4GHz,useful_GBs\n");
  const size_t foots[] = {1ull << 20, 3ull << 20, 16ull << 20, 96ull << 20, 400ull << 20};
  for (size_t F : foots) {
    for (unsigned stride : {64u, 128u}) {
      {  // 4 lanes x 16 B = 64 B taps
        const unsigned n_taps = (unsigned)(F / stride);
        const int iters = 64, U = 8;
        float ms = time_ms([&] { hipLaunchKernelGGL((gather_global<4, 8>), dim3(blocks), dim3(threads), 0, 0, buf, n_taps, stride, iters, sink); });
        double taps = (double)blocks * threads / 4 * iters * U;
        printf("G,4,64,%u,%.0f,%d,%d,%.4f,%.3f,%.3f,%.0f\n", stride, F / 1048576.0, U, iters, ms, taps / (ms * 1e6), taps / (ms * 1e-3 * 2.4e9 * n_cu), taps * 64 / (ms * 1e6));
      }
      if (stride == 128) {  // 8 lanes x 16 B = 128 B taps (full line)
        const unsigned n_taps = (unsigned)(F / stride);
        const int iters = 64, U = 8;
        float ms = time_ms([&] { hipLaunchKernelGGL((gather_global<8, 8>), dim3(blocks), dim3(threads), 0, 0, buf, n_taps, stride, iters, sink); });
        double taps = (double)blocks * threads / 8 * iters * U;
        printf("G,8,128,%u,%.0f,%d,%d,%.4f,%.3f,%.3f,%.0f\n", stride, F / 1048576.0, U, iters, ms, taps / (ms * 1e6), taps / (ms * 1e-3 * 2.4e9 * n_cu), taps * 128 / (ms * 1e6));
      }
      if (stride == 64) {  // unaligned 128 B taps starting on any 64 B boundary
        const unsigned n_taps = (unsigned)(F / stride) - 1;
        const int iters = 64, U = 8;
        float ms = time_ms([&] { hipLaunchKernelGGL((gather_global<8, 8>), dim3(blocks), dim3(threads), 0, 0, buf, n_taps, stride, iters, sink); });
        double taps = (double)blocks * threads / 8 * iters * U;
        printf("G,8,128u,%u,%.0f,%d,%d,%.4f,%.3f,%.3f,%.0f\n", stride, F / 1048576.0, U, iters, ms, taps / (ms * 1e6), taps / (ms * 1e-3 * 2.4e9 * n_cu), taps * 128 / (ms * 1e6));
      }
      {  // int8-style: 4 lanes x 8 B = 32 B taps at stride/2
        const unsigned s2 = stride / 2;
        const unsigned n_taps = (unsigned)(F / s2);
        const int iters = 64, U = 8;
        float ms = time_ms([&] { hipLaunchKernelGGL((gather_global8<4, 8>), dim3(blocks), dim3(threads), 0, 0, buf, n_taps, s2, iters, sink); });
        double taps = (double)blocks * threads / 4 * iters * U;
        printf("G8,4,32,%u,%.0f,%d,%d,%.4f,%.3f,%.3f,%.0f\n", s2, F / 1048576.0, U, iters, ms, taps / (ms * 1e6), taps / (ms * 1e-3 * 2.4e9 * n_cu), taps * 32 / (ms * 1e6));
      }
    }
  }
  // unroll sensitivity at 1 MB / 96 MB, 64 B taps
  for (size_t F : {1ull << 20, 96ull << 20}) {
    const unsigned n_taps = (unsigned)(F / 64);
    {
      float ms = time_ms([&] { hipLaunchKernelGGL((gather_global<4, 2>), dim3(blocks), dim3(threads), 0, 0, buf, n_taps, 64u, 256, sink); });
      double taps = (double)blocks * threads / 4 * 256 * 2;
      printf("G,4,64,64,%.0f,2,256,%.4f,%.3f,%.3f,%.0f\n", F / 1048576.0, ms, taps / (ms * 1e6), taps / (ms * 1e-3 * 2.4e9 * n_cu), taps * 64 / (ms * 1e6));
    }
    {
      float ms = time_ms([&] { hipLaunchKernelGGL((gather_global<4, 16>), dim3(blocks), dim3(threads), 0, 0, buf, n_taps, 64u, 32, sink); });
      double taps = (double)blocks * threads / 4 * 32 * 16;
      printf("G,4,64,64,%.0f,16,32,%.4f,%.3f,%.3f,%.0f\n", F / 1048576.0, ms, taps / (ms * 1e6), taps / (ms * 1e-3 * 2.4e9 * n_cu), taps * 64 / (ms * 1e6));
    }
  }
  // tiny footprints (L1-resident?)
  for (size_t F : {16ull << 10, 64ull << 10, 256ull << 10}) {
    const unsigned n_taps = (unsigned)(F / 64);
    float ms = time_ms([&] { hipLaunchKernelGGL((gather_global<4, 8>), dim3(blocks), dim3(threads), 0, 0, buf, n_taps, 64u, 64, sink); });
    double taps = (double)blocks * threads / 4 * 64 * 8;
    printf("G,4,64,64,%.3f,8,64,%.4f,%.3f,%.3f,%.0f\n", F / 1048576.0, ms, taps / (ms * 1e6), taps / (ms * 1e-3 * 2.4e9 * n_cu), taps * 64 / (ms * 1e6));
  }
  // XCD-affine regions
  for (unsigned region : {1u << 20, 2u << 20, 3u << 20, 4u << 20, 6u << 20, 8u << 20}) {
    for (unsigned stride : {64u, 128u}) {
      if (stride == 64) {
        float ms = time_ms([&] { hipLaunchKernelGGL((gather_xcd<4, 8>), dim3(blocks), dim3(threads), 0, 0, buf, region, stride, 64, sink); });
        double taps = (double)blocks * threads / 4 * 64 * 8;
        printf("X,4,64,%u,%.0f/xcd,8,64,%.4f,%.3f,%.3f,%.0f\n", stride, region / 1048576.0, ms, taps / (ms * 1e6), taps / (ms * 1e-3 * 2.4e9 * n_cu), taps * 64 / (ms * 1e6));
      } else {
        float ms = time_ms([&] { hipLaunchKernelGGL((gather_xcd<8, 8>), dim3(blocks), dim3(threads), 0, 0, buf, region, stride, 64, sink); });
        double taps = (double)blocks * threads / 8 * 64 * 8;
        printf("X,8,128,%u,%.0f/xcd,8,64,%.4f,%.3f,%.3f,%.0f\n", stride, region / 1048576.0, ms, taps / (ms * 1e6), taps / (ms * 1e-3 * 2.4e9 * n_cu), taps * 128 / (ms * 1e6));
      }
    }
  }
  // LDS gather
  for (unsigned lds_bytes : {64u << 10, 128u << 10}) {
    const int iters = 256, U = 8;
    const int lblocks = 256 * (lds_bytes <= (64u << 10) ? 2 : 1) * 4;
    CK(hipFuncSetAttribute((const void *)gather_lds<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10));
    float ms = time_ms([&] { hipLaunchKernelGGL((gather_lds<8>), dim3(lblocks), dim3(1024), lds_bytes, 0, buf, lds_bytes, iters, sink); });
    double taps = (double)lblocks * 1024 / 4 * iters * U;
    printf("L,4,64,64,%.3f,%d,%d,%.4f,%.3f,%.3f,%.0f\n", lds_bytes / 1048576.0, U, iters, ms, taps / (ms * 1e6), taps / (ms * 1e-3 * 2.4e9 * n_cu), taps * 64 / (ms * 1e6));
  }
  return 0;
}
