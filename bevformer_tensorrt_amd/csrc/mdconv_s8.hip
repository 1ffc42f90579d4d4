// INT8 flavour of the modulated deformable convolution (modulatedDeformableConv2dKernel.cu:190-257,463-607,
// 897-978): the reference's integer pipeline as (a) im2col + GEMM, (b) a register-staged fused implicit GEMM and
// (c) the LDS-DMA pipelined implicit GEMM -- all bit-identical to each other (tests/test_mdconv_gpu.py).
// Shared with mdconv.hip through mdconv.h: problem description, workspace layout, LDS image of the LDS-DMA
// kernels, the generic NCHW -> NHWC and weight re-layout kernels.
#include "mdconv.h"

namespace bevops {
namespace {

// int8, HW % 4 == 0 and C % 16 == 0: 128 channels x 128 pixels per block.  A thread loads a 4 x 4 byte block
// (4 pixels of 4 channel rows, lanes along the pixels: 128 contiguous bytes per row and half-wave), transposes
// it in registers and writes 4 dwords (4 channels of one pixel each) into the [pixel][channel] tile; the tile
// leaves as 16-byte vectors, 128 contiguous bytes per pixel.  (The byte-wise 32 x 32 kernel above took 12 us
// for the 8.9 MB stage-3 image -- longer than the fp16 copy of twice the bytes.)
__global__ __launch_bounds__(256) void nchw_to_nhwc_s8v_kernel(const int8_t *__restrict__ in, int8_t *__restrict__ out,
                                                               int C, int HW, unsigned flip4) {
  constexpr int kRow = 128 + 16;   // bytes per tile row (16-byte aligned rows for the b128 reads)
  __shared__ __attribute__((aligned(16))) unsigned char tile[128 * kRow];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 128, c0 = blockIdx.y * 128;
  const int8_t *ib = in + (size_t)b * C * HW;
  int8_t *ob = out + (size_t)b * C * HW;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = threadIdx.x + it * 256;      // 32 channel quads x 32 pixel quads
    const int pq = v & 31, cq = v >> 5;
    const int p = p0 + pq * 4, c = c0 + cq * 4;
    unsigned r[4] = {0u, 0u, 0u, 0u};
    if (p < HW && c < C) {                      // HW % 4 == 0, C % 4 == 0: the 4 x 4 block is all in or all out
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = *reinterpret_cast<const unsigned *>(ib + (size_t)(c + k) * HW + p) ^ flip4;
    }
    // r[k] = 4 pixels of channel c + k  ->  o[j] = 4 channels of pixel p + j
    const unsigned a = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u), e = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);
    const unsigned f = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u), g = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
    const unsigned o[4] = {__builtin_amdgcn_perm(f, a, 0x05040100u), __builtin_amdgcn_perm(f, a, 0x07060302u),
                           __builtin_amdgcn_perm(g, e, 0x05040100u), __builtin_amdgcn_perm(g, e, 0x07060302u)};
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<unsigned *>(&tile[(pq * 4 + j) * kRow + cq * 4]) = o[j];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = threadIdx.x + it * 256;      // 128 pixels x 8 chunks of 16 channels
    const int ch = v & 7, pl = v >> 3;
    if (p0 + pl < HW && c0 + ch * 16 < C)
      *reinterpret_cast<uint4 *>(ob + (size_t)(p0 + pl) * C + c0 + ch * 16) =
          *reinterpret_cast<const uint4 *>(&tile[pl * kRow + ch * 16]);
  }
}

// ---- 6. INT8 flavour (modulatedDeformableConv2dKernel.cu:190-257,463-607,897-978) ----------
// im2col on the NHWC int8 image (16 channels per lane = one 16-byte load per corner; unsigned
// x255 area weights, int32 4-corner dot, T2int8(t/255), then T2int8(val * mask)), one batched
// int8 GEMM on the matrix cores (v_mfma_i32_32x32x32_i8, int32 accumulate), epilogue
// T2int8((acc * s_in*s_w + bias) / s_out) scattered to NCHW.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int q_away(float a) {
  a = fminf(fmaxf(a, -128.f), 127.f);
  return (int)(a + (a > 0.f ? 0.5f : -0.5f));
}
__device__ __forceinline__ int u8w(float a) { return (int)fminf(fmaxf(rintf(a * 255.f), 0.f), 255.f); }

template <int V>
__global__ __launch_bounds__(256) void im2col_nhwc_s8_kernel(const int8_t *__restrict__ xt,
                                                             const int8_t *__restrict__ offset,
                                                             const int8_t *__restrict__ mask,
                                                             int8_t *__restrict__ col, ConvDims d,
                                                             float s_off, float s_mask, int kp) {
  const int vec_per_pix = d.Cin / V;
  const int KK = d.Kh * d.Kw;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int cv = (int)(idx % vec_per_pix);
  const size_t r = idx / vec_per_pix;
  const int t = (int)(r % KK);
  const size_t n = r / KK;
  const int HoWo = d.Ho * d.Wo;
  const size_t N = (size_t)d.B * HoWo;
  if (n >= N) return;
  const int b = (int)(n / HoWo);
  const int pix = (int)(n - (size_t)b * HoWo);
  const int ho = pix / d.Wo, wo = pix - ho * d.Wo;
  const int c = cv * V;
  const int dg = c / (d.Cin / d.DG);
  const int i = t / d.Kw, j = t - i * d.Kw;
  const size_t obase = (((size_t)b * d.DG + dg) * 2 * KK) * HoWo + pix;
  float h_im, w_im, m;
  {
#pragma clang fp contract(off)
    const float off_h = (float)offset[obase + (size_t)(2 * t) * HoWo] * s_off;
    const float off_w = (float)offset[obase + (size_t)(2 * t + 1) * HoWo] * s_off;
    m = (float)mask[(((size_t)b * d.DG + dg) * KK + t) * HoWo + pix] * s_mask;
    h_im = off_h + (float)(ho * d.sh - d.ph + i * d.dh);
    w_im = off_w + (float)(wo * d.sw - d.pw + j * d.dw);
  }
  int acc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = 0;
  const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)d.H && w_im < (float)d.W;
  if (in) {
#pragma clang fp contract(off)
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int h0 = (int)hf, w0 = (int)wf;
    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
    const int aw[4] = {u8w(hh * hw), u8w(hh * lw), u8w(lh * hw), u8w(lh * lw)};
    const bool ok[4] = {h0 >= 0 && w0 >= 0, h0 >= 0 && w0 + 1 < d.W, h0 + 1 < d.H && w0 >= 0,
                        h0 + 1 < d.H && w0 + 1 < d.W};
    const int hs[4] = {h0, h0, h0 + 1, h0 + 1}, ws[4] = {w0, w0 + 1, w0, w0 + 1};
    const int8_t *xb = xt + (size_t)b * d.H * d.W * d.Cin + c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!ok[q]) continue;
      const int8_t *p = xb + ((size_t)hs[q] * d.W + ws[q]) * d.Cin;
      int8_t v[V];
      if constexpr (V == 16) *reinterpret_cast<uint4 *>(v) = *reinterpret_cast<const uint4 *>(p);
      else if constexpr (V == 4) *reinterpret_cast<unsigned *>(v) = *reinterpret_cast<const unsigned *>(p);
      else v[0] = p[0];
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += (int)v[k] * aw[q];
    }
  }
  int8_t res[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
#pragma clang fp contract(off)
    const int val = in ? q_away((float)acc[k] * (1 / 255.f)) : 0;
    res[k] = (int8_t)q_away((float)val * m);
  }
  const int cin_g = d.Cin / d.G;
  const int g = c / cin_g, cg = c - g * cin_g;
  int8_t *o = col + ((size_t)g * N + n) * kp + (size_t)t * cin_g + cg;
  if constexpr (V == 16) *reinterpret_cast<uint4 *>(o) = *reinterpret_cast<const uint4 *>(res);
  else if constexpr (V == 4) *reinterpret_cast<unsigned *>(o) = *reinterpret_cast<const unsigned *>(res);
  else o[0] = res[0];
}

// C[m][n] = sum_k A[m][k] * B[n][k] on int8, 128x128x64 tiles, requantising epilogue
constexpr int kIK = 64, kILd = kIK + 16;  // bytes per LDS row (+16: conflict-free b128 reads)
__global__ __launch_bounds__(256) void gemm_tn_s8_kernel(const int8_t *__restrict__ A,
                                                         const int8_t *__restrict__ Bm,
                                                         const float *__restrict__ bias,
                                                         int8_t *__restrict__ out, int M, int N, int K,
                                                         GemmEpi e, float s_iw, float s_out) {
  __shared__ __attribute__((aligned(16))) int8_t As[kBM][kILd];
  __shared__ __attribute__((aligned(16))) int8_t Bs[kBN][kILd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
  const int r0 = tid >> 2, kc = (tid & 3) * 16, r1 = r0 + 64;  // 128 rows x 4 chunks of 16 B
  i32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
  const int nk = (K + kIK - 1) / kIK;
  uint4 ra0, ra1, rb0, rb1;
  auto ld = [&](const int8_t *p, bool ok) { return ok ? *reinterpret_cast<const uint4 *>(p) : make_uint4(0, 0, 0, 0); };
  auto gload = [&](int kt) {
    const int k = kt * kIK + kc;
    const bool kok = k < K;  // K % 16 == 0 guaranteed by the launcher
    ra0 = ld(A + (size_t)(m0 + r0) * K + k, kok && m0 + r0 < M);
    ra1 = ld(A + (size_t)(m0 + r1) * K + k, kok && m0 + r1 < M);
    rb0 = ld(Bm + (size_t)(n0 + r0) * K + k, kok && n0 + r0 < N);
    rb1 = ld(Bm + (size_t)(n0 + r1) * K + k, kok && n0 + r1 < N);
  };
  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    *reinterpret_cast<uint4 *>(&As[r0][kc]) = ra0;
    *reinterpret_cast<uint4 *>(&As[r1][kc]) = ra1;
    *reinterpret_cast<uint4 *>(&Bs[r0][kc]) = rb0;
    *reinterpret_cast<uint4 *>(&Bs[r1][kc]) = rb1;
    __syncthreads();
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kk = ks * 32 + (lane >> 5) * 16;
      i32x4_t a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const i32x4_t *>(&As[wm * 64 + i * 32 + (lane & 31)][kk]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = *reinterpret_cast<const i32x4_t *>(&Bs[wn * 64 + j * 32 + (lane & 31)][kk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + (lane & 31);
    if (n >= N) continue;
    const int b = n / e.HoWo, pix = n - b * e.HoWo;
    int8_t *ob = out + ((size_t)b * e.Cout + e.co0) * e.HoWo + pix;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M) {
#pragma clang fp contract(off)
          const float v = ((float)acc[i][j][r] * s_iw + (bias ? bias[e.co0 + m] : 0.f)) / s_out;
          ob[(size_t)m * e.HoWo] = (int8_t)q_away(v);
        }
      }
  }
}

// ---- 6b. fused deformable implicit GEMM, int8 ------------------------------------------------
// The int8 flavour of section 5: the column element T2int8(T2int8(sum_corners v a / 255) * mask)
// (modulatedDeformableConv2dKernel.cu:463-548) is produced by the B-tile loader straight into
// LDS and consumed by v_mfma_i32_32x32x32_i8 -- no 80 MB column buffer written and re-read per
// stage-3 call.  Block tile 256 (Cout) x 64 pixels x 64 (one tap, 64 input channels); 4 waves of
// 64 x 64 outputs.  Producer thread = (pixel, 16-channel quarter): four 16-byte corner loads from
// the u8-BIASED channels-last copy (v + 128), 4x4 byte transposes, per channel ONE unsigned dot4
// whose addend -128 * sum(a) removes the bias, the exact integer T2int8(. / 255) of msda_hm4.hip
// (saturating v_mad_i32_i24, result in the top byte; hand-placed: DOT -> other VALU hazard), then
// the reference's float multiply by the mask and round-half-away, 16 result bytes = one
// ds_write_b128.  Out-of-range corners get area weight 0 (so they leave the dot and the bias sum).
// Bit-identical to im2col_nhwc_s8_kernel + gemm_tn_s8_kernel (tests/test_mdconv_gpu.py).
constexpr int kSM = 256, kSN = 64, kSK = 64, kSLd = kSK + 16;

__device__ __forceinline__ void s8_quad(const unsigned (&v)[4], unsigned aw, int neg, int magic, int half,
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
// 4x4 byte transpose: r[k] = 4 channels of corner k  ->  o[c] = channel c of corners 0..3
__device__ __forceinline__ void s8_transpose(unsigned r0, unsigned r1, unsigned r2, unsigned r3, unsigned (&o)[4]) {
  const unsigned a = __builtin_amdgcn_perm(r1, r0, 0x05010400u);
  const unsigned b = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
  const unsigned c = __builtin_amdgcn_perm(r3, r2, 0x05010400u);
  const unsigned e = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
  o[0] = __builtin_amdgcn_perm(c, a, 0x05040100u);
  o[1] = __builtin_amdgcn_perm(c, a, 0x07060302u);
  o[2] = __builtin_amdgcn_perm(e, b, 0x05040100u);
  o[3] = __builtin_amdgcn_perm(e, b, 0x07060302u);
}

// Schedule of a k-step (one set of operand registers, 168 VGPRs = 3 blocks per CU, so the 544 tiles
// of the base stage-3 call run in ONE round; with double-buffered operands the kernel needs 200
// VGPRs = 512 slots and the 32 left-over tiles cost a second round -- measured 124 us, no better
// than the im2col + GEMM pair, profiles/r02):
//   corners of this step arrive -> 4 x (byte transpose + dot / requantise)      [corner registers free]
//   -> gathers of the NEXT step issued -> mask multiply + rounding of the 16 values (~100 VALU)
//   -> barrier, weight + pixel tile to LDS, barrier -> weight loads of the next step issued -> MFMAs.
// A gather's round trip hides behind ~100 VALU instructions, two barriers and the 8 MFMAs; the
// weight loads (L2 hits, coalesced) behind the MFMAs and the next step's front half.
__global__ __launch_bounds__(256, 3) void dcn_fused_s8_kernel(
    const int8_t *__restrict__ xt, const int8_t *__restrict__ offset, const int8_t *__restrict__ mask,
    const int8_t *__restrict__ wt, const float *__restrict__ bias, int8_t *__restrict__ out, ConvDims d, int g,
    int Kp, float s_off, float s_mask, float s_iw, float s_out) {
  __shared__ __attribute__((aligned(16))) int8_t As[kSM][kSLd];
  __shared__ __attribute__((aligned(16))) int8_t Bs[kSN][kSLd];
  constexpr int kOmMax = 64;
  __shared__ int8_t Om[kOmMax][kSN];
  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int KK = d.Kh * d.Kw, cin_g = d.Cin / d.G, cout_g = d.Cout / d.G;
  const int HoWo = d.Ho * d.Wo;
  const int N = d.B * HoWo;
  const int n0 = (int)xcd_remap(blockIdx.x, gridDim.x) * kSN, m0 = blockIdx.y * kSM;
  const int8_t *A = wt + (size_t)g * cout_g * Kp;
  // B-producer role: pixel n0 + tid / 4, channels [cq * 16, +16) of the 64-channel chunk
  const int pp = tid >> 2, cq = tid & 3;
  const int pn = n0 + pp;
  const bool pvalid = pn < N;
  const int pb = pvalid ? pn / HoWo : 0;
  const int ppix = pvalid ? pn - pb * HoWo : 0;
  const int pho = ppix / d.Wo, pwo = ppix - pho * d.Wo;
  // A-loader role: rows (tid >> 2) + 64 i, 16-byte chunk (tid & 3)
  const int ar = tid >> 2, ac = (tid & 3) * 16;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(xt), 0, (unsigned)((size_t)d.B * d.H * d.W * d.Cin), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(A), 0, (unsigned)((size_t)cout_g * Kp), 0x00020000);
  const unsigned ximg_off = (unsigned)((size_t)pb * d.H * d.W * d.Cin + g * cin_g + cq * 16);
  unsigned a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = m0 + ar + 64 * i;
    a_off[i] = r < cout_g ? (unsigned)((size_t)r * Kp + ac) : 0xFFFFFFF0u;
  }
  i32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

  const int chunks = cin_g / kSK;
  const int nsteps = KK * chunks;
  int fidx[4];
  unsigned faw = 0;   // 4 packed u8 area weights (0 for out-of-range corners)
  int fneg = 0;       // -128 * their sum
  float fm = 0.f;     // mask value
  int cur_tap = -1, cur_dg = -1;
  uint4 ra[4], rb[4];

  const int om_per_dg = 3 * KK;
  const bool om_ok = d.DG * om_per_dg <= kOmMax;
  if (om_ok) {
    for (int idx = tid; idx < d.DG * om_per_dg * kSN; idx += 256) {
      const int p = idx % kSN, t = idx / kSN;
      const int dgi = t / om_per_dg, tt = t - dgi * om_per_dg;
      const int n = n0 + p;
      int8_t v = 0;
      if (n < N) {
        const int b = n / HoWo, pix = n - b * HoWo;
        v = tt < 2 * KK ? offset[(((size_t)b * d.DG + dgi) * 2 * KK + tt) * HoWo + pix]
                        : mask[(((size_t)b * d.DG + dgi) * KK + (tt - 2 * KK)) * HoWo + pix];
      }
      Om[t][p] = v;
    }
    __syncthreads();
  }

  // footprint of (pixel, tap): corner offsets, packed area weights, mask -- recomputed when the tap
  // (or the deform group) of `step` differs from the current one
  auto footprint = [&](int step) {
    const int tap = step / chunks, c0 = (step - tap * chunks) * kSK;
    const int dg = (g * cin_g + c0) / (d.Cin / d.DG);
    if (tap == cur_tap && dg == cur_dg) return;
    cur_tap = tap;
    cur_dg = dg;
    const int i = tap / d.Kw, j = tap - i * d.Kw;
    int qoh, qow, qm;
    if (om_ok) {
      qoh = Om[dg * om_per_dg + 2 * tap][pp];
      qow = Om[dg * om_per_dg + 2 * tap + 1][pp];
      qm = Om[dg * om_per_dg + 2 * KK + tap][pp];
    } else {
      const size_t ob = (((size_t)pb * d.DG + dg) * 2 * KK) * HoWo + ppix;
      qoh = offset[ob + (size_t)(2 * tap) * HoWo];
      qow = offset[ob + (size_t)(2 * tap + 1) * HoWo];
      qm = mask[(((size_t)pb * d.DG + dg) * KK + tap) * HoWo + ppix];
    }
    float h_im, w_im;
    {
#pragma clang fp contract(off)
      const float off_h = (float)qoh * s_off, off_w = (float)qow * s_off;
      fm = (float)qm * s_mask;
      h_im = off_h + (float)(pho * d.sh - d.ph + i * d.dh);
      w_im = off_w + (float)(pwo * d.sw - d.pw + j * d.dw);
    }
    const bool in = pvalid && h_im > -1.f && w_im > -1.f && h_im < (float)d.H && w_im < (float)d.W;
    unsigned aw[4] = {0u, 0u, 0u, 0u};
    int hs[4] = {0, 0, 0, 0}, ws[4] = {0, 0, 0, 0};
    if (in) {
#pragma clang fp contract(off)
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h0 = (int)hf, w0 = (int)wf;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool ok[4] = {h0 >= 0 && w0 >= 0, h0 >= 0 && w0 + 1 < d.W, h0 + 1 < d.H && w0 >= 0,
                          h0 + 1 < d.H && w0 + 1 < d.W};
      const int a4[4] = {u8w(hh * hw), u8w(hh * lw), u8w(lh * hw), u8w(lh * lw)};
      const int hq[4] = {h0, h0, h0 + 1, h0 + 1}, wq[4] = {w0, w0 + 1, w0, w0 + 1};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        aw[q] = ok[q] ? (unsigned)a4[q] : 0u;
        hs[q] = ok[q] ? hq[q] : 0;
        ws[q] = ok[q] ? wq[q] : 0;
      }
    }
    faw = aw[0] | (aw[1] << 8) | (aw[2] << 16) | (aw[3] << 24);
    fneg = -(int)((aw[0] + aw[1] + aw[2] + aw[3]) << 7);
#pragma unroll
    for (int q = 0; q < 4; ++q) fidx[q] = (int)(ximg_off + (unsigned)(hs[q] * d.W + ws[q]) * (unsigned)d.Cin);
  };
  auto gather = [&](int step) {
    const int tap = step / chunks, c0 = (step - tap * chunks) * kSK;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      rb[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, fidx[q], c0, 0));
  };
  auto weights = [&](int step) {
    const int tap = step / chunks, c0 = (step - tap * chunks) * kSK;
    const int a_s = tap * cin_g + c0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      ra[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)a_off[i], a_s, 0));
  };

  const int magic = 65793, half = 1 << 23;   // round(2^24 / 255): exact T2int8(t / 255), see msda_hm4.hip
  footprint(0);
  gather(0);
  weights(0);
  for (int step = 0; step < nsteps; ++step) {
    // 1. corners -> requantised bilinear sums of this thread's 16 channels (top bytes of x)
    int x[16];
    {
      const unsigned c0w[4] = {rb[0].x, rb[0].y, rb[0].z, rb[0].w}, c1w[4] = {rb[1].x, rb[1].y, rb[1].z, rb[1].w};
      const unsigned c2w[4] = {rb[2].x, rb[2].y, rb[2].z, rb[2].w}, c3w[4] = {rb[3].x, rb[3].y, rb[3].z, rb[3].w};
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        unsigned tr[4];
        s8_transpose(c0w[v], c1w[v], c2w[v], c3w[v], tr);
        int xq[4];
        s8_quad(tr, faw, fneg, magic, half, xq);
#pragma unroll
        for (int c = 0; c < 4; ++c) x[4 * v + c] = xq[c];
      }
    }
    const float m_cur = fm;
    __builtin_amdgcn_sched_barrier(0);
    // 2. the next step's gathers go out now (corner registers are free)
    if (step + 1 < nsteps) {
      footprint(step + 1);
      gather(step + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    // 3. mask multiply + round half away, pack 16 bytes
    unsigned res[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      int rq[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma clang fp contract(off)
        const float val = (float)(x[4 * v + c] >> 24);   // T2int8(sum / 255)
        rq[c] = q_away(val * m_cur);                     // T2int8(val * mask)
      }
      res[v] = ((unsigned)rq[0] & 0xffu) | (((unsigned)rq[1] & 0xffu) << 8) | (((unsigned)rq[2] & 0xffu) << 16) |
               ((unsigned)rq[3] << 24);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4 *>(&As[ar + 64 * i][ac]) = ra[i];
    *reinterpret_cast<uint4 *>(&Bs[pp][cq * 16]) = make_uint4(res[0], res[1], res[2], res[3]);
    __syncthreads();
    if (step + 1 < nsteps) weights(step + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kk = ks * 32 + (lane >> 5) * 16;
      i32x4_t a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const i32x4_t *>(&As[wm * 64 + i * 32 + (lane & 31)][kk]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = *reinterpret_cast<const i32x4_t *>(&Bs[j * 32 + (lane & 31)][kk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + j * 32 + (lane & 31);
    if (n >= N) continue;
    const int b = n / HoWo, pix = n - b * HoWo;
    int8_t *ob = out + ((size_t)b * d.Cout + g * cout_g) * HoWo + pix;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < cout_g) {
#pragma clang fp contract(off)
          const float v = ((float)acc[i][j][r] * s_iw + (bias ? bias[g * cout_g + m] : 0.f)) / s_out;
          ob[(size_t)m * HoWo] = (int8_t)q_away(v);
        }
      }
  }
}

// ---- 6c. the int8 flavour of section 5c: weights by LDS-DMA, two LDS buffers, ONE barrier per k-step ----
// The register-staged kernel above is latency-bound (two barriers, 64 B of ds_write_b128 per thread and a
// dependent gather round trip per 64-channel step: ~6 000 clk per step for ~2 400 clk of VALU work).  Here
// a k-step is one tap x 128 input channels = a 128-byte LDS row per weight row / pixel, i.e. exactly the
// LDS image, XOR swizzle, DMA piece mapping and fragment reads of dcn_glds_f16_kernel with
// v_mfma_i32_32x32x32_i8 consuming 32 bytes of k where the fp16 kernel consumes 16 halves.  Schedule of
// iteration `step` (tiles of `step` in buffer step & 1, raw corners of step + 1 in registers, a step old):
//   corners of step + 1 -> byte transposes, dots, mask multiply + rounding -> ds_write_b128 of the step + 1
//   pixel row | weight DMA of step + 1 -> other buffer | gathers of step + 2 issued | 8 MFMAs |
//   s_waitcnt vmcnt(4) (the DMA is older than the 4 gathers, which stay in flight across the barrier) |
//   barrier -- the upper half of the waves runs the MFMAs first and the producer work after them.
// The int32 partial sums of the split-K tail are exact in any order.  Bit-identical to the im2col + GEMM pair.
constexpr int kSOmRows = 32;   // 3 * Kh * Kw int8 rows of offsets / mask per pixel tile
template <int WN> struct GldsS8 {
  static constexpr int kLds = Glds<WN>::kLds + kSOmRows * Glds<WN>::kN;
};

// T2int8(T2int8(sum / 255) * mask) of 4 channels whose first requantisation sits in the top byte of x[c]
__device__ __forceinline__ unsigned s8_mask4(const int (&x)[4], float m) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  // (no SDWA byte-select convert here: hand-placed directly behind the v_mad_i32_i24 of s8_quad it read stale
  // registers -- the hazard recogniser does not see inline asm -- and it measured no faster where it was legal)
  float val[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) val[c] = (float)(x[c] >> 24);
  int q[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {   // two channels per packed fp32 multiply / add (separate roundings, as the reference)
    f32x2_t a;
    {
#pragma clang fp contract(off)
      a = f32x2_t{val[2 * h], val[2 * h + 1]} * f32x2_t{m, m};
    }
    const float ax = __builtin_amdgcn_fmed3f(a.x, -128.f, 127.f), ay = __builtin_amdgcn_fmed3f(a.y, -128.f, 127.f);
    // + copysign(0.5, a): a == 0 rounds to 0 with either sign, as (a > 0 ? 0.5 : -0.5) does.  (Scalars on
    // purpose: with the clamped values written back into the 2-vector, hipcc 7.2 emitted ONE v_bfi for the
    // pair and gave both lanes the first lane's sign -- caught by the bit-identity test.)
    const float hx = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, ax) & 0x80000000u) | 0x3f000000u);
    const float hy = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, ay) & 0x80000000u) | 0x3f000000u);
    q[2 * h] = (int)add_rn(ax, hx);
    q[2 * h + 1] = (int)add_rn(ay, hy);
  }
  const unsigned lo = __builtin_amdgcn_perm((unsigned)q[1], (unsigned)q[0], 0x0c0c0400u);   // bytes: q0, q1, 0, 0
  const unsigned hi = __builtin_amdgcn_perm((unsigned)q[3], (unsigned)q[2], 0x04000c0cu);   // bytes: 0, 0, q2, q3
  return lo | hi;
}

// ABL (timing experiments; 1..8 give wrong results): 1 no gathers, 2 no producer VALU, 4 no weight DMA, 8 no MFMA,
// 16 wave halves in opposite phase order (correct results; measured no faster: the kernel is VALU-bound)
// ENG (the INT8 engine's channels-last block, bevops_mdconv_forward_int8_nhwc): `xt` is the caller's SIGNED int8
// [B, H, W, Cin] activation itself (no copy: the +128 bias of the unsigned dot is one v_xor per gathered dword),
// `offset` is the raw fp16 [B, Ho, Wo, om_channels] output of the pack's offset convolution -- its offsets and the
// sigmoid of its mask logits are quantised with s_off / s_mask while they are staged (what TensorRT's Q node in
// front of the plugin does), so the arithmetic below is the plugin's on exactly those int8 operands -- and the
// output leaves as int8 [B, Ho, Wo, Cout] with the ReLU folded into the requantisation.
// FAST (ENG only; the engine's default): ONE requantisation per column element instead of the plugin's two -- the
// mask is folded into the four area weights before they are quantised, a_q = u8(area_q * mask * 255), so the column
// byte is T2int8(sum_q a_q v_q / 255) straight out of the dot + saturating multiply-add; the plugin's second step
// (float multiply by the mask, round half away: ~ 20 VALU operations per 4 channels, the bulk of what makes the
// plugin kernel VALU-bound) disappears.  Same 8-bit weight resolution, one rounding less: NOT bit-identical to the
// plugin (tests compare it within a step), which the exact flavour (FAST = false) remains.
struct S8Eng { int om_channels, relu; };

// 4 consecutive output channels m..m+3 of output pixel n: requantise (ReLU first when asked), one dword store
__device__ __forceinline__ void s8_store4_nhwc(int8_t *__restrict__ out, const float *__restrict__ bias,
                                               const int (&a)[4], int n, int m, int g, int cout_g, int Cout,
                                               float s_iw, float s_out, int relu) {
  int8_t *p = out + (size_t)n * Cout + g * cout_g + m;
  unsigned pk = 0u;
  int q[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma clang fp contract(off)
    float v = ((float)a[e] * s_iw + ((bias && m + e < cout_g) ? bias[g * cout_g + m + e] : 0.f)) / s_out;
    if (relu) v = fmaxf(v, 0.f);
    q[e] = q_away(v);
    pk |= ((unsigned)q[e] & 0xffu) << (8 * e);
  }
  if (m + 3 < cout_g && (((g * cout_g + m) | Cout) & 3) == 0) {
    *reinterpret_cast<unsigned *>(p) = pk;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (m + e < cout_g) p[e] = (int8_t)q[e];
  }
}

template <int WN, int ABL, bool ENG = false, bool FAST = false>
__global__ __launch_bounds__(256 * WN, WN == 2 ? 2 : 1) void dcn_glds_s8_kernel(
    const int8_t *__restrict__ xt, const int8_t *__restrict__ offset, const int8_t *__restrict__ mask,
    const int8_t *__restrict__ wt, const float *__restrict__ bias, int8_t *__restrict__ out, ConvDims d, int g,
    int Kp, float s_off, float s_mask, float s_iw, float s_out, TailPlan tp, S8Eng eng) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [A0][A1][B0][B1][Om]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int KK = d.Kh * d.Kw, cin_g = d.Cin / d.G, cout_g = d.Cout / d.G;
  const int HoWo = d.Ho * d.Wo;
  const int N = d.B * HoWo;
  const int n_tail_blocks = tp.tail_tiles * tp.split;
  const bool is_tail = (int)blockIdx.x < n_tail_blocks;
  int ntile, part = 0;
  if (is_tail) {
    ntile = tp.main_tiles + (int)blockIdx.x / tp.split;
    part = (int)blockIdx.x % tp.split;
  } else {
    ntile = (int)xcd_remap(blockIdx.x - n_tail_blocks, tp.main_tiles);
  }
  constexpr int kN = Glds<WN>::kN, kGB = Glds<WN>::kB, kTh = Glds<WN>::kThreads, kPc = Glds<WN>::kPieces;
  constexpr int kStepK = 128;                      // input channels (= bytes) per k-step
  const int n0 = ntile * kN, m0 = blockIdx.y * kFM;
  const int8_t *A = wt + (size_t)g * cout_g * Kp;
  const int dg = (g * cin_g) / (d.Cin / d.DG);
  int8_t *Om = reinterpret_cast<int8_t *>(smem + Glds<WN>::kLds);   // [3 KK][kN]

  // pixel-producer role: pixel n0 + (tid >> 3), 16 channels cq of the 128-channel chunk
  const int pp = tid >> 3, cq = tid & 7;
  const int pn = n0 + pp;
  const bool pvalid = pn < N;
  const int pb = pvalid ? pn / HoWo : 0;
  const int ppix = pvalid ? pn - pb * HoWo : 0;
  const int pho = ppix / d.Wo, pwo = ppix - pho * d.Wo;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(xt), 0, (unsigned)((size_t)d.B * d.H * d.W * d.Cin), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int8_t *>(A), 0, (unsigned)((size_t)cout_g * Kp), 0x00020000);
  const unsigned ximg_off = (unsigned)((size_t)pb * d.H * d.W * d.Cin + g * cin_g + cq * 16);
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned b_lds = (unsigned)(size_t)((lds_char *)smem) + (unsigned)(2 * kGA) +
                         (unsigned)(pp * 128 + ((cq ^ swz8(pp)) << 4));   // LDS byte address of the thread's B chunk
  // ABL & 32 (round 6, the default order: see the loop): the lower half of the waves issues all the weight pieces
  constexpr bool kLowerDma = (ABL & 32) != 0;
  constexpr int kPcS = kLowerDma ? 2 * kPc : kPc;
  unsigned a_off[kPcS];
#pragma unroll
  for (int j = 0; j < kPcS; ++j) {
    const unsigned row = (unsigned)((wave * kPcS + j) * 8 + (lane >> 3));
    const unsigned chunk = (lane & 7u) ^ swz8(row);
    a_off[j] = (m0 + (int)row) < cout_g ? (unsigned)((size_t)(m0 + row) * Kp + chunk * 16) : 0xFFFFFFF0u;
  }
  unsigned fa[2], fb;
#pragma unroll
  for (int i = 0; i < 2; ++i) fa[i] = (unsigned)(wm * 64 + i * 32 + (lane & 31));
  fb = (unsigned)(wn * 32 + (lane & 31));
  const unsigned hi = lane >> 5;

  i32x16_t acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0;

  // offsets / mask of the tile's pixels -> LDS, once
  if constexpr (ENG) {
    const __half *om = reinterpret_cast<const __half *>(offset);
    for (int idx = tid; idx < 3 * KK * kN; idx += kTh) {
      const int p = idx / (3 * KK), t = idx - p * (3 * KK);   // a pixel's 3 KK values are contiguous
      const int n = n0 + p;
      int q = 0;
      if (n < N) {
        float v = __half2float(om[(size_t)n * eng.om_channels + (size_t)dg * 3 * KK + t]), sc = s_off;
        if (t >= 2 * KK) {   // mask = sigmoid(logit), rounded to fp16 as the fp16 block's tensor is
          v = __half2float(__float2half_rn(1.f / (1.f + __expf(-v))));
          sc = s_mask;
        }
        q = (int)fminf(fmaxf(rintf(v / sc), -127.f), 127.f);
      }
      Om[t * kN + p] = (int8_t)q;
    }
  } else
  for (int idx = tid; idx < 3 * KK * kN; idx += kTh) {
    const int p = idx % kN, t = idx / kN;
    const int n = n0 + p;
    int8_t v = 0;
    if (n < N) {
      const int b = n / HoWo, pix = n - b * HoWo;
      v = t < 2 * KK ? offset[(((size_t)b * d.DG + dg) * 2 * KK + t) * HoWo + pix]
                     : mask[(((size_t)b * d.DG + dg) * KK + (t - 2 * KK)) * HoWo + pix];
    }
    Om[t * kN + p] = v;
  }
  __syncthreads();

  const int chunks = cin_g / kStepK;
  const int taps_per_part = is_tail ? KK / tp.split : KK;
  const bool stagger = (size_t)cout_g * Kp <= (size_t)(2 << 20);
  const int tap_begin = is_tail ? part * taps_per_part : (stagger ? ntile % KK : 0);
  const int n_my_steps = taps_per_part * chunks;

  int fidx[4];
  unsigned f_aw = 0, c_aw = 0;   // packed u8 area weights of the gather / of the corners held in rb
  int f_neg = 0, c_neg = 0;      // -128 * their sum
  float f_m = 0.f, c_m = 0.f;    // mask value
  uint4 rb[4];
  if constexpr (ABL & 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) rb[q] = make_uint4(tid + q, tid * 3u, tid * 5u + q, tid * 7u);
  }
  auto footprint = [&](int tap) {
    const int i = tap / d.Kw, j = tap - i * d.Kw;
    const int qoh = Om[(2 * tap) * kN + pp], qow = Om[(2 * tap + 1) * kN + pp], qm = Om[(2 * KK + tap) * kN + pp];
    float h_im, w_im;
    {
#pragma clang fp contract(off)
      const float off_h = (float)qoh * s_off, off_w = (float)qow * s_off;
      f_m = (float)qm * s_mask;
      h_im = off_h + (float)(pho * d.sh - d.ph + i * d.dh);
      w_im = off_w + (float)(pwo * d.sw - d.pw + j * d.dw);
    }
    const bool in = pvalid && h_im > -1.f && w_im > -1.f && h_im < (float)d.H && w_im < (float)d.W;
    unsigned aw[4] = {0u, 0u, 0u, 0u};
    int hs[4] = {0, 0, 0, 0}, ws[4] = {0, 0, 0, 0};
    if (in) {
#pragma clang fp contract(off)
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h0 = (int)hf, w0 = (int)wf;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool ok[4] = {h0 >= 0 && w0 >= 0, h0 >= 0 && w0 + 1 < d.W, h0 + 1 < d.H && w0 >= 0,
                          h0 + 1 < d.H && w0 + 1 < d.W};
      const float mf = FAST ? f_m : 1.f;
      const int a4[4] = {u8w(hh * hw * mf), u8w(hh * lw * mf), u8w(lh * hw * mf), u8w(lh * lw * mf)};
      const int hq[4] = {h0, h0, h0 + 1, h0 + 1}, wq[4] = {w0, w0 + 1, w0, w0 + 1};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        aw[q] = ok[q] ? (unsigned)a4[q] : 0u;
        hs[q] = ok[q] ? hq[q] : 0;
        ws[q] = ok[q] ? wq[q] : 0;
      }
    }
    f_aw = aw[0] | (aw[1] << 8) | (aw[2] << 16) | (aw[3] << 24);
    f_neg = -(int)((aw[0] + aw[1] + aw[2] + aw[3]) << 7);
#pragma unroll
    for (int q = 0; q < 4; ++q) fidx[q] = (int)(ximg_off + (unsigned)(hs[q] * d.W + ws[q]) * (unsigned)d.Cin);
  };
  // the gather pipeline and the weight pipeline each walk (tap, chunk) with their own counters
  int g_tap = tap_begin, g_chunk = 0, w_tap = tap_begin, w_chunk = 0;
  auto gather_next = [&]() {      // corners of the gather pipeline's current step -> rb; then advance
    if (g_chunk == 0) footprint(g_tap);
    if constexpr (!(ABL & 1)) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        rb[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, fidx[q], g_chunk * kStepK, 0));
    }
    if (++g_chunk == chunks) { g_chunk = 0; g_tap = g_tap + 1 < KK ? g_tap + 1 : 0; }
  };
  typedef __attribute__((address_space(3))) void lds_void;
  auto weights_next = [&](int buf) {
    const int a_s = w_tap * cin_g + w_chunk * kStepK;
    char *adst = smem + buf * kGA + wave * (kPcS * 1024);
    if constexpr (!(ABL & 4)) {
#pragma unroll
      for (int j = 0; j < kPcS; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void *)(adst + j * 1024), 16, (int)a_off[j], a_s, 0, 0);
    }
    if (++w_chunk == chunks) { w_chunk = 0; w_tap = w_tap + 1 < KK ? w_tap + 1 : 0; }
  };
  const int magic = 65793, half = 1 << 23;   // round(2^24 / 255): exact T2int8(t / 255), see msda_hm4.hip
  // corners in rb (footprint c_*) -> the pixel's 16 column bytes in buffer `buf`: per group of 4 channels the
  // byte transposes, the dots with the first requantisation (top byte), the mask multiply + rounding
  auto produce = [&](int buf) {
    constexpr unsigned kX = ENG ? 0x80808080u : 0u;   // signed activation bytes -> the u8-biased form of the dot
    const unsigned c0w[4] = {rb[0].x ^ kX, rb[0].y ^ kX, rb[0].z ^ kX, rb[0].w ^ kX};
    const unsigned c1w[4] = {rb[1].x ^ kX, rb[1].y ^ kX, rb[1].z ^ kX, rb[1].w ^ kX};
    const unsigned c2w[4] = {rb[2].x ^ kX, rb[2].y ^ kX, rb[2].z ^ kX, rb[2].w ^ kX};
    const unsigned c3w[4] = {rb[3].x ^ kX, rb[3].y ^ kX, rb[3].z ^ kX, rb[3].w ^ kX};
    unsigned res[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if constexpr (ABL & 2) {
        res[v] = c0w[v] ^ c1w[v] ^ c2w[v] ^ (c3w[v] + __float_as_uint(c_m));
      } else {
        unsigned tr[4];
        s8_transpose(c0w[v], c1w[v], c2w[v], c3w[v], tr);
        int xq[4];
        s8_quad(tr, c_aw, c_neg, magic, half, xq);
        if constexpr (FAST) {   // the requantised value sits in the top byte of each xq[c]
          res[v] = __builtin_amdgcn_perm((unsigned)xq[1], (unsigned)xq[0], 0x0c0c0703u) |
                   __builtin_amdgcn_perm((unsigned)xq[3], (unsigned)xq[2], 0x07030c0cu);
        } else {
          res[v] = s8_mask4(xq, c_m);
        }
      }
    }
    // hand-written store: behind a compiler-visible LDS store the waitcnt pass drains vmcnt to 0 (it cannot
    // tell the store from the rows the weight DMA is filling), which would stall on loads still in flight;
    // the s_waitcnt lgkmcnt(0) in front of the barrier covers it
    const u32x4_t rv = {res[0], res[1], res[2], res[3]};
    asm volatile("ds_write_b128 %0, %1" ::"v"(b_lds + (unsigned)(buf * kGB)), "v"(rv) : "memory");
  };
  auto next_gathers = [&]() {
    gather_next();
    c_aw = f_aw; c_neg = f_neg; c_m = f_m;
  };

  // prologue: step 0 -> buffer 0, corners of step 1 in flight
  next_gathers();
  produce(0);
  __builtin_amdgcn_sched_barrier(0);
  const bool upper = __builtin_amdgcn_readfirstlane(wave) >= kTh / 128;
  if (!kLowerDma || !upper) weights_next(0);
  if (n_my_steps > 1) next_gathers();
  if (n_my_steps > 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // iteration `step`: tiles of `step` in buffer step & 1, raw corners of step + 1 in rb (a step old).  The
  // weight DMA of step + 1 is always issued BEFORE the gathers of step + 2, so vmcnt(4) in front of the
  // barrier retires it and leaves the gathers in flight.
  // ABL & 16: the two halves of the block's waves run the step's phases in opposite order (every SIMD holds
  // two waves of each half): while the early half does its VALU work the late half runs the MFMAs + fragment
  // reads.  Measured no faster (68.6 vs 67.6 us): the producer VALU work is ~70 % of the SIMD time whatever
  // the order (profiles/r02/int8_dcn_pmc.txt), so the default keeps all waves in the same order.
  const bool late = (ABL & 16) && __builtin_amdgcn_readfirstlane(wave) >= kTh / 128;
  auto mfma_step = [&](int buf) {
    if constexpr (ABL & 8) return;
    const char *Ab = smem + buf * kGA;
    const char *Bb = smem + 2 * kGA + buf * kGB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned c = 2u * ks + hi;
      i32x4_t a[2], b;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const i32x4_t *>(Ab + fa[i] * 128 + ((c ^ swz8(fa[i])) << 4));
      b = *reinterpret_cast<const i32x4_t *>(Bb + fb * 128 + ((c ^ swz8(fb)) << 4));
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b, acc[i], 0, 0, 0);
    }
  };
  // ONE code path for both halves: the late half is the same loop rotated by half an iteration -- its barrier
  // sits between the producer work and the MFMAs, and it runs MFMA(step + 1) where the early half runs
  // MFMA(step); MFMA(0) is peeled.  Between two barriers the early half does producer(s + 1), DMA, gathers,
  // MFMA(s); the late half DMA, gathers, MFMA(s), producer(s + 1).
  if constexpr (kLowerDma) {
    // Round 6 (the fp16 kernel's finding, mdconv.hip / design/dcn.md): a step's vector-memory instructions leave the
    // CU's load path oldest wave first at ~0.4 lines per clock, and a wave cannot start its matrix segment before its
    // own loads are accepted.  The lower half of the waves issues ALL the weight pieces and runs
    //   producer -> DMA -> gathers -> MFMA,    the upper half    MFMA -> producer -> gathers:
    // an upper wave has no load in front of its matrix segment.  One loop body; same buffers, same sums (int32: exact).
    for (int step = 0; step < n_my_steps; ++step) {
      const bool more1 = step + 1 < n_my_steps, more2 = step + 2 < n_my_steps;
      if (!upper && more1) produce((step + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (!upper && more1) weights_next((step + 1) & 1);
      if (!upper && more2) next_gathers();
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(step & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (upper && more1) produce((step + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (upper && more2) next_gathers();
      __builtin_amdgcn_sched_barrier(0);
      if (more2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
  if (late) {
    if (n_my_steps > 1) weights_next(1);
    mfma_step(0);
  }
  for (int step = 0; step < n_my_steps; ++step) {
    const bool more1 = step + 1 < n_my_steps, more2 = step + 2 < n_my_steps;
    if (more1) produce((step + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    if (late) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (late ? more2 : more1) weights_next(late ? (step & 1) : ((step + 1) & 1));
    if (more2) next_gathers();
    __builtin_amdgcn_sched_barrier(0);
    if (late ? more1 : true) mfma_step(late ? ((step + 1) & 1) : (step & 1));
    if (!late) {
      if (more2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  }   // (!kLowerDma)
  if (is_tail) {  // int32 partials (exact in any order), thread-private layout shared with the finish kernel
    int4 *pq = reinterpret_cast<int4 *>(tp.partial) +
               ((((size_t)part * tp.tail_tiles + (ntile - tp.main_tiles)) * gridDim.y + blockIdx.y) * 8) * kTh + tid;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pq[(i * 4 + r) * kTh] = make_int4(acc[i][4 * r], acc[i][4 * r + 1], acc[i][4 * r + 2], acc[i][4 * r + 3]);
    return;
  }
  const int n = n0 + wn * 32 + (lane & 31);
  if (n >= N) return;
  if constexpr (ENG) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int av[4] = {acc[i][4 * rq], acc[i][4 * rq + 1], acc[i][4 * rq + 2], acc[i][4 * rq + 3]};
        s8_store4_nhwc(out, bias, av, n, m0 + wm * 64 + i * 32 + 8 * rq + 4 * (lane >> 5), g, cout_g, d.Cout, s_iw,
                       s_out, eng.relu);
      }
    return;
  }
  const int b = n / HoWo, pix = n - b * HoWo;
  int8_t *ob = out + ((size_t)b * d.Cout + g * cout_g) * HoWo + pix;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m < cout_g) {
#pragma clang fp contract(off)
        const float v = ((float)acc[i][r] * s_iw + (bias ? bias[g * cout_g + m] : 0.f)) / s_out;
        ob[(size_t)m * HoWo] = (int8_t)q_away(v);
      }
    }
}

// grid (tail tiles, Cout tiles, 8): block z sums accumulator quad z = i*4 + rq of every thread, then requantises
template <int WN, bool ENG = false>
__global__ __launch_bounds__(256 * WN) void dcn_tail_finish_s8_kernel(const float *__restrict__ bias,
                                                                 int8_t *__restrict__ out, ConvDims d, int g,
                                                                 float s_iw, float s_out, TailPlan tp, S8Eng eng) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int cout_g = d.Cout / d.G, HoWo = d.Ho * d.Wo, N = d.B * HoWo;
  const int ntile = tp.main_tiles + blockIdx.x;
  constexpr int kTh = Glds<WN>::kThreads;
  const int n0 = ntile * Glds<WN>::kN, m0 = blockIdx.y * kFM;
  const int quad = blockIdx.z, i = quad >> 2, rq = quad & 3;
  int4 a = make_int4(0, 0, 0, 0);
  for (int part = 0; part < tp.split; ++part) {
    const int4 v = reinterpret_cast<const int4 *>(tp.partial)[
        ((((size_t)part * tp.tail_tiles + blockIdx.x) * gridDim.y + blockIdx.y) * 8 + quad) * kTh + tid];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  const int n = n0 + wn * 32 + (lane & 31);
  if (n >= N) return;
  const int b = n / HoWo, pix = n - b * HoWo;
  const int av[4] = {a.x, a.y, a.z, a.w};
  if constexpr (ENG) {
    s8_store4_nhwc(out, bias, av, n, m0 + wm * 64 + i * 32 + 8 * rq + 4 * (lane >> 5), g, cout_g, d.Cout, s_iw, s_out,
                   eng.relu);
    return;
  }
  int8_t *ob = out + ((size_t)b * d.Cout + g * cout_g) * HoWo + pix;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = m0 + wm * 64 + i * 32 + 8 * rq + 4 * (lane >> 5) + e;
    if (m < cout_g) {
#pragma clang fp contract(off)
      const float v = ((float)av[e] * s_iw + (bias ? bias[g * cout_g + m] : 0.f)) / s_out;
      ob[(size_t)m * HoWo] = (int8_t)q_away(v);
    }
  }
}

template <int WN, int ABL, bool ENG = false, bool FAST = false>
int glds_s8_resident_blocks() {
  static thread_local int cached_dev = -1, cached = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev == cached_dev) return cached;
  int per_cu = 0, cus = 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(dcn_glds_s8_kernel<WN, ABL, ENG, FAST>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, GldsS8<WN>::kLds) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dcn_glds_s8_kernel<WN, ABL, ENG, FAST>, Glds<WN>::kThreads,
                                                   GldsS8<WN>::kLds) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return 0;
  cached_dev = dev;
  cached = per_cu * cus;
  return cached;
}

// launch of the int8 LDS-DMA kernel with its tail plan (same plan as launch_glds; int32 partials)
template <int WN, int ABL, bool ENG = false, bool FAST = false>
int launch_glds_s8(const int8_t *xt, const void *offset, const void *mask, const int8_t *wt, const void *bias,
                   void *output, const ConvDims &d, int g, int Kp, char *part_ws, size_t part_room, bool allow_tail,
                   float s_off, float s_mask, float s_iw, float s_out, hipStream_t st, S8Eng eng = S8Eng{0, 0}) {
  const int KK = d.Kh * d.Kw, cout_g = d.Cout / d.G;
  const size_t N = (size_t)d.B * d.Ho * d.Wo;
  const dim3 grid((unsigned)((N + Glds<WN>::kN - 1) / Glds<WN>::kN), (cout_g + kFM - 1) / kFM);
  const int slots = glds_s8_resident_blocks<WN, ABL, ENG, FAST>();
  if (slots <= 0) return BEVOPS_FAILURE;
  TailPlan tp{0, 1, (int)grid.x, nullptr, 0, 0, 0, 0};
  const int blocks = (int)(grid.x * grid.y);
  if (allow_tail && blocks > slots && grid.y == 1) {
    const int left = blocks % slots;
    int split = 0;
    for (int f = KK; f >= 2; --f)
      if (KK % f == 0 && left * f <= slots) { split = f; break; }
    const size_t need = (size_t)split * left * Glds<WN>::kThreads * 32 * sizeof(int);
    if (left > 0 && left * 2 <= slots && split >= 2 && part_room >= need) {
      tp.tail_tiles = left;
      tp.split = split;
      tp.main_tiles = (int)grid.x - left;
      tp.partial = reinterpret_cast<float *>(part_ws);
    }
  }
  const dim3 grid2((unsigned)(tp.tail_tiles * tp.split + tp.main_tiles), grid.y);
  hipLaunchKernelGGL((dcn_glds_s8_kernel<WN, ABL, ENG, FAST>), grid2, dim3(Glds<WN>::kThreads), GldsS8<WN>::kLds, st, xt,
                     (const int8_t *)offset, (const int8_t *)mask, wt, (const float *)bias, (int8_t *)output, d, g,
                     Kp, s_off, s_mask, s_iw, s_out, tp, eng);
  if (tp.tail_tiles)
    hipLaunchKernelGGL((dcn_tail_finish_s8_kernel<WN, ENG>), dim3((unsigned)tp.tail_tiles, grid.y, 8),
                       dim3(Glds<WN>::kThreads), 0, st, (const float *)bias, (int8_t *)output, d, g, s_iw, s_out, tp,
                       eng);
  return launch_status();
}

int run_s8(const void *input, const void *offset, const void *mask, const void *weight,
           const void *bias, void *output, void *workspace, const ConvDims &d, float s_in, float s_off,
           float s_mask, float s_w, float s_out, hipStream_t st, bool weight_is_packed) {
  const WsLayout w = ws_layout(d, 1);
  char *ws = static_cast<char *>(workspace);
  int8_t *xt = reinterpret_cast<int8_t *>(ws + w.xt);
  int8_t *wt = reinterpret_cast<int8_t *>(ws + w.wt);
  int8_t *col = reinterpret_cast<int8_t *>(ws + w.col);
  const int HW = d.H * d.W, KK = d.Kh * d.Kw, cin_g = d.Cin / d.G, cout_g = d.Cout / d.G;
  const size_t N = (size_t)d.B * d.Ho * d.Wo;
  if (N > 0x7FFFFFFFull) return BEVOPS_NOT_SUPPORTED;
  // multiScale...Plugin-style precondition (modulatedDeformableConv2dPlugin.cpp:217-219)
  if (d.Cin % 4 != 0 || cout_g % 4 != 0) return BEVOPS_NOT_SUPPORTED;
  const int Kg = KK * cin_g;
  const int Kp = (int)kpad(d, 1);
  if (Kp != Kg) {  // zero the padding columns once (packed weights + column buffer)
    if (hipMemsetAsync(ws + w.wt, 0, w.total - w.wt, st) != hipSuccess) return BEVOPS_FAILURE;
  }
  if (weight_is_packed)  // [Cout][Kp] image made by bevops_mdconv_pack_weight(BEVOPS_I8, ...), padding zeroed there
    wt = const_cast<int8_t *>(static_cast<const int8_t *>(weight));
  // fused implicit GEMM (no column buffer) when a 64-channel chunk stays inside one group and one
  // deform group; variant 6 keeps the im2col + GEMM pair (A/B reference)
  // ... and when there are enough 256 x 64 tiles to give every CU two blocks: the base stage-3 call
  // (544 tiles) 117 vs 170 us; stage 4 (272 tiles, one 4-wave block per CU) 138 vs 130 us keeps the pair
  // (profiles/r02).  Variant 8 forces the fused kernel.
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const size_t tiles = ((N + kSN - 1) / kSN) * (size_t)((cout_g + kSM - 1) / kSM);
  const bool fits32 = (size_t)d.B * d.H * d.W * d.Cin < 0xFFFFFF00ull && (size_t)cout_g * Kp < 0xFFFFFF00ull;
  // LDS-DMA pipelined kernel (section 6c): 128-channel k-steps inside one group and ONE deform group per
  // conv group; variants 8 (register-staged fused kernel) and 6 (im2col + GEMM) keep the A/B references,
  // variants 20 + mask = timing experiments with parts of the kernel removed (wrong results)
  bool one_dg = cin_g <= d.Cin / d.DG;
  for (int g = 0; g < d.G && one_dg; ++g)
    one_dg = (g * cin_g) / (d.Cin / d.DG) == (g * cin_g + cin_g - 1) / (d.Cin / d.DG);
  // ... and enough 128-pixel tiles to give at least every second CU a block (base stage 3: 272 tiles, 67 us
  // kernel vs 100 us for the register-staged kernel; stage 4: 136 blocks, 95 us per call vs 117 us for the
  // im2col + GEMM pair and 121 us with 64-pixel tiles, profiles/r02/dcn_time.jsonl); variant 9 forces it
  const size_t wide_blocks = ((N + Glds<4>::kN - 1) / Glds<4>::kN) * ((cout_g + kFM - 1) / kFM);
  const bool glds = g_mdconv_variant != 6 && g_mdconv_variant != 8 && fits32 && one_dg && cin_g % 128 == 0 &&
                    3 * KK <= kSOmRows && Kp == Kg && (wide_blocks * 2 >= (size_t)cus || g_mdconv_variant == 9 || g_mdconv_wide);
  const bool fused = glds || (g_mdconv_variant != 6 && cin_g % kSK == 0 && (d.Cin / d.DG) % kSK == 0 && fits32 &&
                              (g_mdconv_variant == 8 || tiles >= (size_t)2 * cus));
  if (HW % 4 == 0 && d.Cin % 16 == 0 && aligned16(input))
    hipLaunchKernelGGL(nchw_to_nhwc_s8v_kernel, dim3((HW + 127) / 128, (d.Cin + 127) / 128, d.B), dim3(256), 0, st,
                       (const int8_t *)input, xt, d.Cin, HW, fused ? 0x80808080u : 0u);
  else
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<int8_t>), dim3((HW + 31) / 32, (d.Cin + 31) / 32, d.B), dim3(256),
                       0, st, (const int8_t *)input, xt, d.Cin, HW, fused ? 0x80 : 0);
  const size_t wtot = (size_t)d.Cout * cin_g * KK;
  if (!weight_is_packed)
    hipLaunchKernelGGL((repack_weight_kernel<int8_t>), dim3((unsigned)((wtot + 255) / 256)), dim3(256), 0, st,
                       (const int8_t *)weight, wt, d.Cout, cin_g, KK, Kp);
  if (glds) {
    const int abl = g_mdconv_variant >= 20 && g_mdconv_variant <= 36 ? g_mdconv_variant - 20 : 0;   // timing experiments
    for (int g = 0; g < d.G; ++g) {
      int rc = BEVOPS_NOT_SUPPORTED;
      char *pw = ws + w.col;
      const size_t room = w.total - w.col;
      const float s_iw = s_in * s_w;
#define BEVOPS_S8_GO(WN_, ABL_) \
  rc = launch_glds_s8<WN_, ABL_>(xt, offset, mask, wt, bias, output, d, g, Kp, pw, room, !g_mdconv_no_tail, s_off, s_mask, s_iw, s_out, st)
      const bool w4 = wide_blocks * 2 >= (size_t)cus || g_mdconv_wide;
      const bool one_order = g_mdconv_rotate == 2;     // variant 13: the rounds 2-5 order (A/B partner of the default)
      if (!w4) { if (one_order) BEVOPS_S8_GO(2, 0); else BEVOPS_S8_GO(2, 32); }
      else
        switch (abl) {
          case 0: if (one_order) BEVOPS_S8_GO(4, 0); else BEVOPS_S8_GO(4, 32); break;
          case 1: BEVOPS_S8_GO(4, 1); break;
          case 2: BEVOPS_S8_GO(4, 2); break;
          case 4: BEVOPS_S8_GO(4, 4); break;
          case 8: BEVOPS_S8_GO(4, 8); break;
          case 7: BEVOPS_S8_GO(4, 7); break;
          case 11: BEVOPS_S8_GO(4, 11); break;
          case 13: BEVOPS_S8_GO(4, 13); break;
          case 14: BEVOPS_S8_GO(4, 14); break;
          case 15: BEVOPS_S8_GO(4, 15); break;
          case 16: BEVOPS_S8_GO(4, 16); break;
          default: break;
        }
#undef BEVOPS_S8_GO
      if (rc != BEVOPS_SUCCESS) return rc;
    }
    return launch_status();
  }
  if (fused) {
    for (int g = 0; g < d.G; ++g)
      hipLaunchKernelGGL(dcn_fused_s8_kernel, dim3((unsigned)((N + kSN - 1) / kSN), (cout_g + kSM - 1) / kSM), dim3(256),
                         0, st, xt, (const int8_t *)offset, (const int8_t *)mask, wt, (const float *)bias,
                         (int8_t *)output, d, g, Kp, s_off, s_mask, s_in * s_w, s_out);
    return launch_status();
  }
  const bool v16 = cin_g % 16 == 0 && (d.Cin / d.DG) % 16 == 0;
  const bool v4 = cin_g % 4 == 0 && (d.Cin / d.DG) % 4 == 0;
  const int V = v16 ? 16 : (v4 ? 4 : 1);
  const size_t blocks = (N * KK * (d.Cin / V) + 255) / 256;
  if (blocks > 0x7FFFFFFFull) return BEVOPS_NOT_SUPPORTED;
  if (v16)
    hipLaunchKernelGGL((im2col_nhwc_s8_kernel<16>), dim3((unsigned)blocks), dim3(256), 0, st, xt,
                       (const int8_t *)offset, (const int8_t *)mask, col, d, s_off, s_mask, Kp);
  else if (v4)
    hipLaunchKernelGGL((im2col_nhwc_s8_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, st, xt,
                       (const int8_t *)offset, (const int8_t *)mask, col, d, s_off, s_mask, Kp);
  else
    hipLaunchKernelGGL((im2col_nhwc_s8_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, xt,
                       (const int8_t *)offset, (const int8_t *)mask, col, d, s_off, s_mask, Kp);
  for (int g = 0; g < d.G; ++g) {
    const GemmEpi e{d.Ho * d.Wo, d.Cout, g * cout_g};
    hipLaunchKernelGGL(gemm_tn_s8_kernel, dim3((unsigned)((N + kBN - 1) / kBN), (cout_g + kBM - 1) / kBM),
                       dim3(256), 0, st, wt + (size_t)g * cout_g * Kp, col + (size_t)g * N * Kp,
                       (const float *)bias, (int8_t *)output, cout_g, (int)N, Kp, e, s_in * s_w, s_out);
  }
  return launch_status();
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" int bevops_mdconv_forward_int8(const void *input, float scale_in, const void *offset,
                                          float scale_offset, const void *mask, float scale_mask,
                                          const void *weight, float scale_weight, const float *bias,
                                          void *output, float scale_out, void *workspace,
                                          size_t workspace_bytes, int B, int Cin, int H, int W, int Cout,
                                          int Kh, int Kw, int stride_h, int stride_w, int pad_h, int pad_w,
                                          int dil_h, int dil_w, int groups, int deform_groups,
                                          void *stream) {
  if (!input || !offset || !mask || !weight || !output || !workspace) return BEVOPS_BAD_PARAM;
  if (!(scale_in > 0.f) || !(scale_offset > 0.f) || !(scale_mask > 0.f) || !(scale_weight > 0.f) ||
      !(scale_out > 0.f))
    return BEVOPS_BAD_PARAM;
  ConvDims d;
  if (!make_dims(d, B, Cin, H, W, Cout, Kh, Kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w,
                 groups, deform_groups))
    return BEVOPS_BAD_PARAM;
  if (workspace_bytes < ws_layout(d, 1).total || !aligned16(workspace)) return BEVOPS_BAD_PARAM;
  return run_s8(input, offset, mask, weight, bias, output, workspace, d, scale_in, scale_offset,
                scale_mask, scale_weight, scale_out, static_cast<hipStream_t>(stream), false);
}

extern "C" int bevops_mdconv_forward_int8_packed(const void *input, float scale_in, const void *offset,
                                                 float scale_offset, const void *mask, float scale_mask,
                                                 const void *packed_weight, float scale_weight, const float *bias,
                                                 void *output, float scale_out, void *workspace,
                                                 size_t workspace_bytes, int B, int Cin, int H, int W, int Cout,
                                                 int Kh, int Kw, int stride_h, int stride_w, int pad_h, int pad_w,
                                                 int dil_h, int dil_w, int groups, int deform_groups,
                                                 void *stream) {
  if (!input || !offset || !mask || !packed_weight || !output || !workspace) return BEVOPS_BAD_PARAM;
  if (!(scale_in > 0.f) || !(scale_offset > 0.f) || !(scale_mask > 0.f) || !(scale_weight > 0.f) ||
      !(scale_out > 0.f))
    return BEVOPS_BAD_PARAM;
  ConvDims d;
  if (!make_dims(d, B, Cin, H, W, Cout, Kh, Kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w,
                 groups, deform_groups))
    return BEVOPS_BAD_PARAM;
  if (workspace_bytes < ws_layout(d, 1).total || !aligned16(workspace) || !aligned16(packed_weight))
    return BEVOPS_BAD_PARAM;
  return run_s8(input, offset, mask, packed_weight, bias, output, workspace, d, scale_in, scale_offset,
                scale_mask, scale_weight, scale_out, static_cast<hipStream_t>(stream), true);
}

// The INT8 engine's channels-last DCNv2 block (not a reference plugin; its arithmetic is the INT8 plugin's,
// modulatedDeformableConv2dKernel.cu:463-607,897-978, on the operands a TensorRT INT8 engine hands it): the
// activation int8 [B, H, W, Cin] in and int8 [B, Ho, Wo, Cout] out, offsets / mask logits the raw fp16
// [B, Ho, Wo, om_channels] output of the pack's offset convolution (quantised with scale_offset / scale_mask
// while staged), weights packed by bevops_mdconv_pack_weight(BEVOPS_I8), ReLU folded into the requantisation.
// Domain of the LDS-DMA kernel: (Cin / groups) % 128 == 0, one deform group per conv group, 3 Kh Kw <= 32.
extern "C" int bevops_mdconv_forward_int8_nhwc(const void *input_nhwc, float scale_in, const void *offset_mask_nhwc,
                                               int offset_mask_channels, float scale_offset, float scale_mask,
                                               const void *packed_weight, float scale_weight, const float *bias,
                                               void *output_nhwc, float scale_out, int relu, int exact, void *workspace,
                                               size_t workspace_bytes, int B, int Cin, int H, int W, int Cout, int Kh,
                                               int Kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h,
                                               int dil_w, int groups, int deform_groups, void *stream) {
  if (!input_nhwc || !offset_mask_nhwc || !packed_weight || !output_nhwc) return BEVOPS_BAD_PARAM;
  if (!(scale_in > 0.f) || !(scale_offset > 0.f) || !(scale_mask > 0.f) || !(scale_weight > 0.f) || !(scale_out > 0.f))
    return BEVOPS_BAD_PARAM;
  ConvDims d;
  if (!make_dims(d, B, Cin, H, W, Cout, Kh, Kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, groups, deform_groups))
    return BEVOPS_BAD_PARAM;
  const int KK = Kh * Kw, cin_g = Cin / groups, cout_g = Cout / groups;
  if (offset_mask_channels < deform_groups * 3 * KK) return BEVOPS_BAD_PARAM;
  // the raw offset-convolution output is read as [2 KK offsets | KK mask logits] of ONE deform group (the layout of
  // mmcv's cat(o1, o2) | mask for deform_groups == 1, modules/cnn/dcn.py:70-74); several groups interleave differently
  if (deform_groups != 1) return BEVOPS_NOT_SUPPORTED;
  if (!aligned16(input_nhwc) || !aligned16(packed_weight) || (reinterpret_cast<uintptr_t>(output_nhwc) & 3u) ||
      (reinterpret_cast<uintptr_t>(offset_mask_nhwc) & 1u))
    return BEVOPS_BAD_PARAM;
  const size_t N = (size_t)B * d.Ho * d.Wo;
  const int Kp = (int)kpad(d, 1);
  bool one_dg = cin_g <= Cin / deform_groups;
  for (int g = 0; g < groups && one_dg; ++g)
    one_dg = (g * cin_g) / (Cin / deform_groups) == (g * cin_g + cin_g - 1) / (Cin / deform_groups);
  const bool fits32 = (size_t)B * H * W * Cin < 0xFFFFFF00ull && (size_t)cout_g * Kp < 0xFFFFFF00ull;
  if (N > 0x7FFFFFFFull || !fits32 || !one_dg || cin_g % 128 != 0 || 3 * KK > kSOmRows || Kp != KK * cin_g ||
      cout_g % 4 != 0)
    return BEVOPS_NOT_SUPPORTED;
  // the split-K tail's int32 partials live in the caller's workspace (none lent: every tile runs its whole k-loop)
  char *pw = static_cast<char *>(workspace);
  const size_t room = (workspace && aligned16(workspace)) ? workspace_bytes : 0;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const size_t wide_blocks = ((N + Glds<4>::kN - 1) / Glds<4>::kN) * ((cout_g + kFM - 1) / kFM);
  const bool w4 = wide_blocks * 2 >= (size_t)cus || g_mdconv_wide;
  const S8Eng eng{offset_mask_channels, relu ? 1 : 0};
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int g = 0; g < groups; ++g) {
#define BEVOPS_S8E(WN_, FAST_) (g_mdconv_rotate == 2 ? BEVOPS_S8E_(WN_, FAST_, 0) : BEVOPS_S8E_(WN_, FAST_, 32))
#define BEVOPS_S8E_(WN_, FAST_, ABL_)                                                                                \
  launch_glds_s8<WN_, ABL_, true, FAST_>((const int8_t *)input_nhwc, offset_mask_nhwc, nullptr, (const int8_t *)packed_weight, \
                                      bias, output_nhwc, d, g, Kp, pw, room, !g_mdconv_no_tail, scale_offset, scale_mask,   \
                                      scale_in * scale_weight, scale_out, st, eng)
    const int rc = exact ? (w4 ? BEVOPS_S8E(4, false) : BEVOPS_S8E(2, false)) : (w4 ? BEVOPS_S8E(4, true) : BEVOPS_S8E(2, true));
#undef BEVOPS_S8E
#undef BEVOPS_S8E_
    if (rc != BEVOPS_SUCCESS) return rc;
  }
  return launch_status();
}

// bytes of workspace bevops_mdconv_forward_int8_nhwc can use for its split-K tail (nothing lent = no tail split):
// the plan of launch_glds_s8 keeps split x tail tiles within the resident block slots (<= 1024 threads per CU),
// each thread parks 32 int32 partial sums
extern "C" size_t bevops_mdconv_int8_nhwc_workspace_size(void) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return (size_t)(cus > 0 ? cus : 256) * 1024 * 32 * sizeof(int);
}
