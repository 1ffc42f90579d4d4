/* cuda_on_cpu/cublas_v2.h -- TEST INFRASTRUCTURE ONLY.  cublasGemmEx as a plain column-major
 * triple loop on the host for the three type combinations the reference's
 * common/cuda_helper.cu instantiates (f32, f16 -> f32 accumulate, rounded once; s8 -> s32
 * exact).  Accumulation order is k-ascending; cuBLAS's is unspecified, so float results are
 * compared with a tolerance, integer results exactly. */
#ifndef CUDA_ON_CPU_CUBLAS_H
#define CUDA_ON_CPU_CUBLAS_H
#include "cuda_fp16.h"

typedef struct cublasContext *cublasHandle_t;
typedef enum { CUBLAS_STATUS_SUCCESS = 0, CUBLAS_STATUS_NOT_SUPPORTED = 15 } cublasStatus_t;
typedef enum { CUBLAS_OP_N = 0, CUBLAS_OP_T = 1 } cublasOperation_t;
typedef enum { CUBLAS_GEMM_DFALT = -1, CUBLAS_GEMM_DFALT_TENSOR_OP = 99 } cublasGemmAlgo_t;
typedef enum { CUDA_R_16F = 2, CUDA_R_32F = 0, CUDA_R_8I = 3, CUDA_R_32I = 10 } cudaDataType;
typedef cudaDataType cublasComputeType_t;

namespace cuda_cpu {
template <class TI, class TO, class ACC, class SC>
inline void gemm(cublasOperation_t ta, cublasOperation_t tb, int m, int n, int k, SC alpha, const TI *A, int lda,
                 const TI *B, int ldb, SC beta, TO *C, int ldc) {
#pragma omp parallel for
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) {
      ACC acc = 0;
      for (int p = 0; p < k; ++p) {
        const TI a = ta == CUBLAS_OP_N ? A[(size_t)p * lda + i] : A[(size_t)i * lda + p];
        const TI b = tb == CUBLAS_OP_N ? B[(size_t)j * ldb + p] : B[(size_t)p * ldb + j];
        acc += (ACC)a * (ACC)b;
      }
      TO &c = C[(size_t)j * ldc + i];
      const ACC b0 = (ACC)beta;
      c = (TO)((ACC)alpha * acc + (b0 == (ACC)0 ? (ACC)0 : b0 * (ACC)c));
    }
}
}  // namespace cuda_cpu

inline cublasStatus_t cublasGemmEx(cublasHandle_t, cublasOperation_t ta, cublasOperation_t tb, int m, int n, int k,
                                   const void *alpha, const void *A, cudaDataType at, int lda, const void *B,
                                   cudaDataType bt, int ldb, const void *beta, void *C, cudaDataType ct, int ldc,
                                   cublasComputeType_t, cublasGemmAlgo_t) {
  if (at == CUDA_R_32F && bt == CUDA_R_32F && ct == CUDA_R_32F) {
    cuda_cpu::gemm<float, float, float, float>(ta, tb, m, n, k, *(const float *)alpha, (const float *)A, lda,
                                               (const float *)B, ldb, *(const float *)beta, (float *)C, ldc);
  } else if (at == CUDA_R_16F && bt == CUDA_R_16F && ct == CUDA_R_16F) {
    cuda_cpu::gemm<__half, __half, float, float>(ta, tb, m, n, k, __half2float(*(const __half *)alpha),
                                                 (const __half *)A, lda, (const __half *)B, ldb,
                                                 __half2float(*(const __half *)beta), (__half *)C, ldc);
  } else if (at == CUDA_R_8I && bt == CUDA_R_8I && ct == CUDA_R_32I) {
    cuda_cpu::gemm<int8_t, int32_t, int32_t, int32_t>(ta, tb, m, n, k, *(const int32_t *)alpha, (const int8_t *)A,
                                                      lda, (const int8_t *)B, ldb, *(const int32_t *)beta,
                                                      (int32_t *)C, ldc);
  } else {
    return CUBLAS_STATUS_NOT_SUPPORTED;
  }
  return CUBLAS_STATUS_SUCCESS;
}

inline cublasStatus_t cublasGemmBatchedEx(cublasHandle_t h, cublasOperation_t ta, cublasOperation_t tb, int m, int n,
                                          int k, const void *alpha, const void *const A[], cudaDataType at, int lda,
                                          const void *const B[], cudaDataType bt, int ldb, const void *beta,
                                          void *const C[], cudaDataType ct, int ldc, int batch, cublasComputeType_t cc,
                                          cublasGemmAlgo_t algo) {
  for (int b = 0; b < batch; ++b) {
    cublasStatus_t s = cublasGemmEx(h, ta, tb, m, n, k, alpha, A[b], at, lda, B[b], bt, ldb, beta, C[b], ct, ldc, cc, algo);
    if (s != CUBLAS_STATUS_SUCCESS) return s;
  }
  return CUBLAS_STATUS_SUCCESS;
}

inline cublasStatus_t cublasGemmStridedBatchedEx(cublasHandle_t h, cublasOperation_t ta, cublasOperation_t tb, int m,
                                                 int n, int k, const void *alpha, const void *A, cudaDataType at,
                                                 int lda, long long sa, const void *B, cudaDataType bt, int ldb,
                                                 long long sb, const void *beta, void *C, cudaDataType ct, int ldc,
                                                 long long sc, int batch, cublasComputeType_t cc,
                                                 cublasGemmAlgo_t algo) {
  const size_t ea = at == CUDA_R_32F ? 4 : (at == CUDA_R_16F ? 2 : 1), ec = ct == CUDA_R_16F ? 2 : 4;
  for (int b = 0; b < batch; ++b) {
    cublasStatus_t s = cublasGemmEx(h, ta, tb, m, n, k, alpha, (const char *)A + b * sa * ea, at, lda,
                                    (const char *)B + b * sb * ea, bt, ldb, beta, (char *)C + b * sc * ec, ct, ldc, cc,
                                    algo);
    if (s != CUBLAS_STATUS_SUCCESS) return s;
  }
  return CUBLAS_STATUS_SUCCESS;
}
#endif
