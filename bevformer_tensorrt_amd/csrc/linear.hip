// bevops_linear_bias_act: out[M, N] = act(A[M, K] . W[N, K]^T + bias[N] + residual[M, N]), fp16 in
// and out, fp32 accumulate and epilogue, ONE library GEMM.  The dense layers that wrap the sampler
// (value_proj / output_proj / FFN; 8f-4: the 1x1 convolutions of the channels-last
// backbone) are plain GEMMs and stay on hipBLASLt's MFMA kernels -- what this entry adds is the
// epilogue a framework cannot ask for in one call: shift + identity + ReLU
// (D = relu(alpha A B + beta C + bias) with C = the residual), so the bottleneck's
// "GEMM, then a shift + identity + ReLU pass" and the attention blocks' "output_proj, then + identity"
// are one launch with no extra trip of the activation through HBM.
// Column-major view used for hipBLASLt: D^T is [N x M] (ld N) = W (stored [N][K] = col-major
// [K x N], ld K, op T) x A (stored [M][K] = col-major [K x M], ld K, op N); the bias runs along the
// rows of D^T, i.e. along N.  Not a reference plugin.
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

#include "common.h"

namespace bevops {
namespace {

struct Plan {
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
  bool ok = false;
};

using Key = std::tuple<int, long long, int, int, int, int, int, size_t>;  // dev, M, N, K, relu, bias, res, ws

std::mutex g_mu;
std::map<int, hipblasLtHandle_t> g_handles;  // one per device, created on first use
std::map<Key, Plan> g_plans;

hipblasLtMatmulDesc_t make_desc(bool relu, const void *bias, bool has_bias) {
  hipblasLtMatmulDesc_t d = nullptr;
  if (hipblasLtMatmulDescCreate(&d, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return nullptr;
  const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
  const hipblasLtEpilogue_t epi = has_bias ? (relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS)
                                           : (relu ? HIPBLASLT_EPILOGUE_RELU : HIPBLASLT_EPILOGUE_DEFAULT);
  const hipDataType bt = HIP_R_16F;
  bool ok = hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)) == HIPBLAS_STATUS_SUCCESS &&
            hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)) == HIPBLAS_STATUS_SUCCESS &&
            hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) == HIPBLAS_STATUS_SUCCESS;
  if (ok && has_bias) {
    ok = hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)) == HIPBLAS_STATUS_SUCCESS;
    if (ok && bias)
      ok = hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) == HIPBLAS_STATUS_SUCCESS;
  }
  if (!ok) {
    hipblasLtMatmulDescDestroy(d);
    return nullptr;
  }
  return d;
}

bool make_plan(hipblasLtHandle_t h, Plan &p, long long M, int N, int K, bool relu, bool has_bias,
               size_t ws_bytes) {
  hipblasLtMatmulDesc_t desc = make_desc(relu, nullptr, has_bias);
  if (!desc) return false;
  struct Guard {
    hipblasLtMatmulDesc_t d;
    ~Guard() { hipblasLtMatmulDescDestroy(d); }
  } guard{desc};
  if (hipblasLtMatrixLayoutCreate(&p.a, HIP_R_16F, (uint64_t)K, (uint64_t)N, K) != HIPBLAS_STATUS_SUCCESS ||  // W
      hipblasLtMatrixLayoutCreate(&p.b, HIP_R_16F, (uint64_t)K, (uint64_t)M, K) != HIPBLAS_STATUS_SUCCESS ||  // A
      hipblasLtMatrixLayoutCreate(&p.c, HIP_R_16F, (uint64_t)N, (uint64_t)M, N) != HIPBLAS_STATUS_SUCCESS)    // C, D
    return false;
  hipblasLtMatmulPreference_t pref = nullptr;
  if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return false;
  const uint64_t wsb = ws_bytes;
  hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb));
  hipblasLtMatmulHeuristicResult_t res[4];
  int n = 0;
  const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(h, desc, p.a, p.b, p.c, p.c, pref, 4, res, &n);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS) return false;
  for (int i = 0; i < n; ++i)
    if (res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize <= ws_bytes) {
      p.algo = res[i].algo;
      p.ws = res[i].workspaceSize;
      p.ok = true;
      return true;
    }
  return false;
}

}  // namespace
}  // namespace bevops

using namespace bevops;

extern "C" size_t bevops_linear_workspace_size(void) { return (size_t)32 << 20; }

extern "C" int bevops_linear_bias_act(int dtype, const void *a, const void *weight, const void *bias,
                                      const void *residual, void *out, long long M, int N, int K, int relu,
                                      void *workspace, size_t workspace_bytes, void *stream) {
  if (!a || !weight || !out || M < 0 || N <= 0 || K <= 0) return BEVOPS_BAD_PARAM;
  if (dtype != BEVOPS_F16) return BEVOPS_NOT_SUPPORTED;
  if (M == 0) return BEVOPS_SUCCESS;
  if (!aligned16(a) || !aligned16(weight) || !aligned16(out) || (residual && !aligned16(residual)) ||
      (workspace_bytes && !aligned16(workspace)))
    return BEVOPS_BAD_PARAM;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return BEVOPS_FAILURE;
  hipblasLtHandle_t h = nullptr;
  Plan plan;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto hit = g_handles.find(dev);
    if (hit == g_handles.end()) {
      if (hipblasLtCreate(&h) != HIPBLAS_STATUS_SUCCESS) return BEVOPS_NOT_INITIALIZED;
      g_handles[dev] = h;
    } else {
      h = hit->second;
    }
    const Key key{dev, M, N, K, relu != 0, bias != nullptr, residual != nullptr, workspace_bytes};
    auto pit = g_plans.find(key);
    if (pit == g_plans.end()) {
      Plan p;
      make_plan(h, p, M, N, K, relu != 0, bias != nullptr, workspace ? workspace_bytes : 0);
      pit = g_plans.emplace(key, p).first;  // failures are cached too: the caller falls back once, not per call
    }
    plan = pit->second;
  }
  if (!plan.ok) return BEVOPS_NOT_SUPPORTED;
  // the bias pointer lives in the matmul descriptor: a fresh one per call keeps the entry re-entrant
  hipblasLtMatmulDesc_t desc = make_desc(relu != 0, bias, bias != nullptr);
  if (!desc) return BEVOPS_FAILURE;
  const float alpha = 1.f, beta = residual ? 1.f : 0.f;
  const void *c = residual ? residual : out;
  const hipblasStatus_t st =
      hipblasLtMatmul(h, desc, &alpha, weight, plan.a, a, plan.b, &beta, c, plan.c, out, plan.c, &plan.algo,
                      workspace, plan.ws, static_cast<hipStream_t>(stream));
  hipblasLtMatmulDescDestroy(desc);
  return st == HIPBLAS_STATUS_SUCCESS ? BEVOPS_SUCCESS : BEVOPS_FAILURE;
}
