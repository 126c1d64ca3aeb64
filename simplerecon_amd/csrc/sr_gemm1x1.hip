// sr_gemm1x1.hip -- 1x1 convolutions over dense channels-last maps as plain library GEMMs (hipBLASLt, fp32).
//
// BASELINE.json's north star puts the 2-D convs "on rocBLAS/MFMA only where it is a dense im2col GEMM": a 1x1 conv over a
// dense [pixels][Cin] map IS that GEMM, Y[pixels][Cout] = X[pixels][Cin] . W^T (+ bias, + residual, SiLU / ReLU) -- no
// halo, no im2col.  The image-prior encoder's MBConv expand / project convs (reference depth_model.py:110-116: timm
// tf_efficientnetv2_s) are short-K or short-N GEMMs with M = 2400 ... 38400 pixels where the hand-written implicit-GEMM
// kernel of sr_conv.hip reaches 38-53 TFLOP/s and hipBLASLt's fp32-MFMA kernels 60-85 (scripts/gemm_probe.py).
// Column-major view handed to hipBLASLt:  D (Cout x M, ld = out pixel stride) = op_T(A = W as Cin x Cout, ld = Cin) .
// B (= X as Cin x M, ld = in pixel stride) [+ C = residual, beta = 1]; bias per row of D, SiLU = the SWISH_EXT epilogue.
// fp32 in / out / accumulate (HIPBLAS_COMPUTE_32F: gfx950 has no xf32 path) -- same arithmetic class as sr_conv.hip.
// The handles (one per DEVICE) and the per-(device, shape) algorithm choices are process-wide caches (created on first use,
// under a mutex): the one place where this library keeps state, never freed.  A shape hipBLASLt cannot serve is cached as
// such (its descriptors released), so later calls fall back to the HIP kernel without repeating the heuristic query.  The choice per shape is made by timing hipBLASLt's heuristic
// candidates once (first call with that shape; it synchronises the stream then and only then).
#include <hipblaslt/hipblaslt.h>

#include <stdlib.h>

#include <map>
#include <mutex>
#include <tuple>

#include "sr_common.h"

namespace {

struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr, ld = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t workspace = 0;
  bool ok = false;
};

// M, Cin, Cout, in_sp, out_sp, res_sp, act, has_bias, device
typedef std::tuple<int, int, int, int, int, int, int, int, int> Key;

std::mutex g_mutex;
std::map<int, hipblasLtHandle_t> g_handles;   // per device
std::map<Key, Plan> g_plans;

void release(Plan& p) {
  if (p.desc) hipblasLtMatmulDescDestroy(p.desc);
  if (p.la) hipblasLtMatrixLayoutDestroy(p.la);
  if (p.lb) hipblasLtMatrixLayoutDestroy(p.lb);
  if (p.lc) hipblasLtMatrixLayoutDestroy(p.lc);
  if (p.ld) hipblasLtMatrixLayoutDestroy(p.ld);
  p.desc = nullptr; p.la = p.lb = p.lc = p.ld = nullptr;
}

constexpr size_t kWorkspace = 32u << 20;

// Arguments of the call that creates a plan: the candidates hipBLASLt's heuristic returns are timed on them once.
struct TuneArgs {
  const float* in; const float* weight; const float* bias; const float* residual; float* out; void* workspace;
  hipStream_t stream;
};

constexpr int kCandidates = 12;

bool make_plan_impl(hipblasLtHandle_t g_handle, const Key& key, Plan& plan, const TuneArgs& ta);

bool make_plan(hipblasLtHandle_t handle, const Key& key, Plan& plan, const TuneArgs& ta) {
  plan.ok = make_plan_impl(handle, key, plan, ta);
  if (!plan.ok) release(plan);   // nothing leaks; the negative result stays cached
  return plan.ok;
}

bool make_plan_impl(hipblasLtHandle_t g_handle, const Key& key, Plan& plan, const TuneArgs& ta) {
  const int M = std::get<0>(key), Cin = std::get<1>(key), Cout = std::get<2>(key), in_sp = std::get<3>(key);
  const int out_sp = std::get<4>(key), res_sp = std::get<5>(key), act = std::get<6>(key), has_bias = std::get<7>(key);
  if (hipblasLtMatmulDescCreate(&plan.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return false;
  const int32_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
  hipblasLtMatmulDescSetAttribute(plan.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT));
  hipblasLtMatmulDescSetAttribute(plan.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN));
  uint32_t epi = HIPBLASLT_EPILOGUE_DEFAULT;
  if (act == 1) epi = has_bias ? HIPBLASLT_EPILOGUE_SWISH_BIAS_EXT : HIPBLASLT_EPILOGUE_SWISH_EXT;
  else if (act == 2) epi = has_bias ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_RELU;
  else if (has_bias) epi = HIPBLASLT_EPILOGUE_BIAS;
  hipblasLtMatmulDescSetAttribute(plan.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi));
  if (act == 1) {
    const float one = 1.0f;   // Swish(x, 1) = x * sigmoid(x) = SiLU
    hipblasLtMatmulDescSetAttribute(plan.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE_ACT_ARG0_EXT, &one, sizeof(one));
  }
  if (hipblasLtMatrixLayoutCreate(&plan.la, HIP_R_32F, Cin, Cout, Cin) != HIPBLAS_STATUS_SUCCESS) return false;
  if (hipblasLtMatrixLayoutCreate(&plan.lb, HIP_R_32F, Cin, M, in_sp) != HIPBLAS_STATUS_SUCCESS) return false;
  if (hipblasLtMatrixLayoutCreate(&plan.lc, HIP_R_32F, Cout, M, res_sp > 0 ? res_sp : out_sp) != HIPBLAS_STATUS_SUCCESS) return false;
  if (hipblasLtMatrixLayoutCreate(&plan.ld, HIP_R_32F, Cout, M, out_sp) != HIPBLAS_STATUS_SUCCESS) return false;
  hipblasLtMatmulPreference_t pref;
  if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return false;
  const uint64_t ws = kWorkspace;
  hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws));
  hipblasLtMatmulHeuristicResult_t res[kCandidates];
  int found = 0;
  const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(g_handle, plan.desc, plan.la, plan.lb, plan.lc, plan.ld, pref,
                                                             kCandidates, res, &found);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS || found < 1) return false;
  int best = 0;
  // The heuristic's first choice is tuned for large square-ish problems; the encoder's GEMMs are short-K / short-N with
  // M = 2400 ... 38400.  Time the candidates once, on the caller's own buffers and stream (the output is simply written
  // several times; C is the residual, never the output).  Skipped inside a stream capture (no synchronisation allowed
  // there) and with SR_GEMM_AUTOTUNE=0.
  const int autotune = sr_opt(SR_OPT_GEMM_AUTOTUNE);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(ta.stream, &cap);
  if (found > 1 && autotune != 0 && cap == hipStreamCaptureStatusNone) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
      if (ta.bias) hipblasLtMatmulDescSetAttribute(plan.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &ta.bias, sizeof(ta.bias));
      const float alpha = 1.0f, beta = ta.residual ? 1.0f : 0.0f;
      float best_ms = -1.0f;
      for (int c = 0; c < found; ++c) {
        if (res[c].workspaceSize > kWorkspace) continue;
        bool ok = true;
        for (int rep = 0; rep < 3 && ok; ++rep) {   // one warm-up, two timed
          if (rep == 1) (void)hipEventRecord(e0, ta.stream);
          ok = hipblasLtMatmul(g_handle, plan.desc, &alpha, ta.weight, plan.la, ta.in, plan.lb, &beta,
                               ta.residual ? ta.residual : ta.out, plan.lc, ta.out, plan.ld, &res[c].algo, ta.workspace,
                               kWorkspace, ta.stream) == HIPBLAS_STATUS_SUCCESS;
        }
        if (!ok) continue;
        (void)hipEventRecord(e1, ta.stream);
        if (hipEventSynchronize(e1) != hipSuccess) continue;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
        if (best_ms < 0.0f || ms < best_ms) { best_ms = ms; best = c; }
      }
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
    }
  }
  plan.algo = res[best].algo;
  plan.workspace = res[best].workspaceSize;
  return true;
}

}  // namespace

extern "C" size_t sr_gemm1x1_workspace_bytes(void) { return kWorkspace; }

// act: 0 none, 1 SiLU, 2 ReLU.  Returns SR_ERR_UNSUPPORTED when hipBLASLt offers no algorithm (the caller falls back to
// the HIP kernel).
extern "C" int sr_gemm1x1_nhwc_fwd(const float* in, int in_pix_stride, const float* weight, const float* bias,
                                   const float* residual, int res_pix_stride, float* out, int out_pix_stride, int M, int Cin,
                                   int Cout, int act, void* workspace, size_t workspace_bytes, void* stream) {
  if (M < 0 || Cin <= 0 || Cout <= 0 || act < 0 || act > 2) return SR_ERR_INVALID_ARGUMENT;
  if (M == 0) return SR_OK;
  if (!in || !weight || !out || in_pix_stride < Cin || out_pix_stride < Cout || (residual && res_pix_stride < Cout))
    return SR_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < kWorkspace || !workspace) return SR_ERR_WORKSPACE_TOO_SMALL;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return SR_ERR_UNSUPPORTED;
  const Key key(M, Cin, Cout, in_pix_stride, out_pix_stride, residual ? res_pix_stride : 0, act, bias ? 1 : 0, device);
  // one lock for the lookup AND the launch: the bias pointer is an attribute of the (shared) descriptor
  std::lock_guard<std::mutex> lock(g_mutex);
  hipblasLtHandle_t& handle = g_handles[device];
  if (!handle && hipblasLtCreate(&handle) != HIPBLAS_STATUS_SUCCESS) { handle = nullptr; return SR_ERR_UNSUPPORTED; }
  auto it = g_plans.find(key);
  if (it == g_plans.end()) {
    Plan p;
    const TuneArgs ta = {in, weight, bias, residual, out, workspace, (hipStream_t)stream};
    make_plan(handle, key, p, ta);
    it = g_plans.emplace(key, p).first;
  }
  const Plan& plan = it->second;
  if (!plan.ok) return SR_ERR_UNSUPPORTED;
  if (bias) hipblasLtMatmulDescSetAttribute(plan.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
  const float alpha = 1.0f, beta = residual ? 1.0f : 0.0f;
  const hipblasStatus_t st = hipblasLtMatmul(handle, plan.desc, &alpha, weight, plan.la, in, plan.lb, &beta,
                                             residual ? residual : out, plan.lc, out, plan.ld, &plan.algo, workspace,
                                             kWorkspace, (hipStream_t)stream);
  return st == HIPBLAS_STATUS_SUCCESS ? SR_OK : SR_ERR_UNSUPPORTED;
}
