// Plain library GEMM (hipBLASLt) for the long-K, narrow-N training shapes.
//
// C[M,N] = A[M,K] W[N,K]^T (+ bias[N]), bf16 operands, fp32 accumulation — exactly od_gemm_nt without epilogue or accumulate.  At
// M = 262,144 the library's hand-scheduled 256x256x64 assembly kernel runs the four N = 512 shapes of a layer (out_proj, proj_o and the
// backward-data products of qkv_proj and proj_vg: K = 1024 ... 3072) at 1.17 - 1.36 PF/s where gemm_nt_big_kernel (same macro tile, same
// fetch volume) reaches 0.90 - 1.08 (profiles/r03_gemm_vs_vendor.txt); everywhere else — K = 512 shapes, every weight gradient, every
// fused epilogue, sampler sizes — the kernels of gemm.hip are level or ahead and stay in use.  The library is bound at run time (dlopen of
// the path the host names), owned by an explicit handle object (od_vendor_gemm_create), given a caller-owned workspace, and optional: every
// failure is a return code and the caller falls back to od_gemm_nt.
#include "od_common.h"
#include "od_api_internal.h"

#if defined(OD_EMU)
extern "C" int od_vendor_gemm_create(void**, const char*) { return OD_ERR_UNSUPPORTED; }
extern "C" int od_vendor_gemm_destroy(void*) { return OD_ERR_UNSUPPORTED; }
extern "C" int od_gemm_nt_vendor(void*, int, const void*, int, const void*, int, const float*, void*, int, int, int, int, void*, long, void*) {
    return OD_ERR_UNSUPPORTED;
}
#else
#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>
#include <map>
#include <tuple>

namespace {

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws = 0;
    bool ok = false;
};

struct VendorGemm {
    void* dl = nullptr;
    hipblasLtHandle_t handle = nullptr;
    decltype(&hipblasLtCreate) Create = nullptr;
    decltype(&hipblasLtDestroy) Destroy = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
    decltype(&hipblasLtMatrixLayoutDestroy) LayoutDestroy = nullptr;
    decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
    decltype(&hipblasLtMatmulDescDestroy) DescDestroy = nullptr;
    decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
    decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
    decltype(&hipblasLtMatmulPreferenceDestroy) PrefDestroy = nullptr;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
    decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
    decltype(&hipblasLtMatmul) Matmul = nullptr;
    std::map<std::tuple<int, int, int, int, int, int, int, long>, Plan> plans;     // (M, N, K, lda, ldw, ldc, bias?, workspace bytes)
};

template <class F> bool bind(void* dl, F& f, const char* name) {
    f = (F)dlsym(dl, name);
    return f != nullptr;
}

void free_plan(VendorGemm* g, Plan& p) {
    if (p.la) g->LayoutDestroy(p.la);
    if (p.lb) g->LayoutDestroy(p.lb);
    if (p.lc) g->LayoutDestroy(p.lc);
    if (p.desc) g->DescDestroy(p.desc);
    p = Plan();
}

// Row-major C[M,N] = A[M,K] W[N,K]^T is, in the library's column-major terms, D[N x M] = op(W)[N x K] * B[K x M] with W read as a K x N
// matrix (leading dimension ldw) transposed, the activations as K x M (leading dimension lda): nothing is copied or re-laid out.
Plan make_plan(VendorGemm* g, int M, int N, int K, int lda, int ldw, int ldc, bool bias, long ws_bytes) {
    Plan p;
    const hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
    if (g->DescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return p;
    bool ok = g->DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT)) == HIPBLAS_STATUS_SUCCESS &&
              g->DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)) == HIPBLAS_STATUS_SUCCESS;
    if (ok && bias) {
        const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
        const hipDataType bt = HIP_R_32F;
        ok = g->DescSet(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) == HIPBLAS_STATUS_SUCCESS &&
             g->DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)) == HIPBLAS_STATUS_SUCCESS;
    }
    ok = ok && g->LayoutCreate(&p.la, HIP_R_16BF, K, N, ldw) == HIPBLAS_STATUS_SUCCESS &&
         g->LayoutCreate(&p.lb, HIP_R_16BF, K, M, lda) == HIPBLAS_STATUS_SUCCESS &&
         g->LayoutCreate(&p.lc, HIP_R_16BF, N, M, ldc) == HIPBLAS_STATUS_SUCCESS;
    hipblasLtMatmulPreference_t pref = nullptr;
    if (ok && g->PrefCreate(&pref) == HIPBLAS_STATUS_SUCCESS) {
        const uint64_t wsz = (uint64_t)ws_bytes;
        hipblasLtMatmulHeuristicResult_t res;
        int n = 0;
        if (g->PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)) == HIPBLAS_STATUS_SUCCESS &&
            g->Heuristic(g->handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, 1, &res, &n) == HIPBLAS_STATUS_SUCCESS && n > 0 &&
            res.workspaceSize <= (size_t)ws_bytes) {
            p.algo = res.algo;
            p.ws = res.workspaceSize;
            p.ok = true;
        }
        g->PrefDestroy(pref);
    }
    if (!p.ok) free_plan(g, p);
    return p;
}

}  // namespace

extern "C" int od_vendor_gemm_create(void** out, const char* libpath) {
    if (!out) return OD_ERR_ARG;
    VendorGemm* g = new VendorGemm();
    g->dl = dlopen((libpath && libpath[0]) ? libpath : "libhipblaslt.so", RTLD_NOW | RTLD_LOCAL);
    const bool ok = g->dl && bind(g->dl, g->Create, "hipblasLtCreate") && bind(g->dl, g->Destroy, "hipblasLtDestroy") &&
                    bind(g->dl, g->LayoutCreate, "hipblasLtMatrixLayoutCreate") && bind(g->dl, g->LayoutDestroy, "hipblasLtMatrixLayoutDestroy") &&
                    bind(g->dl, g->DescCreate, "hipblasLtMatmulDescCreate") && bind(g->dl, g->DescDestroy, "hipblasLtMatmulDescDestroy") &&
                    bind(g->dl, g->DescSet, "hipblasLtMatmulDescSetAttribute") && bind(g->dl, g->PrefCreate, "hipblasLtMatmulPreferenceCreate") &&
                    bind(g->dl, g->PrefDestroy, "hipblasLtMatmulPreferenceDestroy") && bind(g->dl, g->PrefSet, "hipblasLtMatmulPreferenceSetAttribute") &&
                    bind(g->dl, g->Heuristic, "hipblasLtMatmulAlgoGetHeuristic") && bind(g->dl, g->Matmul, "hipblasLtMatmul");
    if (!ok || g->Create(&g->handle) != HIPBLAS_STATUS_SUCCESS) {
        if (g->dl) dlclose(g->dl);
        delete g;
        return OD_ERR_UNSUPPORTED;
    }
    *out = g;
    return 0;
}

extern "C" int od_vendor_gemm_destroy(void* vg) {
    if (!vg) return OD_ERR_ARG;
    VendorGemm* g = (VendorGemm*)vg;
    for (auto& kv : g->plans) free_plan(g, kv.second);
    g->Destroy(g->handle);
    dlclose(g->dl);
    delete g;
    return 0;
}

extern "C" int od_gemm_nt_vendor(void* vg, int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                                 int M, int N, int K, void* workspace, long workspace_bytes, void* stream) {
    if (!vg || !A || !W || !C || M <= 0 || N <= 0 || K <= 0 || workspace_bytes < 0) return OD_ERR_ARG;
    if (dtype != OD_BF16) return OD_ERR_UNSUPPORTED;
    if (lda % 8 || ldw % 8 || ldc % 8) return OD_ERR_ALIGN;
    VendorGemm* g = (VendorGemm*)vg;
    const auto key = std::make_tuple(M, N, K, lda, ldw, ldc, bias ? 1 : 0, workspace_bytes);
    auto it = g->plans.find(key);
    if (it == g->plans.end()) it = g->plans.emplace(key, make_plan(g, M, N, K, lda, ldw, ldc, bias != nullptr, workspace_bytes)).first;
    Plan& p = it->second;
    if (!p.ok) return OD_ERR_UNSUPPORTED;
    if (bias && g->DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) != HIPBLAS_STATUS_SUCCESS) return OD_ERR_UNSUPPORTED;
    const float alpha = 1.f, beta = 0.f;
    const hipblasStatus_t st = g->Matmul(g->handle, p.desc, &alpha, W, p.la, A, p.lb, &beta, C, p.lc, C, p.lc, &p.algo, workspace,
                                         (size_t)workspace_bytes, (hipStream_t)stream);
    return st == HIPBLAS_STATUS_SUCCESS ? 0 : OD_ERR_UNSUPPORTED;
}
#endif
