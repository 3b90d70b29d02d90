// vendor_gemm.hip -- the vendor library as a yardstick INSIDE the step (vb_stream_opts.nt_kernel = 200).
// DEVELOPER LIBRARY ONLY (libvisualbert_hip_dev.so, include/visualbert_hip_dev.h): the product library neither contains this
// file nor accepts nt_kernel = 200 (tests/test_abi.py::test_developer_knobs_are_not_in_the_product_library).
//
// Not the product path: every GEMM of the step runs on the hand-written kernels of gemm.hip.  hipBLASLt's
// hand-scheduled 256x256 stream-K kernel is 8-25 % faster than ours on the step's plain GEMM shapes when timed alone
// (profiles/r03_gemm_vendor_yardstick.txt); this file lets the SAME training step run with those GEMMs -- bias-only, or
// "+ addend" expressed as beta = 1 -- handed to the library, so the question "what would the vendor's schedule be worth
// inside the step, at the clock the step runs at" has a measured answer.  Everything fused
// (GELU + derivative, x GELU' + column sums, split operands, fp32-accumulating weight gradients) stays ours either way.
//
// libhipblaslt.so is opened with dlopen on first use: no link-time dependency, and a box without it gets
// VB_ERR_UNSUPPORTED from this entry (the dispatcher then takes the normal kernel).  This path owns device memory -- one
// 64 MB stream-K workspace per (device, stream), allocated on first use, never freed (a measurement process) -- which is one
// more reason it is not in the product, whose header promises that the library owns no device memory.
#include "vb_rt.h"
#include "../../include/visualbert_hip.h"
#include <hipblaslt/hipblaslt.h>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <tuple>

namespace {

struct Api {
    decltype(&hipblasLtCreate) create;
    decltype(&hipblasLtMatmulDescCreate) desc_create;
    decltype(&hipblasLtMatmulDescSetAttribute) desc_set;
    decltype(&hipblasLtMatrixLayoutCreate) layout_create;
    decltype(&hipblasLtMatmulPreferenceCreate) pref_create;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) pref_set;
    decltype(&hipblasLtMatmulPreferenceDestroy) pref_destroy;
    decltype(&hipblasLtMatmulAlgoGetHeuristic) heuristic;
    decltype(&hipblasLtMatmul) matmul;
    hipblasLtHandle_t handle;
    bool ok;
};

Api* api() {
    static Api a = [] {
        Api x{};
        void* h = dlopen("libhipblaslt.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/libhipblaslt.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return x;
#define VB_SYM(field, name) x.field = (decltype(x.field))dlsym(h, #name); if (!x.field) return x
        VB_SYM(create, hipblasLtCreate);
        VB_SYM(desc_create, hipblasLtMatmulDescCreate);
        VB_SYM(desc_set, hipblasLtMatmulDescSetAttribute);
        VB_SYM(layout_create, hipblasLtMatrixLayoutCreate);
        VB_SYM(pref_create, hipblasLtMatmulPreferenceCreate);
        VB_SYM(pref_set, hipblasLtMatmulPreferenceSetAttribute);
        VB_SYM(pref_destroy, hipblasLtMatmulPreferenceDestroy);
        VB_SYM(heuristic, hipblasLtMatmulAlgoGetHeuristic);
        VB_SYM(matmul, hipblasLtMatmul);
#undef VB_SYM
        x.ok = x.create(&x.handle) == HIPBLAS_STATUS_SUCCESS;
        return x;
    }();
    return a.ok ? &a : nullptr;
}

constexpr size_t kWorkspace = 64u << 20;

struct Plan {
    hipblasLtMatmulDesc_t desc;
    hipblasLtMatrixLayout_t la, lb, lc, ld;
    hipblasLtMatmulAlgo_t algo;
    bool ok;
};
// (M, N, K, lda, ldb, ldc, ld_addend, out_f32, has_bias, has_addend)
typedef std::tuple<int, int, int, long, long, long, long, int, int, int> Key;

std::mutex g_mutex;
std::map<Key, Plan> g_plans;
std::map<std::pair<int, void*>, void*> g_workspaces;           // (device, stream) -> workspace

// Row-major C[M,N] = A[M,K] . B[N,K]^T is, in the library's column-major terms, D[N x M] = op_T(B as K x N) . (A as K x M):
// the "Alik_Bljk" form torch's F.linear issues, so the heuristic lands on the same kernels the yardstick measured.
Plan make_plan(Api* p, const VbVendorGemm& g) {
    Plan pl{};
    const hipDataType tin = HIP_R_16BF, tout = g.out_f32 ? HIP_R_32F : HIP_R_16BF;
    if (p->desc_create(&pl.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return pl;
    const hipblasOperation_t tr = HIPBLAS_OP_T, nt = HIPBLAS_OP_N;
    p->desc_set(pl.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &tr, sizeof(tr));
    p->desc_set(pl.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &nt, sizeof(nt));
    if (g.bias) {
        const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
        const hipDataType bt = HIP_R_32F;
        if (p->desc_set(pl.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) != HIPBLAS_STATUS_SUCCESS) return pl;
        if (p->desc_set(pl.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)) != HIPBLAS_STATUS_SUCCESS) return pl;
    }
    if (p->layout_create(&pl.la, tin, (uint64_t)g.K, (uint64_t)g.N, g.ldb) != HIPBLAS_STATUS_SUCCESS) return pl;
    if (p->layout_create(&pl.lb, tin, (uint64_t)g.K, (uint64_t)g.M, g.lda) != HIPBLAS_STATUS_SUCCESS) return pl;
    if (p->layout_create(&pl.lc, tout, (uint64_t)g.N, (uint64_t)g.M, g.addend ? g.ld_addend : g.ldc) != HIPBLAS_STATUS_SUCCESS) return pl;
    if (p->layout_create(&pl.ld, tout, (uint64_t)g.N, (uint64_t)g.M, g.ldc) != HIPBLAS_STATUS_SUCCESS) return pl;
    hipblasLtMatmulPreference_t pref;
    if (p->pref_create(&pref) != HIPBLAS_STATUS_SUCCESS) return pl;
    const uint64_t ws = kWorkspace;
    p->pref_set(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws));
    if (g.bias) {                                            // the pointer takes part in the heuristic's validity checks
        const void* b = g.bias;
        p->desc_set(pl.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &b, sizeof(b));
    }
    hipblasLtMatmulHeuristicResult_t res[1];
    int found = 0;
    const hipblasStatus_t st = p->heuristic(p->handle, pl.desc, pl.la, pl.lb, pl.lc, pl.ld, pref, 1, res, &found);
    p->pref_destroy(pref);
    if (st != HIPBLAS_STATUS_SUCCESS || found < 1 || res[0].workspaceSize > kWorkspace) return pl;
    pl.algo = res[0].algo;
    pl.ok = true;
    return pl;
}

}  // namespace

int vb_vendor_nt(const VbVendorGemm& g, void* stream) {
    Api* p = api();
    if (!p) return VB_ERR_UNSUPPORTED;
    Plan pl;
    void* ws = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        const Key key(g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.addend ? g.ld_addend : 0L, g.out_f32, g.bias ? 1 : 0, g.addend ? 1 : 0);
        auto it = g_plans.find(key);
        if (it == g_plans.end()) it = g_plans.emplace(key, make_plan(p, g)).first;
        pl = it->second;
        if (!pl.ok) return VB_ERR_UNSUPPORTED;
        int device = 0;
        if (hipGetDevice(&device) != hipSuccess) return VB_ERR_UNSUPPORTED;
        const std::pair<int, void*> wkey(device, stream);
        auto w = g_workspaces.find(wkey);
        if (w == g_workspaces.end()) {
            void* buf = nullptr;
            if (hipMalloc(&buf, kWorkspace) != hipSuccess) return VB_ERR_UNSUPPORTED;
            w = g_workspaces.emplace(wkey, buf).first;
        }
        ws = w->second;
        if (g.bias) {                                        // the descriptor is shared by every call of this shape: set under the lock,
            const void* b = g.bias;                          // and the launch below stays inside it for the same reason
            p->desc_set(pl.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &b, sizeof(b));
        }
        const float alpha = g.alpha, beta = g.addend ? 1.0f : 0.0f;
        const hipblasStatus_t st = p->matmul(p->handle, pl.desc, &alpha, g.B, pl.la, g.A, pl.lb, &beta, g.addend ? g.addend : g.C, pl.lc,
                                             g.C, pl.ld, &pl.algo, ws, kWorkspace, (hipStream_t)stream);
        return st == HIPBLAS_STATUS_SUCCESS ? VB_OK : VB_ERR_LAUNCH;
    }
}
