// dropin.hip -- the host-pointer entry point: pbicgstab.cu:157-409 / :756-922 / :926-1088 in one call (cudamat_solve),
// and the plan cache that lets a second call with the same matrix skip everything but upload + loop.
#include <chrono>
#include <mutex>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "solver.h"

using namespace cm;

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------
// Drop-in host-pointer solve: pbicgstab.cu:157-409 / :756-922 / :926-1088 in one call.
// ---------------------------------------------------------------------------------------
// The reference allocates, analyses, solves and frees per call (pbicgstab.cu:157-409).  Here the solver of the last
// call stays alive: when the next call brings the same matrix (same n, nnz, base and -- compared ON THE DEVICE after
// the upload, 12 bytes per entry read twice: ~2.5 ms at C4 -- the same row pointers, column indices and values), its
// device copies, SpMV plan, value dictionary and ILU(0) factors are reused and the call costs upload + loop.
namespace {
struct PlanCache {
    std::mutex mu;
    cudamat_ctx *ctx = nullptr;
    cudamat_solver *s = nullptr;
    int n = 0, nnz = 0, base = 0;
    bool has_shift = false;
    double *d_d = nullptr;          // the (A0 + I d) diagonal the cached solver points at
};
PlanCache g_cache;

void cache_drop_locked()
{
    if (g_cache.s) cudamat_solver_destroy(g_cache.s);
    if (g_cache.d_d) cudamat_free(g_cache.ctx, g_cache.d_d);
    if (g_cache.ctx) cudamat_ctx_destroy(g_cache.ctx);
    g_cache.s = nullptr;
    g_cache.d_d = nullptr;
    g_cache.ctx = nullptr;
}
}  // namespace

// flag[0] = 1 when a[i] != b[i] for some i (raw 32-bit words)
__global__ __launch_bounds__(kBlock) void k_differs(long long words, const unsigned *a, const unsigned *b, int *flag)
{
    bool diff = false;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < words; i += (long long)gridDim.x * kBlock) diff |= a[i] != b[i];
    if (diff) *flag = 1;
}

static int device_equal(hipStream_t st, const void *a, const void *b, size_t bytes, int *flag_dev)
{
    const long long words = (long long)(bytes / 4);
    if (words == 0) return CUDAMAT_OK;
    long long g = (words + kBlock * 8LL - 1) / (kBlock * 8LL);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_differs, dim3((unsigned)g), dim3(kBlock), 0, st, words, (const unsigned *)a, (const unsigned *)b, flag_dev);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

extern "C" int cudamat_plan_cache_clear(void)
{
    std::lock_guard<std::mutex> lk(g_cache.mu);
    cache_drop_locked();
    return CUDAMAT_OK;
}

// one attempt (the caller holds g_cache.mu)
static int solve_host_locked(const Config &cfg, int n, int nnz, const double *A, const int *iA, const int *jA, const double *d,
                             const double *x0, const double *b, int precond, int loop, int maxit, double tol, int debug,
                             double *x, cudamat_stats *out)
{
    const int base = iA[0];                                        // pbicgstab.cu:201,782,953
    const double t0 = now_s();
    const bool use_cache = cfg.plan_cache != 0;
    if (!use_cache) cache_drop_locked();
    // same shape as the cached system, same switches?  then its context (device, stream, options) carries this call too
    const bool candidate = use_cache && g_cache.s && g_cache.n == n && g_cache.nnz == nnz && g_cache.base == base &&
                           g_cache.ctx->cfg == cfg;
    if (!candidate) cache_drop_locked();
    cudamat_ctx *ctx = candidate ? g_cache.ctx : nullptr;
    if (!ctx) CM_TRY(cudamat_ctx_create(0, nullptr, &ctx));
    int *d_rp = nullptr, *d_ci = nullptr;
    double *d_val = nullptr, *d_b = nullptr, *d_x = nullptr, *d_d = nullptr;
    cudamat_solver *s = nullptr;
    bool reused = false, built_ilu = false;
    int rc = CUDAMAT_OK;
    cudamat_stats st;
    memset(&st, 0, sizeof(st));
    double t_up = 0.0;
    do {
        if ((rc = cudamat_malloc(ctx, sizeof(int) * ((size_t)n + 1), (void **)&d_rp))) break;
        if ((rc = cudamat_malloc(ctx, sizeof(int) * (size_t)nnz, (void **)&d_ci))) break;
        if ((rc = cudamat_malloc(ctx, sizeof(double) * (size_t)nnz, (void **)&d_val))) break;
        if ((rc = cudamat_malloc(ctx, sizeof(double) * (size_t)n, (void **)&d_b))) break;
        if ((rc = cudamat_malloc(ctx, sizeof(double) * (size_t)n, (void **)&d_x))) break;
        if ((rc = cudamat_h2d(ctx, d_rp, iA, sizeof(int) * ((size_t)n + 1)))) break;     // :313-315
        if ((rc = cudamat_h2d(ctx, d_ci, jA, sizeof(int) * (size_t)nnz))) break;
        if ((rc = cudamat_h2d(ctx, d_val, A, sizeof(double) * (size_t)nnz))) break;
        if ((rc = cudamat_h2d(ctx, d_b, b, sizeof(double) * (size_t)n))) break;
        if (x0 && (rc = cudamat_h2d(ctx, d_x, x0, sizeof(double) * (size_t)n))) break;
        if (d) {
            if ((rc = cudamat_malloc(ctx, sizeof(double) * (size_t)n, (void **)&d_d))) break;
            if ((rc = cudamat_h2d(ctx, d_d, d, sizeof(double) * (size_t)n))) break;
        }
        if ((rc = cudamat_ctx_sync(ctx))) break;
        t_up = now_s() - t0;
        if (candidate) {
            // the cached solver holds the matrix rebased to 0: compare the uploaded arrays with it on the device
            int *flag = nullptr, h = 1;
            if ((rc = cudamat_malloc(ctx, sizeof(int), (void **)&flag))) break;
            hipMemsetAsync(flag, 0, sizeof(int), ctx->stream);
            cudamat_solver *c = g_cache.s;
            int *tmp = nullptr;          // rebased copies of the uploaded index arrays
            rc = cudamat_malloc(ctx, sizeof(int) * ((size_t)nnz > (size_t)n + 1 ? (size_t)nnz : (size_t)n + 1), (void **)&tmp);
            if (!rc) rc = launch_rebase(ctx->stream, (int64_t)n + 1, d_rp, -base, tmp);
            if (!rc) rc = device_equal(ctx->stream, tmp, c->rp, sizeof(int) * ((size_t)n + 1), flag);
            if (!rc && nnz) rc = launch_rebase(ctx->stream, nnz, d_ci, -base, tmp);
            if (!rc && nnz) rc = device_equal(ctx->stream, tmp, c->ci, sizeof(int) * (size_t)nnz, flag);
            if (!rc && nnz) rc = device_equal(ctx->stream, d_val, c->val, sizeof(double) * (size_t)nnz, flag);
            if (!rc && hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = CUDAMAT_ERR_HIP;
            if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = CUDAMAT_ERR_HIP;
            if (tmp) cudamat_free(ctx, tmp);
            cudamat_free(ctx, flag);
            if (rc) break;
            if (h == 0) {
                s = g_cache.s;
                reused = true;
            } else {                     // same shape, another matrix: the old solver goes, its context stays
                cudamat_solver_destroy(g_cache.s);
                g_cache.s = nullptr;
            }
        }
        if (g_cache.d_d) { cudamat_free(ctx, g_cache.d_d); g_cache.d_d = nullptr; }
        if (!s && (rc = cudamat_solver_create(ctx, n, n, nnz, d_rp, d_ci, d_val, base, &s))) break;
        if ((rc = cudamat_solver_set_shift(s, d_d))) break;
        if (precond != CUDAMAT_PRECOND_NONE && !(reused && s->has_ilu && !s->ilu_block)) {
            if ((rc = cudamat_solver_ilu0(s))) break;
            built_ilu = true;
        }
        if (precond != CUDAMAT_PRECOND_NONE && debug) {
            printf("analysis lower %f (s), upper %f (s) \n", s->t_analysis_l, s->t_analysis_u);     // :349
            printf("csrilu0 (HIP, level-scheduled) time(s) = %10.8f \n", s->t_factor);            // :355,363
        }
        int flags = (debug ? CUDAMAT_FLAG_DEBUG : 0) | (x0 ? 0 : CUDAMAT_FLAG_X0_ONES);
        if ((rc = cudamat_solver_solve(s, d_b, d_x, precond, loop, maxit, tol, flags, &st))) break;
        if ((rc = cudamat_d2h(ctx, x, d_x, sizeof(double) * (size_t)n))) break;            // :381
    } while (0);
    char saved[512];
    strncpy(saved, cudamat_last_error(), sizeof(saved) - 1);
    saved[sizeof(saved) - 1] = 0;
    st.t_upload = t_up;
    st.plan_reused = reused ? 1 : 0;
    if (reused) { st.t_setup = 0.0; st.t_tune = 0.0; }                       // (they describe the call that built the plan)
    if (reused && !built_ilu) { st.t_analysis = 0.0; st.t_factor = 0.0; }
    // keep the solver for the next call (it owns its own copies of the matrix; the upload buffers go)
    cudamat_solver *const old = g_cache.s;       // the previous call's solver, when it is still alive (may be s itself)
    if (s && rc == CUDAMAT_OK && use_cache) {
        if (old && old != s) cudamat_solver_destroy(old);
        g_cache.ctx = ctx;
        g_cache.s = s;
        g_cache.n = n; g_cache.nnz = nnz; g_cache.base = base;
        g_cache.d_d = d_d;               // the solver points at it (set_shift); replaced by the next call
        d_d = nullptr;
    } else {
        if (s) cudamat_solver_destroy(s);
        if (old && old != s) cudamat_solver_destroy(old);
        g_cache.s = nullptr;
    }
    void *ptrs[] = {d_rp, d_ci, d_val, d_b, d_x, d_d};
    for (void *p : ptrs)
        if (p) cudamat_free(ctx, p);
    if (!g_cache.s) {                    // nothing kept: the context goes too
        if (g_cache.d_d) { cudamat_free(ctx, g_cache.d_d); g_cache.d_d = nullptr; }
        cudamat_ctx_destroy(ctx);
        g_cache.ctx = nullptr;
    }
    if (rc) set_error("%s", saved);
    st.t_total = now_s() - t0;
    if (out) *out = st;
    return rc;
}

extern "C" int cudamat_solve(int n, int nnz, const double *A, const int *iA, const int *jA,
                             const double *d, const double *x0, const double *b, int precond,
                             int loop, int maxit, double tol, int debug, double *x,
                             cudamat_stats *out)
{
    CM_ARG(n > 0 && nnz >= 0 && A && iA && jA && b && x, "null pointer or empty system");
    const int base = iA[0];                                        // pbicgstab.cu:201,782,953
    CM_ARG(base == 0 || base == 1, "iA[0] must be 0 or 1");
    CM_ARG(iA[n] - base == nnz, "nnz != iA[n] - iA[0]");
    if (debug && loop == CUDAMAT_LOOP_PBICGSTAB) printf("N=%d, nnz=%d\n", n, nnz);   // :204
    const Config cfg = config_from_env();                          // no caller-made context: the switches of THIS call
    std::lock_guard<std::mutex> cache_lock(g_cache.mu);            // (the entry points are not re-entrant upstream either)
    int rc = solve_host_locked(cfg, n, nnz, A, iA, jA, d, x0, b, precond, loop, maxit, tol, debug, x, out);
    if (rc == CUDAMAT_ERR_NOMEM && (g_cache.s || g_cache.ctx)) {
        // the solver kept from the previous call (several GB at the BASELINE sizes) may be what is in the way: the
        // reference frees everything per call (pbicgstab.cu:392-405), so release it and try once more
        cache_drop_locked();
        rc = solve_host_locked(cfg, n, nnz, A, iA, jA, d, x0, b, precond, loop, maxit, tol, debug, x, out);
    }
    return rc;
}

