// context.hip -- error reporting, context, device memory helpers, timers and the
// elementary-kernel entry points of the C ABI (include/cudamat.h).
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <new>
#include <set>
#include <utility>

#include "kernels.h"

namespace cm {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail_hip(hipError_t e, const char *what, const char *file, int line)
{
    set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    return CUDAMAT_ERR_HIP;
}

int set_max_lds(const void *fn)
{
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    CM_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, fn})) return CUDAMAT_OK;
    hipFuncAttributes fa;
    CM_HIP(hipFuncGetAttributes(&fa, fn));                 // static LDS counts against the CU's 160 KB too
    CM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)fa.sharedSizeBytes));
    done.insert({dev, fn});
    return CUDAMAT_OK;
}

// ---- optional roctx ranges
namespace {
std::atomic<bool> g_roctx_on{false};
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
        for (const char *n : names) {
            void *h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};
Roctx &roctx()
{
    static Roctx r;
    return r;
}
}  // namespace

void range_enable() { g_roctx_on = true; }
void range_push(const char *name)
{
    if (!g_roctx_on) return;
    Roctx &r = roctx();
    if (r.push) r.push(name);
}
void range_pop()
{
    if (!g_roctx_on) return;
    Roctx &r = roctx();
    if (r.pop) r.pop();
}

}  // namespace cm

using namespace cm;

extern "C" int cudamat_version(void) { return CUDAMAT_VERSION; }
extern "C" const char *cudamat_last_error(void) { return g_err; }

extern "C" int cudamat_device_count(int *count)
{
    CM_ARG(count, "count is NULL");
    *count = 0;
    CM_HIP(hipGetDeviceCount(count));
    return CUDAMAT_OK;
}

extern "C" int cudamat_ctx_create(int device, void *stream, cudamat_ctx **out)
{
    CM_ARG(out, "out is NULL");
    *out = nullptr;
    int count = 0;
    CM_HIP(hipGetDeviceCount(&count));
    if (count <= 0) {
        set_error("no HIP device visible: libcudamat_hip has no CPU fallback");
        return CUDAMAT_ERR_HIP;
    }
    CM_ARG(device >= 0 && device < count, "device index out of range");
    CM_HIP(hipSetDevice(device));
    cudamat_ctx *c = new (std::nothrow) cudamat_ctx();
    if (!c) return CUDAMAT_ERR_NOMEM;
    c->device = device;
    c->cfg = config_from_env();
    if (c->cfg.roctx) range_enable();
    if (stream) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail_hip(e, "hipStreamCreate", __FILE__, __LINE__); }
        c->own_stream = true;
    }
    hipError_t e = hipMalloc((void **)&c->parts, sizeof(double) * 2 * kMaxParts);
    if (e == hipSuccess) e = hipMalloc(&c->scratch, kCtxScratchBytes);
    if (e != hipSuccess) {
        if (c->parts) CM_DROP(hipFree(c->parts));
        if (c->own_stream) CM_DROP(hipStreamDestroy(c->stream));
        delete c;
        return fail_hip(e, "hipMalloc(parts)", __FILE__, __LINE__);
    }
    *out = c;
    return CUDAMAT_OK;
}

extern "C" int cudamat_ctx_destroy(cudamat_ctx *ctx)
{
    if (!ctx) return CUDAMAT_OK;
    CM_DROP(hipSetDevice(ctx->device));
    CM_DROP(hipStreamSynchronize(ctx->stream));
    CM_DROP(hipFree(ctx->parts));
    CM_DROP(hipFree(ctx->scratch));
    if (ctx->own_stream) CM_DROP(hipStreamDestroy(ctx->stream));
    delete ctx;
    return CUDAMAT_OK;
}

// One switch of this context (config.h; names as in cudamat_options_help, with or without the CUDAMAT_ prefix).  Read by
// whatever runs on the context AFTER the call: set SpMV-form switches before the first use of a solver, ILU(0) ones
// before cudamat_solver_ilu0, loop ones before cudamat_solver_solve.
extern "C" int cudamat_ctx_set_option(cudamat_ctx *ctx, const char *name, const char *value)
{
    CM_ARG(ctx && name && value, "null pointer");
    if (!config_set(ctx->cfg, name, value)) return CUDAMAT_ERR_ARG;
    ctx->lds_lane_order = -1;          // (PB_PROBE_FAIL may have changed: probe again at the next blocked SpMV)
    if (ctx->cfg.roctx) range_enable();
    return CUDAMAT_OK;
}

// back to what a context created NOW would hold: the defaults overridden by the CUDAMAT_* environment
extern "C" int cudamat_ctx_reset_options(cudamat_ctx *ctx)
{
    CM_ARG(ctx, "ctx is NULL");
    ctx->cfg = config_from_env();
    ctx->lds_lane_order = -1;
    if (ctx->cfg.roctx) range_enable();
    return CUDAMAT_OK;
}

extern "C" const char *cudamat_options_help(void) { return config_help(); }

// would cudamat_ctx_set_option accept this pair?  (no context, no device needed: a host program can validate its settings)
extern "C" int cudamat_option_check(const char *name, const char *value)
{
    Config scratch;
    return config_set(scratch, name, value) ? CUDAMAT_OK : CUDAMAT_ERR_ARG;
}

extern "C" int cudamat_ctx_sync(cudamat_ctx *ctx)
{
    CM_ARG(ctx, "ctx is NULL");
    CM_HIP(hipStreamSynchronize(ctx->stream));
    return CUDAMAT_OK;
}

extern "C" int cudamat_ctx_stream(cudamat_ctx *ctx, void **stream)
{
    CM_ARG(ctx && stream, "null pointer");
    *stream = (void *)ctx->stream;
    return CUDAMAT_OK;
}

extern "C" int cudamat_malloc(cudamat_ctx *ctx, size_t bytes, void **dev)
{
    CM_ARG(ctx && dev, "null pointer");
    CM_HIP(hipSetDevice(ctx->device));
    *dev = nullptr;
    hipError_t e = hipMalloc(dev, bytes ? bytes : 16);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? CUDAMAT_ERR_NOMEM : CUDAMAT_ERR_HIP;
    }
    return CUDAMAT_OK;
}

extern "C" int cudamat_free(cudamat_ctx *ctx, void *dev)
{
    CM_ARG(ctx, "ctx is NULL");
    if (!dev) return CUDAMAT_OK;
    CM_HIP(hipStreamSynchronize(ctx->stream));
    CM_HIP(hipFree(dev));
    return CUDAMAT_OK;
}

extern "C" int cudamat_h2d(cudamat_ctx *ctx, void *dev, const void *host, size_t bytes)
{
    CM_ARG(ctx && (bytes == 0 || (dev && host)), "null pointer");
    if (!bytes) return CUDAMAT_OK;
    CM_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    CM_HIP(hipStreamSynchronize(ctx->stream));
    return CUDAMAT_OK;
}

extern "C" int cudamat_d2h(cudamat_ctx *ctx, void *host, const void *dev, size_t bytes)
{
    CM_ARG(ctx && (bytes == 0 || (dev && host)), "null pointer");
    if (!bytes) return CUDAMAT_OK;
    CM_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    CM_HIP(hipStreamSynchronize(ctx->stream));
    return CUDAMAT_OK;
}

extern "C" int cudamat_d2d(cudamat_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    CM_ARG(ctx && (bytes == 0 || (dst && src)), "null pointer");
    if (!bytes) return CUDAMAT_OK;
    CM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return CUDAMAT_OK;
}

extern "C" int cudamat_memset(cudamat_ctx *ctx, void *dev, int value, size_t bytes)
{
    CM_ARG(ctx && (bytes == 0 || dev), "null pointer");
    if (!bytes) return CUDAMAT_OK;
    CM_HIP(hipMemsetAsync(dev, value, bytes, ctx->stream));
    return CUDAMAT_OK;
}

// ---- timers: a pair of HIP events recorded on the context's stream ---------------
struct cm_timer { hipEvent_t a, b; };

extern "C" int cudamat_timer_create(cudamat_ctx *ctx, void **timer)
{
    CM_ARG(ctx && timer, "null pointer");
    cm_timer *t = (cm_timer *)calloc(1, sizeof(cm_timer));
    if (!t) return CUDAMAT_ERR_NOMEM;
    CM_HIP(hipEventCreate(&t->a));
    CM_HIP(hipEventCreate(&t->b));
    *timer = t;
    return CUDAMAT_OK;
}
extern "C" int cudamat_timer_start(cudamat_ctx *ctx, void *timer)
{
    CM_ARG(ctx && timer, "null pointer");
    CM_HIP(hipEventRecord(((cm_timer *)timer)->a, ctx->stream));
    return CUDAMAT_OK;
}
extern "C" int cudamat_timer_stop(cudamat_ctx *ctx, void *timer)
{
    CM_ARG(ctx && timer, "null pointer");
    CM_HIP(hipEventRecord(((cm_timer *)timer)->b, ctx->stream));
    return CUDAMAT_OK;
}
extern "C" int cudamat_timer_elapsed_ms(cudamat_ctx *ctx, void *timer, double *ms)
{
    CM_ARG(ctx && timer && ms, "null pointer");
    cm_timer *t = (cm_timer *)timer;
    CM_HIP(hipEventSynchronize(t->b));
    float f = 0.f;
    CM_HIP(hipEventElapsedTime(&f, t->a, t->b));
    *ms = (double)f;
    return CUDAMAT_OK;
}
extern "C" int cudamat_timer_destroy(cudamat_ctx *ctx, void *timer)
{
    (void)ctx;
    if (!timer) return CUDAMAT_OK;
    cm_timer *t = (cm_timer *)timer;
    CM_DROP(hipEventDestroy(t->a));
    CM_DROP(hipEventDestroy(t->b));
    free(t);
    return CUDAMAT_OK;
}

// ---- elementary kernels ---------------------------------------------------------------
extern "C" int cudamat_spmv(cudamat_ctx *ctx, int n, const int *rowptr, const int *colidx,
                            const double *val, int base, double alpha, const double *x,
                            const double *d, double beta, double *y)
{
    CM_ARG(ctx && rowptr && colidx && val && x && y, "null pointer");
    CM_ARG(n >= 0 && (base == 0 || base == 1), "n >= 0, base in {0,1}");
    if (n == 0) return CUDAMAT_OK;
    // nnz is needed only to choose lanes-per-row: read the last row pointer
    int last = 0;
    CM_HIP(hipMemcpyAsync(&last, rowptr + n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    CM_HIP(hipStreamSynchronize(ctx->stream));
    SpmvPlan plan = plan_spmv(ctx->cfg, n, (int64_t)last - base);
    CM_TRY(plan_spmv_refine(ctx->stream, ctx->cfg, n, (int64_t)last - base, rowptr, base, &plan, ctx->scratch));
    SpmvArgs a{};
    a.n = n;
    a.rp = rowptr;
    a.ci = colidx - base;   // rp values (with base) index these shifted arrays directly
    a.val = val - base;
    a.x = x - base;
    a.d = d;
    a.xd = x;
    a.alpha = alpha;
    a.beta = beta;
    a.y = y;
    a.dot = 0;
    a.loop = LoopArgs{nullptr, nullptr, 0, 0, 0};
    a.check = CHECK_NONE;
    int rc = launch_spmv(ctx->stream, plan, a);
    if (plan.tiles) {                      // the tile tables live only for this call
        const int rc_sync = CM_RC(hipStreamSynchronize(ctx->stream));
        if (!rc) rc = rc_sync;
        plan_spmv_free(&plan);
    }
    return rc;
}

extern "C" int cudamat_dot(cudamat_ctx *ctx, int64_t n, const double *x, const double *y,
                           double *out_dev)
{
    CM_ARG(ctx && out_dev && n >= 0 && (n == 0 || (x && y)), "bad argument");
    int np = 0;
    CM_TRY(launch_dot_parts(ctx->stream, n, x, y, ctx->parts, &np));
    return launch_reduce_parts(ctx->stream, ScalarSrc{ctx->parts, np, 1}, 1, out_dev, 0);
}

extern "C" int cudamat_nrm2(cudamat_ctx *ctx, int64_t n, const double *x, double *out_dev)
{
    CM_ARG(ctx && out_dev && n >= 0 && (n == 0 || x), "bad argument");
    int np = 0;
    CM_TRY(launch_dot_parts(ctx->stream, n, x, x, ctx->parts, &np));
    return launch_reduce_parts(ctx->stream, ScalarSrc{ctx->parts, np, 1}, 1, out_dev, 1);
}

extern "C" int cudamat_axpy(cudamat_ctx *ctx, int64_t n, double alpha, const double *x, double *y)
{
    CM_ARG(ctx && n >= 0 && (n == 0 || (x && y)), "bad argument");
    return launch_axpy(ctx->stream, n, alpha, x, y);
}

extern "C" int cudamat_scal(cudamat_ctx *ctx, int64_t n, double alpha, double *x)
{
    CM_ARG(ctx && n >= 0 && (n == 0 || x), "bad argument");
    return launch_scal(ctx->stream, n, alpha, x);
}
