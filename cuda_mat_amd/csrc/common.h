// common.h -- internal declarations shared by the HIP translation units of
// libcudamat_hip.so.  gfx950 only; wave = 64.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "cudamat.h"
#include "config.h"

namespace cm {

// device memory comes from a recycling pool (pool.cpp): every hipMalloc / hipFree of this library's translation units
hipError_t pool_malloc(void **out, size_t bytes);
hipError_t pool_free(void *p);
void pool_shrink(void *p, size_t bytes);      // keep the first `bytes` of a block, the rest goes back to the free list
bool pool_segment_of(const void *p, char **seg_base, size_t *block_bytes);
bool pool_split(void *p, size_t offset);      // one block in use becomes two, each freed on its own
unsigned pool_generation();                   // moves whenever a segment went back to the driver
bool pool_enabled();
void pool_trim();                 // free blocks back to the driver (cudamat_plan_cache_clear, out-of-memory retries)
size_t pool_free_bytes();         // what the current device's pool could hand out without asking the driver

}  // namespace cm
#define hipMalloc(p, bytes) cm::pool_malloc((void **)(p), (bytes))
#define hipFree(p) cm::pool_free((void *)(p))

namespace cm {

void set_error(const char *fmt, ...);
int fail_hip(hipError_t e, const char *what, const char *file, int line);
inline int rc_of(hipError_t e, const char *what, const char *file, int line)
{
    return e == hipSuccess ? CUDAMAT_OK : fail_hip(e, what, file, line);
}
// raise hipFuncAttributeMaxDynamicSharedMemorySize of kernel `fn` to 160 KB on the CURRENT device; remembered per
// (device, kernel), so a second device driven by the same process gets its own call; the return code is checked
int set_max_lds(const void *fn);

#define CM_HIP(expr)                                                        \
    do {                                                                    \
        hipError_t e__ = (expr);                                            \
        if (e__ != hipSuccess) return cm::fail_hip(e__, #expr, __FILE__, __LINE__); \
    } while (0)

// a release or teardown call whose failure has no remedy (hipFree, hip*Destroy, the device selection and the last sync
// of a destructor): the result is dropped on purpose
#define CM_DROP(expr) ((void)(expr))
// a HIP call inside code that keeps its own `rc`: records the failure (message included) and yields the error code
#define CM_RC(expr) cm::rc_of((expr), #expr, __FILE__, __LINE__)

#define CM_TRY(expr)                          \
    do {                                      \
        int rc__ = (expr);                    \
        if (rc__ != CUDAMAT_OK) return rc__;  \
    } while (0)

#define CM_ARG(cond, msg)                     \
    do {                                      \
        if (!(cond)) {                        \
            cm::set_error("bad argument: %s", msg); \
            return CUDAMAT_ERR_ARG;           \
        }                                     \
    } while (0)

// Optional roctx ranges: once a context was created with the option ROCTX = 1 the phases of a solve
// (setup, analysis + factorisation, iteration loop, exchanges) are bracketed by roctxRangePush/Pop, bound at run time
// from librocprofiler-sdk-roctx / libroctx64 -- `rocprofv3 --marker-trace` then shows them.  No-ops otherwise.
void range_enable();
void range_push(const char *name);
void range_pop();
struct Range {
    explicit Range(const char *name) { range_push(name); }
    ~Range() { range_pop(); }
    Range(const Range &) = delete;
    Range &operator=(const Range &) = delete;
};

constexpr int kBlock = 256;          // threads per workgroup (4 waves)
constexpr int kMaxParts = 2048;      // upper bound on partial sums per reduction stage
constexpr int kVecGridMax = 1024;    // workgroups of a streaming vector kernel
constexpr int kSpmvGridMax = 2048;   // workgroups of an SpMV launch (256 CUs x 8)
constexpr size_t kCtxScratchBytes = 64 << 10;   // cudamat_ctx::scratch

// A scalar that lives on the device: either `count` per-workgroup partial sums
// (interleaved with stride `stride`, summed in a fixed order by every consumer
// workgroup) or, when count == 0, one already reduced value (sharded runs: the
// partials were summed by reduce_parts and all-reduced across ranks).
struct ScalarSrc {
    const double *ptr;
    int count;
    int stride;
};

// Device-side state of one solve.  Lives in HBM; every kernel of the loop reads
// `state` first and returns at once when it is non-zero ("freeze on exit"), so
// kernels enqueued past the stopping point change nothing.
struct LoopState {
    int state;       // 0 running, 1 half-step exit, 2 full-step exit, 3 omega breakdown
    int it;          // the reference's loop counter i
    int pad[2];
    double rho[2];   // rho of iteration it (slot it&1) and it-1
    double alpha;
    double omega;
    double nrm0;
    double tolabs;   // tol * nrm0
    double nrm;      // last residual norm evaluated
    double alpha2[2];  // pipelined loop: alpha of iteration k in slot k & 1
};

enum Check { CHECK_NONE = 0, CHECK_HALF = 1, CHECK_FULL = 2 };

}  // namespace cm

struct cudamat_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    double *parts = nullptr;   // kMaxParts * 2 doubles scratch for the standalone dot/nrm2
    // 64 KB of device scratch for the small read-backs of the set-up stages (flags, counters, a value-dictionary probe table):
    // those stages then neither allocate nor free -- hipFree waits for EVERY stream of the device, and the host-pointer entry
    // point runs them while an upload is in flight on another stream (dropin.hip).  One user at a time: the context's stream.
    void *scratch = nullptr;
    cm::Config cfg;            // the switches everything running on this context reads (config.h)
    // does this device's LDS serve equal addresses of one ds_add_f64 in lane order?  (-1 not probed yet, 1 yes, 0 no:
    // spmv_pb.hip pb_strict_for; the blocked SpMV's default phase 2 is bit-exact only when it does)
    int lds_lane_order = -1;
};
