// dropin.hip -- the host-pointer entry point: pbicgstab.cu:157-409 / :756-922 / :926-1088 in one call (cudamat_solve),
// and the plan cache that lets a second call with the same matrix skip everything but upload + loop.
//
// The reference's wrappers allocate, upload, analyse, iterate and download one after the other (pbicgstab.cu:243-381).
// At the BASELINE size the upload is 6.1 GB over PCIe -- four times the loop -- so here the rest of the set-up runs
// BESIDE it:
//   uploader    one thread hands the caller's arrays to the runtime in order -- row pointers, column indices, values, b, x0,
//               d -- on an upload stream; milestones (events) mark "pattern landed", "first values landed", "values up to
//               row R landed", "all landed"
//   this thread every device allocation FIRST (sizes only); at "pattern landed": validation, CSR launch plan, the choice of
//               the SpMV form from the pattern and -- when that is the blocked two-phase form -- its count pass and scans;
//               then the fill pass piece by piece behind the value milestones (or in one pass after the value dictionary
//               is known, when the first values suggest the matrix has one).  No hipMalloc / hipFree while bytes travel.
// so that when the last byte lands only the tail of the fill, the loop and the download remain: at C4 a first call takes
// 0.154-0.161 s (upload 0.113-0.120 s at 51-54 GB/s, 1.4-4 ms of set-up not hidden, loop 0.030 s) where round 3 took 0.237 s.
#include <atomic>
#include <chrono>
#include <mutex>
#include <pthread.h>
#include <sched.h>
#include <ctype.h>
#include <thread>
#include <vector>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "solver.h"

using namespace cm;

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

namespace {

// A two-socket host reaches a GPU through ONE socket's PCIe root: a thread that feeds the runtime's pageable copies from the
// other socket moved 11-18 GB/s where a thread on the near socket moves 54 (round 5: the one-GPU boxes of the pool show 256
// CPUs on two NUMA nodes and hand out GPUs of either).  The uploader thread is the library's own: it runs on the CPUs the
// kernel lists as local to the device (/sys/bus/pci/devices/<bdf>/local_cpulist), within the mask the process already
// has; anything missing or unreadable leaves the thread where it is.
static void bind_thread_near_device(int device)
{
    char bdf[32] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) return;
    for (char *c = bdf; *c; c++) *c = (char)tolower((unsigned char)*c);
    char path[128];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bdf);
    FILE *f = fopen(path, "r");
    if (!f) return;
    char list[1024] = {0};
    const bool got = fgets(list, sizeof(list), f) != nullptr;
    fclose(f);
    if (!got) return;
    cpu_set_t have, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(have), &have) != 0) return;
    int picked = 0;
    for (char *tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a = 0, b = 0;
        const int k = sscanf(tok, "%d-%d", &a, &b);
        if (k < 1) continue;
        if (k == 1) b = a;
        for (int cpu = a; cpu <= b && cpu < CPU_SETSIZE; cpu++)
            if (cpu >= 0 && CPU_ISSET(cpu, &have)) { CPU_SET(cpu, &want); picked++; }
    }
    if (picked > 0) (void)pthread_setaffinity_np(pthread_self(), sizeof(want), &want);
}

// ---------------------------------------------------------------------------------------------------- uploader
// One thread hands the runtime the caller's (pageable) arrays in order, 128 MB at a time (hipMemcpyAsync from pageable
// memory: the runtime pins the pages piece by piece and lets the DMA engines read them: 52-55 GB/s on the development
// boxes); milestones (events recorded behind a piece) tell the set-up what has landed.  A second form -- host threads
// staging 8 MB chunks into pinned slots -- was built and measured in round 4: 50 GB/s alone, 31-37 GB/s beside the set-up
// kernels, 15 ms for its pinned slots; removed (HISTORY.md).
constexpr size_t kPieceBytes = 128u << 20;

struct Uploader {
    struct Chunk { char *dst; const char *src; size_t bytes; int milestone; };     // milestone: recorded AFTER this chunk (-1: none)
    std::vector<Chunk> chunks;
    std::vector<hipEvent_t> ms_event;
    std::vector<std::atomic<int>> ms_recorded;
    std::atomic<int> failed{0};
    hipError_t err = hipSuccess;          // the uploader thread's failing call (read after join(): HIP's last error is per thread)
    const char *err_what = "";
    hipStream_t stream = nullptr;
    int device = 0;
    std::thread thread;
    double t_done = 0.0;

    // append [src, src + bytes) -> dst; returns the milestone id recorded after its last byte
    int add(void *dst, const void *src, size_t bytes)
    {
        const char *sp = (const char *)src;
        char *dp = (char *)dst;
        for (size_t off = 0; off < bytes; off += kPieceBytes)
            chunks.push_back(Chunk{dp + off, sp + off, bytes - off < kPieceBytes ? bytes - off : kPieceBytes, -1});
        return mark();
    }
    // a milestone after everything added so far
    int mark()
    {
        if (chunks.empty()) return -1;
        if (chunks.back().milestone < 0) {
            chunks.back().milestone = (int)ms_event.size();
            ms_event.push_back(nullptr);
        }
        return chunks.back().milestone;
    }

    int start(int dev)
    {
        device = dev;
        CM_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (hipEvent_t &e : ms_event) CM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ms_recorded = std::vector<std::atomic<int>>(ms_event.size());
        for (auto &a : ms_recorded) a = 0;
        thread = std::thread([this] { issue(); });
        return CUDAMAT_OK;
    }

    void issue()
    {
        auto bad = [this](hipError_t e, const char *what) {
            if (e == hipSuccess) return false;
            err = e; err_what = what;
            failed = 1;
            return true;
        };
        if (bad(hipSetDevice(device), "hipSetDevice")) return;
        bind_thread_near_device(device);
        for (size_t c = 0; c < chunks.size(); c++) {
            if (failed) return;
            if (bad(hipMemcpyAsync(chunks[c].dst, chunks[c].src, chunks[c].bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync")) return;
            const int m = chunks[c].milestone;
            if (m >= 0) {
                if (bad(hipEventRecord(ms_event[(size_t)m], stream), "hipEventRecord")) return;
                ms_recorded[(size_t)m].store(1, std::memory_order_release);
            }
        }
        bad(hipStreamSynchronize(stream), "hipStreamSynchronize");
        t_done = now_s();
    }

    // make `st` wait for milestone m (returns once the wait is enqueued; the host does not wait for the bytes)
    int wait_on(hipStream_t st, int m)
    {
        if (m < 0) return CUDAMAT_OK;
        while (!ms_recorded[(size_t)m].load(std::memory_order_acquire)) {
            if (failed) { set_error("upload failed"); return CUDAMAT_ERR_HIP; }
            std::this_thread::sleep_for(std::chrono::microseconds(50));      // (the uploader may share this thread's core)
        }
        CM_HIP(hipStreamWaitEvent(st, ms_event[(size_t)m], 0));
        return CUDAMAT_OK;
    }

    void join()
    {
        if (thread.joinable()) thread.join();
        for (hipEvent_t e : ms_event)
            if (e) CM_DROP(hipEventDestroy(e));
        ms_event.clear();
        if (stream) { CM_DROP(hipStreamSynchronize(stream)); CM_DROP(hipStreamDestroy(stream)); stream = nullptr; }
    }
    int finish()
    {
        join();
        if (failed) {
            set_error("host-to-device upload failed (%s: %s)", err_what, err == hipSuccess ? "stopped early" : hipGetErrorString(err));
            return CUDAMAT_ERR_HIP;
        }
        return CUDAMAT_OK;
    }
    ~Uploader() { if (thread.joinable()) failed = 1; join(); }      // (an early exit: tell the thread to stop; never touches the error string)
};

// ---------------------------------------------------------------------------------------------------- plan cache
// The reference allocates, analyses, solves and frees per call (pbicgstab.cu:157-409).  Here the solver of the last
// call stays alive: when the next call brings the same matrix (same n, nnz, base and -- compared ON THE DEVICE after
// the upload, 12 bytes per entry read twice: ~2.5 ms at C4 -- the same row pointers, column indices and values), its
// device copies, SpMV plan, value dictionary and ILU(0) factors are reused and the call costs upload + loop.
struct PlanCache {
    std::mutex mu;
    cudamat_ctx *ctx = nullptr;
    cudamat_solver *s = nullptr;
    int n = 0, nnz = 0, base = 0;
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
    cache_drop_locked();     // (its memory goes back to the library's pool: cudamat_pool_trim returns that to the driver)
    return CUDAMAT_OK;
}

extern "C" int cudamat_pool_trim(void)
{
    pool_trim();
    return CUDAMAT_OK;
}

extern "C" int cudamat_mem_info(int device, size_t *driver_free, size_t *driver_total, size_t *pool_free)
{
    int before = 0;
    CM_TRY(CM_RC(hipGetDevice(&before)));
    CM_TRY(CM_RC(hipSetDevice(device)));
    size_t f = 0, t = 0;
    const hipError_t e = hipMemGetInfo(&f, &t);
    const size_t pf = pool_free_bytes();
    (void)hipSetDevice(before);
    if (e != hipSuccess) return CM_RC(e);
    if (driver_free) *driver_free = f;
    if (driver_total) *driver_total = t;
    if (pool_free) *pool_free = pf;
    return CUDAMAT_OK;
}

namespace {

struct HostSystem {
    int n, nnz, base;
    int64_t n_cols;       // == n for the drop-in entry point; a row block of a wider matrix for cudamat_solver_create_host
    const double *A;
    const int *iA, *jA;
    const double *d, *x0, *b;
};

// ---- a call whose shape matches the cached solver's: upload into scratch arrays, compare on the device, reuse or rebuild
// from the uploaded copies.  *s_out = the solver to use (the cached one or a new one), *reused says which.
int build_or_reuse_candidate(cudamat_ctx *ctx, const Config &cfg, const HostSystem &h, double *d_b, double *d_x, double *d_d,
                             cudamat_solver **s_out, bool *reused, double *t_up)
{
    const double t0 = now_s();
    const int n = h.n, nnz = h.nnz, base = h.base;
    int *d_rp = nullptr, *d_ci = nullptr, *flag = nullptr, *tmp = nullptr;
    double *d_val = nullptr;
    int rc = CUDAMAT_OK;
    *reused = false;
    *s_out = nullptr;
    do {
        if ((rc = cudamat_malloc(ctx, sizeof(int) * ((size_t)n + 1), (void **)&d_rp))) break;
        if ((rc = cudamat_malloc(ctx, sizeof(int) * (size_t)nnz, (void **)&d_ci))) break;
        if ((rc = cudamat_malloc(ctx, sizeof(double) * (size_t)nnz, (void **)&d_val))) break;
        {
            Uploader up;
            up.add(d_rp, h.iA, sizeof(int) * ((size_t)n + 1));            // pbicgstab.cu:313-315
            up.add(d_ci, h.jA, sizeof(int) * (size_t)nnz);
            up.add(d_val, h.A, sizeof(double) * (size_t)nnz);
            up.add(d_b, h.b, sizeof(double) * (size_t)n);
            if (h.x0) up.add(d_x, h.x0, sizeof(double) * (size_t)n);
            if (h.d) up.add(d_d, h.d, sizeof(double) * (size_t)n);
            if ((rc = up.start(ctx->device))) break;
            if ((rc = up.finish())) break;
        }
        *t_up = now_s() - t0;
        // the cached solver holds the matrix rebased to 0: compare the uploaded arrays with it on the device
        int hflag = 1;
        cudamat_solver *c = g_cache.s;
        if ((rc = cudamat_malloc(ctx, sizeof(int), (void **)&flag))) break;
        // (a flag word that could not be cleared would make any matrix look "different": fail the call instead)
        if ((rc = CM_RC(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream)))) break;
        rc = cudamat_malloc(ctx, sizeof(int) * ((size_t)nnz > (size_t)n + 1 ? (size_t)nnz : (size_t)n + 1), (void **)&tmp);
        if (!rc) rc = launch_rebase(ctx->stream, (int64_t)n + 1, d_rp, -base, tmp);
        if (!rc) rc = device_equal(ctx->stream, tmp, c->rp, sizeof(int) * ((size_t)n + 1), flag);
        if (!rc && nnz) rc = launch_rebase(ctx->stream, nnz, d_ci, -base, tmp);
        if (!rc && nnz) rc = device_equal(ctx->stream, tmp, c->ci, sizeof(int) * (size_t)nnz, flag);
        if (!rc && nnz) rc = device_equal(ctx->stream, d_val, c->val, sizeof(double) * (size_t)nnz, flag);
        if (!rc && hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = CUDAMAT_ERR_HIP;
        if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = CUDAMAT_ERR_HIP;
        if (rc) break;
        if (hflag == 0) {
            *s_out = g_cache.s;
            *reused = true;
        } else {                     // same shape, another matrix: the old solver goes, its context stays
            cudamat_solver_destroy(g_cache.s);
            g_cache.s = nullptr;
            if (g_cache.d_d) { cudamat_free(ctx, g_cache.d_d); g_cache.d_d = nullptr; }
            rc = cudamat_solver_create(ctx, n, n, nnz, d_rp, d_ci, d_val, base, s_out);
        }
    } while (0);
    void *ptrs[] = {d_rp, d_ci, d_val, flag, tmp};
    for (void *p : ptrs)
        if (p) cudamat_free(ctx, p);
    return rc;
}

// ---- a call with a new matrix: the set-up runs beside the upload (see the top of this file).
// Everything is ALLOCATED before the first byte travels (the blocked copy's geometry depends on the sizes only) and
// nothing is freed until the last byte has landed: hipMalloc / hipFree take process-wide locks (hipFree waits for every
// stream of the device), and an upload that shares the process with them was measured at a sixth of its speed.
// speculative = false (the retry after an out-of-memory call): nothing is allocated ahead of the pattern; the SpMV form is
// then chosen, and its copy built, after the upload by ensure_spmv_mode like in the staged cudamat_solver_create path.
// precond == CUDAMAT_PRECOND_ILU0: the pattern-only part of the ILU(0) set-up (diagonal positions, level analysis of L and U:
// 32 ms at C5) runs as soon as the pattern has landed, beside the upload of the values (round 5).
int build_beside_upload(cudamat_ctx *ctx, const Config &cfg, const HostSystem &h, double *d_b, double *d_x, double *d_d,
                        cudamat_solver **s_out, double *t_up, bool speculative = true, int precond = CUDAMAT_PRECOND_NONE,
                        int loop = CUDAMAT_LOOP_PBICGSTAB)
{
    const double t0 = now_s();
    const int n = h.n, nnz = h.nnz, base = h.base;
    const int64_t n_cols = h.n_cols;
    hipStream_t st = ctx->stream;
    cudamat_solver *s = nullptr;
    *s_out = nullptr;
    const bool verbose = cfg.verbose != 0;
    auto stamp = [&](const char *what) {
        if (verbose) fprintf(stderr, "[cudamat] drop-in %-52s at %8.3f ms\n", what, (now_s() - t0) * 1e3);
    };
    CM_TRY(solver_alloc(ctx, n, n_cols, nnz, &s));
    // the blocked copy is a candidate by size (pb_candidate looks at sizes only): allocate it now, decide when the
    // pattern is there; a matrix that ends up with another form frees it after the upload
    PbBuild pb;
    bool pb_open = false;
    // (a row block of a wider matrix will be sharded by cudamat_solver_set_comm, which cuts the column blocks at the ranks'
    // slices: its copy is built then)
    bool want_pb = speculative && n_cols == (int64_t)n && cfg.spmv_mode != 0 && cfg.spmv_mode != 2 && cfg.spmv_mode != 3 &&
                   (cfg.spmv_mode == 1 || !cfg.spmv_tune_full) && pb_candidate(st, n, n_cols, nnz, nullptr, nullptr) &&
                   (cfg.spmv_mode == 1 || (int64_t)nnz >= 8 * (int64_t)n);
    if (want_pb) {
        // the copy is 20 B per entry and the pattern may still turn it down (banded, not scattered): take it ahead of the
        // pattern only while it leaves as much again free -- otherwise a later stage (value dictionary, ILU(0), a cached
        // system) could run out of memory where the staged path would have fitted
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b + pool_free_bytes() < 2 * (size_t)20 * (size_t)nnz) want_pb = false;
    }
    if (want_pb) {
        const int rcb = pb_build_alloc(st, cfg, n, n_cols, nnz, nullptr, &pb);
        pb_open = rcb == CUDAMAT_OK;                     // (no room / outside the form's limits: ensure_spmv_mode decides later)
        if (pb_open && pb_build_values(st, &pb, nullptr) != CUDAMAT_OK) pb_open = false;      // fp64 value array (4 GB at C4); a matrix
    }                                                                                        // with a dictionary swaps it afterwards
    int rc = CUDAMAT_OK;
    if (hipStreamSynchronize(st) != hipSuccess) rc = CUDAMAT_ERR_HIP;
    stamp("allocations done (solver, blocked copy)");
    double t_pb0 = 0.0;
    bool piecewise = false, blocked = false;
    if (rc == CUDAMAT_OK) {
        Uploader up;
        const int m_rp = up.add(s->rp, h.iA, sizeof(int) * ((size_t)n + 1));             // pbicgstab.cu:313-315
        const int m_pattern = nnz ? up.add(s->ci, h.jA, sizeof(int) * (size_t)nnz) : m_rp;
        // the values in pieces that end on row boundaries (~256 MB each): a piece's rows can be placed in the blocked
        // copy as soon as it has landed; the first piece is the value dictionary's sample
        struct Piece { int row_end; int milestone; size_t entries_end; };
        std::vector<Piece> pieces;
        int m_first = m_pattern;
        {
            const size_t piece_entries = (size_t)32 << 20;
            size_t e0 = 0;
            while (e0 < (size_t)nnz) {
                const size_t want = e0 + (pieces.empty() ? (size_t)1 << 20 : piece_entries);
                // the piece ends at the first row boundary at or after `want` (row pointers are the caller's, with its base)
                int lo = 0, hi = n;
                while (lo < hi) {
                    const int mid = lo + (hi - lo) / 2;
                    if ((size_t)(h.iA[mid] - base) >= want) hi = mid; else lo = mid + 1;
                }
                int row_end = lo;                                          // rows [.., row_end) are complete after this piece
                long long cut = (long long)h.iA[row_end] - base;           // (row_end == n: cut == nnz)
                // (row pointers that are not what they should be -- the device-side validation will refuse the matrix --
                // must not send the uploader astray: everything that is left becomes one piece)
                if (cut <= (long long)e0 || cut > (long long)nnz) { cut = nnz; row_end = n; }
                const int m = up.add(s->val + e0, h.A + e0, sizeof(double) * ((size_t)cut - e0));
                if (pieces.empty()) m_first = m;
                pieces.push_back(Piece{row_end, m, (size_t)cut});
                e0 = (size_t)cut;
            }
        }
        if (h.b) up.add(d_b, h.b, sizeof(double) * (size_t)n);
        if (h.x0) up.add(d_x, h.x0, sizeof(double) * (size_t)n);
        if (h.d) up.add(d_d, h.d, sizeof(double) * (size_t)n);
        const int m_all = up.mark();
        do {
            if ((rc = up.start(ctx->device))) break;
            stamp("uploader started");
            // ---- pattern landed: index base, validation, CSR plan, the SpMV form
            if ((rc = up.wait_on(st, m_pattern))) break;
            stamp("pattern milestone recorded");
            if (base) {
                if ((rc = launch_rebase(st, (int64_t)n + 1, s->rp, -base, s->rp))) break;
                if (nnz && (rc = launch_rebase(st, nnz, s->ci, -base, s->ci))) break;
            }
            if ((rc = solver_setup_pattern(s))) break;
            stamp("pattern stage done (validation, CSR plan)");
            // ---- the level analysis of the preconditioner needs the pattern only: its kernels (and the host's waits for their
            // read-backs) fit into the time the values take to arrive; the fill pieces below queue up behind them
            bool loop_in_level_major = false;
            if (precond == CUDAMAT_PRECOND_ILU0 && cfg.early_analysis) {
                if ((rc = ilu0_analyse_early(s))) break;
                stamp("ILU(0) level analysis done (beside the upload)");
                // Both factors hybrid: the preconditioned reference loop runs in the level-major spaces on its OWN blocked copy
                // (rows in L's order, columns in U's positions; loops.hip) and never touches one in the original space -- that
                // copy (12 ms of fill whose tail is not hidden, 10 GB) is then not built; a later un-preconditioned solve on
                // the same matrix builds it at its first SpMV (ensure_spmv_mode).
                loop_in_level_major = cfg.skip_orig_copy && loop == CUDAMAT_LOOP_PBICGSTAB && !h.d && cfg.trsv_perm && ilu0_will_use_level_major(s);
            }
            if (pb_open && !loop_in_level_major && (rc = spmv_mode_is_blocked_early(s, &blocked))) break;
            t_pb0 = now_s();
            if (pb_open && blocked) {
                if ((rc = pb_build_count(st, cfg, s->rp, s->ci, &pb))) { pb_open = false; break; }
                stamp("blocked copy: count pass and scans done");
                // does the matrix look like it has a value dictionary?  (a sample of the first values; arbitrary
                // coefficients overflow the 256-entry table within the first thousands)
                bool many = true;
                if (cfg.value_dict && (int64_t)nnz >= (1 << 20)) {
                    if ((rc = up.wait_on(st, m_first))) break;
                    if ((rc = valdict_sample_overflows(st, (int64_t)pieces[0].entries_end, s->val, ctx->scratch, &many))) break;
                }
                if (many) {
                    s->vd_tried = true;                  // (more than 256 distinct values, or the dictionary is switched off)
                    piecewise = true;
                }
            }
            // ---- values: piece by piece into the blocked copy, or all at once behind the dictionary
            if (piecewise) {
                int sub_done = 0;
                for (const Piece &pc : pieces) {
                    if ((rc = up.wait_on(st, pc.milestone))) break;
                    const int sub_new = pc.row_end >= n ? pb.p.NSUB : pc.row_end / pb.p.SR;
                    if ((rc = pb_build_fill(st, &pb, s->rp, s->ci, s->val, nullptr, sub_done, sub_new))) { pb_open = false; break; }
                    if (sub_new > sub_done) sub_done = sub_new;
                }
                if (rc) break;
                stamp("last fill piece enqueued");
            }
            if ((rc = up.wait_on(st, m_all))) break;
            if ((rc = up.finish())) break;
            stamp("upload finished");
            *t_up = (up.t_done > 0.0 ? up.t_done : now_s()) - t0;
        } while (0);
        if (rc) up.failed = 1;
    }      // (the uploader's threads are joined here whatever happened)
    // ---- the rest may allocate and free again
    if (s) ilu0_flush_deferred(s);
    do {
        if (rc) break;
        if ((rc = solver_setup_values(s))) break;
        if (pb_open && !blocked) { pb_build_abort(&pb); pb_open = false; }        // another form: ensure_spmv_mode decides at the solve
        if (pb_open && !piecewise) {
            if ((rc = ensure_valdict(s))) break;
            if (s->vd.n > 0) {                           // 8-bit indices instead of the fp64 values
                CM_DROP(hipFree(pb.p.pv));
                pb.p.pv = nullptr;
                if ((rc = pb_build_values(st, &pb, &s->vd))) { pb_open = false; break; }
            }
            if ((rc = pb_build_fill(st, &pb, s->rp, s->ci, s->val, &s->vd, 0, pb.p.NSUB))) { pb_open = false; break; }
        }
        if (pb_open) {
            PbPlan plan;
            pb_open = false;
            if ((rc = pb_build_end(st, &pb, &plan))) break;
            spmv_mode_adopt_blocked(s, plan, now_s() - t_pb0);
        }
        // what the statistics call set-up: the part of it that was NOT hidden behind the upload
        if (hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("set-up beside the upload failed"); break; }
        s->t_create = now_s() - (t0 + *t_up);
        s->t_spmv_setup = 0.0;
        stamp("set-up complete");
    } while (0);
    if (rc) {
        char saved[512];
        snprintf(saved, sizeof(saved), "%s", cudamat_last_error());
        if (pb_open) pb_build_abort(&pb);
        cudamat_solver_destroy(s);
        set_error("%s", saved);
        return rc;
    }
    *s_out = s;
    return CUDAMAT_OK;
}

// one attempt (the caller holds g_cache.mu)
int solve_host_locked(const Config &cfg, const HostSystem &h, bool speculative, int precond, int loop, int maxit, double tol, int debug,
                      double *x, cudamat_stats *out)
{
    const int n = h.n, nnz = h.nnz, base = h.base;
    const double t0 = now_s();
    const bool use_cache = cfg.plan_cache != 0;
    if (!use_cache) cache_drop_locked();
    // same shape as the cached system, same switches?  then its context (device, stream, options) carries this call too
    const bool candidate = use_cache && g_cache.s && g_cache.n == n && g_cache.nnz == nnz && g_cache.base == base &&
                           g_cache.ctx->cfg == cfg;
    if (!candidate) cache_drop_locked();
    cudamat_ctx *ctx = candidate ? g_cache.ctx : nullptr;
    if (!ctx) {
        CM_TRY(cudamat_ctx_create(0, nullptr, &ctx));
        ctx->cfg.pb_place = cfg.pb_place;          // (the one switch this entry point overrides, see cudamat_solve)
    }
    double *d_b = nullptr, *d_x = nullptr, *d_d = nullptr;
    cudamat_solver *s = nullptr;
    bool reused = false, built_ilu = false;
    int rc = CUDAMAT_OK;
    cudamat_stats st;
    memset(&st, 0, sizeof(st));
    double t_up = 0.0;
    do {
        if ((rc = cudamat_malloc(ctx, sizeof(double) * (size_t)n, (void **)&d_b))) break;
        if ((rc = cudamat_malloc(ctx, sizeof(double) * (size_t)n, (void **)&d_x))) break;
        if (h.d && (rc = cudamat_malloc(ctx, sizeof(double) * (size_t)n, (void **)&d_d))) break;
        if (candidate) {
            if ((rc = build_or_reuse_candidate(ctx, cfg, h, d_b, d_x, d_d, &s, &reused, &t_up))) break;
        } else {
            if ((rc = build_beside_upload(ctx, cfg, h, d_b, d_x, d_d, &s, &t_up, speculative, precond, loop))) break;
        }
        if (g_cache.d_d && !reused) { cudamat_free(ctx, g_cache.d_d); g_cache.d_d = nullptr; }
        if ((rc = cudamat_solver_set_shift(s, d_d))) break;
        if (precond != CUDAMAT_PRECOND_NONE && !(reused && s->has_ilu && !s->ilu_block)) {
            if ((rc = cudamat_solver_ilu0(s))) break;
            built_ilu = true;
        }
        if (precond != CUDAMAT_PRECOND_NONE && debug) {
            printf("analysis lower %f (s), upper %f (s) \n", s->t_analysis_l, s->t_analysis_u);     // :349
            printf("csrilu0 (HIP, level-scheduled) time(s) = %10.8f \n", s->t_factor);            // :355,363
        }
        int flags = (debug ? CUDAMAT_FLAG_DEBUG : 0) | (h.x0 ? 0 : CUDAMAT_FLAG_X0_ONES);
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
        if (g_cache.d_d && g_cache.d_d != d_d) cudamat_free(ctx, g_cache.d_d);
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
    void *ptrs[] = {d_b, d_x, d_d};
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

}  // namespace

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
    Config cfg = config_from_env();                                // no caller-made context: the switches of THIS call
    // (placing a copy's arrays by memory class costs 10-35 ms of probing before the upload can start -- a fifth of a C4-sized
    // call for a loop of a few iterations: only on request here, PB_PLACE=2)
    cfg.pb_place = cfg.pb_place >= 2 ? 1 : 0;
    const HostSystem h{n, nnz, base, (int64_t)n, A, iA, jA, d, x0, b};
    std::lock_guard<std::mutex> cache_lock(g_cache.mu);            // (the entry points are not re-entrant upstream either)
    int rc = solve_host_locked(cfg, h, true, precond, loop, maxit, tol, debug, x, out);
    if (rc == CUDAMAT_ERR_NOMEM) {
        // the solver kept from the previous call (several GB at the BASELINE sizes) may be what is in the way: the
        // reference frees everything per call (pbicgstab.cu:392-405), so release it and try once more
        cache_drop_locked();
        pool_trim();
        rc = solve_host_locked(cfg, h, false, precond, loop, maxit, tol, debug, x, out);
    }
    return rc;
}

// cudamat_solver_create from HOST arrays: the same staged creation, run beside the upload (what cudamat_solve does for its
// matrix).  For a host program that keeps its system resident (cudamat_solver_*) but holds the matrix in host memory.
extern "C" int cudamat_solver_create_host(cudamat_ctx *ctx, int n_local, int64_t n_cols, int64_t nnz, const int *rowptr,
                                          const int *colidx, const double *val, int base, cudamat_solver **out)
{
    CM_ARG(ctx && out, "null pointer");
    *out = nullptr;
    CM_ARG(base == 0 || base == 1, "base in {0,1}");
    CM_ARG(rowptr && (nnz == 0 || (colidx && val)), "null CSR array");
    CM_ARG(n_local >= 0 && n_cols >= n_local && nnz >= 0 && nnz < (1LL << 31) && n_cols < (1LL << 31), "sizes");
    CM_ARG(rowptr[0] == base && (int64_t)rowptr[n_local] - base == nnz, "row pointers must start at the base and end at base + nnz");
    Range range_create("cudamat: solver create from host arrays (set-up beside the upload)");
    const HostSystem h{n_local, (int)nnz, base, n_cols, val, rowptr, colidx, nullptr, nullptr, nullptr};
    double t_up = 0.0;
    return build_beside_upload(ctx, ctx->cfg, h, nullptr, nullptr, nullptr, out, &t_up);
}
