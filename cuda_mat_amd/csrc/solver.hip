// solver.hip -- the BiCGSTAB driver: one host loop that only ENQUEUES work.
//
// Reference behaviour restated (citations into /root/reference):
//   CUDAMAT_LOOP_PBICGSTAB  = gpu_pbicgstab  pbicgstab.cu:45-154  (ILU(0) or M = I)
//   CUDAMAT_LOOP_PBICGSTAB2 = gpu_pbicgstab2 pbicgstab.cu:581-754 (d variant; with
//                             d == NULL the intended maths of :425-578)
// MI355X-first differences from the reference's structure:
//   * the reference blocks the host 5-6 times per iteration on cuBLAS scalar
//     results; here rho/alpha/omega/norms stay in HBM (LoopState), kernels read
//     them in their prologue, and the host only looks at a snapshot that is
//     kLag iterations old -> the stream never drains.
//   * "freeze on exit": once a stopping test fires on the device, every later
//     kernel returns immediately, so the lagged host check costs no accuracy
//     and the iterate is exactly the one the reference would return.
//   * 21 (plain) / 13 (ILU) vector passes per iteration collapse into 3 fused
//     kernels; the dot products ride on the kernels that stream the operands.
//   * row-sharded operation: the same loop, with the SpMV input gathered and the
//     scalar partials all-reduced through caller-supplied collectives (RCCL).
#include <chrono>
#include <mutex>
#include <utility>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "solver.h"

using namespace cm;

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int dev_alloc(void **p, size_t bytes)
{
    *p = nullptr;
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? CUDAMAT_ERR_NOMEM : CUDAMAT_ERR_HIP;
    }
    return CUDAMAT_OK;
}

static void free_work(cudamat_solver *s)
{
    double **vs[] = {&s->r, &s->rw, &s->p, &s->pw, &s->s, &s->t, &s->v, &s->gather, &s->x0_save, &s->v2,
                     &s->pz, &s->pww, &s->pq, &s->py, &s->pxh, &s->pipeA, &s->pipeB, &s->red_pipe,
                     &s->prh, &s->pwh, &s->psh, &s->pzh, &s->pqh, &s->ptmp};
    for (double **q : vs) {
        if (*q) hipFree(*q);
        *q = nullptr;
    }
}

static int ensure_work(cudamat_solver *s)
{
    if (s->r) return CUDAMAT_OK;
    hipStream_t st = s->ctx->stream;
    const size_t nb = sizeof(double) * (size_t)(s->n_pad > 0 ? s->n_pad : 1);
    double **vs[] = {&s->r, &s->rw, &s->p, &s->pw, &s->s, &s->t, &s->v};
    for (double **q : vs) {
        CM_TRY(dev_alloc((void **)q, nb));
        CM_HIP(hipMemsetAsync(*q, 0, nb, st));
    }
    if (s->sharded) {
        CM_TRY(dev_alloc((void **)&s->gather, nb * (size_t)s->comm.world));
        CM_HIP(hipMemsetAsync(s->gather, 0, nb * (size_t)s->comm.world, st));
    }
    return CUDAMAT_OK;
}

static int ensure_spmv_mode(cudamat_solver *s);

// ---- input validation (once, at creation): a malformed CSR must become an error code, never a stray access
// flags[0]: row pointers not 0 = rp[0] <= rp[1] <= ... <= rp[n] = nnz;  flags[1]: a column id outside [0, n_cols);
// flags[2]: some row's columns are not strictly increasing
__global__ __launch_bounds__(kBlock) void k_check_rowptr(int n, long long nnz, const int *rp, int *flags)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i > n) return;
    bool bad = false;
    if (i == 0 && rp[0] != 0) bad = true;
    if (i == n && rp[n] != nnz) bad = true;
    if (i < n && rp[i] > rp[i + 1]) bad = true;
    if (rp[i] < 0 || rp[i] > nnz) bad = true;
    if (bad) flags[0] = 1;
}

__global__ __launch_bounds__(kBlock) void k_check_columns(int n, long long n_cols, const int *rp, const int *ci, int *flags)
{
    constexpr int L = 8;
    const long long row = ((long long)blockIdx.x * kBlock + threadIdx.x) / L;
    if (row >= n) return;
    const int lane = threadIdx.x & (L - 1);
    const int s = rp[row], e = rp[row + 1];
    bool range = false, order = false;
    for (int k = s + lane; k < e; k += L) {
        const int c = ci[k];
        if (c < 0 || c >= n_cols) range = true;
        if (k > s && ci[k - 1] >= c) order = true;
    }
    if (range) flags[1] = 1;
    if (order) flags[2] = 1;
}

static int validate_csr(cudamat_solver *s)
{
    hipStream_t st = s->ctx->stream;
    int *d = nullptr, h[3] = {0, 0, 0};
    CM_HIP(hipMalloc((void **)&d, sizeof(h)));
    hipMemsetAsync(d, 0, sizeof(h), st);
    hipLaunchKernelGGL(k_check_rowptr, dim3((unsigned)(((long long)s->n + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, s->n,
                       (long long)s->nnz, s->rp, d);
    hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess && !h[0] && s->n > 0 && s->nnz > 0) {
        hipLaunchKernelGGL(k_check_columns, dim3((unsigned)(((long long)s->n * 8 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                           s->n, (long long)s->n_cols, s->rp, s->ci, d);
        hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
        e = hipStreamSynchronize(st);
    }
    hipFree(d);
    if (e != hipSuccess) return fail_hip(e, "CSR validation", __FILE__, __LINE__);
    if (h[0]) { set_error("row pointers must start at the index base, never decrease and end at nnz"); return CUDAMAT_ERR_ARG; }
    if (h[1]) { set_error("a column index lies outside [base, base + n_cols)"); return CUDAMAT_ERR_ARG; }
    s->cols_sorted = h[2] == 0;
    return CUDAMAT_OK;
}

static int ensure_valdict(cudamat_solver *s);

extern "C" int cudamat_solver_create(cudamat_ctx *ctx, int n_local, int64_t n_cols, int64_t nnz,
                                     const int *rowptr, const int *colidx, const double *val,
                                     int base, cudamat_solver **out)
{
    CM_ARG(ctx && out, "null pointer");
    *out = nullptr;
    CM_ARG(n_local >= 0 && n_cols >= n_local && nnz >= 0, "sizes");
    CM_ARG(nnz < (1LL << 31) && n_cols < (1LL << 31), "local nnz and dimension must fit int32");
    CM_ARG(base == 0 || base == 1, "base in {0,1}");
    CM_ARG(rowptr && (nnz == 0 || (colidx && val)), "null CSR array");
    CM_HIP(hipSetDevice(ctx->device));
    Range range_create("cudamat: solver create (copies, validation, CSR plan)");
    const double t_create0 = now_s();
    cudamat_solver *s = new cudamat_solver();
    s->ctx = ctx;
    s->n = n_local;
    s->n_pad = n_local;
    s->n_cols = n_cols;
    s->nnz = nnz;
    hipStream_t st = ctx->stream;
    int rc = CUDAMAT_OK;
    do {
        if ((rc = dev_alloc((void **)&s->rp, sizeof(int) * ((size_t)n_local + 1)))) break;
        if ((rc = dev_alloc((void **)&s->ci, sizeof(int) * (size_t)nnz))) break;
        if ((rc = dev_alloc((void **)&s->val, sizeof(double) * (size_t)nnz))) break;
        if ((rc = launch_rebase(st, (int64_t)n_local + 1, rowptr, -base, s->rp))) break;
        if (nnz) {
            if ((rc = launch_rebase(st, nnz, colidx, -base, s->ci))) break;
            if (hipMemcpyAsync(s->val, val, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToDevice, st) != hipSuccess) {
                rc = CUDAMAT_ERR_HIP; set_error("val copy failed"); break;
            }
        }
        if ((rc = dev_alloc((void **)&s->parts_full, sizeof(double) * 2 * kMaxParts))) break;
        if ((rc = dev_alloc((void **)&s->parts_rv, sizeof(double) * 2 * kMaxParts))) break;
        if ((rc = dev_alloc((void **)&s->parts_half, sizeof(double) * 2 * kMaxParts))) break;
        if ((rc = dev_alloc((void **)&s->parts_tt, sizeof(double) * 2 * kMaxParts))) break;
        if ((rc = dev_alloc((void **)&s->red, sizeof(double) * 16))) break;
        if ((rc = dev_alloc((void **)&s->st, sizeof(LoopState)))) break;
        if (hipHostMalloc((void **)&s->st_ring, sizeof(LoopState) * kRing, hipHostMallocDefault) != hipSuccess) {
            rc = CUDAMAT_ERR_HIP; set_error("hipHostMalloc failed"); break;
        }
        if (hipHostMalloc((void **)&s->snap_host, sizeof(unsigned long long) * kRing, hipHostMallocDefault) != hipSuccess ||
            hipHostGetDevicePointer((void **)&s->snap_dev, s->snap_host, 0) != hipSuccess) {
            rc = CUDAMAT_ERR_HIP; set_error("pinned progress words unavailable"); break;
        }
        if (hipMemsetAsync(s->st, 0, sizeof(LoopState), st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        if (hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("sync after upload failed"); break; }
    } while (0);
    if (rc) {
        cudamat_solver_destroy(s);
        return rc;
    }
    const bool verbose = getenv("CUDAMAT_VERBOSE") != nullptr;
    double t_mark = now_s();
    auto stamp = [&](const char *what) {
        if (!verbose) return;
        const double t = now_s();
        fprintf(stderr, "[cudamat] create %-34s %8.3f ms\n", what, (t - t_mark) * 1e3);
        t_mark = t;
    };
    if (verbose) fprintf(stderr, "[cudamat] create %-34s %8.3f ms\n", "allocations + copies", (t_mark - t_create0) * 1e3);
    if (int rcv = validate_csr(s)) {
        cudamat_solver_destroy(s);
        return rcv;
    }
    stamp("validation");
    s->plan = plan_spmv(n_local, nnz);
    if (int rc2 = plan_spmv_refine(st, n_local, nnz, s->rp, 0, &s->plan)) {
        cudamat_solver_destroy(s);
        return rc2;
    }
    stamp("CSR launch plan (refine)");
    if (int rc3 = plan_spmv_compress(st, n_local, nnz, s->rp, s->ci, &s->plan)) {
        cudamat_solver_destroy(s);
        return rc3;
    }
    stamp("compressed-index attempt");
    if (s->plan.c_off16) {          // compressed stream kernel: 8-bit value indices too when the matrix has a dictionary
        if (int rc4 = ensure_valdict(s)) {
            cudamat_solver_destroy(s);
            return rc4;
        }
        if (s->vd.n > 0) {
            if (int rc5 = plan_spmv_dict(st, n_local, nnz, s->rp, s->vd.idx, s->vd.dict, &s->plan)) {
                cudamat_solver_destroy(s);
                return rc5;
            }
        }
        if (!s->plan.d_pbase) {     // fp64 values: line-aligned copies of the two entry streams
            if (int rc6 = plan_spmv_align(st, n_local, nnz, s->rp, s->val, &s->plan)) {
                cudamat_solver_destroy(s);
                return rc6;
            }
        }
    }
    s->t_create = now_s() - t_create0;
    *out = s;
    return CUDAMAT_OK;
}

extern "C" int cudamat_solver_destroy(cudamat_solver *s)
{
    if (!s) return CUDAMAT_OK;
    hipSetDevice(s->ctx->device);
    hipStreamSynchronize(s->ctx->stream);
    ilu0_release(s);
    pb_free(&s->pb);
    sell_free(&s->sell);
    free_work(s);
    plan_spmv_free(&s->plan);
    void *ptrs[] = {s->rp, s->ci, s->val, s->parts_full, s->parts_rv, s->parts_half, s->parts_tt,
                    s->red, s->st, s->hist};
    for (void *p : ptrs)
        if (p) hipFree(p);
    if (s->st_ring) hipHostFree(s->st_ring);
    if (s->snap_host) hipHostFree(s->snap_host);
    for (int i = 0; i < kRing; i++)
        if (s->ev[i]) hipEventDestroy(s->ev[i]);
    for (hipEvent_t e : s->prof_ev) hipEventDestroy(e);
    for (hipEvent_t e : s->comm_ev) hipEventDestroy(e);
    if (s->need_dev) hipFree(s->need_dev);
    if (s->bar) hipFree(s->bar);
    valdict_free(&s->vd);
    for (int e = 0; e < 2; e++) {
        if (s->ev_red[e]) hipEventDestroy(s->ev_red[e]);
        if (s->ev_red_done[e]) hipEventDestroy(s->ev_red_done[e]);
    }
    if (s->ev_x) hipEventDestroy(s->ev_x);
    for (hipEvent_t e : s->ev_part)
        if (e) hipEventDestroy(e);
    for (hipEvent_t e : s->ev_p1)
        if (e) hipEventDestroy(e);
    for (hipStream_t q : s->part_stream)
        if (q) { hipStreamSynchronize(q); hipStreamDestroy(q); }
    delete s;
    return CUDAMAT_OK;
}

extern "C" int cudamat_solver_set_shift(cudamat_solver *s, const double *d)
{
    CM_ARG(s, "solver is NULL");
    s->d = d;
    return CUDAMAT_OK;
}

extern "C" int cudamat_solver_set_comm(cudamat_solver *s, const cudamat_comm *comm)
{
    CM_ARG(s, "solver is NULL");
    hipStreamSynchronize(s->ctx->stream);
    free_work(s);
    pb_free(&s->pb);
    sell_free(&s->sell);
    ilu0_release(s);           // factors belong to the old partition
    s->spmv_mode = -1;
    s->overlap = false;
    s->agreed = false;
    s->windowed = false;
    s->windows_known = false;
    if (s->need_dev) { hipFree(s->need_dev); s->need_dev = nullptr; }
    const char *force = getenv("CUDAMAT_FORCE_SHARDED");
    const bool forced = comm && comm->world == 1 && force && force[0] == '1';
    if (!comm || (comm->world <= 1 && !forced)) {
        s->sharded = false;
        s->n_pad = s->n;
        return CUDAMAT_OK;
    }
    CM_ARG(comm->allgather && comm->allreduce, "collectives missing");
    CM_ARG(comm->rank >= 0 && comm->rank < comm->world, "rank");
    const int64_t per = (s->n_cols + comm->world - 1) / comm->world;
    int64_t mine = s->n_cols - per * comm->rank;
    if (mine > per) mine = per;
    if (mine < 0) mine = 0;
    if (mine != s->n) {
        set_error("row block mismatch: rank %d of %d must own %lld rows of %lld (uniform blocks of %lld), has %d",
                  comm->rank, comm->world, (long long)mine, (long long)s->n_cols, (long long)per, s->n);
        return CUDAMAT_ERR_ARG;
    }
    s->comm = *comm;
    s->sharded = true;
    s->n_pad = (int)per;
    if (comm->gather_window && comm->world > 1 && comm->world <= 4096)      // compute_windows' exchange buffer: allocated
        CM_TRY(dev_alloc((void **)&s->need_dev, sizeof(double) * ((size_t)2 * comm->world + (size_t)2 * comm->world * comm->world)));  // here, where a failure is still rank-local
    if (comm->gather_part && comm->comm_stream && !s->ev_x) {
        CM_HIP(hipEventCreateWithFlags(&s->ev_x, hipEventDisableTiming));
        for (int c = 0; c < kPbMaxChunks; c++) {
            CM_HIP(hipEventCreateWithFlags(&s->ev_part[c], hipEventDisableTiming));
            CM_HIP(hipEventCreateWithFlags(&s->ev_p1[c], hipEventDisableTiming));
            CM_HIP(hipStreamCreateWithFlags(&s->part_stream[c], hipStreamNonBlocking));
        }
    }
    if (const char *ch = getenv("CUDAMAT_OVERLAP_CHUNKS")) {
        const int v = atoi(ch);
        if (v >= 1 && v <= kPbMaxChunks) s->overlap_chunks = v;
    }
    return CUDAMAT_OK;
}

// profiling of the exchanges: one (start, stop) pair per call, pooled per solver
static hipEvent_t comm_event(cudamat_solver *s)
{
    if (s->comm_used == s->comm_ev.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        s->comm_ev.push_back(e);
    }
    return s->comm_ev[s->comm_used++];
}
static void comm_mark_begin(cudamat_solver *s, int kind, hipStream_t st)
{
    if (!s->profiling) return;
    s->comm_kind.push_back(kind);
    hipEventRecord(comm_event(s), st);
}
static void comm_mark_end(cudamat_solver *s, hipStream_t st)
{
    if (s->profiling) hipEventRecord(comm_event(s), st);
}

// y = (A + diag d) x with x a LOCAL n_pad-long work vector (pad zero); gathers first
// when sharded.  dot/check as in SpmvArgs.
static int spmv_local(cudamat_solver *s, const double *x_local, double *y, int dot, const double *w,
                      double *parts, LoopArgs la, int check, ScalarSrc half)
{
    Range range_spmv(s->sharded ? "cudamat: SpMV + exchange of its input" : "cudamat: SpMV");
    const double *xfull = x_local;
    const bool windowed = s->sharded && s->windowed;
    const bool overlapped = s->sharded && !windowed && s->overlap && s->spmv_mode == 1;
    if (windowed) {
        // only the parts of the other slices that this rank's rows reference travel (a halo for banded matrices)
        hipStream_t st = s->ctx->stream;
        CM_HIP(hipMemcpyAsync(s->gather + (size_t)s->comm.rank * (size_t)s->n_pad, x_local, sizeof(double) * (size_t)s->n_pad,
                              hipMemcpyDeviceToDevice, st));
        comm_mark_begin(s, 2, st);
        if (s->comm.gather_window(s->comm.user, x_local, s->gather, (int64_t)s->n_pad, s->w_send_off.data(), s->w_send_cnt.data(),
                                  s->w_recv_off.data(), s->w_recv_cnt.data()) != 0) {
            set_error("gather_window callback failed");
            return CUDAMAT_ERR_COMM;
        }
        comm_mark_end(s, st);
    } else if (s->sharded && !overlapped) {
        comm_mark_begin(s, 2, s->ctx->stream);
        if (s->comm.allgather(s->comm.user, x_local, s->gather, (int64_t)s->n_pad) != 0) {
            set_error("allgather callback failed");
            return CUDAMAT_ERR_COMM;
        }
        comm_mark_end(s, s->ctx->stream);
    }
    if (s->sharded) xfull = s->gather;
    SpmvArgs a{};
    a.n = s->n;
    a.rp = s->rp;
    a.ci = s->ci;
    a.val = s->val;
    a.x = xfull;
    a.d = s->d;
    a.xd = x_local;
    a.alpha = 1.0;
    a.beta = 0.0;
    a.y = y;
    a.dot = dot;
    a.w = w;
    a.parts = parts;
    a.loop = la;
    a.check = check;
    a.half = half;
    if (overlapped) {
        // The gather in pieces on the communicator's stream, phase 1 piece by piece behind it:
        //   comm stream  :  [wait x ready] piece 0 | piece 1 | ...
        //   part stream c:  [wait x ready, piece c] phase 1 of the blocks of piece c          (c = 0 .. pieces-1)
        //   our stream   :  own slice -> gather buffer, test, phase 1 (local blocks) | wait for every part | phase 2
        // One stream per piece, because a piece's launch alone (and the local slice's: 1/world of the blocks) does
        // not fill the GPU: back to back on one stream they would cost a round of workgroups each (5 rounds instead
        // of 3 at 8 ranks and 4 pieces).  The products do not depend on the order of the phase-1 launches and
        // phase 2 adds them in column order as always, so the result is bit-identical to the plain gather + SpMV.
        // Hazards: every stream waits for `ev_x`, recorded on our stream behind the previous SpMV (the product and
        // gather buffers are free by then); our stream has waited for every part, hence for every piece (= all of
        // this rank's sends), before anything may overwrite x_local.
        hipStream_t st = s->ctx->stream, cst = (hipStream_t)s->comm.comm_stream;
        const PbPlan &p = s->pb;
        CM_HIP(hipEventRecord(s->ev_x, st));
        CM_HIP(hipStreamWaitEvent(cst, s->ev_x, 0));
        comm_mark_begin(s, 0, cst);
        for (int c = 0; c < p.chunks; c++) {
            const int64_t off = (int64_t)c * p.chunk_len;
            int64_t cnt = (int64_t)s->n_pad - off;
            if (cnt > p.chunk_len) cnt = p.chunk_len;
            if (cnt > 0 && s->comm.gather_part(s->comm.user, x_local, s->gather, (int64_t)s->n_pad, off, cnt) != 0) {
                set_error("gather_part callback failed");
                return CUDAMAT_ERR_COMM;
            }
            CM_HIP(hipEventRecord(s->ev_part[c], cst));
        }
        comm_mark_end(s, cst);
        CM_HIP(hipMemcpyAsync(s->gather + (size_t)s->comm.rank * (size_t)s->n_pad, x_local, sizeof(double) * (size_t)s->n_pad,
                              hipMemcpyDeviceToDevice, st));
        CM_TRY(launch_pb_check(st, a));
        CM_TRY(launch_pb_phase1(st, p, a, 0));
        for (int c = 0; c < p.chunks; c++) {
            hipStream_t ps = s->part_stream[c];
            CM_HIP(hipStreamWaitEvent(ps, s->ev_x, 0));
            CM_HIP(hipStreamWaitEvent(ps, s->ev_part[c], 0));
            CM_TRY(launch_pb_phase1(ps, p, a, 1 + c));
            CM_HIP(hipEventRecord(s->ev_p1[c], ps));
        }
        for (int c = 0; c < p.chunks; c++) CM_HIP(hipStreamWaitEvent(st, s->ev_p1[c], 0));
        return launch_pb_phase2(st, p, a);
    }
    if (s->perm_active) return launch_spmv_pb(s->ctx->stream, s->pb_perm, a);      // rows in L's space, columns in U's
    if (s->spmv_mode == 1) return launch_spmv_pb(s->ctx->stream, s->pb, a);
    if (s->spmv_mode == 2) return launch_spmv_sell(s->ctx->stream, s->sell, a);
    return launch_spmv(s->ctx->stream, s->plan, a);
}

// the matrix's value dictionary (valdict.hip), looked for once: large systems only (small ones live in caches and the
// small-system loops keep their values in registers)
static int ensure_valdict(cudamat_solver *s)
{
    if (s->vd_tried) return CUDAMAT_OK;
    s->vd_tried = true;
    if (s->nnz < (1 << 20)) return CUDAMAT_OK;
    return valdict_build(s->ctx->stream, s->nnz, s->val, &s->vd);
}

// number of per-workgroup partial sums an SpMV launch leaves in `parts`
static int spmv_parts(const cudamat_solver *s)
{
    if (s->perm_active) return s->pb_perm.NRB;
    return s->spmv_mode == 1 ? s->pb.NRB : s->spmv_mode == 2 ? s->sell.grid : plan_spmv_parts(s->plan);
}

// How scattered are a row's columns?  Mean of (last - first column) over <= 4096 evenly spaced rows (sorted rows: the
// two ends of a row are its extremes).  One tiny launch; decides whether candidates that cannot win are timed at all.
__global__ __launch_bounds__(kBlock) void k_col_span(int n, const int *rp, const int *ci, int samples, unsigned long long *sum, int *cnt)
{
    const int q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= samples) return;
    const int row = (int)((long long)q * n / samples);
    const int s = rp[row], e = rp[row + 1];
    if (e - s < 2) return;
    atomicAdd(sum, (unsigned long long)(ci[e - 1] - ci[s]));
    atomicAdd(cnt, 1);
}

static int col_span_bytes(cudamat_solver *s, double *out)
{
    *out = 0.0;
    hipStream_t st = s->ctx->stream;
    unsigned long long *d = nullptr, h[2] = {0ULL, 0ULL};
    CM_HIP(hipMalloc((void **)&d, sizeof(h)));
    hipMemsetAsync(d, 0, sizeof(h), st);
    const int samples = s->n < 4096 ? s->n : 4096;
    hipLaunchKernelGGL(k_col_span, dim3((samples + kBlock - 1) / kBlock), dim3(kBlock), 0, st, s->n, s->rp, s->ci, samples, d, (int *)(d + 1));
    hipError_t e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d);
    CM_HIP(e);
    const int cnt = (int)(h[1] & 0xffffffffULL);
    *out = cnt > 0 ? 8.0 * (double)h[0] / cnt : 0.0;
    return CUDAMAT_OK;
}

// Choose the SpMV implementation for this matrix (once), by TIMING the candidates on this device:
//   0  the CSR forms (lanes per row / stream tiles / nnz-balanced tiles, plan_spmv_refine) -- always a candidate;
//   1  the blocked two-phase kernels, when the columns are scattered over a vector far larger than L2 (pb_candidate);
//   2  SELL-C-sigma, for rows of 8 entries and more whose padded copy stays below 1.5 x the entries (banded
//      matrices: 2-2.7 x the lanes-per-row kernel; short rows belong to the stream kernel).
// CUDAMAT_SPMV_MODE=csr|pb|sell overrides.
// Candidates that cannot win are not timed (round 3; the drop-in entry points pay this on every call): when a row's
// columns span far more than the L2s hold (mean span >= 16 MB of x; C4: 77 MB) and rows have >= 8 entries, every gather
// of the lanes-per-row kernel and of SELL misses L2 -- measured 9.8 / 9.1 ms against 2.8 ms blocked at C4, the same
// ratio on every scattered matrix of DESIGN section 4 -- so the blocked copy is selected without building the SELL copy
// or timing anything (a sharded solver still times the blocked form alone: ms_spmv_alone feeds the exposed-gather
// figure).  CUDAMAT_SPMV_TUNE=full restores the timing of every candidate.
static int ensure_spmv_mode_inner(cudamat_solver *s);
static int ensure_spmv_mode(cudamat_solver *s)
{
    if (s->spmv_mode >= 0) return CUDAMAT_OK;
    Range range_mode("cudamat: SpMV form (matrix copies, tuning)");
    const double t0 = now_s();
    const int rc = ensure_spmv_mode_inner(s);
    hipStreamSynchronize(s->ctx->stream);
    s->t_spmv_setup = now_s() - t0;
    if (getenv("CUDAMAT_VERBOSE"))
        fprintf(stderr, "[cudamat] SpMV form %d chosen in %.3f ms (blocked copy %.3f ms, timing %.3f ms)\n", s->spmv_mode,
                s->t_spmv_setup * 1e3, s->pb.build_seconds * 1e3, s->t_spmv_timing * 1e3);
    return rc;
}

static int ensure_spmv_mode_inner(cudamat_solver *s)
{
    hipStream_t st = s->ctx->stream;
    const char *env = getenv("CUDAMAT_SPMV_MODE");
    const bool force_csr = env && !strcmp(env, "csr");
    const bool force_pb = env && !strcmp(env, "pb");
    const bool force_sell = env && !strcmp(env, "sell");
    s->spmv_mode = 0;
    if (force_csr || s->n == 0 || s->nnz == 0) return CUDAMAT_OK;
    bool have[3] = {true, false, false};
    // ---- blocked two-phase copy
    if (!force_sell) {
        if (!s->cols_sorted) {    // the blocked builder ranks entries by runs of equal column block: needs sorted rows
            if (force_pb) { set_error("the blocked SpMV needs rows with increasing column indices"); return CUDAMAT_ERR_ARG; }
        } else if (force_pb || pb_candidate(st, s->n, s->n_cols, s->nnz, s->rp, s->ci)) {
            // a sharded solver cuts the column blocks at the slices (and, for an overlapped gather, the pieces) of the
            // gathered vector; the gather buffer's index IS the column id (uniform slices of n_pad)
            PbCols cols;
            const bool can_overlap = s->sharded && s->comm.world > 1 && s->comm.gather_part && s->comm.comm_stream;
            if (s->sharded && s->comm.world > 1) {
                cols.per = s->n_pad;
                cols.rank = s->comm.rank;
                cols.chunks = can_overlap ? s->overlap_chunks : 1;
            }
            CM_TRY(ensure_valdict(s));
            const int rc = pb_build(st, s->n, s->n_cols, s->nnz, s->rp, s->ci, s->val, &s->pb, &cols, &s->vd);
            if (rc != CUDAMAT_OK && force_pb) return rc;
            have[1] = rc == CUDAMAT_OK;          // e.g. out of memory for the blocked copy: keep the others
        }
        if (force_pb && !s->sharded) { s->spmv_mode = 1; return CUDAMAT_OK; }
        if (force_pb) { have[0] = false; }      // sharded: still time it (ms_spmv_alone feeds the exposed-gather figure)
    }
    // scattered columns: the gather-based forms cannot win (see above)
    bool scattered = false;
    {
        const char *tune = getenv("CUDAMAT_SPMV_TUNE");
        if (have[1] && !force_pb && !force_sell && !(tune && !strcmp(tune, "full")) && s->nnz >= 8 * (int64_t)s->n) {
            CM_TRY(col_span_bytes(s, &s->col_span_bytes));
            scattered = s->col_span_bytes >= 16.0 * 1024 * 1024;
        }
        if (scattered) {
            have[0] = false;
            if (!s->sharded) {
                s->spmv_mode = 1;
                if (getenv("CUDAMAT_VERBOSE"))
                    fprintf(stderr, "cudamat: SpMV: a row's columns span %.1f MB of x on average -> blocked, nothing timed\n", s->col_span_bytes / 1048576.0);
                return CUDAMAT_OK;
            }
        }
    }
    // ---- SELL-C-sigma copy: rows of 8 entries and more (shorter rows belong to the stream kernel, which measures
    // faster there: C3 0.174 vs 0.191 ms) whose padded copy stays below 1.5 x the entries.  Measured on 2e6-row banded
    // matrices (scripts/sell_probe.py): row lengths 14..70 0.46 ms vs 0.94 (CSR forms) / 0.66 (blocked); 13..20 0.18
    // vs 0.40 / 0.26; with scattered columns the blocked form wins (0.48 vs 1.07) -- hence: time them.
    const char *se = getenv("CUDAMAT_SPMV_SELL");
    const bool sell_off = se && se[0] == '0';
    if (force_sell || (!force_pb && !scattered && !sell_off && s->n >= 4096 && s->nnz >= (1 << 16) && s->plan.stream_rows == 0 && s->nnz >= 8 * (int64_t)s->n)) {
        const int rc = sell_build(st, s->n, s->nnz, s->rp, s->ci, s->val, &s->sell, force_sell ? 0.0 : 1.5);
        if (rc != CUDAMAT_OK && force_sell) return rc;
        have[2] = rc == CUDAMAT_OK;
        if (force_sell) { s->spmv_mode = 2; return CUDAMAT_OK; }
    }
    if (!have[1] && !have[2]) return CUDAMAT_OK;
    if (force_pb && !have[1]) { set_error("the blocked copy could not be built"); return CUDAMAT_ERR_NOMEM; }
    if (force_pb) have[2] = false;
    const LoopArgs la_none{nullptr, nullptr, 0, 0, 0};
    const ScalarSrc nosrc{nullptr, 0, 1};
    const double *xin = s->sharded ? s->gather : s->p;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms[3] = {0.f, 0.f, 0.f};
    int rc = CUDAMAT_OK;
    const double t_timing0 = now_s();
    for (int mode = 0; mode < 3 && rc == CUDAMAT_OK; mode++) {
        if (!have[mode]) continue;
        SpmvArgs a{};
        a.n = s->n; a.rp = s->rp; a.ci = s->ci; a.val = s->val; a.x = xin; a.d = nullptr; a.xd = s->p;
        a.alpha = 1.0; a.beta = 0.0; a.y = s->v; a.dot = 0; a.loop = la_none; a.check = CHECK_NONE; a.half = nosrc;
        for (int rep = 0; rep < 3 && rc == CUDAMAT_OK; rep++) {
            if (rep == 1) hipEventRecord(e0, st);
            rc = mode == 1 ? launch_spmv_pb(st, s->pb, a) : mode == 2 ? launch_spmv_sell(st, s->sell, a) : launch_spmv(st, s->plan, a);
        }
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mode], e0, e1);
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    s->t_spmv_timing = now_s() - t_timing0;
    CM_HIP(hipMemsetAsync(s->v, 0, sizeof(double) * (size_t)(s->n_pad > 0 ? s->n_pad : 1), st));
    s->ms_csr = ms[0] / 2;
    s->ms_pb = ms[1] / 2;
    s->ms_sell = ms[2] / 2;
    int best = have[0] ? 0 : 1;
    if (rc == CUDAMAT_OK)
        for (int mode = 1; mode < 3; mode++)
            if (have[mode] && (!have[best] || ms[mode] < ms[best])) best = mode;
    s->spmv_mode = best;
    s->ms_spmv_alone = ms[best] / 2;
    const double sell_fill = s->sell.fill;
    if (best != 1) pb_free(&s->pb);
    if (best != 2) sell_free(&s->sell);
    if (getenv("CUDAMAT_VERBOSE"))
        fprintf(stderr, "cudamat: SpMV auto-tune csr %.3f ms, blocked %s%.3f ms, sell %s%.3f ms (fill %.2f) -> %s\n", s->ms_csr,
                have[1] ? "" : "(n/a) ", s->ms_pb, have[2] ? "" : "(n/a) ", s->ms_sell, sell_fill,
                best == 1 ? "blocked" : best == 2 ? "sell" : "csr");
    return CUDAMAT_OK;
}

static int allreduce(cudamat_solver *s, double *buf, int count)
{
    Range range_ar("cudamat: all-reduce");
    // fault injection for the tests of the failure paths: CUDAMAT_TEST_COMM_FAIL="rank:k" makes the k-th all-reduce
    // of that rank's solver report an error (tests/test_gpu_dist.py: a failing rank must not strand its peers)
    if (const char *inj = getenv("CUDAMAT_TEST_COMM_FAIL")) {
        int r = -1, k = -1;
        if (sscanf(inj, "%d:%d", &r, &k) == 2 && r == s->comm.rank && ++s->test_allreduces == k) {
            set_error("injected all-reduce failure (CUDAMAT_TEST_COMM_FAIL=%s)", inj);
            return CUDAMAT_ERR_COMM;
        }
    }
    comm_mark_begin(s, 3, s->ctx->stream);
    if (s->comm.allreduce(s->comm.user, buf, count) != 0) {
        set_error("allreduce callback failed");
        return CUDAMAT_ERR_COMM;
    }
    comm_mark_end(s, s->ctx->stream);
    return CUDAMAT_OK;
}

// ---- windowed gather (halo): which part of every other rank's slice do the local rows reference?
// lo[q] / hi[q]: smallest / one past the largest column of slice q (relative to the slice) among the local entries.
// Each thread walks a contiguous run of entries and only touches the workgroup's LDS tables when an entry leaves
// the range it has already reported for that slice (rows are mostly sorted, so that is rare).
__global__ __launch_bounds__(kBlock) void k_col_windows(long long nnz, const int *ci, int per, int world, int *lo, int *hi)
{
    extern __shared__ int tab[];              // lo_s[world], hi_s[world]
    int *lo_s = tab, *hi_s = tab + world;
    for (int q = threadIdx.x; q < world; q += kBlock) { lo_s[q] = per; hi_s[q] = 0; }
    __syncthreads();
    const long long per_thread = (nnz + (long long)gridDim.x * kBlock - 1) / ((long long)gridDim.x * kBlock);
    const long long k0 = ((long long)blockIdx.x * kBlock + threadIdx.x) * per_thread;
    const long long k1 = k0 + per_thread < nnz ? k0 + per_thread : nnz;
    int cq = -1, clo = 0, chi = 0;
    for (long long k = k0; k < k1; k++) {
        const int c = ci[k], q = c / per, w = c - q * per;
        if (q == cq && w >= clo && w < chi) continue;
        if (q != cq) { cq = q; clo = w; chi = w + 1; }
        else { clo = w < clo ? w : clo; chi = w + 1 > chi ? w + 1 : chi; }
        atomicMin(&lo_s[q], w);
        atomicMax(&hi_s[q], w + 1);
    }
    __syncthreads();
    for (int q = threadIdx.x; q < world; q += kBlock) {
        if (hi_s[q] > 0) { atomicMin(&lo[q], lo_s[q]); atomicMax(&hi[q], hi_s[q]); }
    }
}

// collective: every rank learns what every rank needs from every slice; decides (identically everywhere) whether
// the windows replace the whole gather.  Every rank-local step comes BEFORE the all-gather and its outcome travels IN
// the payload (a negative entry = "this rank failed"), so a rank whose local work failed still enters the collective
// and every rank returns an error afterwards -- nobody is left alone inside it.
static int compute_windows(cudamat_solver *s)
{
    s->windows_known = true;
    s->windowed = false;
    s->gather_fraction = 1.0;
    const int W = s->comm.world, me = s->comm.rank;
    if (!s->comm.gather_window || W <= 1 || W > 4096) return CUDAMAT_OK;
    hipStream_t st = s->ctx->stream;
    const int per = s->n_pad;
    std::vector<int> h((size_t)2 * W);
    std::vector<double> mine((size_t)2 * W, 0.0), all((size_t)2 * W * W, 0.0);
    char saved[512] = "";
    auto local = [&]() -> int {
        int *d_lohi = nullptr;
        CM_HIP(hipMalloc((void **)&d_lohi, sizeof(int) * 2 * (size_t)W));
        for (int q = 0; q < W; q++) { h[(size_t)q] = per; h[(size_t)W + q] = 0; }
        hipError_t e = hipMemcpyAsync(d_lohi, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess && s->nnz > 0) {
            int grid = (int)((s->nnz + 4096LL * kBlock - 1) / (4096LL * kBlock));
            grid = grid < 1 ? 1 : grid > 4096 ? 4096 : grid;
            hipLaunchKernelGGL(k_col_windows, dim3(grid), dim3(kBlock), sizeof(int) * 2 * (size_t)W, st, (long long)s->nnz, s->ci, per, W,
                               d_lohi, d_lohi + W);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_lohi, sizeof(int) * h.size(), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        hipFree(d_lohi);
        CM_HIP(e);
        for (int q = 0; q < W; q++)
            if (q != me && h[(size_t)W + q] > h[(size_t)q]) { mine[(size_t)2 * q] = h[(size_t)q]; mine[(size_t)2 * q + 1] = h[(size_t)W + q]; }
        return CUDAMAT_OK;
    };
    const int rc_local = local();
    if (rc_local != CUDAMAT_OK) {
        snprintf(saved, sizeof(saved), "%s", cudamat_last_error());
        mine.assign((size_t)2 * W, -1.0);
    }
    // (need_dev was allocated by cudamat_solver_set_comm: no allocation can fail between here and the collective)
    CM_HIP(hipMemcpyAsync(s->need_dev, mine.data(), sizeof(double) * mine.size(), hipMemcpyHostToDevice, st));
    CM_HIP(hipStreamSynchronize(st));
    if (s->comm.allgather(s->comm.user, s->need_dev, s->need_dev + 2 * W, (int64_t)2 * W) != 0) {
        set_error("allgather callback failed");
        return CUDAMAT_ERR_COMM;
    }
    CM_HIP(hipMemcpyAsync(all.data(), s->need_dev + 2 * W, sizeof(double) * all.size(), hipMemcpyDeviceToHost, st));
    CM_HIP(hipStreamSynchronize(st));
    if (rc_local != CUDAMAT_OK) { set_error("%s", saved); return rc_local; }
    for (double v : all)
        if (v < 0.0) { set_error("another rank failed while it looked for its column windows"); return CUDAMAT_ERR_COMM; }
    s->w_send_off.assign((size_t)W, 0); s->w_send_cnt.assign((size_t)W, 0);
    s->w_recv_off.assign((size_t)W, 0); s->w_recv_cnt.assign((size_t)W, 0);
    double worst = 0.0;
    for (int p = 0; p < W; p++) {             // p needs [lo, hi) of slice q
        double tot = 0.0;
        for (int q = 0; q < W; q++) {
            const double lo = all[((size_t)p * W + q) * 2], hi = all[((size_t)p * W + q) * 2 + 1];
            if (q == p || hi <= lo) continue;
            tot += hi - lo;
            if (p == me) { s->w_recv_off[(size_t)q] = (int64_t)lo; s->w_recv_cnt[(size_t)q] = (int64_t)(hi - lo); }
            if (q == me) { s->w_send_off[(size_t)p] = (int64_t)lo; s->w_send_cnt[(size_t)p] = (int64_t)(hi - lo); }
        }
        const double frac = tot / ((double)(W - 1) * (double)per);
        if (p == me) s->gather_fraction = frac;
        worst = frac > worst ? frac : worst;
    }
    const char *env = getenv("CUDAMAT_WINDOWED");
    s->windowed = worst <= 0.5 && !(env && env[0] == '0');
    if (!s->windowed) s->gather_fraction = 1.0;
    if (getenv("CUDAMAT_VERBOSE"))
        fprintf(stderr, "cudamat: rank %d references %.4f of the other slices (worst rank %.4f) -> %s\n", me,
                s->windowed ? s->gather_fraction : worst, worst, s->windowed ? "windowed gather" : "whole gather");
    return CUDAMAT_OK;
}

// Sharded runs: the ranks agree on the outcome of the rank-local setup steps BEFORE the first collective of the
// data path -- a rank whose blocked copy, ILU(0) or allocation failed makes every rank return an error (instead of
// leaving its peers inside a collective for ever), and the gather is overlapped only if every rank runs the
// blocked SpMV (the pieces are exchanged by a different call sequence than the plain all-gather).
static int setup_agree(cudamat_solver *s, int rc_local)
{
    if (!s->sharded) return rc_local;
    char saved[512];
    snprintf(saved, sizeof(saved), "%s", cudamat_last_error());
    hipStream_t st = s->ctx->stream;
    double h[2] = {rc_local != CUDAMAT_OK ? 1.0 : 0.0, s->spmv_mode == 1 ? 1.0 : 0.0};
    // (the only rank-local step in front of the collective is this 16-byte upload; if it fails the device itself is
    // gone and the all-reduce below could not run either -- the host must then abort the communicator,
    // cudamat_rccl_comm_abort, as csrc/sharded.cpp does for every failed rank)
    CM_HIP(hipMemcpyAsync(s->red + 10, h, sizeof(h), hipMemcpyHostToDevice, st));
    CM_HIP(hipStreamSynchronize(st));                       // (h is a stack buffer)
    CM_TRY(allreduce(s, s->red + 10, 2));
    CM_HIP(hipMemcpyAsync(h, s->red + 10, sizeof(h), hipMemcpyDeviceToHost, st));
    CM_HIP(hipStreamSynchronize(st));
    const char *ov = getenv("CUDAMAT_OVERLAP");
    s->overlap = s->comm.world > 1 && s->comm.gather_part && s->comm.comm_stream && s->ev_x && s->pb.chunks >= 1 &&
                 h[1] == (double)s->comm.world && !(ov && ov[0] == '0');
    s->agreed = true;
    if (h[0] == 0.0 && !s->windows_known) CM_TRY(compute_windows(s));       // collective: same call on every rank
    if (h[0] != 0.0) {
        if (rc_local != CUDAMAT_OK) { set_error("%s", saved); return rc_local; }
        set_error("%d rank(s) of the sharded solver failed during setup", (int)h[0]);
        return CUDAMAT_ERR_COMM;
    }
    return CUDAMAT_OK;
}

extern "C" int cudamat_solver_spmv_mode(cudamat_solver *s, int *mode)
{
    CM_ARG(s && mode, "null pointer");
    CM_HIP(hipSetDevice(s->ctx->device));
    CM_TRY(ensure_work(s));
    CM_TRY(ensure_spmv_mode(s));
    *mode = s->spmv_mode;
    return CUDAMAT_OK;
}

extern "C" int cudamat_solver_value_dict(cudamat_solver *s, int *distinct)
{
    CM_ARG(s && distinct, "null pointer");
    CM_HIP(hipSetDevice(s->ctx->device));
    CM_TRY(ensure_work(s));
    CM_TRY(ensure_spmv_mode(s));
    *distinct = 0;
    if (s->spmv_mode == 1 && s->pb.pvi) *distinct = s->pb.ndict;
    else if (s->spmv_mode == 0 && s->plan.d_pbase) *distinct = s->vd.n;
    return CUDAMAT_OK;
}

extern "C" int cudamat_solver_spmv_kernel(cudamat_solver *s, char *name, int cap)
{
    CM_ARG(s && name && cap > 0, "null pointer");
    CM_HIP(hipSetDevice(s->ctx->device));
    CM_TRY(ensure_work(s));
    CM_TRY(ensure_spmv_mode(s));
    const SpmvPlan &p = s->plan;
    if (s->spmv_mode == 1) snprintf(name, (size_t)cap, "%s + k_pb_phase2", s->pb.pvi ? "k_pb_phase1_dict" : "k_pb_phase1");
    else if (s->spmv_mode == 2) snprintf(name, (size_t)cap, "k_spmv_sell");
    else if (p.tiles > 0) snprintf(name, (size_t)cap, "k_spmv_tiles");
    else if (p.stream_rows && p.c_off16) snprintf(name, (size_t)cap, "%s<%d>", p.d_pbase ? "k_spmv_stream_d" : "k_spmv_stream_c", p.stream_rows);
    else if (p.stream_rows) snprintf(name, (size_t)cap, "k_spmv_stream<%d>", p.stream_rows);
    else snprintf(name, (size_t)cap, "k_spmv<%d>", p.lanes);
    return CUDAMAT_OK;
}

extern "C" int cudamat_solver_spmv(cudamat_solver *s, const double *x_local, double *y_local)
{
    CM_ARG(s && x_local && y_local, "null pointer");
    CM_HIP(hipSetDevice(s->ctx->device));
    {
        int rc_setup = ensure_work(s);
        if (rc_setup == CUDAMAT_OK) rc_setup = ensure_spmv_mode(s);
        if (s->sharded && !s->agreed) rc_setup = setup_agree(s, rc_setup);
        CM_TRY(rc_setup);
    }
    const double *xin = x_local;
    if (s->sharded) {   // the gather needs n_pad entries with a zero pad
        CM_HIP(hipMemcpyAsync(s->pw, x_local, sizeof(double) * (size_t)s->n, hipMemcpyDeviceToDevice,
                              s->ctx->stream));
        xin = s->pw;
    }
    return spmv_local(s, xin, y_local, 0, nullptr, nullptr, LoopArgs{nullptr, nullptr, 0, 0, 0},
                      CHECK_NONE, ScalarSrc{nullptr, 0, 1});
}

static hipEvent_t prof_event(cudamat_solver *s, size_t i)
{
    while (s->prof_ev.size() <= i) {
        hipEvent_t e;
        hipEventCreate(&e);
        s->prof_ev.push_back(e);
    }
    return s->prof_ev[i];
}

// native: `in` is in L's level-major space and `out` leaves in U's (the loop that runs in those spaces); otherwise both
// are in the caller's row numbering (level-major factors then permute on the way in and out)
static int precond_apply(cudamat_solver *s, const double *in, double *tmp, double *out, bool native = false)
{
    if (s->L.lm && !native) return precond_apply_original(s, in, tmp, out);
    CM_TRY(trsv_apply(s, s->L, false, in, tmp));    // pbicgstab.cu:92-94 / :121-123
    CM_TRY(trsv_apply(s, s->U, true, tmp, out));    // pbicgstab.cu:96-98 / :125-127
    return CUDAMAT_OK;
}

extern "C" int cudamat_solver_precond_apply(cudamat_solver *s, const double *in, double *out)
{
    CM_ARG(s && in && out, "null pointer");
    CM_ARG(s->has_ilu, "call cudamat_solver_ilu0 / cudamat_solver_block_ilu0 first");
    CM_HIP(hipSetDevice(s->ctx->device));
    CM_TRY(ensure_work(s));
    return precond_apply(s, in, s->t, out);
}

static int solve_once(cudamat_solver *s, const double *b, double *x, int precond, int loop, int maxit, double tol,
                      int flags, cudamat_stats *out, bool *precond_gave_up, bool *resident_gave_up, double abs_tol)
{
    CM_ARG(s && b && x, "null pointer");
    CM_ARG(precond == CUDAMAT_PRECOND_NONE || precond == CUDAMAT_PRECOND_ILU0 || precond == CUDAMAT_PRECOND_BLOCK_ILU0,
           "precond");
    CM_ARG(loop == CUDAMAT_LOOP_PBICGSTAB || loop == CUDAMAT_LOOP_PBICGSTAB2 || loop == CUDAMAT_LOOP_PIPELINED, "loop");
    CM_ARG(maxit >= 0, "maxit");
    CM_ARG(!(precond == CUDAMAT_PRECOND_ILU0 && s->sharded),
           "ILU(0) of the whole matrix is single-GPU only (SURVEY 8e); sharded runs take CUDAMAT_PRECOND_BLOCK_ILU0");
    CM_ARG(!(precond && s->d), "the (A0 + I d) variant has no preconditioner (pbicgstab.h:110)");
    CM_HIP(hipSetDevice(s->ctx->device));
    const double t_begin = now_s();
    hipStream_t st = s->ctx->stream;
    {
        int rc_setup = ensure_work(s);
        if (rc_setup == CUDAMAT_OK) rc_setup = ensure_spmv_mode(s);
        if (rc_setup == CUDAMAT_OK && precond && (!s->has_ilu || (s->sharded && !s->ilu_block)))
            rc_setup = ilu0_setup(s, precond == CUDAMAT_PRECOND_BLOCK_ILU0);
        CM_TRY(setup_agree(s, rc_setup));        // sharded: every rank learns of a failure on any rank
    }
    // The loop in LEVEL-MAJOR SPACES (round 3).  With the hybrid triangular solves the factors live in level-major index
    // spaces: L reads and writes streams in L's order, U writes a stream in U's order.  The reference loop
    // (pbicgstab.cu:45-154) only ever combines vectors element by element within two families -- r, rw, p, v, t
    // (residual side: outputs of A, inputs of L) and M^-1 p, M^-1 r, x (solution side: outputs of U, inputs of A) -- so
    // the first family is kept in L's order, the second in U's, and A is stored with rows in L's order and columns in
    // U's positions (ilu_perm_matrix).  Then no vector is permuted inside the loop: per M^-1 application the only indexed
    // access left besides the near gathers is U reading its right-hand side from L's space (1 per row instead of 4).
    // b and x0 are permuted on the way in, x on the way out.  One GPU, reference loop; CUDAMAT_TRSV_PERM=0 disables.
    bool perm = false;
    {
        const char *pe = getenv("CUDAMAT_TRSV_PERM");
        perm = precond == CUDAMAT_PRECOND_ILU0 && !s->sharded && loop == CUDAMAT_LOOP_PBICGSTAB && s->L.lm && s->U.lm && !s->d &&
               !(pe && pe[0] == '0');
        if (perm && s->perm_failed) perm = false;
        if (perm && !s->perm_ready) {
            const int rcp = ilu_perm_matrix(s);
            if (rcp == CUDAMAT_ERR_NOMEM) { perm = false; s->perm_failed = true; }    // no room for the second blocked copy: permute per
            else CM_TRY(rcp);                                                        // application, and do not try again on every solve
        }
    }
    s->perm_active = perm;
    struct PermOff { cudamat_solver *s; ~PermOff() { s->perm_active = false; } } perm_off{s};
    double *const x_user = x;
    if (perm) {
        CM_TRY(perm_to_space(s, false, b, s->b_perm));
        if (!(flags & CUDAMAT_FLAG_X0_ONES)) CM_TRY(perm_to_space(s, true, x, s->x_perm));
        b = s->b_perm;
        x = s->x_perm;
    }

    // residual history: two entries per iteration (half / full step) or one; capped at 2^20 entries (8 MB) -- a solve
    // with a larger maxit keeps the first 2^20 (the kernels check the capacity)
    const long long want_hist = (long long)(loop != CUDAMAT_LOOP_PBICGSTAB2 ? 2 : 1) * (maxit > 0 ? maxit : 1);
    const int need_hist = (int)(want_hist < (1LL << 20) ? want_hist : (1LL << 20));
    // a restart segment (abs_tol > 0) appends to the history of the segments before it (the kernels check the capacity)
    const int hist_base = abs_tol > 0.0 ? (s->hist_count < s->hist_cap ? s->hist_count : s->hist_cap) : 0;
    if (hist_base == 0 && need_hist > s->hist_cap) {
        if (s->hist) { CM_HIP(hipStreamSynchronize(st)); hipFree(s->hist); s->hist = nullptr; }
        CM_TRY(dev_alloc((void **)&s->hist, sizeof(double) * (size_t)need_hist));
        s->hist_cap = need_hist;
    }
    if (s->hist_cap > hist_base)
        CM_HIP(hipMemsetAsync(s->hist + hist_base, 0xFF, sizeof(double) * (size_t)(s->hist_cap - hist_base), st));  // NaN fill
    s->last_loop = loop;
    const bool profile = (flags & CUDAMAT_FLAG_PROFILE) != 0;
    const bool sharded = s->sharded;
    const int n = s->n;
    LoopArgs la{s->st, s->hist + hist_base, s->hist_cap - hist_base, loop, (flags & CUDAMAT_FLAG_NO_EXIT) ? 1 : 0, s->snap_dev, kRing, 0};
    for (int i = 0; i < kRing; i++) s->snap_host[i] = 0ULL;
    const LoopArgs la_none{nullptr, nullptr, 0, 0, 0};
    const ScalarSrc nosrc{nullptr, 0, 1};
    size_t pe = 0;   // profiling events used

    s->comm_used = 0;
    s->comm_kind.clear();
    s->profiling = profile && sharded;
    struct ProfilingOff { cudamat_solver *s; ~ProfilingOff() { s->profiling = false; } } profiling_off{s};
    Range range_loop("cudamat: iteration loop (enqueue + lagged checks)");
    const double t_loop0 = now_s();
    if (flags & CUDAMAT_FLAG_X0_ONES) CM_TRY(launch_fill(st, n, 1.0, x));
    // r = A x0 (pbicgstab.cu:67 / :645-646); x may be a caller buffer without pad
    CM_HIP(hipMemcpyAsync(s->pw, x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
    CM_TRY(spmv_local(s, s->pw, s->r, 0, nullptr, nullptr, la_none, CHECK_NONE, nosrc));
    int np_full = 0, np_half = 0;
    // reference loop in its five-launch form: the half step's x += alpha pw (pbicgstab.cu:110) is carried out by k_full of the
    // same iteration (CUDAMAT_DEFER_X=0: by k_half, as the reference orders it) -- same operations on the same operands
    const bool defer_x = [] { const char *e = getenv("CUDAMAT_DEFER_X"); return !(e && e[0] == '0'); }();
    const double *pw_last = nullptr;
    CM_TRY(launch_init(st, n, b, s->r, s->rw, s->p, s->parts_full, &np_full));   // :69-74
    ScalarSrc full_src{s->parts_full, np_full, 2};
    if (sharded) {
        CM_TRY(launch_reduce_parts(st, full_src, 2, s->red + 4, 0));
        CM_TRY(allreduce(s, s->red + 4, 2));
        full_src = ScalarSrc{s->red + 4, 0, 1};
    }
    CM_TRY(launch_init_finish(st, s->st, full_src, tol, abs_tol));

    // Pipelined BiCGStab (kernels.hip): extra vectors, w0 = A rh0 (with rw.w0), t0 = A wh0, and the seed
    // [rw.r0, rw.w0, 0, 0, r0.r0] of the first k_pipe_a.  A reduction phase = the per-workgroup partials of a
    // kernel summed (and, sharded, all-reduced) into red_pipe: on the communicator's reduce stream when it has one,
    // so that it runs beside the SpMV that follows the kernel.
    // With a preconditioner (ILU(0), or block-Jacobi ILU(0) when sharded: SURVEY 8 f4) the hatted vectors M^-1 r,
    // M^-1 w, M^-1 s, M^-1 z, M^-1 q are carried too and M^-1 is applied in front of each SpMV, where
    // pbicgstab.cu:92-98,121-127 apply it.
    const bool pipelined = loop == CUDAMAT_LOOP_PIPELINED;
    const bool pipe_pc = pipelined && precond != CUDAMAT_PRECOND_NONE;
    // residual replacement period (Cools & Vanroose): every rr-th iteration r, w, s, z (and their hatted forms, and v)
    // are recomputed from x and p, which discards the rounding errors the recurrences have accumulated
    int pipe_rr = kPipeRR;
    if (const char *e = getenv("CUDAMAT_PIPE_RR")) pipe_rr = atoi(e);
    ScalarSrc pipeB_src{nullptr, 0, 1};
    hipStream_t rst = nullptr;
    if (pipelined) {
        const size_t nb = sizeof(double) * (size_t)(s->n_pad > 0 ? s->n_pad : 1);
        if (!s->pz) {
            double **vs[] = {&s->pz, &s->pww, &s->pq, &s->py, &s->pxh};
            for (double **q : vs) {
                CM_TRY(dev_alloc((void **)q, nb));
                CM_HIP(hipMemsetAsync(*q, 0, nb, st));
            }
            CM_TRY(dev_alloc((void **)&s->pipeA, sizeof(double) * 3 * kVecGridMax));
            CM_TRY(dev_alloc((void **)&s->pipeB, sizeof(double) * 5 * kVecGridMax));
            CM_TRY(dev_alloc((void **)&s->red_pipe, sizeof(double) * 16));
        }
        if (pipe_pc && !s->prh) {
            double **vs[] = {&s->prh, &s->pwh, &s->psh, &s->pzh, &s->pqh, &s->ptmp};
            for (double **q : vs) {
                CM_TRY(dev_alloc((void **)q, nb));
                CM_HIP(hipMemsetAsync(*q, 0, nb, st));
            }
        }
        if (sharded && s->comm.allreduce_side && s->comm.reduce_stream) {
            rst = (hipStream_t)s->comm.reduce_stream;
            for (int e = 0; e < 2; e++) {
                if (!s->ev_red[e]) CM_HIP(hipEventCreateWithFlags(&s->ev_red[e], hipEventDisableTiming));
                if (!s->ev_red_done[e]) CM_HIP(hipEventCreateWithFlags(&s->ev_red_done[e], hipEventDisableTiming));
            }
        }
        const LoopArgs la_freeze{s->st, nullptr, 0, loop, 0};      // (returns at once when the initial guess already passes: restarts)
        const double *rh0 = s->r;
        if (pipe_pc) { CM_TRY(precond_apply(s, s->r, s->ptmp, s->prh)); rh0 = s->prh; }              // rh0 = M^-1 r0
        CM_TRY(spmv_local(s, rh0, s->pww, 1, s->rw, s->parts_rv, la_freeze, CHECK_NONE, nosrc));      // w0 = A rh0, rw.w0
        ScalarSrc rww{s->parts_rv, spmv_parts(s), 2};
        if (sharded) {
            CM_TRY(launch_reduce_parts(st, rww, 1, s->red + 0, 0));
            CM_TRY(allreduce(s, s->red + 0, 1));
            rww = ScalarSrc{s->red + 0, 0, 1};
        }
        const double *wh0 = s->pww;
        if (pipe_pc) { CM_TRY(precond_apply(s, s->pww, s->ptmp, s->pwh)); wh0 = s->pwh; }            // wh0 = M^-1 w0
        CM_TRY(spmv_local(s, wh0, s->t, 0, nullptr, nullptr, la_freeze, CHECK_NONE, nosrc));          // t0 = A wh0
        CM_TRY(launch_pipe_seed(st, full_src, rww, s->red_pipe + 8));
        pipeB_src = ScalarSrc{s->red_pipe + 8, 0, 1};
    }
    // one reduction phase of the pipelined loop: partials -> K sums in `out` (all-reduced when sharded); returns
    // the source the consumer kernel reads.  With a reduce stream the work is queued there, behind `slot`'s event.
    auto pipe_reduce = [&](ScalarSrc parts, int K, double *out, int slot) -> int {
        if (!sharded) return CUDAMAT_OK;                     // the consumer sums the partials itself
        if (rst) {
            CM_HIP(hipEventRecord(s->ev_red[slot], st));
            CM_HIP(hipStreamWaitEvent(rst, s->ev_red[slot], 0));
            CM_TRY(launch_reduce_parts(rst, parts, K, out, 0));
            comm_mark_begin(s, 3, rst);
            if (s->comm.allreduce_side(s->comm.user, out, K) != 0) { set_error("allreduce_side callback failed"); return CUDAMAT_ERR_COMM; }
            comm_mark_end(s, rst);
            CM_HIP(hipEventRecord(s->ev_red_done[slot], rst));
        } else {
            CM_TRY(launch_reduce_parts(st, parts, K, out, 0));
            CM_TRY(allreduce(s, out, K));
        }
        return CUDAMAT_OK;
    };
    auto pipe_wait = [&](int slot) -> int {
        if (sharded && rst) CM_HIP(hipStreamWaitEvent(st, s->ev_red_done[slot], 0));
        return CUDAMAT_OK;
    };
    int np_a = 0, np_b = 0;

    // Small systems (vectors resident in L2): three launches per iteration instead of five -- the vector updates
    // in front of the two SpMVs are folded into them (kernels.hip, "fused loop"); p, v and r are double-buffered.
    bool fused = false;
    {
        // Measured (bench.py, one MI355X): 5-point stencil rows 38.5 -> 46.3 k it/s at 1e4 rows, 37.7 -> 43.8 k at 4e4,
        // 30.4 -> 32.1 k at 1.6e5, even at 4.9e5, slower beyond; with 50 entries per row the three gathers per entry
        // cost more than the two launches save (29.9 -> 24.1 k it/s at 2e4 rows).  So: short rows (the stream-tile
        // plan) up to 3e5 rows.  CUDAMAT_FUSED=0 disables, CUDAMAT_FUSED=N forces it for every supported plan up to N rows.
        const char *fe = getenv("CUDAMAT_FUSED");
        const bool forced = fe && fe[0] != '\0';
        const long long max_rows = forced ? atoll(fe) : 300000;
        fused = loop != CUDAMAT_LOOP_PIPELINED && !sharded && !precond && s->spmv_mode == 0 && fused_spmv_supported(s->plan) && n > 0 && n <= max_rows &&
                (forced || s->plan.stream_rows > 0);
        if (fused && !s->v2) {
            const size_t nb = sizeof(double) * (size_t)(s->n_pad > 0 ? s->n_pad : 1);
            CM_TRY(dev_alloc((void **)&s->v2, nb));
            CM_HIP(hipMemsetAsync(s->v2, 0, nb, st));
        }
    }
    double *p_a = s->p, *p_b = s->pw, *v_a = s->v, *v_b = s->v2;

    // Very small systems (one stream tile per workgroup, at most one workgroup per compute unit): the whole loop in ONE
    // launch, grid barriers instead of launch boundaries (kernels.hip, "resident loop").  CUDAMAT_RESIDENT=0 disables.
    *resident_gave_up = false;
    int loop_form = fused ? 1 : 0;
    bool resident = false;
    {
        const char *re = getenv("CUDAMAT_RESIDENT");
        resident = fused && !profile && !s->resident_off && !(re && re[0] == '0') && resident_loop_supported(s->plan, n);
        if (resident) {      // all workgroups must be resident at once: at most one per two compute units of THIS device
            if (s->device_cus == 0 &&
                hipDeviceGetAttribute(&s->device_cus, hipDeviceAttributeMultiprocessorCount, s->ctx->device) != hipSuccess)
                s->device_cus = -1;
            resident = s->device_cus > 0 && 2 * s->plan.grid <= s->device_cus;
        }
    }
    if (resident) {
        loop_form = 2;
        if (!s->bar) CM_TRY(dev_alloc((void **)&s->bar, 2 * sizeof(unsigned)));
        SpmvArgs a{};
        a.n = n; a.rp = s->rp; a.ci = s->ci; a.val = s->val; a.x = nullptr; a.d = s->d; a.xd = nullptr;
        a.alpha = 1.0; a.beta = 0.0; a.check = CHECK_NONE; a.half = nosrc;
        a.loop = la;
        a.loop.snap = nullptr;               // no per-iteration progress words: the host waits for the launch
        int done = 0;
        while (done < maxit) {
            const int c = maxit - done < 8192 ? maxit - done : 8192;      // ~0.1 s of iterations per launch
            CM_HIP(hipMemsetAsync(s->bar, 0, 2 * sizeof(unsigned), st));
            ResidentArgs q{};
            q.iters = c; q.first_count = done == 0 ? np_full : s->plan.grid; q.bar = s->bar;
            q.spin_limit = 1u << 22;
            if (const char *lim = getenv("CUDAMAT_RESIDENT_SPIN_LIMIT")) q.spin_limit = (unsigned)atoi(lim);
            q.p_a = p_a; q.p_b = p_b; q.v_a = v_a; q.v_b = v_b; q.r = s->r; q.s = s->s; q.t = s->t; q.x = x; q.rw = s->rw;
            q.parts_rv = s->parts_rv; q.parts_tt = s->parts_tt; q.parts_half = s->parts_half; q.parts_full = s->parts_full;
            CM_TRY(launch_resident_loop(st, s->plan, a, q));
            unsigned bar_host[2] = {0u, 0u};
            CM_HIP(hipMemcpyAsync(&s->st_ring[0], s->st, sizeof(LoopState), hipMemcpyDeviceToHost, st));
            CM_HIP(hipMemcpyAsync(bar_host, s->bar, sizeof(bar_host), hipMemcpyDeviceToHost, st));
            CM_HIP(hipStreamSynchronize(st));
            if (bar_host[1] != 0u) {         // a barrier wait ran into its bound: this attempt is void
                *resident_gave_up = true;
                *precond_gave_up = false;
                return CUDAMAT_OK;
            }
            full_src = ScalarSrc{s->parts_full, s->plan.grid, 2};
            if (s->st_ring[0].state != 0) break;
            if (c & 1) {
                std::swap(p_a, p_b);
                std::swap(v_a, v_b);
                std::swap(s->r, s->s);
            }
            done += c;
        }
    }

    int k = 0;
    for (; !resident && k < maxit; k++) {
        if (k >= kLag) {   // lagged, deterministic look at the device state: the progress word of
            const int j = k - kLag;   // iteration j, published by its k_full through pinned memory
            volatile unsigned long long *slot = &s->snap_host[j % kRing];
            unsigned long long w = *slot;
            if ((unsigned)(w >> 32) != (unsigned)(j + 1)) {
                const double t_wait = now_s();
                while ((unsigned)((w = *slot) >> 32) != (unsigned)(j + 1)) {
                    __builtin_ia32_pause();
                    if (now_s() - t_wait > 30.0) {
                        // give queued work a bounded chance to drain (a query, not a wait: the device may be wedged),
                        // then fail the solve; in a sharded run the caller must exit so that its peers are torn down
                        const double t_drain = now_s();
                        while (hipStreamQuery(st) == hipErrorNotReady && now_s() - t_drain < 5.0) __builtin_ia32_pause();
                        set_error("iteration %d did not report progress within 30 s%s", j,
                                  sharded ? " (sharded run: this rank must exit, its peers are waiting in a collective)" : "");
                        return CUDAMAT_ERR_HIP;
                    }
                }
            }
            if ((unsigned)(w & 0xffffffffULL) != 0u) break;
        }
        la.k = k;
        if (pipelined) {
            const PipeHatA hat_a = pipe_pc ? PipeHatA{s->prh, s->pwh, s->pzh, s->psh, s->pqh} : PipeHatA{nullptr, nullptr, nullptr, nullptr, nullptr};
            const PipeHatB hat_b = pipe_pc ? PipeHatB{s->pqh, s->pwh, s->pzh, s->prh} : PipeHatB{nullptr, nullptr, nullptr, nullptr};
            double *const rh = pipe_pc ? s->prh : s->r, *const wh = pipe_pc ? s->pwh : s->pww;
            double *const sh = pipe_pc ? s->psh : s->s, *const zh = pipe_pc ? s->pzh : s->pz;
            // full-step test of iteration k-1, beta, alpha, the recurrences; dots (q.y, y.y, q.q)
            CM_TRY(launch_pipe_a(st, la, pipeB_src, n, s->r, s->pww, s->t, s->v, s->p, s->s, s->pz, s->pq, s->py, x, s->pxh,
                                 s->pipeA, &np_a, hat_a));
            ScalarSrc a_src{s->pipeA, np_a, 3};
            CM_TRY(pipe_reduce(a_src, 3, s->red_pipe + 0, 0));
            if (sharded) a_src = ScalarSrc{s->red_pipe + 0, 0, 1};
            if (pipe_pc) {                                                                            // zh = M^-1 z   :92-98
                if (profile) hipEventRecord(prof_event(s, pe++), st);
                CM_TRY(precond_apply(s, s->pz, s->ptmp, s->pzh));
                if (profile) hipEventRecord(prof_event(s, pe++), st);
            }
            if (profile) hipEventRecord(prof_event(s, pe++), st);
            CM_TRY(spmv_local(s, zh, s->v, 0, nullptr, nullptr, la, CHECK_NONE, nosrc));              // v = A zh
            if (profile) hipEventRecord(prof_event(s, pe++), st);
            CM_TRY(pipe_wait(0));
            // half-step test, omega, x, r, rh, w; dots (rw.r, rw.w, rw.s, rw.z, r.r); i++
            CM_TRY(launch_pipe_b(st, la, a_src, n, s->pq, s->py, s->t, s->v, s->rw, s->s, s->pz, s->pxh, x, s->r, s->pww,
                                 s->pipeB, &np_b, hat_b));
            if (pipe_rr > 0 && (k + 1) % pipe_rr == 0) {
                // Residual replacement.  q and y are free until the next k_pipe_a; pw is not used by this loop.  The
                // kernels below return at once when the loop is frozen (`la`), the triangular solves do not look.
                // (every kernel of this block returns at once when the loop is frozen, so r and the phase-B partials stay
                // those of the returned iterate; the copy of x only fills the scratch vector pw)
                CM_HIP(hipMemcpyAsync(s->pw, x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
                CM_TRY(spmv_local(s, s->pw, s->pq, 0, nullptr, nullptr, la, CHECK_NONE, nosrc));      // q = A x
                CM_TRY(launch_residual(st, la, n, b, s->pq, s->r));                                   // r = b - A x
                if (pipe_pc) CM_TRY(precond_apply(s, s->r, s->ptmp, s->prh));                         // rh = M^-1 r
                CM_TRY(spmv_local(s, rh, s->pww, 0, nullptr, nullptr, la, CHECK_NONE, nosrc));        // w = A rh
                CM_TRY(spmv_local(s, s->p, s->s, 0, nullptr, nullptr, la, CHECK_NONE, nosrc));        // s = A ph
                if (pipe_pc) CM_TRY(precond_apply(s, s->s, s->ptmp, s->psh));                         // sh = M^-1 s
                CM_TRY(spmv_local(s, sh, s->pz, 0, nullptr, nullptr, la, CHECK_NONE, nosrc));         // z = A sh
                if (pipe_pc) CM_TRY(precond_apply(s, s->pz, s->ptmp, s->pzh));                        // zh = M^-1 z
                CM_TRY(spmv_local(s, zh, s->v, 0, nullptr, nullptr, la, CHECK_NONE, nosrc));          // v = A zh
                CM_TRY(launch_pipe_dots(st, la, n, s->rw, s->r, s->pww, s->s, s->pz, s->pipeB, &np_b));
            }
            pipeB_src = ScalarSrc{s->pipeB, np_b, 5};
            CM_TRY(pipe_reduce(pipeB_src, 5, s->red_pipe + 8, 1));
            if (sharded) pipeB_src = ScalarSrc{s->red_pipe + 8, 0, 1};
            if (pipe_pc) {                                                                            // wh = M^-1 w   :121-127
                if (profile) hipEventRecord(prof_event(s, pe++), st);
                CM_TRY(precond_apply(s, s->pww, s->ptmp, s->pwh));
                if (profile) hipEventRecord(prof_event(s, pe++), st);
            }
            if (profile) hipEventRecord(prof_event(s, pe++), st);
            CM_TRY(spmv_local(s, wh, s->t, 0, nullptr, nullptr, la, CHECK_NONE, nosrc));              // t = A wh
            if (profile) hipEventRecord(prof_event(s, pe++), st);
            CM_TRY(pipe_wait(1));
            continue;
        }
        if (fused) {
            SpmvArgs a{};
            a.n = n; a.rp = s->rp; a.ci = s->ci; a.val = s->val; a.x = nullptr; a.d = s->d; a.xd = nullptr;
            a.alpha = 1.0; a.beta = 0.0; a.loop = la; a.check = CHECK_NONE; a.half = nosrc;
            const int np = plan_spmv_parts(s->plan);
            // rho, beta, full-step test, p' = r + beta (p - omega v), v' = A p', rw.v'            :80-89, :104-106
            FuseArgs f1{};
            f1.mode = 1; f1.r = s->r; f1.p_old = p_a; f1.v_old = v_a; f1.p_out = p_b; f1.src = full_src;
            a.y = v_b; a.dot = 1; a.w = s->rw; a.parts = s->parts_rv;
            if (profile) hipEventRecord(prof_event(s, pe++), st);
            CM_TRY(launch_fused_spmv(st, s->plan, a, f1));
            if (profile) hipEventRecord(prof_event(s, pe++), st);
            // alpha, s = r - alpha v', x += alpha p', t = A s, (t.s, t.t), ||s||^2                :107-111, :132-136
            FuseArgs f2{};
            f2.mode = 2; f2.r = s->r; f2.v = v_b; f2.s_out = s->s; f2.xsol = x; f2.p = p_b;
            f2.src = ScalarSrc{s->parts_rv, np, 2}; f2.parts_half = s->parts_half;
            a.y = s->t; a.dot = 2; a.w = nullptr; a.parts = s->parts_tt;
            if (profile) hipEventRecord(prof_event(s, pe++), st);
            CM_TRY(launch_fused_spmv(st, s->plan, a, f2));
            if (profile) hipEventRecord(prof_event(s, pe++), st);
            // half-step test, omega, x += omega s, r = s - omega t, (rw.r, ||r||^2), i++          :116, :137-151
            CM_TRY(launch_full(st, la, ScalarSrc{s->parts_tt, np, 2}, n, x, s->s, s->s, s->t, s->rw, s->parts_full, &np_full,
                               ScalarSrc{s->parts_half, np, 1}));
            full_src = ScalarSrc{s->parts_full, np_full, 2};
            std::swap(p_a, p_b);
            std::swap(v_a, v_b);
            std::swap(s->r, s->s);        // the new residual was written over s
            continue;
        }
        // rho, beta, p = r + beta (p - omega v)                     :80-89
        CM_TRY(launch_update_p(st, la, full_src, n, s->r, s->p, s->v));
        const double *pw = s->p;
        if (precond) {                                            // :92-98
            if (profile) hipEventRecord(prof_event(s, pe++), st);
            CM_TRY(precond_apply(s, s->p, s->t, s->pw, perm));
            if (profile) hipEventRecord(prof_event(s, pe++), st);
            pw = s->pw;
        }
        // v = A pw, rw.v                                            :104-106
        if (profile) hipEventRecord(prof_event(s, pe++), st);
        CM_TRY(spmv_local(s, pw, s->v, 1, s->rw, s->parts_rv, la, CHECK_NONE, nosrc));
        if (profile) hipEventRecord(prof_event(s, pe++), st);
        ScalarSrc rv_src{s->parts_rv, spmv_parts(s), 2};
        if (sharded) {
            CM_TRY(launch_reduce_parts(st, rv_src, 1, s->red + 0, 0));
            CM_TRY(allreduce(s, s->red + 0, 1));
            rv_src = ScalarSrc{s->red + 0, 0, 1};
        }
        // alpha, r -= alpha v, x += alpha pw, ||r||                 :107-111
        // (x += alpha pw rides in k_full -- x is then streamed once per iteration, not twice; an exit at the half step
        // applies it after the loop)
        CM_TRY(launch_half(st, la, rv_src, n, s->r, s->v, defer_x ? nullptr : x, pw, s->parts_half, &np_half));
        pw_last = pw;
        const ScalarSrc half_src{s->parts_half, np_half, 1};
        const double *sv = s->r;
        ScalarSrc tt_src{s->parts_tt, spmv_parts(s), 2};
        if (!sharded) {
            if (precond) {                                        // :116, :121-127
                CM_TRY(launch_check(st, la, half_src, CHECK_HALF));
                if (profile) hipEventRecord(prof_event(s, pe++), st);
                CM_TRY(precond_apply(s, s->r, s->t, s->s, perm));
                if (profile) hipEventRecord(prof_event(s, pe++), st);
                sv = s->s;
                if (profile) hipEventRecord(prof_event(s, pe++), st);
                CM_TRY(spmv_local(s, sv, s->t, 2, s->r, s->parts_tt, la, CHECK_NONE, nosrc));
                if (profile) hipEventRecord(prof_event(s, pe++), st);
            } else {
                // half-step test fused into the SpMV prologue      :116, :132-136
                if (profile) hipEventRecord(prof_event(s, pe++), st);
                CM_TRY(spmv_local(s, sv, s->t, 2, s->r, s->parts_tt, la, CHECK_HALF, half_src));
                if (profile) hipEventRecord(prof_event(s, pe++), st);
            }
        } else {
            // The SpMV changes only t, so the half-step test may ride with the
            // (t.r, t.t) all-reduce: one collective instead of two.
            CM_TRY(launch_reduce_parts(st, half_src, 1, s->red + 1, 0));
            if (precond) {   // block-Jacobi: local triangular solves, no collective (they only write s and t, so an
                             // exit at the half step, noticed after the all-reduce below, leaves x and r untouched)
                if (profile) hipEventRecord(prof_event(s, pe++), st);
                CM_TRY(precond_apply(s, s->r, s->t, s->s));
                if (profile) hipEventRecord(prof_event(s, pe++), st);
                sv = s->s;
            }
            if (profile) hipEventRecord(prof_event(s, pe++), st);
            CM_TRY(spmv_local(s, sv, s->t, 2, s->r, s->parts_tt, la, CHECK_NONE, nosrc));
            if (profile) hipEventRecord(prof_event(s, pe++), st);
            CM_TRY(launch_reduce_parts(st, tt_src, 2, s->red + 2, 0));
            CM_TRY(allreduce(s, s->red + 1, 3));
            CM_TRY(launch_check(st, la, ScalarSrc{s->red + 1, 0, 1}, CHECK_HALF));
            tt_src = ScalarSrc{s->red + 2, 0, 1};
        }
        // omega, x += omega s, r -= omega t, (rw.r, ||r||), i++     :137-151
        CM_TRY(launch_full(st, la, tt_src, n, x, sv, s->r, s->t, s->rw, s->parts_full, &np_full, ScalarSrc{nullptr, 0, 1},
                           defer_x ? pw : nullptr));
        full_src = ScalarSrc{s->parts_full, np_full, 2};
        if (sharded) {
            CM_TRY(launch_reduce_parts(st, full_src, 2, s->red + 4, 0));
            CM_TRY(allreduce(s, s->red + 4, 2));
            full_src = ScalarSrc{s->red + 4, 0, 1};
        }
    }
    // the full-step test of the last iteration has not been looked at yet
    if (pipelined)      // (check_full wants (., r.r): the last two of the five phase-B scalars)
        CM_TRY(launch_check(st, la, ScalarSrc{pipeB_src.ptr + 3, pipeB_src.count, pipeB_src.stride}, CHECK_FULL));
    else
        CM_TRY(launch_check(st, la, full_src, CHECK_FULL));
    CM_HIP(hipMemcpyAsync(&s->st_ring[0], s->st, sizeof(LoopState), hipMemcpyDeviceToHost, st));
    CM_HIP(hipStreamSynchronize(st));                              // :372
    const double t_loop1 = now_s();
    // A dependency-driven triangular solve that gave up waiting (another spin-waiting kernel shared the GPU,
    // see DESIGN.md section 4) invalidates this attempt; in a sharded run every rank must learn of it.
    *precond_gave_up = false;
    if (precond) {
        int bad = trsv_status(s) != CUDAMAT_OK ? 1 : 0;
        if (sharded) {
            const double mine = (double)bad;
            double all = 0.0;
            CM_HIP(hipMemcpy(s->red + 7, &mine, sizeof(double), hipMemcpyHostToDevice));
            CM_TRY(allreduce(s, s->red + 7, 1));
            CM_HIP(hipStreamSynchronize(st));
            CM_HIP(hipMemcpy(&all, s->red + 7, sizeof(double), hipMemcpyDeviceToHost));
            bad = all != 0.0;
        }
        if (bad) {
            *precond_gave_up = true;
            return CUDAMAT_OK;
        }
    }
    if (defer_x && pw_last && s->st_ring[0].state == 1) {      // left through the half step: pbicgstab.cu:110 is still due
        CM_TRY(launch_axpy(st, n, s->st_ring[0].alpha, pw_last, x));
        CM_HIP(hipStreamSynchronize(st));
    }
    if (perm) {                              // the iterate leaves U's space
        CM_TRY(perm_from_space(s, true, x, x_user));
        CM_HIP(hipStreamSynchronize(st));
    }
    const LoopState fin = s->st_ring[0];
    if (pipelined && fin.state == 1) {      // left through the half step: the iterate is x + alpha p, kept in xh
        CM_HIP(hipMemcpyAsync(x, s->pxh, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
        CM_HIP(hipStreamSynchronize(st));
    }
    s->hist_count = hist_base + ((loop != CUDAMAT_LOOP_PBICGSTAB2) ? 2 * fin.it + (fin.state == 1 ? 1 : 0) : fin.it);
    if (s->hist_count > s->hist_cap) s->hist_count = s->hist_cap;

    cudamat_stats stt;
    memset(&stt, 0, sizeof(stt));
    stt.iters = fin.it;
    stt.half_exit = fin.state == 1;
    stt.converged = fin.state == 1 || fin.state == 2;
    stt.breakdown = fin.state == 3;
    stt.nrm0 = fin.nrm0;
    stt.nrm = fin.nrm;
    stt.t_analysis = s->t_analysis;
    stt.t_factor = s->t_factor;
    stt.t_solve = t_loop1 - t_loop0;
    stt.n_levels_l = s->L.nlevels;
    stt.n_levels_u = s->U.nlevels;
    stt.trsv_form = precond ? trsv_form_code(s) : 0;
    stt.trsv_fallbacks = s->trsv_fallbacks;
    if (precond) trsv_group_counts(s, &stt.trsv_groups_l, &stt.trsv_groups_u);
    stt.loop_form = loop_form;
    stt.loop_fallbacks = s->loop_fallbacks;
    stt.overlapped = sharded && s->windowed ? 2 : (sharded && s->overlap && s->spmv_mode == 1) ? 1 : 0;
    stt.gather_fraction = sharded ? s->gather_fraction : 0.0;
    stt.ms_spmv_alone = s->ms_spmv_alone;
    stt.t_setup = s->t_create + s->t_spmv_setup;
    stt.t_tune = s->t_spmv_timing;
    stt.spmv_mode = s->spmv_mode;
    if (s->profiling) {
        // exposed part of an overlapped gather: the waits (kind 1), clipped to the gather they wait for only by
        // construction -- the solver's stream idles there for nothing else
        for (size_t i = 0; i < s->comm_kind.size() && 2 * i + 1 < s->comm_used; i++) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, s->comm_ev[2 * i], s->comm_ev[2 * i + 1]);
            switch (s->comm_kind[i]) {
            case 0: stt.ms_gather += ms; stt.n_gather++; break;
            case 2: stt.ms_gather += ms; stt.ms_gather_exposed += ms; stt.n_gather++; break;
            default: stt.ms_allreduce += ms; stt.n_allreduce++; break;
            }
        }
    }
    if (profile) {
        // events come in (start, stop) pairs; trsv pairs and spmv pairs alternate as recorded
        size_t i = 0;
        const int per_it_pairs = precond ? 4 : 2;
        for (; i + 1 < pe; i += 2) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, s->prof_ev[i], s->prof_ev[i + 1]);
            const size_t pair = (i / 2) % per_it_pairs;
            const bool is_trsv = precond && (pair == 0 || pair == 2);
            if (is_trsv) { stt.ms_trsv += ms; stt.n_trsv += 2; }
            else {
                stt.ms_spmv += ms;
                stt.n_spmv += 1;
                // overlapped gather: what an SpMV took beyond the same SpMV with x already in place (the tuner's
                // timing) is the part of the exchange that was NOT hidden behind it
                if (stt.overlapped == 1 && s->ms_spmv_alone > 0.0 && ms > s->ms_spmv_alone) stt.ms_gather_exposed += ms - s->ms_spmv_alone;
            }
        }
    }

    if (flags & CUDAMAT_FLAG_DEBUG) {
        std::vector<double> h((size_t)(s->hist_count > 0 ? s->hist_count : 1));
        if (s->hist_count > 0)
            hipMemcpy(h.data(), s->hist, sizeof(double) * (size_t)s->hist_count, hipMemcpyDeviceToHost);
        if (loop != CUDAMAT_LOOP_PBICGSTAB2) {
            printf("gpu, init residual:norm %20.16f\n", fin.nrm0);            // :77
            for (int i = hist_base; i < s->hist_count; i++) {                  // (a restart segment prints its own part)
                if ((i & 1) == 0) printf("i = %d, residual norm (before precond) = %g\n", i / 2, h[i]);  // :114
                else printf("i = %d, residual norm = %g\n", i / 2, h[i]);      // :145
            }
        } else {
            printf("initial norm = %g\n", fin.nrm0);                           // :659
            for (int i = 0; i < s->hist_count; i++) printf("k = %d, norm = %g\n", i, h[i]);  // :727
            if (fin.state == 3)
                printf("omega is close to zero, cannot continue\nomega = %g\n", fin.omega);   // :737
        }
        fflush(stdout);
    }
    stt.t_total = now_s() - t_begin;
    if (out) *out = stt;
    return CUDAMAT_OK;
}

static int solve_guarded(cudamat_solver *s, const double *b, double *x, int precond, int loop, int maxit, double tol,
                         int flags, cudamat_stats *out, double abs_tol)
{
    CM_ARG(s && b && x, "null pointer");
    // keep the caller's x0 while the dependency-driven preconditioner is in use: if one of its waits times
    // out, the solve is redone from x0 with the level-by-level kernels (same results, bit for bit)
    // (the single-launch loop of very small systems can be voided the same way: <= 65536 rows, the copy is nothing)
    const bool keep_x0 = (precond != CUDAMAT_PRECOND_NONE || (!s->sharded && !s->resident_off && s->n <= 65536)) &&
                         !(flags & CUDAMAT_FLAG_X0_ONES);
    if (keep_x0) {
        CM_HIP(hipSetDevice(s->ctx->device));
        if (!s->x0_save) CM_TRY(dev_alloc((void **)&s->x0_save, sizeof(double) * (size_t)(s->n > 0 ? s->n : 1)));
        CM_HIP(hipMemcpyAsync(s->x0_save, x, sizeof(double) * (size_t)s->n, hipMemcpyDeviceToDevice, s->ctx->stream));
    }
    bool gave_up = false, resident_gave_up = false;
    CM_TRY(solve_once(s, b, x, precond, loop, maxit, tol, flags, out, &gave_up, &resident_gave_up, abs_tol));
    if (resident_gave_up) {
        // the grid barrier of the single-launch loop ran into its bound (its workgroups were not all resident: the GPU
        // is shared): from now on this solver uses the three-launch loop; the solve is redone from x0
        s->resident_off = true;
        s->loop_fallbacks++;
        if (getenv("CUDAMAT_VERBOSE"))
            fprintf(stderr, "cudamat: the single-launch loop's grid barrier timed out (GPU shared?); redoing the solve with "
                            "one launch per phase\n");
        if (keep_x0)
            CM_HIP(hipMemcpyAsync(x, s->x0_save, sizeof(double) * (size_t)s->n, hipMemcpyDeviceToDevice, s->ctx->stream));
        CM_TRY(solve_once(s, b, x, precond, loop, maxit, tol, flags, out, &gave_up, &resident_gave_up, abs_tol));
    }
    if (!gave_up) return CUDAMAT_OK;
    if (!trsv_syncfree_active(s)) {
        set_error("triangular solve reported a timeout although the level-by-level kernels were in use");
        return CUDAMAT_ERR_HIP;
    }
    trsv_disable_syncfree(s);
    s->trsv_fallbacks++;          // reported in cudamat_stats: a redo must not pass for a slow solve
    if (getenv("CUDAMAT_VERBOSE"))
        fprintf(stderr, "cudamat: a dependency-driven triangular solve timed out (GPU shared with another spin-waiting "
                        "kernel?); redoing the solve with one launch per level\n");
    if (keep_x0)
        CM_HIP(hipMemcpyAsync(x, s->x0_save, sizeof(double) * (size_t)s->n, hipMemcpyDeviceToDevice, s->ctx->stream));
    CM_TRY(solve_once(s, b, x, precond, loop, maxit, tol, flags, out, &gave_up, &resident_gave_up, abs_tol));
    if (gave_up) {
        set_error("triangular solve timed out twice");
        return CUDAMAT_ERR_HIP;
    }
    return CUDAMAT_OK;
}

extern "C" int cudamat_solver_solve(cudamat_solver *s, const double *b, double *x, int precond,
                                    int loop, int maxit, double tol, int flags, cudamat_stats *out)
{
    cudamat_stats st0;
    CM_TRY(solve_guarded(s, b, x, precond, loop, maxit, tol, flags, &st0, 0.0));
    // The pipelined loop carries r, w = A r, s = A p, z = A s by recurrences; over a few hundred iterations their rounding
    // errors can let the recursive residual pass the test while the true one is orders of magnitude away (seen: 5e-3
    // against a tolerance of 1e-9).  So an iterate that this loop calls converged is VERIFIED: a restart from it computes
    // the true residual b - A x (one SpMV); within twice the target it is accepted, otherwise the loop goes on from
    // there towards the same absolute target -- at most three times, within the caller's maxit.
    if (loop == CUDAMAT_LOOP_PIPELINED && st0.converged && !(flags & CUDAMAT_FLAG_NO_EXIT) && st0.nrm0 > 0.0) {
        const double target = tol * st0.nrm0;
        for (int r = 0; r < 3 && st0.converged && st0.iters < maxit; r++) {
            cudamat_stats st2;
            CM_TRY(solve_guarded(s, b, x, precond, loop, maxit - st0.iters, tol, flags & ~CUDAMAT_FLAG_X0_ONES, &st2, target));
            st0.t_solve += st2.t_solve;
            st0.t_total += st2.t_total;
            st0.nrm = st2.nrm;                       // the true residual of the verified iterate (st2.nrm0) or the loop's last
            if (st2.iters == 0 && st2.converged) break;
            st0.restarts++;
            st0.iters += st2.iters;
            st0.converged = st2.converged;
            st0.half_exit = st2.half_exit;
            st0.breakdown = st2.breakdown;
        }
    }
    if (out) *out = st0;
    return CUDAMAT_OK;
}

extern "C" int cudamat_solver_history(cudamat_solver *s, double *hist_host, int cap, int *count)
{
    CM_ARG(s && count, "null pointer");
    int c = s->hist_count < cap ? s->hist_count : cap;
    if (c < 0) c = 0;
    if (c > 0) {
        CM_ARG(hist_host, "hist_host is NULL");
        CM_HIP(hipMemcpy(hist_host, s->hist, sizeof(double) * (size_t)c, hipMemcpyDeviceToHost));
    }
    *count = c;
    return CUDAMAT_OK;
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

bool cache_enabled()
{
    const char *e = getenv("CUDAMAT_PLAN_CACHE");
    return !(e && e[0] == '0');
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

extern "C" int cudamat_solve(int n, int nnz, const double *A, const int *iA, const int *jA,
                             const double *d, const double *x0, const double *b, int precond,
                             int loop, int maxit, double tol, int debug, double *x,
                             cudamat_stats *out)
{
    CM_ARG(n > 0 && nnz >= 0 && A && iA && jA && b && x, "null pointer or empty system");
    const int base = iA[0];                                        // pbicgstab.cu:201,782,953
    CM_ARG(base == 0 || base == 1, "iA[0] must be 0 or 1");
    CM_ARG(iA[n] - base == nnz, "nnz != iA[n] - iA[0]");
    const double t0 = now_s();
    if (debug && loop == CUDAMAT_LOOP_PBICGSTAB) printf("N=%d, nnz=%d\n", n, nnz);   // :204
    std::lock_guard<std::mutex> cache_lock(g_cache.mu);            // (the entry points are not re-entrant upstream either)
    const bool use_cache = cache_enabled();
    if (!use_cache) cache_drop_locked();
    // same shape as the cached system?  then its context (device, stream) carries this call too
    const bool candidate = use_cache && g_cache.s && g_cache.n == n && g_cache.nnz == nnz && g_cache.base == base;
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
