// sharded.cpp -- cudamat_solve_sharded: the host-pointer solve (cudamat_solve) over several GPUs of one
// node from ONE process.  Host code only (C ABI calls + threads).
//
// The reference is single-GPU (pbicgstab.cu:223-240).  Here the matrix is cut into uniform row blocks
// (SURVEY.md section 8e), rank g = one host thread driving device g: its own context and stream, its row
// block and vector slices in that device's HBM, a communicator of the library's RCCL binding
// (csrc/comm_rccl.hip), and the same C++ loop as everywhere else (cudamat_solver_solve on a solver with
// cudamat_solver_set_comm).  host/example.cpp reaches it through -G<n>.
//
// CUDAMAT_SHARDED_ONE_DEVICE=1 (debugging aid for machines with fewer GPUs than ranks): every rank uses
// device 0 and the collectives are host-synchronised device copies between the ranks' buffers (RCCL refuses
// two ranks on one device).  Same loop, same kernels, same results; no overlap, no speed.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace {

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// A barrier a failed rank can BREAK: every thread waiting in it, or arriving later, returns false.  (With a
// pthread barrier a rank thread that left after an error would leave its peers waiting for ever.)
struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int n = 1, count = 0;
    unsigned gen = 0;
    bool broken = false;
    bool wait()
    {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const unsigned g = gen;
        if (++count == n) {
            count = 0;
            gen++;
            cv.notify_all();
            return true;
        }
        cv.wait(lk, [&] { return gen != g || broken; });
        return gen != g;
    }
    void break_all()
    {
        std::lock_guard<std::mutex> lk(m);
        broken = true;
        cv.notify_all();
    }
};

struct Shared {
    int world = 1;
    Barrier bar;
    std::atomic<int> failed{0};
    std::atomic<int> first_bad{-1};       // the rank whose failure came first: its message is the one reported
    char id[CUDAMAT_RCCL_ID_BYTES];
    // the ranks' RCCL communicators, so that a rank that fails in the solve can abort EVERY rank's (its peers may be
    // inside a collective that waits for it)
    std::mutex comms_mu;
    std::vector<cudamat_comm *> comms;
    // one-device emulation
    bool emulate = false;
    std::vector<const double *> send;
    std::vector<std::vector<double>> vals;
};

// a rank's solve failed: nobody may be left waiting for it
void fail_everyone(Shared *sh, int rank)
{
    int none = -1;
    sh->first_bad.compare_exchange_strong(none, rank);
    sh->failed.store(1);
    sh->bar.break_all();
    std::lock_guard<std::mutex> lk(sh->comms_mu);
    for (cudamat_comm *c : sh->comms)
        if (c) cudamat_rccl_comm_abort(c);
}

struct EmuComm {                 // `user` of the emulated collectives
    Shared *sh;
    int rank;
    cudamat_ctx *ctx;
};

int emu_allgather(void *user, const double *send, double *recv, int64_t count)
{
    EmuComm *c = (EmuComm *)user;
    if (cudamat_ctx_sync(c->ctx)) return 1;                       // my producer kernels are done
    c->sh->send[(size_t)c->rank] = send;
    if (!c->sh->bar.wait()) return 1;
    for (int r = 0; r < c->sh->world; r++)
        if (cudamat_d2d(c->ctx, recv + (size_t)count * (size_t)r, c->sh->send[(size_t)r], sizeof(double) * (size_t)count)) return 1;
    if (cudamat_ctx_sync(c->ctx)) return 1;
    if (!c->sh->bar.wait()) return 1;                             // nobody overwrites a send buffer early
    return 0;
}

int emu_allreduce(void *user, double *buf, int count)
{
    EmuComm *c = (EmuComm *)user;
    std::vector<double> &mine = c->sh->vals[(size_t)c->rank];
    mine.assign((size_t)count, 0.0);
    if (cudamat_d2h(c->ctx, mine.data(), buf, sizeof(double) * (size_t)count)) return 1;
    if (!c->sh->bar.wait()) return 1;
    std::vector<double> tot((size_t)count, 0.0);
    for (int r = 0; r < c->sh->world; r++)                        // fixed order: identical on every rank
        for (int k = 0; k < count; k++) tot[(size_t)k] += c->sh->vals[(size_t)r][(size_t)k];
    if (!c->sh->bar.wait()) return 1;
    return cudamat_h2d(c->ctx, buf, tot.data(), sizeof(double) * (size_t)count) ? 1 : 0;
}

struct Job {
    Shared *sh;
    int rank;
    // the whole system (host)
    int n, base;
    const double *A;
    const int *iA, *jA;
    const double *d, *x0, *b;
    int precond, loop, maxit, debug;
    double tol;
    double *x;
    // result
    int rc = CUDAMAT_OK;
    cudamat_stats st;
    char err[512];
};

// a stage that every rank must pass before the next collective: returns true when ALL ranks passed
bool all_ok(Job *j, int rc)
{
    if (rc != CUDAMAT_OK) {
        if (j->rc == CUDAMAT_OK) {
            j->rc = rc;
            snprintf(j->err, sizeof(j->err), "%s", cudamat_last_error());
        }
        int none = -1;
        j->sh->first_bad.compare_exchange_strong(none, j->rank);
        j->sh->failed.store(1);
    }
    bool ok = j->sh->bar.wait();
    ok = ok && j->sh->failed.load() == 0;
    ok = j->sh->bar.wait() && ok;
    if (!ok && j->rc == CUDAMAT_OK) {
        j->rc = CUDAMAT_ERR_COMM;
        snprintf(j->err, sizeof(j->err), "another rank of the sharded solve failed");
    }
    return ok;
}

void rank_main(Job *j)
{
    Shared *sh = j->sh;
    const int world = sh->world, rank = j->rank;
    memset(&j->st, 0, sizeof(j->st));
    j->err[0] = 0;
    const int64_t per = ((int64_t)j->n + world - 1) / world;
    int64_t row0 = per * rank, row1 = row0 + per;
    if (row0 > j->n) row0 = j->n;
    if (row1 > j->n) row1 = j->n;
    const int nloc = (int)(row1 - row0);
    const int k0 = j->iA[row0] - j->base, k1 = j->iA[row1] - j->base;
    const int nnz_loc = k1 - k0;

    cudamat_ctx *ctx = nullptr;
    cudamat_solver *s = nullptr;
    int *d_rp = nullptr, *d_ci = nullptr;
    double *d_val = nullptr, *d_b = nullptr, *d_x = nullptr, *d_d = nullptr;
    cudamat_comm comm;
    memset(&comm, 0, sizeof(comm));
    EmuComm emu{sh, rank, nullptr};
    bool native = false;
    int rc = CUDAMAT_OK;
    do {
        // ---- stage 1: device, row block, solver (rank-local)
        rc = cudamat_ctx_create(sh->emulate ? 0 : rank, nullptr, &ctx);
        if (rc == CUDAMAT_OK) {
            std::vector<int> rp((size_t)nloc + 1);
            for (int i = 0; i <= nloc; i++) rp[(size_t)i] = j->iA[row0 + i] - j->iA[row0] + j->base;
            const size_t nn = (size_t)(nnz_loc > 0 ? nnz_loc : 1), nv = (size_t)(nloc > 0 ? nloc : 1);
            if (!rc) rc = cudamat_malloc(ctx, sizeof(int) * ((size_t)nloc + 1), (void **)&d_rp);
            if (!rc) rc = cudamat_malloc(ctx, sizeof(int) * nn, (void **)&d_ci);
            if (!rc) rc = cudamat_malloc(ctx, sizeof(double) * nn, (void **)&d_val);
            if (!rc) rc = cudamat_malloc(ctx, sizeof(double) * nv, (void **)&d_b);
            if (!rc) rc = cudamat_malloc(ctx, sizeof(double) * nv, (void **)&d_x);
            if (!rc) rc = cudamat_h2d(ctx, d_rp, rp.data(), sizeof(int) * ((size_t)nloc + 1));
            if (!rc && nnz_loc > 0) rc = cudamat_h2d(ctx, d_ci, j->jA + k0, sizeof(int) * (size_t)nnz_loc);
            if (!rc && nnz_loc > 0) rc = cudamat_h2d(ctx, d_val, j->A + k0, sizeof(double) * (size_t)nnz_loc);
            if (!rc && nloc > 0) rc = cudamat_h2d(ctx, d_b, j->b + row0, sizeof(double) * (size_t)nloc);
            if (!rc && nloc > 0 && j->x0) rc = cudamat_h2d(ctx, d_x, j->x0 + row0, sizeof(double) * (size_t)nloc);
            if (!rc && j->d) {
                rc = cudamat_malloc(ctx, sizeof(double) * nv, (void **)&d_d);
                if (!rc && nloc > 0) rc = cudamat_h2d(ctx, d_d, j->d + row0, sizeof(double) * (size_t)nloc);
            }
            if (!rc) rc = cudamat_solver_create(ctx, nloc, j->n, nnz_loc, d_rp, d_ci, d_val, j->base, &s);
            if (!rc && d_d) rc = cudamat_solver_set_shift(s, d_d);
        }
        if (!all_ok(j, rc)) break;
        // ---- stage 2: the communicator (collective)
        if (sh->emulate) {
            emu.ctx = ctx;
            comm.rank = rank;
            comm.world = world;
            comm.user = &emu;
            comm.allgather = emu_allgather;
            comm.allreduce = emu_allreduce;
        } else {
            if (rank == 0) rc = cudamat_rccl_unique_id(sh->id);
            if (!all_ok(j, rc)) break;                           // (also publishes the id to the other threads)
            rc = cudamat_rccl_comm_create(ctx, sh->id, rank, world, &comm);
            native = rc == CUDAMAT_OK;
            if (native) {
                std::lock_guard<std::mutex> lk(sh->comms_mu);
                sh->comms[(size_t)rank] = &comm;
            }
        }
        if (!all_ok(j, rc)) break;
        rc = cudamat_solver_set_comm(s, &comm);
        if (!all_ok(j, rc)) break;
        // ---- stage 3: the solve (its own setup failures are agreed upon inside, solver.hip setup_agree)
        const int flags = (j->debug && rank == 0 ? CUDAMAT_FLAG_DEBUG : 0) | (j->x0 ? 0 : CUDAMAT_FLAG_X0_ONES);
        rc = cudamat_solver_solve(s, d_b, d_x, j->precond, j->loop, j->maxit, j->tol, flags, &j->st);
        if (rc == CUDAMAT_OK && nloc > 0) rc = cudamat_d2h(ctx, j->x + row0, d_x, sizeof(double) * (size_t)nloc);
        if (rc != CUDAMAT_OK) {
            // No barrier follows a failed solve, and the peers may be inside a collective that waits for this rank
            // (or, having timed out on it, about to synchronise streams that hold one): break the emulated
            // barriers and abort EVERY rank's communicators before anything here waits for a stream.
            if (j->rc == CUDAMAT_OK) {
                j->rc = rc;
                snprintf(j->err, sizeof(j->err), "%s", cudamat_last_error());
            }
            fail_everyone(sh, rank);
        }
    } while (0);
    if (native) {                        // (from here on nobody aborts through a pointer into this frame)
        std::lock_guard<std::mutex> lk(sh->comms_mu);
        sh->comms[(size_t)rank] = nullptr;
    }
    if (sh->failed.load() && native) cudamat_rccl_comm_abort(&comm);
    if (s) cudamat_solver_destroy(s);
    if (native) cudamat_rccl_comm_destroy(&comm);
    void *ptrs[] = {d_rp, d_ci, d_val, d_b, d_x, d_d};
    for (void *p : ptrs)
        if (p) cudamat_free(ctx, p);
    if (ctx) cudamat_ctx_destroy(ctx);
}

}  // namespace

extern "C" int cudamat_solve_sharded(int ngpu, int n, int nnz, const double *A, const int *iA, const int *jA,
                                     const double *d, const double *x0, const double *b, int precond, int loop,
                                     int maxit, double tol, int debug, double *x, cudamat_stats *out)
{
    if (ngpu <= 1) return cudamat_solve(n, nnz, A, iA, jA, d, x0, b, precond, loop, maxit, tol, debug, x, out);
    CM_ARG(n > 0 && nnz >= 0 && A && iA && jA && b && x, "null pointer or empty system");
    const int base = iA[0];
    CM_ARG((base == 0 || base == 1) && iA[n] - base == nnz, "iA[0] must be 0 or 1 and nnz == iA[n] - iA[0]");
    CM_ARG(precond != CUDAMAT_PRECOND_ILU0, "ILU(0) of the whole matrix does not shard: use CUDAMAT_PRECOND_BLOCK_ILU0");
    CM_ARG(ngpu <= n, "more ranks than rows");
    const double t0 = now_s();
    cudamat_plan_cache_clear();          // what cudamat_solve keeps on device 0 (several GB) is needed by the ranks
    Shared sh;
    sh.world = ngpu;
    sh.emulate = cm::config_from_env().sharded_one_device != 0;
    if (!sh.emulate) {
        int have = 0;
        CM_TRY(cudamat_device_count(&have));
        if (have < ngpu) {
            cm::set_error("%d GPUs requested, %d visible", ngpu, have);
            return CUDAMAT_ERR_ARG;
        }
    }
    sh.send.assign((size_t)ngpu, nullptr);
    sh.vals.assign((size_t)ngpu, std::vector<double>());
    sh.comms.assign((size_t)ngpu, nullptr);
    sh.bar.n = ngpu;
    std::vector<Job> jobs((size_t)ngpu);
    std::vector<std::thread> th;
    for (int r = 0; r < ngpu; r++) {
        Job &j = jobs[(size_t)r];
        j.sh = &sh; j.rank = r; j.n = n; j.base = base; j.A = A; j.iA = iA; j.jA = jA; j.d = d; j.x0 = x0; j.b = b;
        j.precond = precond; j.loop = loop; j.maxit = maxit; j.debug = debug; j.tol = tol; j.x = x;
        th.emplace_back(rank_main, &j);
    }
    for (std::thread &t : th) t.join();
    int rc = CUDAMAT_OK;
    const int fb = sh.first_bad.load();          // the root cause first; the others failed because of it
    if (fb >= 0 && jobs[(size_t)fb].rc != CUDAMAT_OK) {
        rc = jobs[(size_t)fb].rc;
        cm::set_error("rank %d of %d: %s", fb, ngpu, jobs[(size_t)fb].err);
    }
    for (Job &j : jobs)
        if (j.rc != CUDAMAT_OK && rc == CUDAMAT_OK) {
            rc = j.rc;
            cm::set_error("rank %d of %d: %s", j.rank, ngpu, j.err);
        }
    if (out) {
        *out = jobs[0].st;                       // every rank takes the same decisions (all-reduced scalars)
        out->t_total = now_s() - t0;
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------------------
// "Dry" communicator: every exchange is SKIPPED (the streams and the call sequence are real).  For timing one
// rank's share of a sharded solve on a single GPU (scripts/rank_probe.py): the iterates are meaningless.
namespace {
struct DryComm {
    hipStream_t side = nullptr, red = nullptr;
};
int dry_allgather(void *, const double *, double *, int64_t) { return 0; }
int dry_allreduce(void *, double *, int) { return 0; }
int dry_gather_part(void *, const double *, double *, int64_t, int64_t, int64_t) { return 0; }
}  // namespace

extern "C" int cudamat_comm_dry_create(cudamat_ctx *ctx, int rank, int world, cudamat_comm *out)
{
    CM_ARG(ctx && out && world >= 1 && rank >= 0 && rank < world, "bad argument");
    CM_HIP(hipSetDevice(ctx->device));
    DryComm *c = new DryComm();
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->red, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        cm::set_error("hipStreamCreate failed");
        return CUDAMAT_ERR_HIP;
    }
    memset(out, 0, sizeof(*out));
    out->rank = rank;
    out->world = world;
    out->user = c;
    out->allgather = dry_allgather;
    out->allreduce = dry_allreduce;
    out->gather_part = dry_gather_part;
    out->comm_stream = c->side;
    out->allreduce_side = dry_allreduce;
    out->reduce_stream = c->red;
    return CUDAMAT_OK;
}

extern "C" int cudamat_comm_dry_destroy(cudamat_comm *comm)
{
    if (!comm || !comm->user) return CUDAMAT_OK;
    CM_ARG(comm->allgather == dry_allgather, "not a dry communicator");
    DryComm *c = (DryComm *)comm->user;
    CM_DROP(hipStreamSynchronize(c->side));
    CM_DROP(hipStreamSynchronize(c->red));
    CM_DROP(hipStreamDestroy(c->side));
    CM_DROP(hipStreamDestroy(c->red));
    delete c;
    memset(comm, 0, sizeof(*comm));
    return CUDAMAT_OK;
}
