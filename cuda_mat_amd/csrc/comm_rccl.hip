// comm_rccl.hip -- the library's own collectives for a row-sharded solve: RCCL over xGMI.
//
// The reference is single-GPU (pbicgstab.cu:223-240: one device, default stream); SURVEY.md section 8e
// adds row-block sharding with an all-gather of the SpMV input and all-reduces of the dot products.
// This file implements the three cudamat_comm callbacks directly on RCCL, so the solver's hot loop
// never leaves C++ (no interpreter, no tensor wrappers):
//   allgather    ncclAllGather on the solver's stream
//   allreduce    ncclAllReduce (sum, in place) on the solver's stream
//   gather_part  one piece of the gather as grouped point-to-point transfers on a second stream: on the
//                full xGMI mesh every peer is one hop away, so a rank sends its piece to all world-1 peers
//                at once (all links busy) instead of passing it round a ring; the solver overlaps these
//                pieces with phase 1 of the blocked SpMV (solver.hip, spmv_local)
// Three communicators, one per stream: the collectives on the solver's stream, the pieces on the communicator's
// own stream, and the all-reduces the pipelined loop runs beside an SpMV on a third stream -- so no two streams
// ever serialise on one communicator's launch order.
//
// librccl is bound at run time (dlopen "librccl.so.1"): a process that has already loaded an RCCL (a host
// program built on PyTorch) gets that very copy, a plain C++ program gets /opt/rocm/lib's; a machine
// without RCCL can still use every single-GPU entry point of this library.
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and enums only: every entry point is resolved with dlsym
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "common.h"

namespace cm {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

static RcclApi g_api;

static int load_rccl()
{
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (g_api.lib) return CUDAMAT_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        set_error("RCCL is not available (dlopen librccl.so.1: %s)", dlerror());
        return CUDAMAT_ERR_COMM;
    }
    RcclApi a;
    a.lib = h;
#define CM_SYM(field, name)                                              \
    a.field = (decltype(a.field))dlsym(h, name);                         \
    if (!a.field) { set_error("librccl lacks %s", name); dlclose(h); return CUDAMAT_ERR_COMM; }
    CM_SYM(GetUniqueId, "ncclGetUniqueId")
    CM_SYM(CommInitRank, "ncclCommInitRank")
    CM_SYM(CommDestroy, "ncclCommDestroy")
    CM_SYM(CommAbort, "ncclCommAbort")
    CM_SYM(AllGather, "ncclAllGather")
    CM_SYM(AllReduce, "ncclAllReduce")
    CM_SYM(Send, "ncclSend")
    CM_SYM(Recv, "ncclRecv")
    CM_SYM(GroupStart, "ncclGroupStart")
    CM_SYM(GroupEnd, "ncclGroupEnd")
    CM_SYM(GetErrorString, "ncclGetErrorString")
#undef CM_SYM
    g_api = a;
    return CUDAMAT_OK;
}

#define CM_NCCL(expr)                                                                       \
    do {                                                                                    \
        ncclResult_t r__ = (expr);                                                          \
        if (r__ != ncclSuccess) {                                                           \
            set_error("RCCL error %d (%s): %s", (int)r__, g_api.GetErrorString(r__), #expr); \
            return CUDAMAT_ERR_COMM;                                                        \
        }                                                                                   \
    } while (0)

struct RcclComm {
    ncclComm_t coll = nullptr;       // all-gather / all-reduce, on the solver's stream
    ncclComm_t p2p = nullptr;        // gather_part, on `side`
    ncclComm_t red = nullptr;        // allreduce_side, on `rstream`
    int rank = 0, world = 1, device = 0;
    hipStream_t main = nullptr;      // the context's stream (not owned)
    hipStream_t side = nullptr;      // owned
    hipStream_t rstream = nullptr;   // owned
    // Abort may come from another thread than the rank's (sharded.cpp: the thread of the rank that failed aborts
    // EVERY rank's communicators).  ncclCommAbort frees the communicator, so an enqueue path must never hold a handle
    // that abort is tearing down: every enqueue registers itself under `mu` (RcclUse: refused once `aborted`, handles
    // snapshotted, `inflight` counted), and abort first closes the door, then waits until the calls in flight have
    // returned before it touches the handles.
    std::mutex mu;
    std::condition_variable cv;
    int inflight = 0;
    bool aborted = false;            // guarded by mu
};

// one enqueue's registration with its communicator (see RcclComm::mu)
struct RcclUse {
    RcclComm *c;
    bool ok;
    ncclComm_t coll = nullptr, p2p = nullptr, red = nullptr;
    explicit RcclUse(RcclComm *c_) : c(c_)
    {
        std::lock_guard<std::mutex> lock(c->mu);
        ok = !c->aborted;
        if (ok) { c->inflight++; coll = c->coll; p2p = c->p2p; red = c->red; }
        else set_error("the communicator was aborted");
    }
    ~RcclUse()
    {
        if (!ok) return;
        std::lock_guard<std::mutex> lock(c->mu);
        if (--c->inflight == 0) c->cv.notify_all();
    }
    RcclUse(const RcclUse &) = delete;
    RcclUse &operator=(const RcclUse &) = delete;
};

// a group that met an error must still be closed: an open group would swallow every later call of this thread
#define CM_NCCL_IN_GROUP(expr)                                                              \
    do {                                                                                    \
        ncclResult_t r__ = (expr);                                                          \
        if (r__ != ncclSuccess) {                                                           \
            g_api.GroupEnd();                                                               \
            set_error("RCCL error %d (%s): %s", (int)r__, g_api.GetErrorString(r__), #expr); \
            return CUDAMAT_ERR_COMM;                                                        \
        }                                                                                   \
    } while (0)

static int rccl_allgather(void *user, const double *send, double *recv, int64_t count)
{
    RcclComm *c = (RcclComm *)user;
    RcclUse use(c);
    if (!use.ok) return CUDAMAT_ERR_COMM;
    CM_NCCL(g_api.AllGather(send, recv, (size_t)count, ncclDouble, use.coll, c->main));
    return 0;
}

static int rccl_allreduce(void *user, double *buf, int count)
{
    RcclComm *c = (RcclComm *)user;
    RcclUse use(c);
    if (!use.ok) return CUDAMAT_ERR_COMM;
    CM_NCCL(g_api.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, use.coll, c->main));
    return 0;
}

static int rccl_allreduce_side(void *user, double *buf, int count)
{
    RcclComm *c = (RcclComm *)user;
    RcclUse use(c);
    if (!use.ok) return CUDAMAT_ERR_COMM;
    CM_NCCL(g_api.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, use.red, c->rstream));
    return 0;
}

// rank q's send[offset, offset + count) -> recv[q * stride + offset, ...) on every other rank
static int rccl_gather_part(void *user, const double *send, double *recv, int64_t stride, int64_t offset, int64_t count)
{
    RcclComm *c = (RcclComm *)user;
    if (c->world <= 1 || count <= 0) return 0;
    RcclUse use(c);
    if (!use.ok) return CUDAMAT_ERR_COMM;
    CM_NCCL(g_api.GroupStart());
    for (int d = 1; d < c->world; d++) {
        // peer order rotated by rank: at every position of the group the world's sends hit distinct receivers
        const int to = (c->rank + d) % c->world, from = (c->rank - d + c->world) % c->world;
        CM_NCCL_IN_GROUP(g_api.Send(send + offset, (size_t)count, ncclDouble, to, use.p2p, c->side));
        CM_NCCL_IN_GROUP(g_api.Recv(recv + (size_t)from * (size_t)stride + (size_t)offset, (size_t)count, ncclDouble, from, use.p2p, c->side));
    }
    CM_NCCL(g_api.GroupEnd());
    return 0;
}

// windowed gather (halo): per-peer ranges, one group on the solver's stream
static int rccl_gather_window(void *user, const double *send, double *recv, int64_t stride, const int64_t *send_off,
                              const int64_t *send_cnt, const int64_t *recv_off, const int64_t *recv_cnt)
{
    RcclComm *c = (RcclComm *)user;
    if (c->world <= 1) return 0;
    RcclUse use(c);
    if (!use.ok) return CUDAMAT_ERR_COMM;
    CM_NCCL(g_api.GroupStart());
    for (int d = 1; d < c->world; d++) {
        const int to = (c->rank + d) % c->world, from = (c->rank - d + c->world) % c->world;
        if (send_cnt[to] > 0)
            CM_NCCL_IN_GROUP(g_api.Send(send + send_off[to], (size_t)send_cnt[to], ncclDouble, to, use.coll, c->main));
        if (recv_cnt[from] > 0)
            CM_NCCL_IN_GROUP(g_api.Recv(recv + (size_t)from * (size_t)stride + (size_t)recv_off[from], (size_t)recv_cnt[from], ncclDouble,
                                        from, use.coll, c->main));
    }
    CM_NCCL(g_api.GroupEnd());
    return 0;
}

}  // namespace cm

using namespace cm;

extern "C" int cudamat_rccl_available(void) { return load_rccl() == CUDAMAT_OK ? 1 : 0; }

extern "C" int cudamat_rccl_unique_id(void *id)
{
    CM_ARG(id, "id is NULL");
    CM_TRY(load_rccl());
    static_assert(CUDAMAT_RCCL_ID_BYTES == 3 * sizeof(ncclUniqueId), "three communicators, three ids");
    for (int k = 0; k < 3; k++) {
        ncclUniqueId a;
        CM_NCCL(g_api.GetUniqueId(&a));
        memcpy((char *)id + (size_t)k * sizeof(a), &a, sizeof(a));
    }
    return CUDAMAT_OK;
}

extern "C" int cudamat_rccl_comm_create(cudamat_ctx *ctx, const void *id, int rank, int world, cudamat_comm *out)
{
    CM_ARG(ctx && id && out, "null pointer");
    CM_ARG(world >= 1 && rank >= 0 && rank < world, "rank / world");
    CM_TRY(load_rccl());
    CM_HIP(hipSetDevice(ctx->device));
    RcclComm *c = new RcclComm();
    c->rank = rank;
    c->world = world;
    c->device = ctx->device;
    c->main = ctx->stream;
    int rc = CUDAMAT_OK;
    do {
        if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&c->rstream, hipStreamNonBlocking) != hipSuccess) {
            set_error("hipStreamCreate failed for the communicator's stream");
            rc = CUDAMAT_ERR_HIP;
            break;
        }
        ncclUniqueId a, b, d;
        memcpy(&a, id, sizeof(a));
        memcpy(&b, (const char *)id + sizeof(a), sizeof(b));
        memcpy(&d, (const char *)id + 2 * sizeof(a), sizeof(d));
        ncclResult_t r = g_api.CommInitRank(&c->coll, world, a, rank);
        if (r == ncclSuccess) r = g_api.CommInitRank(&c->p2p, world, b, rank);
        if (r == ncclSuccess) r = g_api.CommInitRank(&c->red, world, d, rank);
        if (r != ncclSuccess) {
            set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_api.GetErrorString(r));
            rc = CUDAMAT_ERR_COMM;
            break;
        }
    } while (0);
    if (rc != CUDAMAT_OK) {
        if (c->coll) g_api.CommDestroy(c->coll);
        if (c->p2p) g_api.CommDestroy(c->p2p);
        if (c->red) g_api.CommDestroy(c->red);
        if (c->side) CM_DROP(hipStreamDestroy(c->side));
        if (c->rstream) CM_DROP(hipStreamDestroy(c->rstream));
        delete c;
        return rc;
    }
    out->rank = rank;
    out->world = world;
    out->user = c;
    out->allgather = rccl_allgather;
    out->allreduce = rccl_allreduce;
    out->gather_part = rccl_gather_part;
    out->comm_stream = c->side;
    out->allreduce_side = rccl_allreduce_side;
    out->reduce_stream = c->rstream;
    out->gather_window = rccl_gather_window;
    return CUDAMAT_OK;
}

// wait for a stream for at most `seconds` (after an abort its kernels are told to leave; a wedged device must not
// turn a reported failure into a hang)
static bool bounded_sync(hipStream_t st, double seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    while (hipStreamQuery(st) == hipErrorNotReady) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    return true;
}

// Abort: tear the three communicators down WITHOUT waiting for their streams -- collectives already enqueued (here or,
// once every rank of the job has aborted, on the peers) stop waiting for this rank and return.  For a rank that
// failed while its peers sit inside a collective (csrc/sharded.cpp); callable from any thread; idempotent.
extern "C" int cudamat_rccl_comm_abort(cudamat_comm *comm)
{
    if (!comm || !comm->user) return CUDAMAT_OK;
    CM_ARG(comm->allgather == rccl_allgather, "not a communicator made by cudamat_rccl_comm_create");
    RcclComm *c = (RcclComm *)comm->user;
    ncclComm_t h[3];
    int device;
    {
        std::unique_lock<std::mutex> lock(c->mu);
        device = c->device;
        if (c->aborted) return CUDAMAT_OK;
        c->aborted = true;               // no new enqueue gets a handle from here on
        // Enqueues return within microseconds; one that does not is blocked INSIDE RCCL waiting for a rank that will never
        // come (connection set-up of a first send/recv), and ncclCommAbort from another thread is what RCCL offers to
        // release it -- so the wait is bounded, and only in that case is a handle aborted while a call still holds it.
        c->cv.wait_for(lock, std::chrono::seconds(2), [c] { return c->inflight == 0; });
        h[0] = c->coll; h[1] = c->p2p; h[2] = c->red;
        c->coll = c->p2p = c->red = nullptr;
    }
    CM_DROP(hipSetDevice(device));
    for (ncclComm_t q : h)
        if (q) g_api.CommAbort(q);
    return CUDAMAT_OK;
}

extern "C" int cudamat_rccl_comm_destroy(cudamat_comm *comm)
{
    if (!comm || !comm->user) return CUDAMAT_OK;
    CM_ARG(comm->allgather == rccl_allgather, "not a communicator made by cudamat_rccl_comm_create");
    RcclComm *c = (RcclComm *)comm->user;
    CM_DROP(hipSetDevice(c->device));
    bool was_aborted;
    { std::lock_guard<std::mutex> lock(c->mu); was_aborted = c->aborted; }
    if (was_aborted) {              // (the communicators are gone; their kernels were told to leave)
        bounded_sync(c->main, 5.0);
        bounded_sync(c->side, 5.0);
        bounded_sync(c->rstream, 5.0);
    } else {
        CM_DROP(hipStreamSynchronize(c->main));
        CM_DROP(hipStreamSynchronize(c->side));
        CM_DROP(hipStreamSynchronize(c->rstream));
    }
    if (c->coll) g_api.CommDestroy(c->coll);
    if (c->p2p) g_api.CommDestroy(c->p2p);
    if (c->red) g_api.CommDestroy(c->red);
    if (c->side) CM_DROP(hipStreamDestroy(c->side));
    if (c->rstream) CM_DROP(hipStreamDestroy(c->rstream));
    delete c;
    memset(comm, 0, sizeof(*comm));
    return CUDAMAT_OK;
}
