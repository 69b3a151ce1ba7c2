// loops.hip -- the iteration loops of cudamat_solver_solve: a host that only ENQUEUES work.
//
// Reference behaviour restated (citations into /root/reference):
//   CUDAMAT_LOOP_PBICGSTAB  = gpu_pbicgstab  pbicgstab.cu:45-154  (ILU(0) or M = I)
//   CUDAMAT_LOOP_PBICGSTAB2 = gpu_pbicgstab2 pbicgstab.cu:581-754 (d variant; with d == NULL the intended maths of
//                             :425-578)
//   CUDAMAT_LOOP_PIPELINED  = the same recurrences re-arranged (Cools & Vanroose 2017, Alg. 4): not a reference loop
//
// One solve = Solve::setup (validation, set-up on first use, r0 and the LoopState: pbicgstab.cu:67-74), then ONE of
// the loop forms, then Solve::finish (exit bookkeeping, statistics):
//   run_resident      the whole loop in one launch with grid barriers   (<= 128 stream tiles: C2)
//   iterate_fused     three launches per iteration                      (short rows, <= 3e5 rows, one GPU, M = I)
//   iterate_reference five launches per iteration                       (everything else; + M^-1, + exchanges when sharded)
//   iterate_pipelined four launches per iteration, reductions beside the SpMVs (sharded runs)
// The host-side loop (run_host_loop) is shared by the last three: it enqueues iteration k and looks at the progress
// word iteration k - kLag published through pinned memory, so the stream never drains; once a stopping test fires on
// the device every later kernel returns immediately ("freeze on exit"), so the lagged look costs no accuracy and the
// iterate is exactly the one the reference would return.
#include <chrono>
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

namespace {

struct Solve {
    // ---- the call
    cudamat_solver *s;
    const double *b;
    double *x;
    int precond, loop, maxit;
    double tol;
    int flags;
    double abs_tol;
    // ---- set by setup()
    const Config *cfg = nullptr;
    hipStream_t st = nullptr;
    int n = 0;
    bool profile = false, sharded = false, perm = false, pipelined = false, pipe_pc = false;
    double *x_user = nullptr;          // the caller's x (x itself is the level-major copy while perm)
    int hist_base = 0;
    LoopArgs la{};
    LoopArgs la_none{nullptr, nullptr, 0, 0, 0};
    ScalarSrc nosrc{nullptr, 0, 1};
    size_t pe = 0;                     // profiling events used
    double t_begin = 0.0, t_loop0 = 0.0, t_loop1 = 0.0;
    int np_full = 0, np_half = 0, np_a = 0, np_b = 0;
    ScalarSrc full_src{nullptr, 0, 1};
    const double *pw_last = nullptr;   // reference loop: the M^-1 p of the last half step (x += alpha pw is applied by k_full)
    int loop_form = 0;                 // 0 five launches, 1 three (fused), 2 one (resident)
    // fused / resident loops: p, v and r are double-buffered
    double *p_a = nullptr, *p_b = nullptr, *v_a = nullptr, *v_b = nullptr;
    // pipelined loop
    int pipe_rr = kPipeRR;
    ScalarSrc pipeB_src{nullptr, 0, 1};
    hipStream_t rst = nullptr;         // the communicator's reduce stream (sharded runs)

    void prof_mark() { if (profile && hipEventRecord(prof_event(s, pe++), st) != hipSuccess) s->prof_failed = true; }

    int setup();
    int pipelined_prologue();
    bool wants_fused() const;
    bool wants_resident();
    int run_resident(bool *gave_up);
    int run_host_loop();
    int iterate_reference();
    int iterate_fused();
    int iterate_pipelined(int k);
    int pipe_reduce(ScalarSrc parts, int K, double *out, int slot);
    int pipe_wait(int slot);
    int finish(bool *precond_gave_up, cudamat_stats *out);
};

// ---- validation, set-up on first use, index spaces, history, r0 and the LoopState (pbicgstab.cu:67-74)
int Solve::setup()
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
    cfg = &s->ctx->cfg;
    t_begin = now_s();
    st = s->ctx->stream;
    n = s->n;
    x_user = x;
    {
        int rc_setup = ensure_work(s);
        // (a solver whose SpMV form is still undecided -- the drop-in call leaves it so when the preconditioned loop will run
        // in the level-major spaces on its own copy of the matrix -- gets its factors first: the choice below may not be needed)
        const bool defer_form = s->spmv_mode < 0 && precond == CUDAMAT_PRECOND_ILU0 && !s->sharded;
        if (rc_setup == CUDAMAT_OK && !defer_form) rc_setup = ensure_spmv_mode(s);
        if (rc_setup == CUDAMAT_OK && precond && (!s->has_ilu || (s->sharded && !s->ilu_block)))
            rc_setup = ilu0_setup(s, precond == CUDAMAT_PRECOND_BLOCK_ILU0);
        CM_TRY(setup_agree(s, rc_setup));        // sharded: every rank learns of a failure on any rank
    }
    // The loop in LEVEL-MAJOR SPACES.  With the hybrid triangular solves the factors live in level-major index spaces: L
    // reads and writes streams in L's order, U writes a stream in U's order.  The reference loop (pbicgstab.cu:45-154)
    // only ever combines vectors element by element within two families -- r, rw, p, v, t (residual side: outputs of A,
    // inputs of L) and M^-1 p, M^-1 r, x (solution side: outputs of U, inputs of A) -- so the first family is kept in L's
    // order, the second in U's, and A is stored with rows in L's order and columns in U's positions (ilu_perm_matrix).
    // Then no vector is permuted inside the loop: per M^-1 application the only indexed access left besides the near
    // gathers is U reading its right-hand side from L's space (1 per row instead of 4).  b and x0 are permuted on the way
    // in, x on the way out.  One GPU, reference loop; the option TRSV_PERM = 0 disables.
    perm = precond == CUDAMAT_PRECOND_ILU0 && !s->sharded && loop == CUDAMAT_LOOP_PBICGSTAB && s->L.lm && s->U.lm && !s->d &&
           cfg->trsv_perm && !s->perm_failed;
    if (perm && !s->perm_ready) {
        const int rcp = ilu_perm_matrix(s);
        if (rcp == CUDAMAT_ERR_NOMEM) { perm = false; s->perm_failed = true; }    // no room for the second blocked copy: permute per
        else CM_TRY(rcp);                                                        // application, and do not try again on every solve
    }
    s->perm_active = perm;
    if (!perm && s->spmv_mode < 0) CM_TRY(ensure_spmv_mode(s));     // (deferred above; the loop runs in the original space after all)
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
    hist_base = abs_tol > 0.0 ? (s->hist_count < s->hist_cap ? s->hist_count : s->hist_cap) : 0;
    if (hist_base == 0 && need_hist > s->hist_cap) {
        if (s->hist) { CM_HIP(hipStreamSynchronize(st)); CM_DROP(hipFree(s->hist)); s->hist = nullptr; }
        CM_TRY(dev_alloc((void **)&s->hist, sizeof(double) * (size_t)need_hist));
        s->hist_cap = need_hist;
    }
    if (s->hist_cap > hist_base)
        CM_HIP(hipMemsetAsync(s->hist + hist_base, 0xFF, sizeof(double) * (size_t)(s->hist_cap - hist_base), st));  // NaN fill
    s->last_loop = loop;
    profile = (flags & CUDAMAT_FLAG_PROFILE) != 0;
    sharded = s->sharded;
    pipelined = loop == CUDAMAT_LOOP_PIPELINED;
    pipe_pc = pipelined && precond != CUDAMAT_PRECOND_NONE;
    la = LoopArgs{s->st, s->hist + hist_base, s->hist_cap - hist_base, loop, (flags & CUDAMAT_FLAG_NO_EXIT) ? 1 : 0, s->snap_dev, kRing, 0};
    for (int i = 0; i < kRing; i++) s->snap_host[i] = 0ULL;

    s->comm_used = 0;
    s->comm_kind.clear();
    s->profiling = profile && sharded;
    s->prof_failed = false;
    t_loop0 = now_s();
    if (flags & CUDAMAT_FLAG_X0_ONES) CM_TRY(launch_fill(st, n, 1.0, x));
    // r = A x0 (pbicgstab.cu:67 / :645-646); x may be a caller buffer without pad
    CM_HIP(hipMemcpyAsync(s->pw, x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
    CM_TRY(spmv_local(s, s->pw, s->r, 0, nullptr, nullptr, la_none, CHECK_NONE, nosrc));
    CM_TRY(launch_init(st, n, b, s->r, s->rw, s->p, s->parts_full, &np_full));   // :69-74
    full_src = ScalarSrc{s->parts_full, np_full, 2};
    if (sharded) {
        CM_TRY(launch_reduce_parts(st, full_src, 2, s->red + 4, 0));
        CM_TRY(allreduce(s, s->red + 4, 2));
        full_src = ScalarSrc{s->red + 4, 0, 1};
    }
    CM_TRY(launch_init_finish(st, s->st, full_src, tol, abs_tol));
    return CUDAMAT_OK;
}

// ============================================================================================ pipelined BiCGStab
// Extra vectors, w0 = A rh0 (with rw.w0), t0 = A wh0, and the seed [rw.r0, rw.w0, 0, 0, r0.r0] of the first k_pipe_a.
// A reduction phase = the per-workgroup partials of a kernel summed (and, sharded, all-reduced) into red_pipe: on the
// communicator's reduce stream when it has one, so that it runs beside the SpMV that follows the kernel.  With a
// preconditioner (ILU(0), or block-Jacobi ILU(0) when sharded: SURVEY 8 f4) the hatted vectors M^-1 r, M^-1 w, M^-1 s,
// M^-1 z, M^-1 q are carried too and M^-1 is applied in front of each SpMV, where pbicgstab.cu:92-98,121-127 apply it.
int Solve::pipelined_prologue()
{
    // residual replacement period (Cools & Vanroose): every rr-th iteration r, w, s, z (and their hatted forms, and v)
    // are recomputed from x and p, which discards the rounding errors the recurrences have accumulated
    pipe_rr = cfg->pipe_rr >= 0 ? cfg->pipe_rr : kPipeRR;
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
    return CUDAMAT_OK;
}

// one reduction phase of the pipelined loop: partials -> K sums in `out` (all-reduced when sharded).  With a reduce
// stream the work is queued there, behind `slot`'s event.  One GPU: the consumer kernel sums the partials itself.
int Solve::pipe_reduce(ScalarSrc parts, int K, double *out, int slot)
{
    if (!sharded) return CUDAMAT_OK;
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
}

int Solve::pipe_wait(int slot)
{
    if (sharded && rst) CM_HIP(hipStreamWaitEvent(st, s->ev_red_done[slot], 0));
    return CUDAMAT_OK;
}

int Solve::iterate_pipelined(int k)
{
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
        prof_mark();
        CM_TRY(precond_apply(s, s->pz, s->ptmp, s->pzh));
        prof_mark();
    }
    prof_mark();
    CM_TRY(spmv_local(s, zh, s->v, 0, nullptr, nullptr, la, CHECK_NONE, nosrc));              // v = A zh
    prof_mark();
    CM_TRY(pipe_wait(0));
    // half-step test, omega, x, r, rh, w; dots (rw.r, rw.w, rw.s, rw.z, r.r); i++
    CM_TRY(launch_pipe_b(st, la, a_src, n, s->pq, s->py, s->t, s->v, s->rw, s->s, s->pz, s->pxh, x, s->r, s->pww,
                         s->pipeB, &np_b, hat_b));
    if (pipe_rr > 0 && (k + 1) % pipe_rr == 0) {
        // Residual replacement.  q and y are free until the next k_pipe_a; pw is not used by this loop.  Every kernel of
        // this block returns at once when the loop is frozen (`la`), so r and the phase-B partials stay those of the
        // returned iterate; the copy of x only fills the scratch vector pw, the triangular solves only scratch vectors.
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
        prof_mark();
        CM_TRY(precond_apply(s, s->pww, s->ptmp, s->pwh));
        prof_mark();
    }
    prof_mark();
    CM_TRY(spmv_local(s, wh, s->t, 0, nullptr, nullptr, la, CHECK_NONE, nosrc));              // t = A wh
    prof_mark();
    CM_TRY(pipe_wait(1));
    return CUDAMAT_OK;
}

// ============================================================================================ small systems
// Vectors resident in L2: three launches per iteration instead of five -- the vector updates in front of the two SpMVs
// are folded into them (small_loops.hip); p, v and r are double-buffered.
// Measured (bench.py, one MI355X): 5-point stencil rows 38.5 -> 46.3 k it/s at 1e4 rows, 37.7 -> 43.8 k at 4e4, 30.4 ->
// 32.1 k at 1.6e5, even at 4.9e5, slower beyond; with 50 entries per row the three gathers per entry cost more than the
// two launches save (29.9 -> 24.1 k it/s at 2e4 rows).  So: short rows (the stream-tile plan) up to 3e5 rows.  The option
// FUSED = 0 disables, FUSED = N forces it for every supported plan up to N rows.
bool Solve::wants_fused() const
{
    const bool forced = cfg->fused >= 0;
    const long long max_rows = forced ? cfg->fused : 300000;
    return loop != CUDAMAT_LOOP_PIPELINED && !sharded && !precond && s->spmv_mode == 0 && fused_spmv_supported(s->plan) && n > 0 &&
           n <= max_rows && (forced || s->plan.stream_rows > 0);
}

// Very small systems (one stream tile per workgroup, at most one workgroup per two compute units): the whole loop in ONE
// launch, grid barriers instead of launch boundaries (small_loops.hip).  The option RESIDENT = 0 disables.
bool Solve::wants_resident()
{
    if (loop_form != 1 || profile || s->resident_off || !cfg->resident || !resident_loop_supported(s->plan, n)) return false;
    // all workgroups must be resident at once: at most one per two compute units of THIS device
    if (s->device_cus == 0 && hipDeviceGetAttribute(&s->device_cus, hipDeviceAttributeMultiprocessorCount, s->ctx->device) != hipSuccess)
        s->device_cus = -1;
    return s->device_cus > 0 && 2 * s->plan.grid <= s->device_cus;
}

int Solve::run_resident(bool *gave_up)
{
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
        q.spin_limit = cfg->resident_spin_limit >= 0 ? (unsigned)cfg->resident_spin_limit : 1u << 22;
        q.p_a = p_a; q.p_b = p_b; q.v_a = v_a; q.v_b = v_b; q.r = s->r; q.s = s->s; q.t = s->t; q.x = x; q.rw = s->rw;
        q.parts_rv = s->parts_rv; q.parts_tt = s->parts_tt; q.parts_half = s->parts_half; q.parts_full = s->parts_full;
        CM_TRY(launch_resident_loop(st, s->plan, a, q));
        unsigned bar_host[2] = {0u, 0u};
        CM_HIP(hipMemcpyAsync(&s->st_ring[0], s->st, sizeof(LoopState), hipMemcpyDeviceToHost, st));
        CM_HIP(hipMemcpyAsync(bar_host, s->bar, sizeof(bar_host), hipMemcpyDeviceToHost, st));
        CM_HIP(hipStreamSynchronize(st));
        if (bar_host[1] != 0u) {         // a barrier wait ran into its bound: this attempt is void
            *gave_up = true;
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
    return CUDAMAT_OK;
}

int Solve::iterate_fused()
{
    SpmvArgs a{};
    a.n = n; a.rp = s->rp; a.ci = s->ci; a.val = s->val; a.x = nullptr; a.d = s->d; a.xd = nullptr;
    a.alpha = 1.0; a.beta = 0.0; a.loop = la; a.check = CHECK_NONE; a.half = nosrc;
    const int np = plan_spmv_parts(s->plan);
    // rho, beta, full-step test, p' = r + beta (p - omega v), v' = A p', rw.v'            :80-89, :104-106
    FuseArgs f1{};
    f1.mode = 1; f1.r = s->r; f1.p_old = p_a; f1.v_old = v_a; f1.p_out = p_b; f1.src = full_src;
    a.y = v_b; a.dot = 1; a.w = s->rw; a.parts = s->parts_rv;
    prof_mark();
    CM_TRY(launch_fused_spmv(st, s->plan, a, f1));
    prof_mark();
    // alpha, s = r - alpha v', x += alpha p', t = A s, (t.s, t.t), ||s||^2                :107-111, :132-136
    FuseArgs f2{};
    f2.mode = 2; f2.r = s->r; f2.v = v_b; f2.s_out = s->s; f2.xsol = x; f2.p = p_b;
    f2.src = ScalarSrc{s->parts_rv, np, 2}; f2.parts_half = s->parts_half;
    a.y = s->t; a.dot = 2; a.w = nullptr; a.parts = s->parts_tt;
    prof_mark();
    CM_TRY(launch_fused_spmv(st, s->plan, a, f2));
    prof_mark();
    // half-step test, omega, x += omega s, r = s - omega t, (rw.r, ||r||^2), i++          :116, :137-151
    CM_TRY(launch_full(st, la, ScalarSrc{s->parts_tt, np, 2}, n, x, s->s, s->s, s->t, s->rw, s->parts_full, &np_full,
                       ScalarSrc{s->parts_half, np, 1}));
    full_src = ScalarSrc{s->parts_full, np_full, 2};
    std::swap(p_a, p_b);
    std::swap(v_a, v_b);
    std::swap(s->r, s->s);        // the new residual was written over s
    return CUDAMAT_OK;
}

// ============================================================================================ the reference loop
// One iteration of pbicgstab.cu:80-151 (gpu_pbicgstab) / :665-747 (gpu_pbicgstab2) as five launches:
//   k_update_p | [M^-1] SpMV (+ rw.v) | k_half | [M^-1] SpMV (+ t.r, t.t; half-step test in its prologue) | k_full
int Solve::iterate_reference()
{
    // rho, beta, p = r + beta (p - omega v)                     :80-89
    CM_TRY(launch_update_p(st, la, full_src, n, s->r, s->p, s->v));
    const double *pw = s->p;
    if (precond) {                                            // :92-98
        prof_mark();
        CM_TRY(precond_apply(s, s->p, s->t, s->pw, perm));
        prof_mark();
        pw = s->pw;
    }
    // v = A pw, rw.v                                            :104-106
    prof_mark();
    CM_TRY(spmv_local(s, pw, s->v, 1, s->rw, s->parts_rv, la, CHECK_NONE, nosrc));
    prof_mark();
    ScalarSrc rv_src{s->parts_rv, spmv_parts(s), 2};
    if (sharded) {
        CM_TRY(launch_reduce_parts(st, rv_src, 1, s->red + 0, 0));
        CM_TRY(allreduce(s, s->red + 0, 1));
        rv_src = ScalarSrc{s->red + 0, 0, 1};
    }
    // alpha, r -= alpha v, ||r||                                :107-111
    // (x += alpha pw, :110, rides in k_full -- x is streamed once per iteration, not twice; an exit at the half step
    // applies it after the loop, finish())
    CM_TRY(launch_half(st, la, rv_src, n, s->r, s->v, s->parts_half, &np_half));
    pw_last = pw;
    const ScalarSrc half_src{s->parts_half, np_half, 1};
    const double *sv = s->r;
    ScalarSrc tt_src{s->parts_tt, spmv_parts(s), 2};
    if (!sharded) {
        if (precond) {                                        // :116, :121-127
            CM_TRY(launch_check(st, la, half_src, CHECK_HALF));
            prof_mark();
            CM_TRY(precond_apply(s, s->r, s->t, s->s, perm));
            prof_mark();
            sv = s->s;
            prof_mark();
            CM_TRY(spmv_local(s, sv, s->t, 2, s->r, s->parts_tt, la, CHECK_NONE, nosrc));
            prof_mark();
        } else {
            // half-step test fused into the SpMV prologue      :116, :132-136
            prof_mark();
            CM_TRY(spmv_local(s, sv, s->t, 2, s->r, s->parts_tt, la, CHECK_HALF, half_src));
            prof_mark();
        }
    } else {
        // The SpMV changes only t, so the half-step test may ride with the (t.r, t.t) all-reduce: one collective
        // instead of two.
        CM_TRY(launch_reduce_parts(st, half_src, 1, s->red + 1, 0));
        if (precond) {   // block-Jacobi: local triangular solves, no collective (they only write s and t, so an
                         // exit at the half step, noticed after the all-reduce below, leaves x and r untouched)
            prof_mark();
            CM_TRY(precond_apply(s, s->r, s->t, s->s));
            prof_mark();
            sv = s->s;
        }
        prof_mark();
        CM_TRY(spmv_local(s, sv, s->t, 2, s->r, s->parts_tt, la, CHECK_NONE, nosrc));
        prof_mark();
        CM_TRY(launch_reduce_parts(st, tt_src, 2, s->red + 2, 0));
        CM_TRY(allreduce(s, s->red + 1, 3));
        CM_TRY(launch_check(st, la, ScalarSrc{s->red + 1, 0, 1}, CHECK_HALF));
        tt_src = ScalarSrc{s->red + 2, 0, 1};
    }
    // omega, x += alpha pw, x += omega s, r -= omega t, (rw.r, ||r||), i++     :110, :137-151
    CM_TRY(launch_full(st, la, tt_src, n, x, sv, s->r, s->t, s->rw, s->parts_full, &np_full, ScalarSrc{nullptr, 0, 1}, pw));
    full_src = ScalarSrc{s->parts_full, np_full, 2};
    if (sharded) {
        CM_TRY(launch_reduce_parts(st, full_src, 2, s->red + 4, 0));
        CM_TRY(allreduce(s, s->red + 4, 2));
        full_src = ScalarSrc{s->red + 4, 0, 1};
    }
    return CUDAMAT_OK;
}

// ============================================================================================ the host side of a loop
int Solve::run_host_loop()
{
    for (int k = 0; k < maxit; k++) {
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
        if (pipelined) CM_TRY(iterate_pipelined(k));
        else if (loop_form == 1) CM_TRY(iterate_fused());
        else CM_TRY(iterate_reference());
    }
    return CUDAMAT_OK;
}

// ---- the last stopping test, exit bookkeeping, statistics
int Solve::finish(bool *precond_gave_up, cudamat_stats *out)
{
    // the full-step test of the last iteration has not been looked at yet
    if (pipelined)      // (check_full wants (., r.r): the last two of the five phase-B scalars)
        CM_TRY(launch_check(st, la, ScalarSrc{pipeB_src.ptr + 3, pipeB_src.count, pipeB_src.stride}, CHECK_FULL));
    else
        CM_TRY(launch_check(st, la, full_src, CHECK_FULL));
    CM_HIP(hipMemcpyAsync(&s->st_ring[0], s->st, sizeof(LoopState), hipMemcpyDeviceToHost, st));
    CM_HIP(hipStreamSynchronize(st));                              // :372
    t_loop1 = now_s();
    // A dependency-driven triangular solve that gave up waiting (another spin-waiting kernel shared the GPU)
    // invalidates this attempt; in a sharded run every rank must learn of it.
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
    if (pw_last && s->st_ring[0].state == 1) {      // left through the half step: pbicgstab.cu:110 is still due
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
    stt.spmv_mode = s->spmv_mode < 0 && s->perm_active ? 1 : s->spmv_mode;      // (the permuted copy is a blocked two-phase copy)
    if (s->profiling) {
        // exposed part of an overlapped gather: the waits (kind 1), clipped to the gather they wait for only by
        // construction -- the solver's stream idles there for nothing else
        for (size_t i = 0; i < s->comm_kind.size() && 2 * i + 1 < s->comm_used; i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s->comm_ev[2 * i], s->comm_ev[2 * i + 1]) != hipSuccess) s->prof_failed = true;
            switch (s->comm_kind[i]) {
            case 0: stt.ms_gather += ms; stt.n_gather++; break;
            case 2: stt.ms_gather += ms; stt.ms_gather_exposed += ms; stt.n_gather++; break;
            default: stt.ms_allreduce += ms; stt.n_allreduce++; break;
            }
        }
    }
    if (profile) {
        // events come in (start, stop) pairs; trsv pairs and spmv pairs alternate as recorded
        const int per_it_pairs = precond ? 4 : 2;
        for (size_t i = 0; i + 1 < pe; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s->prof_ev[i], s->prof_ev[i + 1]) != hipSuccess) s->prof_failed = true;
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

    if (s->prof_failed) {          // an event call failed: no timing is better than a wrong one
        stt.ms_spmv = stt.ms_trsv = stt.ms_gather = stt.ms_gather_exposed = stt.ms_allreduce = 0.0;
        stt.n_spmv = stt.n_trsv = stt.n_gather = stt.n_allreduce = 0;
        if (s->ctx->cfg.verbose) fprintf(stderr, "[cudamat] per-launch timing dropped: a hipEvent call failed\n");
    }

    if (flags & CUDAMAT_FLAG_DEBUG) {
        std::vector<double> h((size_t)(s->hist_count > 0 ? s->hist_count : 1));
        if (s->hist_count > 0)
            CM_HIP(hipMemcpy(h.data(), s->hist, sizeof(double) * (size_t)s->hist_count, hipMemcpyDeviceToHost));
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

// One attempt at a solve.  *precond_gave_up / *resident_gave_up: the attempt is void (a bounded wait of a
// dependency-driven triangular solve / of a grid barrier ran out) and the caller redoes it in another form.
int solve_once(cudamat_solver *s, const double *b, double *x, int precond, int loop, int maxit, double tol,
               int flags, cudamat_stats *out, bool *precond_gave_up, bool *resident_gave_up, double abs_tol)
{
    *precond_gave_up = *resident_gave_up = false;
    Solve q{s, b, x, precond, loop, maxit, tol, flags, abs_tol};
    struct Off { cudamat_solver *s; ~Off() { if (s) { s->perm_active = false; s->profiling = false; } } } off{s};
    CM_TRY(q.setup());
    Range range_loop("cudamat: iteration loop (enqueue + lagged checks)");
    if (q.pipelined) CM_TRY(q.pipelined_prologue());
    q.loop_form = q.wants_fused() ? 1 : 0;
    if (q.loop_form == 1 && !s->v2) {
        const size_t nb = sizeof(double) * (size_t)(s->n_pad > 0 ? s->n_pad : 1);
        CM_TRY(dev_alloc((void **)&s->v2, nb));
        CM_HIP(hipMemsetAsync(s->v2, 0, nb, q.st));
    }
    q.p_a = s->p; q.p_b = s->pw; q.v_a = s->v; q.v_b = s->v2;
    if (q.wants_resident()) {
        CM_TRY(q.run_resident(resident_gave_up));
        if (*resident_gave_up) return CUDAMAT_OK;
    } else {
        CM_TRY(q.run_host_loop());
    }
    return q.finish(precond_gave_up, out);
}

int solve_guarded(cudamat_solver *s, const double *b, double *x, int precond, int loop, int maxit, double tol,
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
        if (s->ctx->cfg.verbose)
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
    if (s->ctx->cfg.verbose)
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

}  // namespace

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
