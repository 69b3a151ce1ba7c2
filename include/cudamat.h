/*
 * cudamat.h -- C ABI of libcudamat_hip.so: the MI355X (gfx950) BiCGSTAB hot path.
 *
 * This is the drop-in boundary for the reference's solver path.  Every entry
 * point takes plain pointers and sizes (no C++/torch types) and returns an
 * int status (CUDAMAT_OK == 0); nothing here ever calls exit() (the reference
 * does, helper_cuda.h:999-1014).  cudamat_last_error() describes the last
 * failure on the calling thread.  There is NO CPU fallback: without a HIP
 * device every compute entry point fails with CUDAMAT_ERR_HIP.
 *
 * What each group replaces in /root/reference:
 *   cudamat_solve                 bicgstab / bicgstab / bicgstab_lu_precond
 *                                 (pbicgstab.h:113,116,119; pbicgstab.cu:157-409,
 *                                 756-922, 926-1088): host CSR in, host x out.
 *   cudamat_solver_*              the same solve with everything resident in HBM
 *                                 (what pbicgstab.cu:365-374 times as dtAlg), plus
 *                                 row-sharded multi-GPU operation through two
 *                                 caller-supplied collectives (new: the reference
 *                                 is single-GPU).
 *   cudamat_spmv                  cusparseDcsrmv      pbicgstab.cu:67,104,132,469,
 *                                                     501,528,646,676,704 (+ the
 *                                                     mult_spec diagonal term :36-42)
 *   cudamat_dot / cudamat_nrm2    cublasDdot/Dnrm2    pbicgstab.cu:74,81,106,111,...
 *   cudamat_axpy / cudamat_scal   cublasDaxpy/Dscal   pbicgstab.cu:69-70,86-88,...
 *   cudamat_ilu0 / cudamat_trsv   cusparseDcsrilu0 / cusparseDcsrsv_analysis+solve
 *                                 (via cudamat_solver_ilu0 / _precond_apply)
 *   cudamat_load_mtx              loadMMSparseMatrix  mmio_wrapper.h:133-348
 *   cudamat_to_dense_vector       toDenseVector       pbicgstab.cu:1101-1115
 *   cudamat_gen_*                 synthetic inputs of SURVEY section 8d (the
 *                                 reference's O(n^2) rand() generators,
 *                                 pbicgstab.h:32-76, cannot reach 1e7 rows)
 *
 * CSR conventions are the reference's: fp64 values, int32 indices, index base
 * taken from rowptr[0] (0 or 1, pbicgstab.cu:201), columns strictly increasing
 * inside a row (mmio_wrapper.h:123).
 */
#ifndef CUDAMAT_H
#define CUDAMAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CUDAMAT_VERSION 1

/* status codes */
#define CUDAMAT_OK              0
#define CUDAMAT_ERR_HIP         1   /* HIP runtime / no device                      */
#define CUDAMAT_ERR_ARG         2   /* bad argument                                  */
#define CUDAMAT_ERR_ZERO_PIVOT  3   /* ILU(0): zero or structurally missing diagonal */
#define CUDAMAT_ERR_NOMEM       4
#define CUDAMAT_ERR_IO          5
#define CUDAMAT_ERR_COMM        6   /* a caller-supplied collective failed           */

/* preconditioner (cudamat_solve / cudamat_solver_solve) */
#define CUDAMAT_PRECOND_NONE    0
#define CUDAMAT_PRECOND_ILU0    1
#define CUDAMAT_PRECOND_BLOCK_ILU0 2 /* block-Jacobi: ILU(0) of each rank's diagonal block.
                                       Not in the reference (single GPU); with one rank it IS
                                       CUDAMAT_PRECOND_ILU0, with more it is a different (weaker)
                                       preconditioner with its own parity statement.            */

/* which reference loop's stopping rules to follow */
#define CUDAMAT_LOOP_PBICGSTAB  0   /* pbicgstab.cu:45-154: half-step + full-step exits
                                       against tol*||r0||, no breakdown guard           */
#define CUDAMAT_LOOP_PBICGSTAB2 1   /* pbicgstab.cu:581-754: end-of-iteration exit,
                                       |omega| < 1e-5 / NaN guard => not converged      */
#define CUDAMAT_LOOP_PIPELINED  2   /* pipelined BiCGStab (Cools & Vanroose 2017, Alg. 4): the recurrences of
                                     * LOOP_PBICGSTAB re-arranged so that both reduction phases of an iteration run
                                     * while an SpMV runs; same stopping rules and history layout;
                                     * no preconditioner; iterates equal BiCGSTAB's up to rounding              */

/* flags for cudamat_solver_solve */
#define CUDAMAT_FLAG_DEBUG      1   /* print the reference's debug lines                */
#define CUDAMAT_FLAG_PROFILE    2   /* bracket every SpMV / trsv launch with HIP events */
#define CUDAMAT_FLAG_NO_EXIT    4   /* evaluate the stopping tests but never leave:
                                       fixed-length timing windows (bench.py)           */
#define CUDAMAT_FLAG_X0_ONES    8   /* start from x0 = 1 (pbicgstab.cu:306-308,827-831)
                                       instead of the contents of x                     */

typedef struct cudamat_ctx cudamat_ctx;         /* device + stream + reduction workspace  */
typedef struct cudamat_solver cudamat_solver;   /* one HBM-resident linear system          */

typedef struct cudamat_stats {
    int iters;          /* the reference loop counter on exit                            */
    int half_exit;      /* left through the half-step test (pbicgstab.cu:116)            */
    int converged;      /* a stopping test fired                                         */
    int breakdown;      /* omega guard fired (pbicgstab.cu:735)                          */
    double nrm0;        /* ||r0||                                                        */
    double nrm;         /* last residual norm the loop computed                          */
    double t_analysis;  /* s, level analysis           (pbicgstab.cu:335-347)            */
    double t_factor;    /* s, ILU(0) factorisation     (pbicgstab.cu:356-363), the factors' layout and, at the first
                         * preconditioned solve of a big system, the matrix copy in the factors' index spaces */
    double t_solve;     /* s, iteration loop = dtAlg   (pbicgstab.cu:365-374)            */
    double t_total;     /* s, whole call incl. H2D/D2H                                   */
    /* CUDAMAT_FLAG_PROFILE: device time by kernel class inside the loop                */
    double ms_spmv;     int n_spmv;
    double ms_trsv;     int n_trsv;
    int n_levels_l;     int n_levels_u;
    /* triangular-solve form this solve ended with: 0 one launch per level, 1 dependency-driven
     * launches (rows wait inside the launch), 2 one workgroup with the vector in LDS          */
    int trsv_form;
    /* solves of this solver that were discarded and redone level by level because a wait of a
     * dependency-driven launch ran into its bound (0 in a healthy run; the result is the same) */
    int trsv_fallbacks;
    /* CUDAMAT_FLAG_PROFILE on a row-sharded solver: device time of the exchanges of this solve.
     * ms_gather: the all-gathers of the SpMV inputs (on the communicator's stream when the
     * gather is overlapped); ms_gather_exposed: the part of it that was not hidden behind
     * the SpMV (all of it without overlap); hidden = ms_gather - ms_gather_exposed.         */
    int n_gather;       int n_allreduce;
    double ms_gather;   double ms_gather_exposed;   double ms_allreduce;
    int overlapped;     /* 1: the gather ran in pieces behind phase 1 of the blocked SpMV; 2: only the windows
                         * of the other slices that this rank's rows reference were exchanged (halo)            */
    int loop_form;      /* how the iterations were issued: 0 five launches each, 1 three (vector updates folded into the
                         * SpMVs, small systems), 2 the whole loop in ONE launch with grid barriers (<= 32768 short rows) */
    double gather_fraction; /* doubles received per SpMV / doubles of a whole gather ((world-1) slices)        */
    double ms_spmv_alone;  /* the selected SpMV form with its input in place, as timed when it was selected
                            * (0: never timed); an overlapped gather's exposed part is what the SpMVs of the
                            * loop took beyond this                                                            */
    int loop_fallbacks;    /* solves of this solver that were discarded and redone with the three-launch loop because a
                            * grid barrier of the single-launch loop ran into its bound (GPU shared; 0 in a healthy run) */
    int restarts;          /* pipelined loop only: how often an iterate it called converged failed the check of its TRUE
                            * residual (one SpMV) and the loop was restarted from it (0 in most solves, at most 3)       */
    /* where a call's time went beyond t_analysis / t_factor / t_solve (the reference prints "total delta time" next to
     * "algorithm delta time", example.cpp:364-365; pbicgstab.cu:365-374):                                              */
    double t_upload;       /* s, cudamat_solve / cudamat_solve_sharded: device allocations + host-to-device copies       */
    double t_setup;        /* s, solver creation (copies, validation, CSR launch plan) + choice of the SpMV form incl. the
                            * matrix copies in other layouts; 0 when the plan of the previous call was reused            */
    double t_tune;         /* s, of t_setup: timed candidate launches (0 when the choice needed none)                    */
    int spmv_mode;         /* the SpMV form the solve used: 0 CSR forms, 1 blocked two-phase, 2 SELL-C-sigma             */
    int plan_reused;       /* cudamat_solve: 1 = the previous call brought the same matrix (pattern AND values): its solver
                            * -- device copies, SpMV plan, value dictionary, ILU(0) factors -- was reused; 0 = built anew */
    /* hybrid triangular solves (big factors with wide, scattered levels): groups of consecutive levels per factor; one
     * application of L^-1 (U^-1) = `groups` dependency-driven launches over the entries inside a group + `groups - 1`
     * blocked two-phase SpMVs over the entries whose column lies in an earlier group.  0: the factor is not split.    */
    int trsv_groups_l;     int trsv_groups_u;
} cudamat_stats;

/* Collectives for a row-sharded solve.  Either supplied by the host program (e.g.
 * torch.distributed) or the library's own RCCL binding
 * (cudamat_rccl_comm_create below).  allgather / allreduce are enqueued on / ordered
 * with the context's stream by the callee and must not block the host longer than the
 * enqueue.  Pointers are device pointers.  Return 0 on success.
 *   allgather: every rank contributes `count` doubles; recv holds world*count.
 *   allreduce: in-place sum of `count` doubles.
 *   gather_part (optional, NULL = the gather is never overlapped): one PIECE of the
 *     all-gather.  Every rank calls it with the same (stride, offset, count): rank q's
 *     send[offset .. offset+count) must arrive at recv[q*stride + offset ..) on every
 *     OTHER rank (a rank places its own piece itself).  The exchange is enqueued on
 *     `comm_stream`, a hipStream_t owned by the communicator; the solver orders it
 *     against its own stream with events, so that phase 1 of the blocked SpMV runs on
 *     the pieces that have arrived while the next ones are in flight.                  */
typedef int (*cudamat_allgather_fn)(void *user, const double *send, double *recv, int64_t count);
typedef int (*cudamat_allreduce_fn)(void *user, double *buf, int count);
typedef int (*cudamat_gather_part_fn)(void *user, const double *send, double *recv, int64_t stride,
                                      int64_t offset, int64_t count);
/* optional: a WINDOWED gather for matrices whose rows reference only part of every other slice (banded matrices:
 * a halo).  This rank sends send[send_off[q] .. + send_cnt[q]) to rank q and receives rank q's
 * [recv_off[q] .. + recv_cnt[q]) into recv[q*stride + recv_off[q] ..); the four arrays have `world` entries
 * (host memory, valid during the call; entry `rank` is ignored; counts may be 0).  Enqueued on the context's stream. */
typedef int (*cudamat_gather_window_fn)(void *user, const double *send, double *recv, int64_t stride,
                                        const int64_t *send_off, const int64_t *send_cnt,
                                        const int64_t *recv_off, const int64_t *recv_cnt);
typedef struct cudamat_comm {
    int rank;
    int world;
    void *user;
    cudamat_allgather_fn allgather;
    cudamat_allreduce_fn allreduce;
    cudamat_gather_part_fn gather_part;   /* may be NULL */
    void *comm_stream;                    /* hipStream_t gather_part enqueues on (NULL with gather_part NULL) */
    /* optional: allreduce enqueued on `reduce_stream` (a third stream of the communicator) instead of the
     * context's stream, so that CUDAMAT_LOOP_PIPELINED can run a reduction while an SpMV runs              */
    cudamat_allreduce_fn allreduce_side;  /* may be NULL: the pipelined loop then reduces on its own stream */
    void *reduce_stream;
    cudamat_gather_window_fn gather_window;   /* may be NULL: every SpMV input is gathered whole */
} cudamat_comm;

/* ---- the library's own communicator: RCCL over xGMI, bound at run time ---------------- */
/* (csrc/comm_rccl.hip; the reference has no counterpart: single device, pbicgstab.cu:223-240)
 * One rank per GPU, as threads of one process (host/example.cpp -G<n>) or as processes
 * (bench.py under torch.distributed.run).  Rank 0 makes an id and hands its bytes to the
 * other ranks by whatever channel the host program has; every rank then creates its
 * communicator (collective: returns when all `world` ranks have joined), passes it to
 * cudamat_solver_set_comm and destroys it after the solver.  The collectives run on the
 * context's stream, the pieces of an overlapped gather on a stream the communicator owns.  */
#define CUDAMAT_RCCL_ID_BYTES 384
int cudamat_rccl_available(void);                  /* 1 when librccl could be loaded */
int cudamat_rccl_unique_id(void *id);              /* fills CUDAMAT_RCCL_ID_BYTES bytes */
int cudamat_rccl_comm_create(cudamat_ctx *ctx, const void *id, int rank, int world, cudamat_comm *out);
int cudamat_rccl_comm_destroy(cudamat_comm *comm);
/* Tear the communicator down without waiting for its streams: for a rank that failed while its peers wait inside a
 * collective (every rank of the job aborts, then destroys).  Any thread; idempotent; destroy must still follow.      */
int cudamat_rccl_comm_abort(cudamat_comm *comm);

/* A communicator whose exchanges are SKIPPED (streams and call sequence are real): times one rank's share of a
 * sharded solve on a single GPU (scripts/rank_probe.py).  The iterates of such a solve are meaningless.        */
int cudamat_comm_dry_create(cudamat_ctx *ctx, int rank, int world, cudamat_comm *out);
int cudamat_comm_dry_destroy(cudamat_comm *comm);

int         cudamat_version(void);
const char *cudamat_last_error(void);
int         cudamat_device_count(int *count);

/* ---- context ------------------------------------------------------------------ */
/* stream: a hipStream_t created by the caller (e.g. torch's current stream) or
 * NULL to let the context create and own one.                                       */
int cudamat_ctx_create(int device, void *stream, cudamat_ctx **out);
int cudamat_ctx_destroy(cudamat_ctx *ctx);
/* Switches (testing / probing; the defaults are what ships).  A context reads CUDAMAT_<NAME> from the environment ONCE,
 * when it is created; afterwards a switch changes only through this call (name with or without the CUDAMAT_ prefix,
 * value as it would stand in the environment).  Whatever runs on the context after the call sees the new value: set
 * SpMV-form switches before the first use of a solver, ILU(0) ones before cudamat_solver_ilu0, loop ones before
 * cudamat_solver_solve.  CUDAMAT_ERR_ARG for an unknown name or a value outside the option's range.  cudamat_solve and
 * cudamat_solve_sharded (no caller-made context) read the environment per call.  cudamat_options_help(): one line per
 * switch -- name, accepted values, meaning (the table of csrc/config.cpp).                                            */
int cudamat_ctx_set_option(cudamat_ctx *ctx, const char *name, const char *value);
int cudamat_ctx_reset_options(cudamat_ctx *ctx);   /* back to what a context created now would hold (defaults + environment) */
const char *cudamat_options_help(void);
int cudamat_option_check(const char *name, const char *value);   /* CUDAMAT_OK when cudamat_ctx_set_option would accept the pair (no device needed) */
int cudamat_ctx_sync(cudamat_ctx *ctx);
int cudamat_ctx_stream(cudamat_ctx *ctx, void **stream);

/* device memory for hosts without an allocator of their own                         */
int cudamat_malloc(cudamat_ctx *ctx, size_t bytes, void **dev);
int cudamat_free(cudamat_ctx *ctx, void *dev);
int cudamat_h2d(cudamat_ctx *ctx, void *dev, const void *host, size_t bytes);
int cudamat_d2h(cudamat_ctx *ctx, void *host, const void *dev, size_t bytes);
int cudamat_d2d(cudamat_ctx *ctx, void *dst, const void *src, size_t bytes);   /* async, stream-ordered */
int cudamat_memset(cudamat_ctx *ctx, void *dev, int value, size_t bytes);

/* stream-ordered timers (HIP events on the context's stream)                        */
int cudamat_timer_create(cudamat_ctx *ctx, void **timer);
int cudamat_timer_start(cudamat_ctx *ctx, void *timer);
int cudamat_timer_stop(cudamat_ctx *ctx, void *timer);
int cudamat_timer_elapsed_ms(cudamat_ctx *ctx, void *timer, double *ms);  /* syncs */
int cudamat_timer_destroy(cudamat_ctx *ctx, void *timer);

/* ---- elementary kernels on device pointers ------------------------------------- */
/* y = alpha*(A x + d .* x) + beta*y ; d may be NULL ; base = index base of rowptr /
 * colidx (they are read as given).  n rows; x has as many entries as A has columns. */
int cudamat_spmv(cudamat_ctx *ctx, int n, const int *rowptr, const int *colidx,
                 const double *val, int base, double alpha, const double *x,
                 const double *d, double beta, double *y);
/* *out_dev = sum x_i y_i (device scalar, fixed reduction order => reproducible)     */
int cudamat_dot(cudamat_ctx *ctx, int64_t n, const double *x, const double *y, double *out_dev);
/* *out_dev = sqrt(sum x_i^2)                                                        */
int cudamat_nrm2(cudamat_ctx *ctx, int64_t n, const double *x, double *out_dev);
int cudamat_axpy(cudamat_ctx *ctx, int64_t n, double alpha, const double *x, double *y);
int cudamat_scal(cudamat_ctx *ctx, int64_t n, double alpha, double *x);

/* ---- HBM-resident system --------------------------------------------------------- */
/* Local row block of a (possibly row-sharded) square system.
 *   n_local : rows owned by this rank (== n for a single GPU)
 *   n_cols  : number of columns = global dimension
 *   rowptr/colidx/val : DEVICE pointers, base taken from `base`; they are copied
 *   (and rebased to 0) into the solver's own HBM arrays, so the caller may free them.
 * Work vectors and the reduction workspace are allocated here, once.
 * The arrays are validated on the device: row pointers that are not monotone from base to base + nnz or a
 * column outside [base, base + n_cols) => CUDAMAT_ERR_ARG.  Rows whose columns are not strictly increasing
 * are accepted for the un-preconditioned loops (lanes-per-row / tile SpMV); ILU(0) refuses them.        */
int cudamat_solver_create(cudamat_ctx *ctx, int n_local, int64_t n_cols, int64_t nnz,
                          const int *rowptr, const int *colidx, const double *val,
                          int base, cudamat_solver **out);
/* The same from HOST arrays (any pageable memory): creation runs beside the upload -- validation and the SpMV plan start
 * when the row pointers and column indices have landed, a blocked copy is filled behind the values.
 * rowptr[0] must equal base.                                                                                           */
int cudamat_solver_create_host(cudamat_ctx *ctx, int n_local, int64_t n_cols, int64_t nnz,
                               const int *rowptr_host, const int *colidx_host, const double *val_host,
                               int base, cudamat_solver **out);
int cudamat_solver_destroy(cudamat_solver *s);
/* (A0 + diag(d)) variant, pbicgstab.h:116; d is a device vector of n_local doubles
 * that must stay alive; NULL removes it.                                             */
int cudamat_solver_set_shift(cudamat_solver *s, const double *d);
/* ILU(0) of the local block sharing A's pattern + level analysis of L and U
 * (pbicgstab.cu:336-359).  Single rank only.                                          */
int cudamat_solver_ilu0(cudamat_solver *s);
/* the same on the rank's DIAGONAL BLOCK (rows and columns it owns) -- allowed in a sharded solver;
 * no collective is involved, neither here nor in the preconditioning steps            */
int cudamat_solver_block_ilu0(cudamat_solver *s);
/* copies the LU values (nnz doubles, same ordering as val; for the block variant: the
 * entries of the diagonal block in row order) to a device buffer                       */
int cudamat_solver_ilu0_values(cudamat_solver *s, double *out_dev);
/* how many values that is (nnz, or the entry count of the diagonal block)                */
int cudamat_solver_ilu0_nnz(cudamat_solver *s, int64_t *count);
/* which triangular-solve kernels the next preconditioning step uses: 1 = dependency-driven (one launch per
 * group of levels, rows wait for their dependencies inside the launch), 0 = one launch per level.  The first
 * form assumes no OTHER spin-waiting kernel shares the GPU (one stream per GPU); its waits are bounded, and a
 * solve that sees a timeout is redone with the second form, which then stays selected.                      */
int cudamat_solver_trsv_form(cudamat_solver *s, int *form);
/* out = U^-1 L^-1 in  (what one preconditioning step applies, pbicgstab.cu:92-98)     */
int cudamat_solver_precond_apply(cudamat_solver *s, const double *in, double *out);
/* row-sharded operation; comm is copied.  world == 1 or NULL => single GPU (unless the
 * environment sets CUDAMAT_FORCE_SHARDED=1, which keeps the collective path for testing). */
int cudamat_solver_set_comm(cudamat_solver *s, const cudamat_comm *comm);
/* which SpMV implementation the analysis chose for this matrix: 0 = one group of lanes per row on
 * the CSR arrays, 1 = blocked two-phase kernels (x / y tiles in LDS); decided at the first use.   */
int cudamat_solver_spmv_mode(cudamat_solver *s, int *mode);
/* Value dictionary: *distinct = number of distinct fp64 bit patterns among the matrix values when the SELECTED SpMV form
 * reads 8-bit indices into a dictionary of them instead of the 8-byte values (large matrices with at most 256 distinct
 * values; results are bit-identical, HBM traffic drops by 7 bytes per entry), 0 when it reads the values themselves.
 * The switch VALUE_DICT = 0 (cudamat_ctx_set_option / CUDAMAT_VALUE_DICT in the environment) disables the dictionary. */
int cudamat_solver_value_dict(cudamat_solver *s, int *distinct);
/* name(s) of the HIP kernel(s) one SpMV launch of this solver runs (e.g. "k_pb_phase1 + k_pb_phase2", "k_spmv_stream_c<256>",
 * "k_spmv<32>"): what a rocprofv3 kernel trace of the loop shows, for the bench line's `roofline.kernel`            */
int cudamat_solver_spmv_kernel(cudamat_solver *s, char *name, int cap);
/* Where the blocked copy's arrays went (round 5, csrc/spmv_pb.hip): device memory comes in classes, and a kernel that reads one
 * array while it writes another loses 5-7 % when both lie in one class, so the product stream of a large copy is placed in a
 * class of its own.  *placed: 1 placed, 0 searched without finding an arrangement, -1 not tried (another SpMV form, a copy
 * below 2 GB of products, PB_PLACE = 0, a drop-in call without PB_PLACE = 2); *slabs: 16 GB slabs classified; *seconds: what
 * the search took; classes: their 2 GB blocks by class ("00000000 11112222"), then the final check's timing (values read beside
 * writes into the product stream) between the fastest and slowest pair met, at most cap - 1 characters.            */
int cudamat_solver_placement(cudamat_solver *s, int *placed, int *slabs, double *seconds, char *classes, int cap);
/* y_local = (A + diag(d)) x ; x is the LOCAL slice, gathered through comm if sharded  */
int cudamat_solver_spmv(cudamat_solver *s, const double *x_local, double *y_local);
/* Solve.  b, x: device vectors of n_local doubles; x holds the initial guess on entry
 * (unless CUDAMAT_FLAG_X0_ONES) and the solution on return.                           */
int cudamat_solver_solve(cudamat_solver *s, const double *b, double *x, int precond,
                         int loop, int maxit, double tol, int flags, cudamat_stats *st);
/* residual-norm history of the last solve: LOOP_PBICGSTAB: hist[2i], hist[2i+1] =
 * norm after the half / full step of iteration i; LOOP_PBICGSTAB2: hist[i].  (LOOP_PIPELINED
 * as LOOP_PBICGSTAB; when that loop was restarted after the check of its true residual --
 * cudamat_stats.restarts -- the segments follow one another.)
 * Copies min(cap, available) doubles to the HOST buffer, returns the count in *count. */
int cudamat_solver_history(cudamat_solver *s, double *hist_host, int cap, int *count);

/* ---- drop-in host-pointer solve ---------------------------------------------------- */
/* One call = what the three reference entry points do: upload, (analyse, factor,)
 * iterate, download.  d / x0 may be NULL (x0 NULL => ones).  x receives the iterate
 * even when not converged.  Returns CUDAMAT_OK whenever the solve ran; convergence
 * is reported in st (may be NULL).                                                    */
int cudamat_solve(int n, int nnz, const double *A, const int *iA, const int *jA,
                  const double *d, const double *x0, const double *b, int precond,
                  int loop, int maxit, double tol, int debug, double *x,
                  cudamat_stats *st);

/* cudamat_solve keeps the solver of its LAST call (CSR copies, SpMV plan, ILU(0) factors: device memory on device 0 --
 * about 16 GB at 1e7 rows x 50 entries, 50 GB with ILU(0)) so that a caller who solves with the same matrix again --
 * time steps, several right-hand sides: the reference's per-call shape, pbicgstab.cu:157-409 -- pays the upload and a
 * device-side comparison but no analysis.  The next call with a different matrix (or different CUDAMAT_* switches)
 * replaces it; a call that runs out of device memory releases it and tries once more; cudamat_solve_sharded releases
 * it before it starts; cudamat_plan_cache_clear() (or CUDAMAT_PLAN_CACHE=0 in the environment) releases / disables it.  */
int cudamat_plan_cache_clear(void);

/* Device memory of this library comes from a recycling pool (csrc/pool.cpp): what a solver, a context or
 * cudamat_plan_cache_clear() frees stays with the library and serves its later requests -- on this platform a fresh
 * hipMalloc is not uniformly cheap (0.3 ms for 8 GB, but 26 ms per GB on some boxes once a process holds more than ~40 GB,
 * and seconds for single calls), and a host program that solves system after system would pay that every time.  The pool
 * keeps at most 96 GB free per device and empties itself when an allocation would fail.  cudamat_pool_trim() hands every
 * free block back to the driver -- for a host program that shares the device with other allocators; CUDAMAT_POOL=0 (read
 * once per process) turns the pool off.  The reference allocates and frees per call (pbicgstab.cu:243-255,392-405).   */
int cudamat_pool_trim(void);
/* what the driver reports free / in total on `device` (hipMemGetInfo) and, of the memory the driver counts as used, how
 * much the library's pool could hand out again without asking it; any pointer may be NULL */
int cudamat_mem_info(int device, size_t *driver_free, size_t *driver_total, size_t *pool_free);

/* The same solve over `ngpu` GPUs of this node, from one process: uniform row blocks, one host
 * thread and one RCCL rank per device (csrc/sharded.cpp).  precond: NONE or BLOCK_ILU0 (each
 * rank's diagonal block; ILU0 of the whole matrix does not shard => CUDAMAT_ERR_ARG).
 * ngpu <= 1 is cudamat_solve.  st receives rank 0's statistics (all ranks decide alike).     */
int cudamat_solve_sharded(int ngpu, int n, int nnz, const double *A, const int *iA, const int *jA,
                          const double *d, const double *x0, const double *b, int precond,
                          int loop, int maxit, double tol, int debug, double *x,
                          cudamat_stats *st);

/* ---- synthetic inputs, generated in HBM ------------------------ */
int64_t cudamat_poisson5_nnz(int nx, int ny);
int cudamat_gen_poisson5(cudamat_ctx *ctx, int nx, int ny, int64_t row0, int64_t row1,
                         int base, int *rowptr, int *colidx, double *val);
int cudamat_rand_row_nnz(int64_t n, int per_row);
int cudamat_gen_rand_rows(cudamat_ctx *ctx, int64_t n, int per_row, uint64_t seed,
                          int64_t row0, int64_t row1, int base, int *rowptr,
                          int *colidx, double *val);
int cudamat_gen_xstar(cudamat_ctx *ctx, int64_t i0, int64_t i1, uint64_t seed, double *x);

/* ---- Matrix Market I/O (host only) -------------------------------------------------- */
/* semantics of loadMMSparseMatrix(filename,'d',csr,...) mmio_wrapper.h:133-348:
 * returns 0 ok / CUDAMAT_ERR_IO; outputs are malloc'd, release with cudamat_host_free
 * (or free()).                                                                        */
int cudamat_load_mtx(const char *filename, int csr_format, int *m, int *n, int *nnz,
                     double **val, int **row, int **col);
void cudamat_host_free(void *p);
void cudamat_to_dense_vector(int n, int nnz, const double *A, const int *IA, double *out);

#ifdef __cplusplus
}
#endif
#endif /* CUDAMAT_H */
