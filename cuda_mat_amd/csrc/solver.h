// solver.h -- the HBM-resident system behind cudamat_solver_* (internal).
#pragma once
#include <vector>

#include "kernels.h"
#include "spmv_pb.h"
#include "valdict.h"
#include "spmv_sell.h"
#include "spmv_pat.h"

namespace cm {

constexpr int kLag = 2;            // host looks at the state written kLag iterations ago
constexpr int kRing = kLag + 2;
constexpr int kPipeRR = 32;         // residual replacement period of the pipelined loop (oracle.py PIPE_RR)

struct TriFactor {                 // one triangular factor in level-major storage
    int nlevels = 0;
    std::vector<int> level_ptr;    // host: rows of level l are [level_ptr[l], level_ptr[l+1])
    int *rp = nullptr;             // device, n+1, rows in level order, 0-based
    int *ci = nullptr;             // device, column ids (original numbering)
    double *val = nullptr;         // device
    int *row_of = nullptr;         // device: original row id of permuted row
    double *dinv = nullptr;        // device: 1/diag in permuted order (upper only)
    int64_t nnz = 0;
    // index spaces of a solve (trsv.hip trsv_rows): permuted row pr reads rhs[rhs_of[pr]], writes out[out_of[pr]]
    // (nullptr = pr itself); `ci` holds indices into out.  Original space: both = row_of.  Level-major space (lm,
    // hybrid factors): out_of = nullptr; rhs_of = nullptr for L, the U-position -> L-position map (owned) for U.
    int *rhs_of = nullptr, *out_of = nullptr;
    bool lm = false;
};

}  // namespace cm

struct cudamat_solver {
    cudamat_ctx *ctx = nullptr;
    int n = 0;                 // local rows
    int n_pad = 0;             // rows per rank in a sharded run (== n when single)
    int64_t n_cols = 0;
    int64_t nnz = 0;
    int *rp = nullptr, *ci = nullptr;
    double *val = nullptr;
    const double *d = nullptr;
    cm::SpmvPlan plan{};
    int spmv_mode = -1;        // -1 undecided, 0 CSR forms, 1 blocked two-phase kernels, 2 SELL-C-sigma, 3 row-pattern dictionary
    bool cols_sorted = true;   // every row's columns strictly increasing (checked at creation)
    cm::PbPlan pb{};
    cm::SellPlan sell{};
    cm::PatPlan pat{};
    double ms_csr = 0.0, ms_pb = 0.0, ms_sell = 0.0, ms_pat = 0.0;   // auto-tune timings
    double t_create = 0.0;       // s: upload-side copies, validation, CSR launch plan (cudamat_solver_create)
    double t_create0 = 0.0;      // wall clock at the start of the creation
    double t_spmv_setup = 0.0;   // s: ensure_spmv_mode in all (copies of the matrix in other layouts + timing of candidates)
    double t_spmv_timing = 0.0;  // s: of that, the timed candidate launches
    double col_span_bytes = -1.0; // mean (last - first column) * 8 over sampled rows (-1: not sampled)

    // work vectors (n_pad doubles each, pad kept zero)
    double *r = nullptr, *rw = nullptr, *p = nullptr, *pw = nullptr, *s = nullptr, *t = nullptr,
           *v = nullptr;
    double *gather = nullptr;  // world * n_pad doubles (sharded runs)
    double *x0_save = nullptr; // the caller's x0, kept while a dependency-driven preconditioner may have to be redone
    double *v2 = nullptr;      // second v buffer of the fused small-system loop (p and r double-buffer in pw and s)
    // CUDAMAT_LOOP_PIPELINED: z = A s, w = A r, q, y = A q, xh = x + alpha p (n_pad each), partials of k_pipe_a / k_pipe_b,
    // reduced scalars ([0..2] phase A, [8..12] phase B), events ordering the side-stream reductions
    double *pz = nullptr, *pww = nullptr, *pq = nullptr, *py = nullptr, *pxh = nullptr;
    double *prh = nullptr, *pwh = nullptr, *psh = nullptr, *pzh = nullptr, *pqh = nullptr, *ptmp = nullptr;   // preconditioned pipelined loop: M^-1 r, w, s, z, q; scratch
    double *pipeA = nullptr, *pipeB = nullptr, *red_pipe = nullptr;
    hipEvent_t ev_red[2] = {}, ev_red_done[2] = {};

    // reduction workspace: four stages of per-workgroup partials + reduced scalars
    double *parts_full = nullptr, *parts_rv = nullptr, *parts_half = nullptr, *parts_tt = nullptr;
    double *red = nullptr;     // 8 doubles
    cm::LoopState *st = nullptr;       // device
    cm::LoopState *st_ring = nullptr;  // pinned host, kRing slots
    hipEvent_t ev[cm::kRing] = {};
    unsigned long long *snap_host = nullptr;   // pinned: per-iteration progress words written by k_full
    unsigned long long *snap_dev = nullptr;    // the same memory as the device sees it
    double *hist = nullptr;    // device
    int hist_cap = 0;
    int hist_count = 0;
    int last_loop = 0;

    // profiling events
    std::vector<hipEvent_t> prof_ev;
    // CUDAMAT_FLAG_PROFILE on a sharded solver: (start, stop) event pairs around the exchanges, by kind:
    // 0 pieces of an overlapped gather (communicator's stream), 1 the solver's stream waiting for a piece,
    // 2 a plain all-gather, 3 an all-reduce
    std::vector<hipEvent_t> comm_ev;
    std::vector<int> comm_kind;
    size_t comm_used = 0;
    bool profiling = false;
    bool prof_failed = false;          // an event call of the (optional) per-launch timing failed: the timings of this solve are void

    // row sharding
    bool sharded = false;
    cudamat_comm comm{};
    // overlapped gather (blocked SpMV + a communicator with gather_part): phase 1 runs on the pieces of the
    // gathered vector that have arrived while the next ones are in flight on the communicator's stream
    bool overlap = false;
    bool agreed = false;       // the ranks have compared their setup outcomes since the last set_comm
    int overlap_chunks = 4;
    hipEvent_t ev_x = nullptr, ev_part[cm::kPbMaxChunks] = {};
    // phase 1 of piece c runs on its own stream (waits: x ready, piece arrived), so that the small launches of the
    // pieces and of the local slice fill the GPU together instead of one after the other
    hipStream_t part_stream[cm::kPbMaxChunks] = {};
    hipEvent_t ev_p1[cm::kPbMaxChunks] = {};
    double ms_spmv_alone = 0.0;   // the chosen SpMV form with x in place, as the tuner timed it (0: not timed)
    // windowed gather (halo): the part [lo, hi) of every other slice this rank's rows reference, exchanged with the
    // ranks at setup (setup_agree); used in place of the whole gather when every rank needs less than half of it
    bool windowed = false;
    bool windows_known = false;
    std::vector<int64_t> w_send_off, w_send_cnt, w_recv_off, w_recv_cnt;
    double gather_fraction = 1.0;
    double *need_dev = nullptr;   // 2 * world (mine) + 2 * world^2 (everybody's) doubles

    // ILU(0)
    bool has_ilu = false;
    // the matrix the preconditioner is built from: the solver's own CSR, or -- block-Jacobi in a sharded
    // run -- a copy of the rank's diagonal block with local column ids
    int *pm_rp = nullptr, *pm_ci = nullptr;
    double *pm_val = nullptr;
    int64_t pm_nnz = 0;
    bool pm_owned = false;
    bool ilu_block = false;    // factors belong to CUDAMAT_PRECOND_BLOCK_ILU0
    double *lu = nullptr;      // pm_nnz doubles on that matrix's pattern
    int *diag_pos = nullptr;   // position of the diagonal in each row
    cm::TriFactor L, U;
    void *ilu_plans = nullptr;  // launch plans (ilu.h) owned by ilu.hip
    // the loop in level-major spaces (hybrid factors, one GPU): the matrix with rows in L's order and columns in U's
    // positions as a blocked copy, b in L's space, x in U's
    cm::PbPlan pb_perm{};
    cm::ValDict vd_perm;
    double *x_perm = nullptr, *b_perm = nullptr;
    bool perm_ready = false;    // pb_perm is built
    bool perm_failed = false;   // ... could not be (out of memory): the loop permutes around every M^-1 instead
    bool perm_active = false;   // the solve in progress runs in the level-major spaces (spmv_local uses pb_perm)
    double t_perm_matrix = 0.0;
    double t_analysis = 0.0, t_factor = 0.0, t_analysis_l = 0.0, t_analysis_u = 0.0;
    int trsv_fallbacks = 0;     // solves redone level by level after a dependency-driven wait timed out
    cm::ValDict vd;             // value dictionary of the matrix (n == 0: more than 256 distinct values, or not looked yet)
    bool vd_tried = false;
    unsigned *bar = nullptr;    // grid barrier words of the single-launch loop (device)
    int test_allreduces = 0;    // fault injection (CUDAMAT_TEST_COMM_FAIL)
    bool resident_off = false;  // a barrier wait ran into its bound once: keep to the three-launch loop
    int device_cus = 0;         // compute units of the device (0: not asked yet)
    int loop_fallbacks = 0;     // solves redone with the three-launch loop for that reason
};

namespace cm {
// ---- solver.hip: what the loops (loops.hip) and the drop-in entry point (dropin.hip) use of the resident system
int dev_alloc(void **p, size_t bytes);
// creation in stages (cudamat_solver_create = all three on device arrays; dropin.hip runs them beside the upload)
int solver_alloc(cudamat_ctx *ctx, int n_local, int64_t n_cols, int64_t nnz, cudamat_solver **out);   // rp / ci / val allocated, EMPTY
int solver_setup_pattern(cudamat_solver *s);         // rp, ci in place (0-based): validation, CSR plan, compressed indices
int solver_setup_values(cudamat_solver *s);          // val in place: value-dependent parts of the CSR plan
// the blocked form is the clear choice for this matrix (scattered columns: ensure_spmv_mode would select it without timing
// anything), judged from the pattern alone; *blocked = false: leave the choice to ensure_spmv_mode
int spmv_mode_is_blocked_early(cudamat_solver *s, bool *blocked);
// take a finished blocked copy as this solver's SpMV form (built beside the upload, dropin.hip)
void spmv_mode_adopt_blocked(cudamat_solver *s, const PbPlan &pb, double seconds);
int ensure_valdict(cudamat_solver *s);               // the matrix's value dictionary, looked for once
int ensure_work(cudamat_solver *s);                  // the seven work vectors (+ the gather buffer when sharded)
int ensure_spmv_mode(cudamat_solver *s);             // choose the SpMV form (once per system / partition)
int setup_agree(cudamat_solver *s, int rc_local);    // sharded: the ranks compare their set-up outcomes (collective)
// y = (A + diag d) x with x a LOCAL n_pad-long work vector (pad zero); exchanges first when sharded; dot / check as SpmvArgs
int spmv_local(cudamat_solver *s, const double *x_local, double *y, int dot, const double *w, double *parts, LoopArgs la, int check,
               ScalarSrc half);
int spmv_parts(const cudamat_solver *s);             // per-workgroup partial sums an SpMV launch leaves in `parts`
int allreduce(cudamat_solver *s, double *buf, int count);
void comm_mark_begin(cudamat_solver *s, int kind, hipStream_t st);
void comm_mark_end(cudamat_solver *s, hipStream_t st);
hipEvent_t prof_event(cudamat_solver *s, size_t i);
// native: `in` is in L's level-major space and `out` leaves in U's (the loop that runs in those spaces); otherwise both are in
// the caller's row numbering
int precond_apply(cudamat_solver *s, const double *in, double *tmp, double *out, bool native = false);

constexpr int kSortRowMax = 1024;     // longest row k_sort_rows stages (level-major index spaces need every row below it)
int launch_sort_rows(hipStream_t st, int nrows, const int *src_rp, const int *src_of, const int *dst_rp, const int *ci,
                     const double *val, const int *colmap, int *out_ci, double *out_val);
int ilu0_setup(cudamat_solver *s, bool block);
// the pattern-only part of ilu0_setup (diagonal positions, level analysis of L and U), kept for the ilu0_setup that follows
int ilu0_analyse_early(cudamat_solver *s);
void ilu0_flush_deferred(cudamat_solver *s);
bool ilu0_will_use_level_major(cudamat_solver *s);    // ... and the temporaries it kept alive while the upload ran
int ilu0_release(cudamat_solver *s);
// rhs / out in the factor's own index spaces (TriFactor::rhs_of / out_of)
int trsv_apply(cudamat_solver *s, const TriFactor &F, bool upper, const double *rhs, double *out);
// level-major factors (TriFactor::lm): M^-1 on original-numbering vectors; vectors to / from L's (upper: U's) space;
// the matrix of the loop that runs in those spaces
int precond_apply_original(cudamat_solver *s, const double *in, double *tmp, double *out);
int perm_to_space(cudamat_solver *s, bool upper, const double *in, double *out);
int perm_from_space(cudamat_solver *s, bool upper, const double *in, double *out);
int ilu_perm_matrix(cudamat_solver *s);
int trsv_status(cudamat_solver *s);   // after a stream sync: did a dependency-driven solve give up waiting?
bool trsv_syncfree_active(cudamat_solver *s);
int trsv_form_code(cudamat_solver *s);           // 0 level launches, 1 dependency-driven, 2 single workgroup in LDS
void trsv_group_counts(cudamat_solver *s, int *groups_l, int *groups_u);   // hybrid factors: groups of levels (0: not split)
void trsv_disable_syncfree(cudamat_solver *s);   // sticky: level-by-level kernels from now on
}  // namespace cm
