// kernels.h -- launch wrappers of the hand-written gfx950 kernels (internal API).
#pragma once
#include "common.h"

namespace cm {

struct LoopArgs {
    LoopState *st = nullptr;   // NULL: kernel used outside a solve (no freeze / checks)
    double *hist = nullptr;
    int hist_cap = 0;
    int loop = 0;              // CUDAMAT_LOOP_*
    int no_exit = 0;
    // per-iteration progress word in PINNED HOST memory: k_full publishes (k+1) << 32 | state with one
    // 8-byte system-scope store, so the host can look at a lagged state without a copy or an event
    unsigned long long *snap = nullptr;
    int snap_slots = 0;
    int k = 0;                 // the host's iteration index of this launch
};

struct SpmvPlan {
    int lanes;            // lanes cooperating on one row (2..64)
    int grid;             // workgroups
    int rows_per_block;   // contiguous rows owned by a workgroup
    int stream_rows;      // > 0: LDS-staged stream kernel, this many rows per LDS tile (short rows)
    // > 0: nnz-balanced tiles of 2048 entries (skewed row lengths); rows_per_block = tiles per workgroup;
    // the tables are owned by the plan (plan_spmv_free)
    int tiles = 0;
    int tile_nspan = 0, tile_fix_grid = 0;
    int *tile_S = nullptr, *tile_span = nullptr;
    double *tile_heads = nullptr, *tile_tails = nullptr;
    double lane_cost = 0.0;   // lane-iterations of the lanes-per-row kernel / nnz (1 = perfectly balanced)
    // stream kernel with compressed indices (banded matrices): 16-bit column offsets from the tile's first row,
    // 8-bit row lengths, one entry offset per tile; owned by the plan
    short *c_off16 = nullptr;
    unsigned char *c_len8 = nullptr;
    int *c_tile_base = nullptr;
    // line-aligned copies for the compressed stream kernel (plan_spmv_align, round 3): every tile's entries start on a
    // 64-entry boundary, so each wave's 512-byte value load and 128-byte offset load covers whole 128-byte lines
    int *a_base = nullptr;              // ntiles + 1: first entry slot of every tile
    short *a_off16 = nullptr;
    double *a_val = nullptr;
    // the matrix's value dictionary (valdict.h; 256 doubles on the device, not owned) when the copies below exist
    const double *c_dict = nullptr;
    // dictionary form of the compressed stream kernel (plan_spmv_dict): per tile, the entries' 16-bit offsets and 8-bit
    // value indices padded to a multiple of 8 entries, so that a thread fetches its 8 entries with one 16-byte and one
    // 8-byte load (owned by the plan)
    int *d_pbase = nullptr;             // ntiles + 1: first padded entry of every tile
    short *d_off16 = nullptr;
    unsigned char *d_val8 = nullptr;
};
SpmvPlan plan_spmv(const Config &cfg, int n_rows, int64_t nnz);
void plan_spmv_free(SpmvPlan *plan);
// per-workgroup dot partials one launch leaves behind
inline int plan_spmv_parts(const SpmvPlan &p) { return p.grid + (p.tiles ? p.tile_fix_grid : 0); }
// short rows (mean <= 12): switch the plan to the LDS-staged stream kernel when every tile of
// `stream_rows` consecutive rows holds at most kStreamNnz entries (checked on the device); otherwise, when
// the lanes-per-row kernel would spend > 2.5 lane-iterations per entry (skewed row lengths), switch it to the
// nnz-balanced tile kernel.  CUDAMAT_SPMV_FORM=lanes|tiles forces one of the two.
// scratch: >= 64 bytes of device memory for the plan's read-backs (cudamat_ctx::scratch); NULL: allocated and freed here
int plan_spmv_refine(hipStream_t s, const Config &cfg, int n_rows, int64_t nnz, const int *rp, int base, SpmvPlan *plan, void *scratch = nullptr);

// y = alpha*(A x + d .* xd) + beta*y  on 0- or 1-based CSR (base folded into the
// pointers by the caller).  dot: 0 none, 1: parts[2b] = sum y*w, 2: also
// parts[2b+1] = sum y*y.  check: CHECK_HALF evaluates the half-step stopping test
// from `half` in the prologue.
struct SpmvArgs {
    int n;
    const int *rp;
    const int *ci;
    const double *val;
    const double *x;      // indexed by column id
    const double *d;      // optional diagonal shift (local rows)
    const double *xd;     // x restricted to the local rows (for d .* x)
    double alpha, beta;
    double *y;
    int dot;
    const double *w;
    double *parts;
    LoopArgs loop;
    int check;
    ScalarSrc half;
    int pb_strict;        // blocked form, phase 2: add a row's products of one wave instruction rank by rank (Config::pb_strict)
};
int launch_spmv(hipStream_t s, const SpmvPlan &plan, const SpmvArgs &a);

// fused loop of small systems: the SpMV computes its input vector on the fly (small_loops.hip)
struct FuseArgs {
    int mode;                              // 1: p' = r + beta (p - omega v);  2: s = r - alpha v
    const double *r;
    const double *p_old, *v_old;           // mode 1 inputs
    double *p_out;                         // mode 1: p' stored by the row owners
    const double *v;                       // mode 2 input (the v just computed)
    double *s_out, *xsol;                  // mode 2: s stored by the row owners, x += alpha p'
    const double *p;
    ScalarSrc src;                         // mode 1: (rw.r, ||r||^2) partials;  mode 2: rw.v partials
    double *parts_half;                    // mode 2: ||s||^2 partial of every workgroup
};
// the whole loop in one launch (small systems; small_loops.hip)
struct ResidentArgs {
    int iters;                 // iterations this launch may run (it stops early when a test fires)
    int first_count;           // partial sums behind parts_full at this launch's first iteration
    unsigned spin_limit;       // polls a barrier wait may take (2^22 ~ seconds; CUDAMAT_RESIDENT_SPIN_LIMIT: tests)
    unsigned *bar;             // [0] barrier arrivals, [1] set when a barrier wait ran into its bound; zeroed by the host
    double *p_a, *p_b, *v_a, *v_b, *r, *s, *t, *x;
    const double *rw;
    double *parts_rv, *parts_tt, *parts_half, *parts_full;
};
bool resident_loop_supported(const SpmvPlan &plan, int n);
int launch_resident_loop(hipStream_t s, const SpmvPlan &plan, const SpmvArgs &a, const ResidentArgs &q);
bool fused_spmv_supported(const SpmvPlan &plan);
int launch_fused_spmv(hipStream_t s, const SpmvPlan &plan, const SpmvArgs &a, const FuseArgs &f);
// stream plans only: build the compressed index copy when every offset fits (no-op otherwise); rp/ci 0-based
int plan_spmv_compress(hipStream_t s, const Config &cfg, int n_rows, int64_t nnz, const int *rp, const int *ci, SpmvPlan *plan);
// compressed stream plans of a matrix with a value dictionary (valdict.h): vidx = 8-bit value index per entry in CSR
// order, dict = 256 doubles on the device (not owned); no-op when the plan has no compressed copy
int plan_spmv_align(hipStream_t s, const Config &cfg, int n_rows, int64_t nnz, const int *rp, const double *val, SpmvPlan *plan);
int plan_spmv_dict(hipStream_t s, int n_rows, int64_t nnz, const int *rp, const unsigned char *vidx, const double *dict,
                   SpmvPlan *plan);

int vec_grid(int64_t n);

// r = b - r (r holds A x0 on entry), rw = r, p = r; parts[2b] = parts[2b+1] = sum r^2
int launch_init(hipStream_t s, int64_t n, const double *b, double *r, double *rw, double *p,
                double *parts, int *nparts);
// one workgroup: set up LoopState from the initial reduction
// abs_tol > 0: the loop stops at this absolute residual norm (tol is ignored) and starts out 'converged' when the
// initial residual is within twice of it
int launch_init_finish(hipStream_t s, LoopState *st, ScalarSrc init, double tol, double abs_tol = 0.0);
// p = r + beta (p - omega v), preceded by the full-step test of the previous iteration
int launch_update_p(hipStream_t s, LoopArgs la, ScalarSrc full, int64_t n, const double *r,
                    double *p, const double *v);
// alpha = rho / (rw.v); r -= alpha v; parts[b] = sum r^2   (the half step's x += alpha pw rides in launch_full(..., pw))
int launch_half(hipStream_t s, LoopArgs la, ScalarSrc rv, int64_t n, double *r, const double *v, double *parts, int *nparts);
// omega = (t.r)/(t.t); x += omega s; r -= omega t; parts = (rw.r, r.r); it++
// half.ptr != NULL: evaluate the half-step stopping test from `half` first (fused small-system loop)
// pw != NULL: x += alpha pw (the half step's update, with the alpha k_half stored) before x += omega s
int launch_full(hipStream_t s, LoopArgs la, ScalarSrc tt, int64_t n, double *x, const double *sv,
                double *r, const double *t, const double *rw, double *parts, int *nparts,
                ScalarSrc half = ScalarSrc{nullptr, 0, 1}, const double *pw = nullptr);
// pipelined BiCGStab (pipelined.hip): partials of k_pipe_a have stride 3, of k_pipe_b stride 5
int launch_pipe_seed(hipStream_t s, ScalarSrc init, ScalarSrc rww, double *out5);
// the hatted (M^-1-applied) vectors of the preconditioned form; all NULL without a preconditioner
struct PipeHatA { const double *rh, *wh, *zh; double *sh, *qh; };
struct PipeHatB { const double *qh, *wh, *zh; double *rh; };
int launch_pipe_a(hipStream_t s, LoopArgs la, ScalarSrc B, int64_t n, const double *r, const double *w, const double *t,
                  const double *v, double *p, double *sv, double *z, double *q, double *y, const double *x, double *xh,
                  double *parts, int *nparts, PipeHatA hat);
int launch_pipe_b(hipStream_t s, LoopArgs la, ScalarSrc A, int64_t n, const double *q, const double *y, const double *t,
                  const double *v, const double *rw, const double *sv, const double *z, const double *xh, double *x, double *r,
                  double *w, double *parts, int *nparts, PipeHatB hat);
// residual replacement: r = f - ax; the five dots of k_pipe_b recomputed (stride 5 partials)
int launch_residual(hipStream_t s, const LoopArgs &la, int64_t n, const double *f, const double *ax, double *r);
int launch_pipe_dots(hipStream_t s, const LoopArgs &la, int64_t n, const double *rw, const double *r, const double *w, const double *sv,
                     const double *z, double *parts, int *nparts);
// standalone stopping tests (one workgroup)
int launch_check(hipStream_t s, LoopArgs la, ScalarSrc src, int which);
// out[k] = sum of partials, k < K (one workgroup)
int launch_reduce_parts(hipStream_t s, ScalarSrc in, int K, double *out, int sqrt_it);
// generic streaming kernels
int launch_dot_parts(hipStream_t s, int64_t n, const double *x, const double *y, double *parts,
                     int *nparts);
int launch_axpy(hipStream_t s, int64_t n, double alpha, const double *x, double *y);
int launch_scal(hipStream_t s, int64_t n, double alpha, double *x);
int launch_fill(hipStream_t s, int64_t n, double value, double *x);
int launch_rebase(hipStream_t s, int64_t n, const int *in, int shift, int *out);

}  // namespace cm
