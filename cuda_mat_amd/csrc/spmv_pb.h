// spmv_pb.h -- two-phase "propagation blocking" SpMV for matrices whose columns are scattered
// over a vector much larger than an XCD's 4 MiB L2 (internal API; see spmv_pb.hip).
#pragma once
#include "kernels.h"
#include "valdict.h"

namespace cm {

constexpr int kPbMaxChunks = 16;

// How the column range is cut before it is tiled into column blocks (x tiles).  A row-sharded solver gathers
// its SpMV input from `world` ranks, slice by slice and -- when the gather is overlapped with phase 1 --
// piece by piece: column blocks never straddle a slice or a piece, so phase 1 of the blocks of a piece
// can start as soon as that piece has arrived.
struct PbCols {
    int64_t per = 0;     // columns per slice (a rank's share of the gathered vector); 0: one slice = all columns
    int chunks = 1;      // pieces per slice
    int rank = 0;        // the local slice (needs no exchange)
};

struct PbPlan {
    int n = 0;            // local rows
    int64_t n_cols = 0;
    int64_t nnz = 0;
    int CB = 0, NCB = 0;  // columns per block (x tile in LDS, upper bound), number of column blocks
    int64_t per = 0;      // columns per slice, pieces per slice, columns per piece, blocks per piece
    int chunks = 1;
    int64_t chunk_len = 0;
    int bpc = 0;
    int RB = 0, NRB = 0;  // rows per block, number of row blocks (= workgroups of phase 2)
    int NW = 0;           // waves per phase-2 workgroup = sub-blocks per row block
    int SR = 0;           // rows per sub-block (one wave owns them)
    int LPS = 64;         // lanes per (sub-block, column block) segment in phase 2
    int depth = 4;        // segment loads in flight per wave in phase 2
    int NSUB = 0;         // NRB * NW
    // entries in (column block, row block, row, column) order
    double *pv = nullptr;          // values (nullptr when the matrix has a value dictionary:)
    unsigned char *pvi = nullptr;  // ... then one 8-bit index per entry into `dict` (valdict.h; not owned, <= 256 doubles)
    const double *dict = nullptr;
    int ndict = 0;
    unsigned short *pc = nullptr;  // column - first column of its block
    unsigned short *pr = nullptr;  // row - sub*SR
    double *P = nullptr;           // products val * x[col], same order (phase 1 -> phase 2)
    int *cstart = nullptr;         // NCB+1: first entry of every column block
    int *col0 = nullptr;           // NCB+1: first column of every column block (blocks tile [0, n_cols) in order)
    int *sstart = nullptr;         // [NSUB][NCB]: first entry of (sub, cb)
    int *slen = nullptr;           // [NSUB][NCB]: entries of (sub, cb)
    // phase-1 launch parts: part 0 = the blocks of the local slice, part 1 + c = piece c of every other slice
    int *order = nullptr;          // NCB block ids, part after part
    int part_off[kPbMaxChunks + 2] = {0};
    int placed = -1;               // -1: not tried (small copy / PB_PLACE=0); 0: no arrangement found; 1: product stream in a memory class of its own
    int place_slabs = 0;           // slabs of device memory the placement search classified
    double place_seconds = 0.0;
    float place_check_ms = 0.f, place_fast_ms = 0.f, place_slow_ms = 0.f;      // the final check's timing between the two groups' extremes
    char place_classes[192] = {0};  // their 2 GB blocks by class, e.g. "00000000 11112222"
    bool pc_zeroed = false;        // (construction: the alignment pads of pc are in place)
    double build_seconds = 0.0;
};

struct PbCut {          // device copy of the column cut: slice -> piece -> block
    int per, chunks, chunk_len, bpc, CB;
};
// a blocked copy under construction (pb_build_begin .. pb_build_values .. pb_build_fill .. pb_build_end)
struct PbBuild {
    PbPlan p;
    int *bins = nullptr;       // [column block][sub-block] fill cursors (device)
    PbCut cut{};
    size_t cap = 0;            // entries of the value / column / row / product arrays (with the blocks' alignment pads)
    double t0 = 0.0;
    bool verbose = false;      // Config::verbose at the count pass: pb_build_end prints the plan and its segment lengths
    // two-pass fill (k_pb_group + k_pb_scatter): groups of GB column blocks, NG groups; one packed (block, column, row) word
    // per entry between the passes (the values travel in the product stream), first entry of every (sub-block, group) bucket
    bool two_pass = false;
    int place = 0;             // Config::pb_place
    bool report_classes = false;      // VERBOSE >= 2
    double place_max_seconds = 0.3;
    int fill_occ = 0;          // Config::pb_fill_occ: resident waves per CU of pass A (0: default)
    int GB = 1, NG = 1;
    unsigned long long *smeta = nullptr;
    int *gstart = nullptr;
};
int pb_build_begin(hipStream_t st, const Config &cfg, int n, int64_t n_cols, int64_t nnz, const int *rp, const int *ci,
                   const PbCols *cols, PbBuild *b);
// pb_build_begin in its two halves: every allocation (sizes only) / count pass + scans (needs rp, ci; allocates nothing)
int pb_build_alloc(hipStream_t st, const Config &cfg, int n, int64_t n_cols, int64_t nnz, const PbCols *cols, PbBuild *b);
int pb_build_count(hipStream_t st, const Config &cfg, const int *rp, const int *ci, PbBuild *b);
int pb_build_values(hipStream_t st, PbBuild *b, const ValDict *vd);
int pb_build_fill(hipStream_t st, PbBuild *b, const int *rp, const int *ci, const double *val, const ValDict *vd, int sub0, int sub1);
int pb_build_end(hipStream_t st, PbBuild *b, PbPlan *out);
void pb_build_abort(PbBuild *b);

// decide whether the matrix is a candidate (large x, scattered columns) -- cheap estimate
bool pb_candidate(hipStream_t st, int n, int64_t n_cols, int64_t nnz, const int *rp, const int *ci);
// build the blocked copy from 0-based CSR on the device
// vd (optional): the value dictionary of `val` (indices in the same CSR order) -- the copy then stores 1 byte per value
int pb_build(hipStream_t st, const Config &cfg, int n, int64_t n_cols, int64_t nnz, const int *rp, const int *ci,
             const double *val, PbPlan *out, const PbCols *cols = nullptr, const ValDict *vd = nullptr);
void pb_free(PbPlan *p);
// SpmvArgs::pb_strict for launches on this context: the option PB_STRICT, or 1 when the device's LDS does NOT serve equal
// addresses of one ds_add_f64 in lane order (probed once per context by a one-wave kernel, ~30 us; see spmv_pb.hip)
int pb_strict_for(cudamat_ctx *ctx, int *strict);
// y = alpha*(A x + d.*xd) + beta*y with the same fused dot / prologue options as launch_spmv;
// args.rp/ci/val are ignored (the plan holds the matrix)
int launch_spmv_pb(hipStream_t st, const PbPlan &plan, const SpmvArgs &a);
// the same in pieces (overlapped gather): the stopping test, phase 1 of launch part `part`
// (0 .. plan.chunks; see PbPlan::order), phase 2
int launch_pb_check(hipStream_t st, const SpmvArgs &a);
int launch_pb_phase1(hipStream_t st, const PbPlan &plan, const SpmvArgs &a, int part);
int launch_pb_phase2(hipStream_t st, const PbPlan &plan, const SpmvArgs &a);

}  // namespace cm
