// spmv_pb.h -- two-phase "propagation blocking" SpMV for matrices whose columns are scattered
// over a vector much larger than an XCD's 4 MiB L2 (internal API; see spmv_pb.hip).
#pragma once
#include "kernels.h"

namespace cm {

struct PbPlan {
    int n = 0;            // local rows
    int64_t n_cols = 0;
    int64_t nnz = 0;
    int CB = 0, NCB = 0;  // columns per block (x tile in LDS), number of column blocks
    int RB = 0, NRB = 0;  // rows per block, number of row blocks (= workgroups of phase 2)
    int NW = 0;           // waves per phase-2 workgroup = sub-blocks per row block
    int SR = 0;           // rows per sub-block (one wave owns them)
    int LPS = 64;         // lanes per (sub-block, column block) segment in phase 2
    int NSUB = 0;         // NRB * NW
    // entries in (column block, row block, row, column) order
    double *pv = nullptr;          // values
    unsigned short *pc = nullptr;  // column - cb*CB
    unsigned short *pr = nullptr;  // row - sub*SR
    double *P = nullptr;           // products val * x[col], same order (phase 1 -> phase 2)
    int *cstart = nullptr;         // NCB+1: first entry of every column block
    int *sstart = nullptr;         // [NSUB][NCB]: first entry of (sub, cb)
    int *slen = nullptr;           // [NSUB][NCB]: entries of (sub, cb)
    double build_seconds = 0.0;
};

// decide whether the matrix is a candidate (large x, scattered columns) -- cheap estimate
bool pb_candidate(hipStream_t st, int n, int64_t n_cols, int64_t nnz, const int *rp, const int *ci);
// build the blocked copy from 0-based CSR on the device
int pb_build(hipStream_t st, int n, int64_t n_cols, int64_t nnz, const int *rp, const int *ci,
             const double *val, PbPlan *out);
void pb_free(PbPlan *p);
// y = alpha*(A x + d.*xd) + beta*y with the same fused dot / prologue options as launch_spmv;
// args.rp/ci/val are ignored (the plan holds the matrix)
int launch_spmv_pb(hipStream_t st, const PbPlan &plan, const SpmvArgs &a);

}  // namespace cm
