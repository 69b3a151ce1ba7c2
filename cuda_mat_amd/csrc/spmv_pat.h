// spmv_pat.h -- row-pattern dictionary copy of a CSR matrix and its SpMV (internal API; see spmv_pat.hip).
#pragma once
#include "kernels.h"

namespace cm {

constexpr int kPatMaxLen = 15;       // longest row a pattern may describe (a table row = 16 ints: length + offsets)
constexpr int kPatMax = 256;         // distinct patterns an 8-bit id can name

struct PatPlan {
    int n = 0;
    int64_t nnz = 0;
    int W = 0;                       // value slots per row = the longest row
    int npat = 0;                    // distinct row patterns
    int nchunks = 0;                 // chunks of 64 rows (the last one padded with empty rows)
    double fill = 1.0;               // stored value slots / nnz
    int grid = 0, tiles_per_block = 0;
    double *val = nullptr;           // [nchunks][W][64]: slot-major inside a chunk
    unsigned char *pid = nullptr;    // [nchunks * 64]: pattern of every row
    int *tab = nullptr;              // [kPatMax][16]: length, then the column offsets (column - row) in column order
    double build_seconds = 0.0;
};

// max_fill > 0: give up (CUDAMAT_ERR_ARG, nothing kept) when the padded copy would hold more than max_fill x the entries.
// Also CUDAMAT_ERR_ARG (not an error for the caller: "this matrix has no such form") when a row is longer than
// kPatMaxLen or the rows show more than kPatMax - 1 distinct patterns.
int pat_build(hipStream_t st, int n, int64_t nnz, const int *rp, const int *ci, const double *val, PatPlan *out,
              double max_fill = 0.0);
void pat_free(PatPlan *p);
// y = alpha*(A x + d.*xd) + beta*y with the same fused dot / prologue options as launch_spmv
int launch_spmv_pat(hipStream_t st, const PatPlan &plan, const SpmvArgs &a);

}  // namespace cm
