// spmv_pat.h -- row-pattern dictionary copy of a CSR matrix and its SpMV (internal API; see spmv_pat.hip).
#pragma once
#include "kernels.h"
#include "valdict.h"

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
    double *val = nullptr;           // [nchunks][W][64]: slot-major inside a chunk (NULL when the copy holds dictionary indices)
    // value-dictionary form (the matrix has <= 256 distinct values, valdict.h): per row ONE word of 8 (W <= 8) or 16 bytes
    // holding the 8-bit dictionary indices of its entries in column order -- 8 or 16 B per row instead of 8 B per entry
    unsigned char *vidx = nullptr;   // [nchunks * 64][vword]
    int vword = 0;                   // 8 or 16
    const double *dict = nullptr;    // the dictionary (owned by the ValDict it came from)
    int ndict = 0;
    unsigned char *pid = nullptr;    // [nchunks * 64]: pattern of every row
    int *tab = nullptr;              // [kPatMax][16]: length, then the column offsets (column - row) in column order
    double build_seconds = 0.0;
};

// max_fill > 0: give up (CUDAMAT_ERR_ARG, nothing kept) when the padded copy would hold more than max_fill x the entries.
// Also CUDAMAT_ERR_ARG (not an error for the caller: "this matrix has no such form") when a row is longer than
// kPatMaxLen or the rows show more than kPatMax - 1 distinct patterns.
// vd (optional): the value dictionary of `val` (indices in the same CSR order) -- the copy then stores one byte per value
int pat_build(hipStream_t st, int n, int64_t nnz, const int *rp, const int *ci, const double *val, PatPlan *out,
              double max_fill = 0.0, const ValDict *vd = nullptr);
void pat_free(PatPlan *p);
// y = alpha*(A x + d.*xd) + beta*y with the same fused dot / prologue options as launch_spmv
int launch_spmv_pat(hipStream_t st, const PatPlan &plan, const SpmvArgs &a);

}  // namespace cm
