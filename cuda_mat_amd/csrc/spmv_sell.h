// spmv_sell.h -- SELL-C-sigma copy of a CSR matrix and its SpMV (internal API; see spmv_sell.hip).
#pragma once
#include "kernels.h"

namespace cm {

struct SellPlan {
    int n = 0;
    int64_t nnz = 0;
    int nchunks = 0;               // chunks of 64 rows (the last window is padded with empty rows)
    long long slots = 0;           // stored slots incl. padding
    double fill = 1.0;             // slots / nnz
    int grid = 0, chunks_per_block = 0;
    double *val = nullptr;         // [slots] column-major inside a chunk
    int *col = nullptr;
    int *perm = nullptr;           // [nchunks * 64] original row of every chunk lane, -1 = padding row
    int *len = nullptr;            // [nchunks * 64] entries of that row
    long long *chunk_off = nullptr;   // [nchunks + 1] first column (of 64 slots) of every chunk
    double build_seconds = 0.0;
};

// max_fill > 0: give up (CUDAMAT_ERR_ARG, nothing allocated for the entries) when slots / nnz would exceed it
int sell_build(hipStream_t st, int n, int64_t nnz, const int *rp, const int *ci, const double *val, SellPlan *out,
               double max_fill = 0.0);
void sell_free(SellPlan *p);
// y = alpha*(A x + d.*xd) + beta*y with the same fused dot / prologue options as launch_spmv
int launch_spmv_sell(hipStream_t st, const SellPlan &plan, const SpmvArgs &a);

}  // namespace cm
