// valdict.h -- value dictionary of a sparse matrix (internal API; see valdict.hip).
#pragma once
#include "common.h"

namespace cm {

constexpr int kDictMax = 256;      // distinct values an 8-bit index can name

struct ValDict {
    int n = 0;                       // distinct values (0: the matrix has more than kDictMax, no dictionary)
    double *dict = nullptr;          // device, kDictMax doubles (ascending bit patterns; the unused tail is 0)
    unsigned char *idx = nullptr;    // device, one index per entry in CSR order: val[k] == dict[idx[k]] bit for bit
};

// Scan the values (device array, nnz entries); when at most kDictMax distinct bit patterns occur, build the
// dictionary and the per-entry indices.  out->n == 0 afterwards means "no dictionary" (not an error).
int valdict_build(hipStream_t st, const Config &cfg, int64_t nnz, const double *val, ValDict *out);
void valdict_free(ValDict *d);
// Do the first `count` values show more than kDictMax distinct bit patterns?  Uses `scratch` (>= 33 KB of device memory:
// cudamat_ctx::scratch) for the probe table -- allocates nothing, frees nothing.  *many = true: no dictionary possible.
int valdict_sample_overflows(hipStream_t st, int64_t count, const double *val, void *scratch, bool *many);

}  // namespace cm
