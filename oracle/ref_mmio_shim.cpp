// ref_mmio_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// Exposes the UNMODIFIED reference loader (mmio_wrapper.h:133-348, compiled in
// place from /root/reference together with its mmio.c) behind a C symbol so
// tests/golden/make_golden.py can record what the reference itself produces.
// Built only where /root/reference exists; output goes to oracle/_ref/.
#include "mmio_wrapper.h"

extern "C" int ref_loadMMSparseMatrix(char *filename, int csr_format, int *m, int *n,
                                      int *nnz, double **val, int **row, int **col)
{
    return loadMMSparseMatrix(filename, 'd', csr_format != 0, m, n, nnz, val, row, col);
}
extern "C" void ref_free(void *p) { free(p); }
