/*
 * mmio_wrapper.h -- drop-in declaration of the Matrix Market loader.
 *
 * Signature and semantics of the reference's loadMMSparseMatrix (mmio_wrapper.h:133-142): read a
 * real/integer coordinate .mtx file, expand symmetric / skew-symmetric storage, sort, auto-detect
 * the index base and return CSR (csrFormat = true) or CSC arrays allocated with malloc() -- the
 * caller frees them with free() (example.cpp:370-374).  Returns 0 on success, 1 on any error.
 * The definition lives in libcuda_mat.so (cuda_mat_amd/host/pbicgstab.cpp -> cudamat_load_mtx);
 * the reference defines it inside its header on top of NIST mmio.c, which this library does not use.
 */
#pragma once

int loadMMSparseMatrix(char *filename, char elem_type, bool csrFormat, int *m, int *n, int *nnz,
                       double **aVal, int **aRowInd, int **aColInd);
