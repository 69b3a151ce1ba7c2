/*
 * pbicgstab.h -- C++ interface of libcuda_mat.so on MI355X.
 *
 * Drop-in for the header of the same name in Russoul/cuda-mat: every declaration a caller of the
 * reference library uses is here with the same name, parameter order and meaning
 *   enum Base                                  reference pbicgstab.h:23-26
 *   rand_float_0_1 / rand_float                :28-30
 *   gen_rand_csr_matrix<Base> / fill_csr_matrix<Base>   :32-76
 *   gen_rand_vector, dump_vector<T>, toDenseVector      :78-91
 *   bicgstab (2 overloads), bicgstab_lu_precond         :113-120
 * but none of the CUDA / Windows includes (<cuda_runtime.h>, <conio.h>), which this implementation does
 * not need.  The solvers are shims over the C ABI of include/cudamat.h (cudamat_solve); the arithmetic
 * runs in hand-written gfx950 kernels.
 *
 * Observable differences, all deliberate: bicgstab(A, b) works (upstream never forms
 * r0 = b - A x0, pbicgstab.cu:471-478); nothing calls exit(); cudamat_last_stats() reports what the
 * reference throws away (iterations, residual norms, phase times).
 */
#pragma once

#include <cmath>
#include <cstdlib>
#include <functional>
#include <sstream>
#include <string>
#include <vector>

#include "cudamat.h"

enum Base { Base0 = 0, Base1 = 1 };

/* libc rand() scaled to [0,1] / [min,max]; seeding with srand() reproduces the reference's draws */
double rand_float_0_1();
double rand_float(double min, double max);

namespace cudamat_detail {
/* shared body of the two dense-scan CSR builders: visit (i, j) row-major, ask `entry` whether the
 * position holds a value, append it with its column index in the requested base */
template <Base base, typename Entry>
int dense_scan_to_csr(int rows, int cols, std::vector<double> *values, std::vector<int> *row_ptr,
                      std::vector<int> *col_idx, Entry entry)
{
    int cursor = static_cast<int>(base);
    row_ptr->push_back(cursor);
    for (int i = 0; i < rows; ++i) {
        for (int j = 0; j < cols; ++j) {
            double value = 0.0;
            if (!entry(i, j, &value)) continue;
            values->push_back(value);
            col_idx->push_back(j + static_cast<int>(base));
            ++cursor;
        }
        row_ptr->push_back(cursor);
    }
    return static_cast<int>(values->size());
}
}  // namespace cudamat_detail

/* random CSR by dense scan: a position is zero with the given probability, otherwise uniform in
 * [min,max], redrawn while |value| < eps.  O(n*m) rand() calls -- meant for small demos. */
template <Base base>
int gen_rand_csr_matrix(int n, int m, std::vector<double> *A, std::vector<int> *IA, std::vector<int> *JA,
                        double probability_of_zero, double min, double max, double eps)
{
    return cudamat_detail::dense_scan_to_csr<base>(n, m, A, IA, JA, [&](int, int, double *out) {
        if (rand_float_0_1() <= probability_of_zero) return false;
        double r = rand_float(min, max);
        while (std::fabs(r) < eps) r = rand_float(min, max);
        *out = r;
        return true;
    });
}

/* CSR of the dense function f(i,j), keeping the positions with |f| > eps */
template <Base base>
int fill_csr_matrix(int n, int m, std::vector<double> *A, std::vector<int> *IA, std::vector<int> *JA,
                    std::function<double(int, int)> f, double eps)
{
    return cudamat_detail::dense_scan_to_csr<base>(n, m, A, IA, JA, [&](int i, int j, double *out) {
        *out = f(i, j);
        return std::fabs(*out) > eps;
    });
}

void gen_rand_vector(int n, double *vector, double probability_of_zero, double min, double max);

/* "(v0 v1 ... )", elements formatted by std::to_string */
template <typename T>
void dump_vector(std::ostringstream &stream, int n, T *vector)
{
    std::string text = "(";
    for (int i = 0; i < n; ++i) text += std::to_string(vector[i]) + " ";
    stream << text << ")";
}

/* an n x 1 CSR matrix (what loadMMSparseMatrix returns for a vector file) to a dense array */
void toDenseVector(int n, int nnz, double *A, int *IA, double *out);

/* wall-clock seconds since the epoch */
double second(void);

/* argument direction markers kept for source compatibility */
#define IN
#define OUT

/*
 * Arguments shared by the three solvers: n = dimension, nnz = stored entries, A/iA/jA = CSR values,
 * row pointers (iA[0] is the index base, 0 or 1) and column indices in that base, b = right-hand side,
 * maxit / tol = iteration limit and tolerance relative to the initial residual, debug = print the
 * residual trace, x = result (written also when the method does not converge), dtAlg = seconds spent
 * in the device iteration loop (uploads and analysis excluded).
 */

/* A x = b, no preconditioner, x0 = 1 */
bool bicgstab(int n, int nnz,
              double IN(*A), int IN(*iA), int IN(*jA),
              double IN(*b),
              int maxit, double tol, bool debug,
              double OUT(*x), double OUT(*dtAlg));

/* (A0 + I*d) x = b from the initial guess x0, no preconditioner */
bool bicgstab(int n, int nnz,
              double IN(*A0), int IN(*iA0), int IN(*jA0),
              double IN(*d), double IN(*x0), double IN(*b),
              int maxit, double tol, bool debug,
              double OUT(*x), double OUT(*dtAlg));

/* A x = b with the ILU(0) preconditioner; every diagonal entry A[i,i] must be stored and non-zero */
bool bicgstab_lu_precond(int n, int nnz,
                         double IN(*A), int IN(*iA), int IN(*jA),
                         double IN(*b),
                         int maxit, double tol, bool debug,
                         double OUT(*x), double OUT(*dtAlg));

/* statistics of the last solve issued by the calling thread */
const cudamat_stats *cudamat_last_stats();

/* Addition to the reference's interface: spread the solves of this process over `ngpu` GPUs of the
 * node (uniform row blocks, RCCL; cudamat_solve_sharded).  The ILU(0) entry point then factors every
 * GPU's diagonal block (block-Jacobi: a weaker preconditioner than ILU(0) of the whole matrix, which
 * does not shard).  Default 1 = the reference's single-device behaviour. */
void cudamat_use_gpus(int ngpu);
