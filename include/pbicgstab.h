/*
 * pbicgstab.h -- drop-in C++ interface of the solver library (libcuda_mat.so).
 *
 * Same entry points, argument order and meaning as the reference header
 * (/root/reference/pbicgstab.h:23-30,78,91,113-120), so a caller such as the reference's
 * example.cpp compiles against this header unchanged -- minus the CUDA / <conio.h> includes,
 * which this header does not need.  The three solvers are thin shims over the C ABI
 * (include/cudamat.h: cudamat_solve); all arithmetic runs in hand-written gfx950 kernels.
 *
 * Differences a caller can observe (all deliberate):
 *   - bicgstab(A, b) works: the reference's version never forms r0 = b - A x0
 *     (pbicgstab.cu:471-478 is commented out) and fails on iteration 0.
 *   - no call ever exit()s the process; device errors make the solvers return false.
 *   - cudamat_last_stats() exposes iterations / residuals, which the reference discards.
 */
#pragma once

#include <cmath>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "cudamat.h"

enum Base {
    Base0 = 0,
    Base1 = 1
};

/* uniform draws from libc rand(), exactly the reference's recipe (pbicgstab.cu:413-423) so that a
 * caller who seeds srand() sees the same matrices and vectors */
double rand_float_0_1();
double rand_float(double min, double max);

/* dense-scan random CSR (pbicgstab.h:32-55 of the reference): every (i,j) draws one uniform to decide
 * "zero", non-zeros draw values until |value| >= eps.  O(n*m) by construction. */
template <Base base>
int gen_rand_csr_matrix(int n, int m, std::vector<double> *A, std::vector<int> *IA, std::vector<int> *JA,
                        double probability_of_zero, double min, double max, double eps)
{
    int next = base;
    IA->push_back(next);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < m; ++j) {
            if (rand_float_0_1() <= probability_of_zero) continue;
            double r = rand_float(min, max);
            while (std::fabs(r) < eps) r = rand_float(min, max);
            A->push_back(r);
            JA->push_back(j + base);
            ++next;
        }
        IA->push_back(next);
    }
    return (int)A->size();
}

/* CSR from a dense generator f(i,j), keeping |f| > eps (reference pbicgstab.h:57-76) */
template <Base base>
int fill_csr_matrix(int n, int m, std::vector<double> *A, std::vector<int> *IA, std::vector<int> *JA,
                    std::function<double(int, int)> f, double eps)
{
    int next = base;
    IA->push_back(next);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < m; ++j) {
            const double el = f(i, j);
            if (std::fabs(el) > eps) {
                A->push_back(el);
                JA->push_back(j + base);
                ++next;
            }
        }
        IA->push_back(next);
    }
    return (int)A->size();
}

void gen_rand_vector(int n, double *vector, double probability_of_zero, double min, double max);

/* "(v0 v1 ... )" with std::to_string formatting, as the reference prints results */
template <typename T>
void dump_vector(std::ostringstream &stream, int n, T *vector)
{
    stream << "(";
    for (int i = 0; i < n; ++i) stream << std::to_string(vector[i]) << " ";
    stream << ")";
}

/* n x 1 CSR "column vector" (as loadMMSparseMatrix returns for a vector file) -> dense */
void toDenseVector(int n, int nnz, double *A, int *IA, double *out);

/* wall-clock seconds (helper_cusolver.h:148-153 of the reference) */
double second(void);

#define IN
#define OUT

/* Common arguments (reference pbicgstab.h:96-110): n dimension; nnz non-zeros; A values; iA row
 * pointers (iA[0] = index base 0 or 1); jA column indices in that base; b right-hand side; maxit;
 * tol relative to the initial residual; debug prints the reference's trace; x solution (written even
 * when not converged); dtAlg seconds spent in the device iteration loop. */

/* solve A x = b, no preconditioner, x0 = 1 */
bool bicgstab(int n, int nnz, double IN(*A), int IN(*iA), int IN(*jA), double IN(*b), int maxit, double tol,
              bool debug, double OUT(*x), double OUT(*dtAlg));

/* solve (A0 + I*d) x = b from x0, no preconditioner */
bool bicgstab(int n, int nnz, double IN(*A0), int IN(*iA0), int IN(*jA0), double IN(*d), double IN(*x0),
              double IN(*b), int maxit, double tol, bool debug, double OUT(*x), double OUT(*dtAlg));

/* solve A x = b with the ILU(0) preconditioner; requires A[i,i] != 0 */
bool bicgstab_lu_precond(int n, int nnz, double IN(*A), int IN(*iA), int IN(*jA), double IN(*b), int maxit,
                         double tol, bool debug, double OUT(*x), double OUT(*dtAlg));

/* what the last solve on this thread did (iterations, residual norms, phase times) */
const cudamat_stats *cudamat_last_stats();
