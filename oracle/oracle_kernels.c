/*
 * oracle_kernels.c -- CPU restatement of the elementary operations of the path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Citations are into /root/reference.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* bicstab_omp/bicstab.cpp:69-80: row loop `omp parallel for`, b[i] = 0 then a
 * left-to-right accumulation over the row.  The reference program is 0-based
 * only; the base (rp[0]) is honoured here the way cusparse's descriptor does
 * it for the GPU path (pbicgstab.cu:296-303). */
void orc_spmv(int n, const int *rp, const int *ci, const double *v,
              const double *x, double *y)
{
    const int base = rp[0];
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int j = rp[i] - base; j < rp[i + 1] - base; j++)
            s += v[j] * x[ci[j] - base];
        y[i] = s;
    }
}

/* cusparseDcsrmv(alpha, beta): pbicgstab.cu:67 (1,0) :469 (-1,0) :646 (-1,1)
 * :676,704 (1,1). */
void orc_csrmv(int n, const int *rp, const int *ci, const double *v, double alpha,
               const double *x, double beta, double *y)
{
    const int base = rp[0];
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int j = rp[i] - base; j < rp[i + 1] - base; j++)
            s += v[j] * x[ci[j] - base];
        y[i] = (beta == 0.0) ? alpha * s : alpha * s + beta * y[i];
    }
}

/* bicstab_omp/bicstab.cpp:83-91 (an `omp parallel for reduction(+)`, whose order of combination -- and so the
 * last bits of the sum, and near a tolerance the iteration count -- changes with the thread count and from run to
 * run).  A checker has to give the same answer every time: fixed blocks of ORC_DOT_BLOCK elements, each summed in
 * index order, combined in block order -- identical for every thread count. */
#define ORC_DOT_BLOCK 4096
double orc_dot(int n, const double *a, const double *b)
{
    const int nb = (n + ORC_DOT_BLOCK - 1) / ORC_DOT_BLOCK;
    double sum = 0.0;
    if (nb <= 1) {
        for (int i = 0; i < n; i++)
            sum += a[i] * b[i];
        return sum;
    }
    double *part = (double *)malloc((size_t)nb * sizeof(double));
    if (!part) {                      /* out of memory: the same blocks, one after the other */
        for (int k = 0; k < nb; k++) {
            const int hi = (k + 1) * ORC_DOT_BLOCK < n ? (k + 1) * ORC_DOT_BLOCK : n;
            double s = 0.0;
            for (int i = k * ORC_DOT_BLOCK; i < hi; i++)
                s += a[i] * b[i];
            sum += s;
        }
        return sum;
    }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nb; k++) {
        const int hi = (k + 1) * ORC_DOT_BLOCK < n ? (k + 1) * ORC_DOT_BLOCK : n;
        double s = 0.0;
        for (int i = k * ORC_DOT_BLOCK; i < hi; i++)
            s += a[i] * b[i];
        part[k] = s;
    }
    for (int k = 0; k < nb; k++)
        sum += part[k];
    free(part);
    return sum;
}

/* cublasDnrm2 (pbicgstab.cu:74,111,142,655,723).  cuBLAS uses a scaled
 * sum-of-squares; at the magnitudes of this path (no overflow/underflow of
 * x^2) that equals sqrt(sum x^2) to rounding, which is what is restated. */
double orc_nrm2(int n, const double *a) { return sqrt(orc_dot(n, a, a)); }

void orc_axpy(int n, double alpha, const double *x, double *y)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        y[i] += alpha * x[i];
}

void orc_scal(int n, double alpha, double *x)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        x[i] *= alpha;
}

/* pbicgstab.cu:36-42 */
void orc_mult_spec(int n, const double *a, const double *b, double k, double *c)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        c[i] = a[i] * b[i] * k;
}

/* pbicgstab.cu:1101-1115 toDenseVector: n x 1 CSR -> dense */
void orc_to_dense_vector(int n, int nnz, const double *A, const int *IA, double *out)
{
    (void)nnz;
    int sum = IA[0];
    int count = 0;
    for (int i = 0; i < n; ++i) {
        if (IA[i + 1] - sum > 0) {
            out[i] = A[count++];
            sum = IA[i + 1];
        } else {
            out[i] = 0.0;
        }
    }
}

void orc_free(void *p) { free(p); }

/* -------------------------------------------------------------------------
 * ILU(0) and triangular solves: cusparseDcsrilu0 (pbicgstab.cu:359) works in
 * place on a copy of A's values (:316) that shares A's rowptr/colidx
 * (:357-358); no pivoting, no boosting.  Restated as the textbook IKJ ILU(0)
 * on sorted rows (the loader guarantees strictly increasing columns,
 * mmio_wrapper.h:123).
 * ------------------------------------------------------------------------- */
int orc_ilu0(int n, const int *rp, const int *ci, double *vals)
{
    const int base = rp[0];
    int *diag = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        diag[i] = -1;
        for (int j = rp[i] - base; j < rp[i + 1] - base; j++)
            if (ci[j] - base == i) { diag[i] = j; break; }
    }
    int err = 0;
    for (int i = 0; i < n && !err; i++) {
        const int rs = rp[i] - base, re = rp[i + 1] - base;
        for (int kk = rs; kk < re; kk++) {
            const int k = ci[kk] - base;
            if (k >= i) break;
            if (diag[k] < 0 || vals[diag[k]] == 0.0) { err = k + 1; break; }
            const double lik = vals[kk] / vals[diag[k]];
            vals[kk] = lik;
            /* row_i[j] -= l_ik * row_k[j] for j > k present in both rows */
            int jj = kk + 1;
            int pp = diag[k] + 1;
            const int pe = rp[k + 1] - base;
            while (jj < re && pp < pe) {
                const int cj = ci[jj], cp = ci[pp];
                if (cj == cp) { vals[jj] -= lik * vals[pp]; jj++; pp++; }
                else if (cj < cp) jj++;
                else pp++;
            }
        }
        if (!err && (diag[i] < 0 || vals[diag[i]] == 0.0)) err = i + 1;
    }
    free(diag);
    return err;
}

/* cusparseDcsrsv_solve, FILL_MODE_LOWER + DIAG_TYPE_UNIT (pbicgstab.cu:92-94) */
void orc_trsv_lower_unit(int n, const int *rp, const int *ci, const double *vm,
                         const double *rhs, double *out)
{
    const int base = rp[0];
    for (int i = 0; i < n; i++) {
        double s = rhs[i];
        for (int j = rp[i] - base; j < rp[i + 1] - base; j++) {
            const int c = ci[j] - base;
            if (c >= i) break;
            s -= vm[j] * out[c];
        }
        out[i] = s;
    }
}

/* cusparseDcsrsv_solve, FILL_MODE_UPPER + DIAG_TYPE_NON_UNIT (pbicgstab.cu:96-98) */
void orc_trsv_upper(int n, const int *rp, const int *ci, const double *vm,
                    const double *rhs, double *out)
{
    const int base = rp[0];
    for (int i = n - 1; i >= 0; i--) {
        double s = rhs[i];
        double dg = 1.0;
        for (int j = rp[i] - base; j < rp[i + 1] - base; j++) {
            const int c = ci[j] - base;
            if (c < i) continue;
            if (c == i) dg = vm[j];
            else s -= vm[j] * out[c];
        }
        out[i] = s / dg;
    }
}

/* level sets of the strict-lower (upper=0) or strict-upper (upper=1) pattern:
 * level(i) = 1 + max level of the rows it depends on (0 for none). */
int orc_levels(int n, const int *rp, const int *ci, int upper, int *lev)
{
    const int base = rp[0];
    int nlev = 0;
    if (!upper) {
        for (int i = 0; i < n; i++) {
            int l = 0;
            for (int j = rp[i] - base; j < rp[i + 1] - base; j++) {
                const int c = ci[j] - base;
                if (c >= i) break;
                if (lev[c] + 1 > l) l = lev[c] + 1;
            }
            lev[i] = l;
            if (l + 1 > nlev) nlev = l + 1;
        }
    } else {
        for (int i = n - 1; i >= 0; i--) {
            int l = 0;
            for (int j = rp[i] - base; j < rp[i + 1] - base; j++) {
                const int c = ci[j] - base;
                if (c <= i) continue;
                if (lev[c] + 1 > l) l = lev[c] + 1;
            }
            lev[i] = l;
            if (l + 1 > nlev) nlev = l + 1;
        }
    }
    return nlev;
}
