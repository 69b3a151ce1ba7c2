/*
 * oracle_gen.c -- CPU definition of the synthetic inputs.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  The device generators in
 * cuda_mat_amd/csrc/ must reproduce these bit for bit; tests compare them.
 *
 * The reference's own generators are unusable at scale: fill_csr_matrix
 * (pbicgstab.h:57-76) and bicstab_omp/generator.cpp are O(n^2) and draw from
 * unseeded rand().  Entries are small integers so that the reference CPU
 * program's int-transpose defect (bicstab.cpp:37,57) is harmless
 * and SpMV results are exact in fp64 regardless of summation order.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

/* splitmix64 finaliser (Steele, Lea, Flood 2014), counter based */
uint64_t orc_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline uint64_t mulhi64(uint64_t a, uint64_t b)
{
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
}

/* ---- 5-point Laplacian, row-major grid, i = y*nx + x ---------------------- */
int64_t orc_poisson5_nnz(int nx, int ny)
{
    return 5LL * nx * ny - 2LL * nx - 2LL * ny;
}

void orc_poisson5(int nx, int ny, int base, int *rp, int *ci, double *v)
{
    int64_t k = 0;
    const int64_t n = (int64_t)nx * ny;
    rp[0] = base;
    for (int64_t i = 0; i < n; i++) {
        const int x = (int)(i % nx), y = (int)(i / nx);
        if (y > 0)      { ci[k] = (int)(i - nx) + base; v[k++] = -1.0; }
        if (x > 0)      { ci[k] = (int)(i - 1) + base;  v[k++] = -1.0; }
        ci[k] = (int)i + base; v[k++] = 4.0;
        if (x < nx - 1) { ci[k] = (int)(i + 1) + base;  v[k++] = -1.0; }
        if (y < ny - 1) { ci[k] = (int)(i + nx) + base; v[k++] = -1.0; }
        rp[i + 1] = (int)k + base;
    }
}

/* ---- random matrix, fixed nnz per row -------------------------------------
 * row i: key = mix64(seed + i); draw a = 0,1,2,...: h = mix64(key + a),
 * column c = floor(h * n / 2^64), value {-2,-1,1,2}[h & 3]; a draw is rejected
 * if c == i or c was already taken; stop after min(per_row-1, n-1) accepted
 * draws.  Columns sorted increasing; diagonal = 1 + sum |offdiag| inserted at
 * its sorted position.  Every row therefore has exactly orc_rand_row_nnz()
 * entries and the matrix is strictly diagonally dominant (non-singular, ILU(0)
 * exists without pivoting). */
int orc_rand_row_nnz(int64_t n, int per_row)
{
    int64_t off = per_row - 1;
    if (off > n - 1) off = n - 1;
    if (off < 0) off = 0;
    return (int)off + 1;
}

void orc_rand_rows(int64_t n, int per_row, uint64_t seed, int64_t row0,
                   int64_t row1, int base, int *rp, int *ci, double *v)
{
    static const double vals[4] = {-2.0, -1.0, 1.0, 2.0};
    const int rn = orc_rand_row_nnz(n, per_row);
    const int noff = rn - 1;
    rp[0] = base;
    for (int64_t i = row0; i < row1; i++)
        rp[i - row0 + 1] = base + (int)((i - row0 + 1) * rn);
#pragma omp parallel
    {
        int64_t *cols = (int64_t *)malloc(sizeof(int64_t) * (size_t)(noff + 1));
        double *cv = (double *)malloc(sizeof(double) * (size_t)(noff + 1));
#pragma omp for schedule(static)
        for (int64_t i = row0; i < row1; i++) {
            const uint64_t key = orc_mix64(seed + (uint64_t)i);
            int cnt = 0;
            double absum = 0.0;
            for (uint64_t a = 0; cnt < noff; a++) {
                const uint64_t h = orc_mix64(key + a);
                const int64_t c = (int64_t)mulhi64(h, (uint64_t)n);
                if (c == i) continue;
                /* sorted insert with duplicate rejection */
                int pos = cnt;
                while (pos > 0 && cols[pos - 1] > c) pos--;
                if (pos > 0 && cols[pos - 1] == c) continue;
                for (int q = cnt; q > pos; q--) { cols[q] = cols[q - 1]; cv[q] = cv[q - 1]; }
                cols[pos] = c;
                cv[pos] = vals[h & 3];
                absum += cv[pos] < 0 ? -cv[pos] : cv[pos];
                cnt++;
            }
            int *oc = ci + (i - row0) * rn;
            double *ov = v + (i - row0) * rn;
            int k = 0, placed = 0;
            for (int q = 0; q < noff; q++) {
                if (!placed && cols[q] > i) { oc[k] = (int)i + base; ov[k++] = 1.0 + absum; placed = 1; }
                oc[k] = (int)cols[q] + base; ov[k++] = cv[q];
            }
            if (!placed) { oc[k] = (int)i + base; ov[k++] = 1.0 + absum; }
        }
        free(cols);
        free(cv);
    }
}

/* x*_i = 1 + (mix64(seed + i) & 7) / 8  in {1, 1.125, ..., 1.875}: with integer
 * matrix entries b = A x* is exact in fp64 for any summation order. */
void orc_xstar(int64_t i0, int64_t i1, uint64_t seed, double *x)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = i0; i < i1; i++)
        x[i - i0] = 1.0 + (double)(orc_mix64(seed + (uint64_t)i) & 7) * 0.125;
}

/* ---- the reference CLI's own random system (example.cpp:173-180,274-288,339) ----------------
 * What `example` solves when no -M is given: fill_csr_matrix<Base1>(dim, dim, f, 1e-3)
 * (pbicgstab.h:57-76: dense scan, row-major, positions with |f(i,j)| > eps kept, 1-based) with
 *   f(i,i) = rand_float(1,10);  f(i,j) = rand_float_0_1() >= p_zero ? rand_float(1,10) : 0.0
 * (example.cpp:274-285), rand_float_0_1 = rand() / RAND_MAX, rand_float = that * (max-min) + min
 * (pbicgstab.cu:413-423); the right-hand side afterwards from the same rand() stream:
 * b_i = rand_float_0_1() <= p_zero_vec ? 0 : rand_float(1,5)  (gen_rand_vector, pbicgstab.cu:1093-1097;
 * example.cpp:339 with P(0) = 0.2).  The reference never calls srand(): its draws are those of
 * srand(1).  `seed` is passed to srand() here (1 = the reference's stream).
 * The matrix is neither diagonally dominant nor symmetric: entries in [1,10] everywhere.
 * Two passes over the same stream: orc_example_count sizes the arrays. */
static double ex_rand01(void) { return (double)rand() / (double)RAND_MAX; }
static double ex_rand(double lo, double hi) { return ex_rand01() * (hi - lo) + lo; }

static int64_t example_scan(int dim, double p_zero, unsigned seed, int *rp, int *ci, double *v)
{
    int64_t k = 0;
    srand(seed);
    if (rp) rp[0] = 1;
    for (int i = 0; i < dim; i++) {
        for (int j = 0; j < dim; j++) {
            double el;
            if (i == j) el = ex_rand(1, 10);                                 /* example.cpp:275 */
            else if (ex_rand01() >= p_zero) el = ex_rand(1, 10);             /* :277-278 */
            else el = 0.0;                                                   /* :280 */
            if ((el < 0 ? -el : el) > 1e-3) {                                /* pbicgstab.h:65 */
                if (ci) { ci[k] = j + 1; v[k] = el; }
                k++;
            }
        }
        if (rp) rp[i + 1] = (int)k + 1;
    }
    return k;
}

int64_t orc_example_count(int dim, double p_zero, unsigned seed)
{
    return example_scan(dim, p_zero, seed, NULL, NULL, NULL);
}

/* fills rp[dim+1], ci[nnz], v[nnz] (1-based, as fill_csr_matrix<Base1>) and b[dim]; returns nnz */
int64_t orc_example_system(int dim, double p_zero, double p_zero_vec, unsigned seed,
                           int *rp, int *ci, double *v, double *b)
{
    const int64_t nnz = example_scan(dim, p_zero, seed, rp, ci, v);
    for (int i = 0; i < dim; i++)                                            /* pbicgstab.cu:1093-1097 */
        b[i] = ex_rand01() <= p_zero_vec ? 0.0 : ex_rand(1, 5.0);
    return nnz;
}
