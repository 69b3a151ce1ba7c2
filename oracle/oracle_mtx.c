/*
 * oracle_mtx.c -- CPU restatement of the reference Matrix Market loader.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Follows
 *   mmio.c:103-186   mm_read_banner         (banner tokens, lower-casing)
 *   mmio.c:198-225   mm_read_mtx_crd_size   (skip % comments, read M N nz)
 *   mmio.c:277-311   mm_read_mtx_crd_data   (indices kept exactly as in file)
 *   mmio_wrapper.h:133-348 loadMMSparseMatrix (symmetrise, sort, base
 *                    auto-detect, compress, verify)
 * for real / integer coordinate matrices (complex is read, real part kept, as
 * mmio_wrapper.h:324-326 does).
 */
#include "oracle.h"
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int i, j, p; } coo_t;

static int cmp_csr(const void *a, const void *b)   /* mmio_wrapper.h:56-67 */
{
    const coo_t *s = (const coo_t *)a, *t = (const coo_t *)b;
    if (s->i < t->i) return -1;
    if (s->i > t->i) return 1;
    return s->j - t->j;
}
static int cmp_csc(const void *a, const void *b)   /* mmio_wrapper.h:69-80 */
{
    const coo_t *s = (const coo_t *)a, *t = (const coo_t *)b;
    if (s->j < t->j) return -1;
    if (s->j > t->j) return 1;
    return s->i - t->i;
}

static void lower(char *p) { for (; *p; p++) *p = (char)tolower((unsigned char)*p); }

int orc_mtx_load(const char *filename, int csr_format, int *m, int *n, int *nnz,
                 double **val, int **row, int **col)
{
    FILE *f = fopen(filename, "r");
    if (!f) return 1;
    char line[1100], banner[80], mtx[80], crd[80], dtype[80], sym[80];
    if (!fgets(line, 1025, f)) { fclose(f); return 1; }
    if (sscanf(line, "%64s %64s %64s %64s %64s", banner, mtx, crd, dtype, sym) != 5) { fclose(f); return 1; }
    lower(mtx); lower(crd); lower(dtype); lower(sym);
    if (strncmp(banner, "%%MatrixMarket", 14) != 0 || strcmp(mtx, "matrix") != 0) { fclose(f); return 1; }
    const int is_sparse = strcmp(crd, "coordinate") == 0;
    const int is_real = strcmp(dtype, "real") == 0, is_int = strcmp(dtype, "integer") == 0;
    const int is_cplx = strcmp(dtype, "complex") == 0, is_pat = strcmp(dtype, "pattern") == 0;
    const int is_gen = strcmp(sym, "general") == 0, is_sym = strcmp(sym, "symmetric") == 0;
    const int is_herm = strcmp(sym, "hermitian") == 0, is_skew = strcmp(sym, "skew-symmetric") == 0;
    if (!(is_real || is_int || is_cplx || is_pat) || !(is_gen || is_sym || is_herm || is_skew)) { fclose(f); return 1; }
    /* mm_is_valid (mmio.c:92-101) + sparse only (mmio.c:352) */
    if (!is_sparse || (is_real && is_herm) || (is_pat && (is_herm || is_skew))) { fclose(f); return 1; }
    /* mmio_wrapper.h:161-169: complex needs 'z'/'c' (caller passes 'd'), pattern rejected */
    if (is_cplx || is_pat) { fclose(f); return 1; }

    int M = 0, N = 0, nz = 0;
    do {
        if (!fgets(line, 1025, f)) { fclose(f); return 1; }
    } while (line[0] == '%');
    if (sscanf(line, "%d %d %d", &M, &N, &nz) != 3) {
        int got;
        do {
            got = fscanf(f, "%d %d %d", &M, &N, &nz);
            if (got == EOF) { fclose(f); return 1; }
        } while (got != 3);
    }
    int *ti = (int *)malloc(sizeof(int) * (size_t)(nz > 0 ? nz : 1));
    int *tj = (int *)malloc(sizeof(int) * (size_t)(nz > 0 ? nz : 1));
    double *tv = (double *)malloc(sizeof(double) * (size_t)(nz > 0 ? nz : 1));
    for (int k = 0; k < nz; k++)
        if (fscanf(f, "%d %d %lg\n", &ti[k], &tj[k], &tv[k]) != 3) {
            free(ti); free(tj); free(tv); fclose(f);
            return 1;
        }
    fclose(f);

    /* symmetrise (mmio_wrapper.h:172-230) */
    int *ri = ti, *rj = tj;
    double *rv = tv;
    if (is_sym || is_herm || is_skew) {
        int count = 0;
        for (int k = 0; k < nz; k++) if (ti[k] != tj[k]) count++;
        ri = (int *)malloc(sizeof(int) * (size_t)(nz + count + 1));
        rj = (int *)malloc(sizeof(int) * (size_t)(nz + count + 1));
        rv = (double *)malloc(sizeof(double) * (size_t)(nz + count + 1));
        int j = 0;
        for (int k = 0; k < nz; k++) {
            ri[j] = ti[k]; rj[j] = tj[k]; rv[j] = tv[k]; j++;
            if (ti[k] != tj[k]) {
                ri[j] = tj[k]; rj[j] = ti[k];
                rv[j] = is_skew ? -tv[k] : tv[k];
                j++;
            }
        }
        nz += count;
        free(ti); free(tj); free(tv);
    }

    /* sort (mmio_wrapper.h:240-264) */
    coo_t *work = (coo_t *)malloc(sizeof(coo_t) * (size_t)(nz > 0 ? nz : 1));
    for (int k = 0; k < nz; k++) { work[k].i = ri[k]; work[k].j = rj[k]; work[k].p = k; }
    qsort(work, (size_t)nz, sizeof(coo_t), csr_format ? cmp_csr : cmp_csc);

    /* base auto-detect (mmio_wrapper.h:266-289) */
    int base0 = 0, base1 = 0;
    for (int k = 0; k < nz; k++) {
        if (work[k].i == 0 || work[k].j == 0) base0 = 1;
        if (work[k].i == M || work[k].j == N) base1 = 1;
    }
    if (base0 && base1) { free(work); free(ri); free(rj); free(rv); return 1; }
    const int base = base1 ? 1 : 0;

    /* compress (mmio_wrapper.h:24-46, 291-327) */
    const int dim = csr_format ? M : N;
    int *ptr = (int *)calloc((size_t)dim + 1, sizeof(int));
    ptr[0] = base;
    for (int k = 0; k < nz; k++) ptr[(csr_format ? work[k].i : work[k].j) + (1 - base)]++;
    for (int k = 0; k < dim; k++) ptr[k + 1] += ptr[k];
    int *idx = (int *)malloc(sizeof(int) * (size_t)(nz > 0 ? nz : 1));
    double *ov = (double *)malloc(sizeof(double) * (size_t)(nz > 0 ? nz : 1));
    for (int k = 0; k < nz; k++) {
        idx[k] = csr_format ? work[k].j : work[k].i;
        ov[k] = rv[work[k].p];
    }
    free(work); free(ri); free(rj); free(rv);

    /* verify_pattern (mmio_wrapper.h:91-130) */
    int bad = (nz != ptr[dim] - ptr[0]);
    for (int r = 0; !bad && r < dim; r++) {
        int s = ptr[r] - base, e = ptr[r + 1] - base;
        if (s > e) bad = 1;
        for (int c = s; !bad && c < e; c++) {
            if (idx[c] < base) bad = 1;
            if (c < e - 1 && idx[c] >= idx[c + 1]) bad = 1;
        }
    }
    if (bad) { free(ptr); free(idx); free(ov); return 1; }

    *m = M; *n = N; *nnz = nz; *val = ov;
    if (csr_format) { *row = ptr; *col = idx; }
    else            { *col = ptr; *row = idx; }
    return 0;
}
