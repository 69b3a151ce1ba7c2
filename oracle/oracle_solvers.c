/*
 * oracle_solvers.c -- CPU restatement of the three solver loops of the path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Citations are into /root/reference.
 */
#include "oracle.h"
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

static double *dalloc(int n)
{
    return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
}

/* =========================================================================
 * bicstab_omp/bicstab.cpp:93-196  BiCG  (the directory says "bicstab" but the
 * algorithm is bi-conjugate gradient with A^T).
 * ========================================================================= */

/* bicstab.cpp:35-66 Transpose2: counting sort by column.  With int_transpose
 * the value travels through an `int` exactly as `V = A.Value[j]` does (:37,57),
 * i.e. truncation toward zero (defect D6).  0-based, like the program. */
static void transpose2(int n, int nz, const int *rp, const int *ci, const double *v,
                       int *trp, int *tci, double *tv, int int_transpose)
{
    memset(trp, 0, sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < nz; i++)
        trp[ci[i] + 1]++;
    int S = 0;
    for (int i = 1; i <= n; i++) {
        int tmp = trp[i];
        trp[i] = S;
        S += tmp;
    }
    for (int i = 0; i < n; i++) {
        for (int j = rp[i]; j < rp[i + 1]; j++) {
            double V = int_transpose ? (double)(int)v[j] : v[j];
            int r = ci[j];
            int pos = trp[r + 1];
            tv[pos] = V;
            tci[pos] = i;
            trp[r + 1]++;
        }
    }
}

/* The same A^T arrays (a stable counting sort by column gives one result) built with every OpenMP thread:
 * per-thread column histograms over contiguous row ranges, offsets by (column, thread), scatter in row order.
 * NOT part of the reference's path -- its transposition is the serial loop above, ~70 s for the 5e8 entries of
 * the 1e7-row bench matrix; bench.py's CPU baseline uses this one so that it can time the reference's
 * iteration loop (bicstab.cpp:146-182) on the FULL matrix within seconds.  Checked against transpose2 in
 * tests/test_oracle_golden.py. */
static void transpose2_par(int n, int nz, const int *rp, const int *ci, const double *v,
                           int *trp, int *tci, double *tv, int int_transpose)
{
    const int T = omp_get_max_threads();
    int *hist = (int *)calloc((size_t)T * (size_t)(n + 1), sizeof(int));
    if (!hist) { transpose2(n, nz, rp, ci, v, trp, tci, tv, int_transpose); return; }
#pragma omp parallel num_threads(T)
    {
        const int t = omp_get_thread_num();
        const long long r0 = (long long)n * t / T, r1 = (long long)n * (t + 1) / T;
        int *h = hist + (size_t)t * (size_t)(n + 1);
        for (long long i = r0; i < r1; i++)
            for (int j = rp[i]; j < rp[i + 1]; j++) h[ci[j]]++;
#pragma omp barrier
        /* column totals -> trp[c + 1]; per-thread counts -> exclusive prefix inside the column */
#pragma omp for schedule(static)
        for (int c = 0; c < n; c++) {
            int run = 0;
            for (int q = 0; q < T; q++) {
                int *e = hist + (size_t)q * (size_t)(n + 1) + c;
                const int cnt = *e;
                *e = run;
                run += cnt;
            }
            trp[c + 1] = run;
        }
#pragma omp single
        {
            trp[0] = 0;
            for (int c = 0; c < n; c++) trp[c + 1] += trp[c];
        }
        for (long long i = r0; i < r1; i++)
            for (int j = rp[i]; j < rp[i + 1]; j++) {
                const int c = ci[j];
                const int pos = trp[c] + h[c]++;
                tv[pos] = int_transpose ? (double)(int)v[j] : v[j];
                tci[pos] = (int)i;
            }
    }
    free(hist);
}

static double wall_s(void) { return omp_get_wtime(); }

int orc_bicg(int n, const int *rp_in, const int *ci_in, const double *v,
             const double *b, double *x, int maxit, double eps,
             int int_transpose, int parallel_vec, int *iters)
{
    return orc_bicg_timed(n, rp_in, ci_in, v, b, x, maxit, eps, int_transpose, parallel_vec, iters, 0, NULL, NULL);
}

/* orc_bicg with (optionally) the threaded transposition and with the seconds spent in the transposition and in
 * the iteration loop (bicstab.cpp:146-182) reported separately */
int orc_bicg_timed(int n, const int *rp_in, const int *ci_in, const double *v,
                   const double *b, double *x, int maxit, double eps,
                   int int_transpose, int parallel_vec, int *iters, int fast_transpose,
                   double *t_transpose, double *t_loop)
{
    return orc_bicg_teams(n, rp_in, ci_in, v, b, x, maxit, eps, int_transpose, parallel_vec, fast_transpose,
                          0, NULL, iters, t_transpose, t_loop);
}

/* The BiCG program once per OpenMP team size: ONE transposition (with the team in force at the call), then for every
 * entry of teams[] the program's set-up (:135-144) and iteration loop (:146-182) from x = 1 with that many threads;
 * iters[k] / t_loop[k] report each run, x holds the last one.  n_teams == 0: one run with the current team (this is
 * orc_bicg_timed).  bench.py's CPU baseline picks its team on the loop it reports with this. */
int orc_bicg_teams(int n, const int *rp_in, const int *ci_in, const double *v,
                   const double *b, double *x, int maxit, double eps,
                   int int_transpose, int parallel_vec, int fast_transpose,
                   int n_teams, const int *teams, int *iters, double *t_transpose, double *t_loop)
{
    /* the reference program is 0-based (bicstab.cpp:198-214); rebase if needed */
    const int base = rp_in[0];
    const int nz = rp_in[n] - base;
    int *rp = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    int *ci = (int *)malloc(sizeof(int) * (size_t)(nz > 0 ? nz : 1));
    for (int i = 0; i <= n; i++) rp[i] = rp_in[i] - base;
    for (int i = 0; i < nz; i++) ci[i] = ci_in[i] - base;

    int *trp = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    int *tci = (int *)malloc(sizeof(int) * (size_t)(nz > 0 ? nz : 1));
    double *tv = (double *)malloc(sizeof(double) * (size_t)(nz > 0 ? nz : 1));
    const double tt0 = wall_s();
    if (fast_transpose) transpose2_par(n, nz, rp, ci, v, trp, tci, tv, int_transpose);
    else transpose2(n, nz, rp, ci, v, trp, tci, tv, int_transpose);     /* :99  */
    if (t_transpose) *t_transpose = wall_s() - tt0;

    double *R = dalloc(n), *biR = dalloc(n), *nR = dalloc(n), *nbiR = dalloc(n);
    double *P = dalloc(n), *biP = dalloc(n), *nP = dalloc(n), *nbiP = dalloc(n);
    double *multAP = dalloc(n), *multAtbiP = dalloc(n), *tmp;
    double alfa, beta, numerator, denominator, check, norm;
    int i, iter;
    const int team_before = omp_get_max_threads();

    for (int run = 0; run < (n_teams > 0 ? n_teams : 1); run++) {
    if (n_teams > 0) omp_set_num_threads(teams[run] > 0 ? teams[run] : 1);
    norm = sqrt(orc_dot(n, b, b));                                      /* :135 */
    for (i = 0; i < n; i++) x[i] = 1.0;                                 /* :139 */
    orc_spmv(n, rp, ci, v, x, multAP);                                  /* :142 */
    for (i = 0; i < n; i++)
        R[i] = biR[i] = P[i] = biP[i] = b[i] - multAP[i];               /* :144 */

    const double tl0 = wall_s();
    for (iter = 0; iter < maxit; iter++) {                              /* :146 */
        orc_spmv(n, rp, ci, v, P, multAP);                              /* :147 */
        orc_spmv(n, trp, tci, tv, biP, multAtbiP);                      /* :148 */
        numerator = orc_dot(n, biR, R);                                 /* :149 */
        denominator = orc_dot(n, biP, multAP);                          /* :150 */
        alfa = numerator / denominator;                                 /* :151 */
#pragma omp parallel for schedule(static) if (parallel_vec)
        for (i = 0; i < n; i++) nR[i] = R[i] - alfa * multAP[i];        /* :152 */
#pragma omp parallel for schedule(static) if (parallel_vec)
        for (i = 0; i < n; i++) nbiR[i] = biR[i] - alfa * multAtbiP[i]; /* :154 */
        denominator = numerator;                                        /* :156 */
        numerator = orc_dot(n, nbiR, nR);                               /* :157 */
        beta = numerator / denominator;                                 /* :158 */
#pragma omp parallel for schedule(static) if (parallel_vec)
        for (i = 0; i < n; i++) nP[i] = nR[i] + beta * P[i];            /* :159 */
#pragma omp parallel for schedule(static) if (parallel_vec)
        for (i = 0; i < n; i++) nbiP[i] = nbiR[i] + beta * biP[i];      /* :161 */
        /* the test uses the OLD R and leaves before x is updated (:164-166) */
        check = sqrt(orc_dot(n, R, R)) / norm;                          /* :164 */
        if (check < eps) break;                                         /* :165 */
#pragma omp parallel for schedule(static) if (parallel_vec)
        for (i = 0; i < n; i++) x[i] += alfa * P[i];                    /* :167 */
        tmp = R; R = nR; nR = tmp;                                      /* :170 */
        tmp = P; P = nP; nP = tmp;
        tmp = biR; biR = nbiR; nbiR = tmp;
        tmp = biP; biP = nbiP; nbiP = tmp;                              /* :181 */
    }
    if (iters) iters[run] = iter;
    if (t_loop) t_loop[run] = wall_s() - tl0;
    }
    if (n_teams > 0) omp_set_num_threads(team_before);

    free(R); free(biR); free(nR); free(nbiR);
    free(P); free(biP); free(nP); free(nbiP);
    free(multAP); free(multAtbiP);
    free(trp); free(tci); free(tv); free(rp); free(ci);
    return 0;
}

/* =========================================================================
 * pbicgstab.cu:45-154  gpu_pbicgstab : right-preconditioned BiCGSTAB, M = LU
 * from ILU(0) (or M = I when vm == NULL).  Every cuBLAS/cuSPARSE call of the
 * reference appears as one call below, in the reference's order.
 * ========================================================================= */
static void precond_apply(int n, const int *rp, const int *ci, const double *vm,
                          const double *in, double *t, double *out)
{
    if (vm) {
        orc_trsv_lower_unit(n, rp, ci, vm, in, t);   /* :92-94  / :121-123 */
        orc_trsv_upper(n, rp, ci, vm, t, out);       /* :96-98  / :125-127 */
    } else {
        memcpy(t, in, sizeof(double) * (size_t)n);   /* L = I */
        memcpy(out, t, sizeof(double) * (size_t)n);  /* U = I */
    }
}

/* sum |a_i b_i|: the scale the rounding error of orc_dot(a, b) is relative to */
static double abs_dot(int n, const double *a, const double *b)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) s += fabs(a[i] * b[i]);
    return s;
}

int orc_pbicgstab(int n, const int *rp, const int *ci, const double *a,
                  const double *vm, const double *f, double *x,
                  int maxit, double tol, double *hist, int hist_cap,
                  orc_stats *st)
{
    return orc_pbicgstab_ex(n, rp, ci, a, vm, f, x, maxit, tol, hist, hist_cap, NULL, 0, st);
}

/* the same loop, also recording its scalars: trace[8 i + ...] = rho (:81), sum |rw_j r_j|, rw.v (:106), sum |rw_j v_j|,
 * alpha (:107), t.r (:135), t.t (:136), omega (:137) of iteration i (NaN where the iteration ended at the half step).
 * The reference has no breakdown guard in this loop (:81,107,137 divide whatever they get): a caller can tell from the
 * trace when rho or rw.v fell below the rounding error of its own summation (|rho| <= n eps sum|rw_j r_j|), i.e. when the
 * loop began to divide noise by noise -- tests/test_gpu_nondominant.py classifies outcomes with it. */
int orc_pbicgstab_ex(int n, const int *rp, const int *ci, const double *a,
                     const double *vm, const double *f, double *x,
                     int maxit, double tol, double *hist, int hist_cap,
                     double *trace, int trace_iters, orc_stats *st)
{
    double *r = dalloc(n), *rw = dalloc(n), *p = dalloc(n), *pw = dalloc(n);
    double *s = dalloc(n), *t = dalloc(n), *v = dalloc(n);
    double rho, rhop, beta, alpha = 0.0, negalpha, omega = 0.0, negomega, temp, temp2;
    double nrmr = 0.0, nrmr0;
    int i = 0, half_exit = 0, converged = 0;
    rho = 0.0;

    orc_csrmv(n, rp, ci, a, 1.0, x, 0.0, r);                 /* :67  r = A x      */
    orc_scal(n, -1.0, r);                                     /* :69               */
    orc_axpy(n, 1.0, f, r);                                   /* :70  r = f - A x  */
    memcpy(rw, r, sizeof(double) * (size_t)n);                /* :72               */
    memcpy(p, r, sizeof(double) * (size_t)n);                 /* :73               */
    nrmr0 = orc_nrm2(n, r);                                   /* :74               */
    nrmr = nrmr0;

    for (i = 0; i < maxit;) {                                 /* :79               */
        rhop = rho;                                           /* :80               */
        rho = orc_dot(n, rw, r);                              /* :81               */
        double *tr = (trace && i < trace_iters) ? trace + 8 * (size_t)i : NULL;
        if (tr) {
            for (int q = 0; q < 8; q++) tr[q] = NAN;
            tr[0] = rho; tr[1] = abs_dot(n, rw, r);
        }
        if (i > 0) {                                          /* :83               */
            beta = (rho / rhop) * (alpha / omega);            /* :84               */
            negomega = -omega;
            orc_axpy(n, negomega, v, p);                      /* :86               */
            orc_scal(n, beta, p);                             /* :87               */
            orc_axpy(n, 1.0, r, p);                           /* :88               */
        }
        precond_apply(n, rp, ci, vm, p, t, pw);               /* :92-98            */
        orc_csrmv(n, rp, ci, a, 1.0, pw, 0.0, v);             /* :104 v = A pw     */
        temp = orc_dot(n, rw, v);                             /* :106              */
        alpha = rho / temp;                                   /* :107              */
        if (tr) { tr[2] = temp; tr[3] = abs_dot(n, rw, v); tr[4] = alpha; }
        negalpha = -alpha;
        orc_axpy(n, negalpha, v, r);                          /* :109              */
        orc_axpy(n, alpha, pw, x);                            /* :110              */
        nrmr = orc_nrm2(n, r);                                /* :111              */
        if (hist && 2 * i < hist_cap) hist[2 * i] = nrmr;
        if (nrmr < tol * nrmr0) {                             /* :116              */
            half_exit = 1; converged = 1;
            break;
        }
        precond_apply(n, rp, ci, vm, r, t, s);                /* :121-127          */
        orc_csrmv(n, rp, ci, a, 1.0, s, 0.0, t);              /* :132 t = A s      */
        temp = orc_dot(n, t, r);                              /* :135              */
        temp2 = orc_dot(n, t, t);                             /* :136              */
        omega = temp / temp2;                                 /* :137              */
        if (tr) { tr[5] = temp; tr[6] = temp2; tr[7] = omega; }
        negomega = -omega;
        orc_axpy(n, omega, s, x);                             /* :139              */
        orc_axpy(n, negomega, t, r);                          /* :140              */
        nrmr = orc_nrm2(n, r);                                /* :142              */
        if (hist && 2 * i + 1 < hist_cap) hist[2 * i + 1] = nrmr;
        if (nrmr < tol * nrmr0) {                             /* :147              */
            i++;
            converged = 1;
            break;
        }
        i++;                                                  /* :151              */
    }
    if (st) {
        st->iters = i; st->half_exit = half_exit; st->converged = converged;
        st->breakdown = 0; st->nrm0 = nrmr0; st->nrm = nrmr;
    }
    free(r); free(rw); free(p); free(pw); free(s); free(t); free(v);
    return 1; /* the reference path always reports success (pbicgstab.cu:408) */
}

/* =========================================================================
 * pbicgstab.cu:581-754  gpu_pbicgstab2 (d variant).  With d == NULL the
 * mult_spec terms vanish, which is the intended maths of the plain variant
 * (:425-578) whose `r += b; r0 = r` lines are commented out (:471-478, D1).
 * The copy/scal/axpy triplets are kept exactly as the reference issues them.
 * ========================================================================= */
static void shifted_mv(int n, const int *rp, const int *ci, const double *a0,
                       const double *d, double k, double alpha,
                       const double *x, double *y)
{
    if (d) {
        orc_mult_spec(n, x, d, k, y);                    /* :645 / :675 / :703 */
        orc_csrmv(n, rp, ci, a0, alpha, x, 1.0, y);      /* :646 / :676 / :704 */
    } else {
        orc_csrmv(n, rp, ci, a0, alpha, x, 0.0, y);      /* :469 / :501 / :528 */
    }
}

int orc_pbicgstab2(int n, const int *rp, const int *ci, const double *a0,
                   const double *d, const double *x0_in, const double *b,
                   int maxit, double tol, double *x, double *hist, int hist_cap,
                   orc_stats *st)
{
    const size_t nb = sizeof(double) * (size_t)n;
    double *x0 = dalloc(n), *r0 = dalloc(n), *r = dalloc(n), *r_ = dalloc(n);
    double *v = dalloc(n), *v_ = dalloc(n), *p = dalloc(n), *p_ = dalloc(n);
    double *s = dalloc(n), *t = dalloc(n), *h = dalloc(n);
    double omega = 1, alpha = 1, beta = 0, rho = 1, rho_ = rho;   /* :614-618 */
    double norm0, norm = 0.0;
    int result = 0, breakdown = 0, converged = 0, it = 0;
    (void)beta;

    memcpy(x0, x0_in, nb);
    memset(x, 0, nb);                                             /* :1001 */

    shifted_mv(n, rp, ci, a0, d, -1.0, -1.0, x0, r);              /* :645-646 */
    orc_axpy(n, 1.0, b, r);                                       /* :649 */
    memcpy(r0, r, nb);                                            /* :652 */
    norm0 = orc_nrm2(n, r);                                       /* :655 */
    norm = norm0;

    for (int i = 0; i < maxit; i++) {                             /* :662 */
        it = i;
        rho_ = orc_dot(n, r0, r);                                 /* :665 */
        beta = (rho_ / rho) * (alpha / omega);                    /* :666 */
        double momega = -omega;
        memcpy(p_, v, nb);                                        /* :668 */
        orc_scal(n, momega, p_);                                  /* :669 */
        orc_axpy(n, 1.0, p, p_);                                  /* :670 */
        orc_scal(n, beta, p_);                                    /* :671 */
        orc_axpy(n, 1.0, r, p_);                                  /* :672 */

        shifted_mv(n, rp, ci, a0, d, 1.0, 1.0, p_, v_);           /* :675-676 */

        double dot_r_v = orc_dot(n, r0, v_);                      /* :688 */
        alpha = rho_ / dot_r_v;                                   /* :689 */
        double malpha = -alpha;

        memcpy(h, p_, nb);                                        /* :694 */
        orc_scal(n, alpha, h);                                    /* :695 */
        orc_axpy(n, 1.0, x0, h);                                  /* :696 */

        memcpy(s, v_, nb);                                        /* :698 */
        orc_scal(n, malpha, s);                                   /* :699 */
        orc_axpy(n, 1.0, r, s);                                   /* :700 */

        shifted_mv(n, rp, ci, a0, d, 1.0, 1.0, s, t);             /* :703-704 */

        double num = orc_dot(n, t, s);                            /* :708 */
        double denum = orc_dot(n, t, t);                          /* :709 */
        omega = num / denum;                                      /* :710 */
        momega = -omega;

        memcpy(x, s, nb);                                         /* :714 */
        orc_scal(n, omega, x);                                    /* :715 */
        orc_axpy(n, 1.0, h, x);                                   /* :716 */

        memcpy(r_, t, nb);                                        /* :718 */
        orc_scal(n, momega, r_);                                  /* :719 */
        orc_axpy(n, 1.0, s, r_);                                  /* :720 */

        norm = orc_nrm2(n, r_);                                   /* :723 */
        if (hist && i < hist_cap) hist[i] = norm;
        it = i + 1;
        if (norm < tol * norm0) {                                 /* :730 */
            result = 1; converged = 1;
            break;
        }
        if (fabs(omega) < 1e-5 || isnan(omega)) {                 /* :735 */
            breakdown = 1;
            break;
        }
        memcpy(r, r_, nb);                                        /* :744 */
        memcpy(p, p_, nb);                                        /* :745 */
        memcpy(v, v_, nb);                                        /* :746 */
        memcpy(x0, x, nb);                                        /* :747 */
        rho = rho_;                                               /* :748 */
    }
    if (st) {
        st->iters = it; st->half_exit = 0; st->converged = converged;
        st->breakdown = breakdown; st->nrm0 = norm0; st->nrm = norm;
    }
    free(x0); free(r0); free(r); free(r_); free(v); free(v_);
    free(p); free(p_); free(s); free(t); free(h);
    return result;
}


/* =========================================================================
 * Pipelined BiCGStab (second half).  NOT in the reference: a re-arrangement of the
 * recurrences of pbicgstab.cu:45-154 (M = I) after Cools & Vanroose, "The communication-hiding pipelined
 * BiCGStab method for the parallel solution of large unsymmetric linear systems", Parallel Computing 65
 * (2017), Alg. 4, so that each of the two reduction phases of an iteration can run WHILE an SpMV runs.
 * In exact arithmetic the iterates are those of BiCGSTAB; in floating point they differ by rounding.
 *   s = A p, z = A s, v = A z, w = A r, t = A w, q = r - alpha s ("s" of pbicgstab.cu:109), y = A q
 *   p_i = r_i + beta (p - omega s);  s_i = w_i + beta (s - omega z);  z_i = t_i + beta (z - omega v)
 *   q_i = r_i - alpha s_i;  y_i = w_i - alpha z_i;               [dots (q,y) (y,y) (q,q)]  ||  v_i = A z_i
 *   omega = (q,y)/(y,y);  x += alpha p + omega q;  r' = q - omega y;  w' = y - omega (t - alpha v)
 *                                    [dots (rw,r') (rw,w') (rw,s) (rw,z) (r',r')]  ||  t' = A w'
 *   beta = (alpha/omega) (rw,r')/(rw,r);  alpha' = (rw,r') / ((rw,w') + beta (rw,s) - beta omega (rw,z))
 * Stopping tests as in the reference loop: half step on ||q|| (:116, x += alpha p only), full step on ||r'||
 * (:147), both against tol * ||r0||; x0 as passed.  hist as orc_pbicgstab.
 * ========================================================================= */
int orc_pipelined_bicgstab(int n, const int *rp, const int *ci, const double *a, const double *f, double *x,
                           int maxit, double tol, double *hist, int hist_cap, orc_stats *st)
{
    return orc_ppipelined_bicgstab(n, rp, ci, a, NULL, f, x, maxit, tol, 0, hist, hist_cap, st);
}

/* =========================================================================
 * The PRECONDITIONED pipelined loop with RESIDUAL REPLACEMENT ("lifts the 8-GPU preconditioned
 * case").  Cools & Vanroose 2017, the preconditioned form of their Alg. 4 (right preconditioning, M^-1 applied where
 * pbicgstab.cu:92-98,121-127 apply it: in front of each of the two SpMVs of an iteration) and their residual
 * replacement.  Hatted vectors are M^-1 times the plain ones; with vm == NULL (M = I) they ARE the plain ones and the
 * loop is the un-preconditioned one above, operation for operation.
 *   rh = M^-1 r, w = A rh, wh = M^-1 w, t = A wh;   s = A ph, z = A sh, v = A zh
 *   ph = rh + beta (ph - omega sh);  s = w + beta (s - omega z);  sh = wh + beta (sh - omega zh);  z = t + beta (z - omega v)
 *   q = r - alpha s;  qh = rh - alpha sh;  y = w - alpha z;      [dots (q,y) (y,y) (q,q)]  ||  zh = M^-1 z, v = A zh
 *   omega = (q,y)/(y,y);  x += alpha ph + omega qh;  r' = q - omega y;  rh' = qh - omega (wh - alpha zh);
 *   w' = y - omega (t - alpha v)              [dots (rw,r') (rw,w') (rw,s) (rw,z) (r',r')]  ||  wh' = M^-1 w', t' = A wh'
 * Residual replacement, every `rr` iterations (rr = 0: never), right after x, r', rh', w' of that iteration:
 *   r' = f - A x;  rh' = M^-1 r';  w' = A rh';  s = A ph;  sh = M^-1 s;  z = A sh;  zh = M^-1 z;  v = A zh
 * -- every carried vector is recomputed from x and ph, so the rounding errors the recurrences have accumulated are
 * discarded (the published replacement renews r, w, s, z; zh and v are renewed here too, so that the next z recurrence
 * is consistent with the renewed z: one more solve and product per replacement).
 * Stopping tests on the UN-preconditioned residuals ||q||, ||r'|| as in the reference loop (:116, :147).
 * ========================================================================= */
int orc_ppipelined_bicgstab(int n, const int *rp, const int *ci, const double *a, const double *vm, const double *f, double *x,
                            int maxit, double tol, int rr, double *hist, int hist_cap, orc_stats *st)
{
    double *r = dalloc(n), *rw = dalloc(n), *ph = dalloc(n), *s = dalloc(n), *z = dalloc(n), *v = dalloc(n);
    double *w = dalloc(n), *t = dalloc(n), *q = dalloc(n), *y = dalloc(n), *tmp = dalloc(n);
    /* the hatted copies exist only with a preconditioner */
    double *rh = vm ? dalloc(n) : r, *wh = vm ? dalloc(n) : w, *sh = vm ? dalloc(n) : s, *zh = vm ? dalloc(n) : z;
    double *qh = vm ? dalloc(n) : q;
    double alpha = 0.0, beta = 0.0, omega = 0.0, rho, rho_new, rw_w, rw_s = 0.0, rw_z = 0.0, nrmr, nrmr0;
    int i = 0, k, half_exit = 0, converged = 0;

    orc_csrmv(n, rp, ci, a, 1.0, x, 0.0, r);
    for (k = 0; k < n; k++) r[k] = f[k] - r[k];               /* r0 = f - A x0 (:67-70) */
    memcpy(rw, r, sizeof(double) * (size_t)n);                /* shadow residual (:72)  */
    nrmr0 = orc_nrm2(n, r);
    nrmr = nrmr0;
    if (vm) precond_apply(n, rp, ci, vm, r, tmp, rh);         /* rh0 = M^-1 r0 */
    orc_csrmv(n, rp, ci, a, 1.0, rh, 0.0, w);                 /* w0 = A rh0 */
    if (vm) precond_apply(n, rp, ci, vm, w, tmp, wh);         /* wh0 = M^-1 w0 */
    orc_csrmv(n, rp, ci, a, 1.0, wh, 0.0, t);                 /* t0 = A wh0 */
    rho = orc_dot(n, rw, r);
    rw_w = orc_dot(n, rw, w);
    if (nrmr0 == 0.0) { converged = 1; maxit = 0; }
    for (i = 0; i < maxit;) {
        if (i == 0) {
            alpha = rho / rw_w;
            for (k = 0; k < n; k++) { ph[k] = rh[k]; s[k] = w[k]; z[k] = t[k]; }
            if (vm) memcpy(sh, wh, sizeof(double) * (size_t)n);
        } else {
            alpha = rho / (rw_w + beta * rw_s - beta * omega * rw_z);
            for (k = 0; k < n; k++) {
                ph[k] = rh[k] + beta * (ph[k] - omega * sh[k]);
                if (vm) sh[k] = wh[k] + beta * (sh[k] - omega * zh[k]);
                s[k] = w[k] + beta * (s[k] - omega * z[k]);
                z[k] = t[k] + beta * (z[k] - omega * v[k]);
            }
        }
        for (k = 0; k < n; k++) {
            q[k] = r[k] - alpha * s[k];
            if (vm) qh[k] = rh[k] - alpha * sh[k];
            y[k] = w[k] - alpha * z[k];
        }
        const double qy = orc_dot(n, q, y), yy = orc_dot(n, y, y);
        nrmr = orc_nrm2(n, q);
        if (vm) precond_apply(n, rp, ci, vm, z, tmp, zh);     /* zh = M^-1 z (:92-98) */
        orc_csrmv(n, rp, ci, a, 1.0, zh, 0.0, v);             /* v = A zh  (overlaps the reduction) */
        if (hist && 2 * i < hist_cap) hist[2 * i] = nrmr;
        if (nrmr < tol * nrmr0) {                             /* half-step exit (:116) */
            for (k = 0; k < n; k++) x[k] += alpha * ph[k];
            half_exit = 1; converged = 1;
            break;
        }
        omega = qy / yy;
        for (k = 0; k < n; k++) {
            x[k] += alpha * ph[k] + omega * qh[k];
            r[k] = q[k] - omega * y[k];
            if (vm) rh[k] = qh[k] - omega * (wh[k] - alpha * zh[k]);
            w[k] = y[k] - omega * (t[k] - alpha * v[k]);
        }
        if (rr > 0 && (i + 1) % rr == 0) {                    /* residual replacement */
            orc_csrmv(n, rp, ci, a, 1.0, x, 0.0, tmp);
            for (k = 0; k < n; k++) r[k] = f[k] - tmp[k];
            if (vm) precond_apply(n, rp, ci, vm, r, tmp, rh);
            orc_csrmv(n, rp, ci, a, 1.0, rh, 0.0, w);
            orc_csrmv(n, rp, ci, a, 1.0, ph, 0.0, s);
            if (vm) precond_apply(n, rp, ci, vm, s, tmp, sh);
            orc_csrmv(n, rp, ci, a, 1.0, sh, 0.0, z);
            if (vm) precond_apply(n, rp, ci, vm, z, tmp, zh);
            orc_csrmv(n, rp, ci, a, 1.0, zh, 0.0, v);
        }
        rho_new = orc_dot(n, rw, r);
        rw_w = orc_dot(n, rw, w);
        rw_s = orc_dot(n, rw, s);
        rw_z = orc_dot(n, rw, z);
        nrmr = orc_nrm2(n, r);
        if (vm) precond_apply(n, rp, ci, vm, w, tmp, wh);     /* wh = M^-1 w (:121-127) */
        orc_csrmv(n, rp, ci, a, 1.0, wh, 0.0, t);             /* t = A wh  (overlaps the reduction) */
        beta = (alpha / omega) * (rho_new / rho);
        rho = rho_new;
        if (hist && 2 * i + 1 < hist_cap) hist[2 * i + 1] = nrmr;
        i++;
        if (nrmr < tol * nrmr0) { converged = 1; break; }     /* full-step exit (:147) */
    }
    if (st) {
        st->iters = i; st->half_exit = half_exit; st->converged = converged; st->breakdown = 0;
        st->nrm0 = nrmr0; st->nrm = nrmr;
    }
    free(r); free(rw); free(ph); free(s); free(z); free(v); free(w); free(t); free(q); free(y); free(tmp);
    if (vm) { free(rh); free(wh); free(sh); free(zh); free(qh); }
    return 0;
}
