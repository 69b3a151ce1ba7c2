// small.hip -- whole BiCGSTAB solve in ONE launch of ONE workgroup, for systems that fit a CU's reach.
//
// mat10000.mtx (BASELINE configs[1]: 10 000 rows, 49 600 entries, 0.6 MB) is L2-resident; with five
// launches per iteration the multi-kernel loop is bound by launch + dependent-kernel-boundary latency
// (~25 us / iteration), i.e. by nothing the matrix costs.  Here a 1024-thread workgroup runs the entire
// loop of pbicgstab.cu:45-154 (or :581-754) with workgroup barriers between the phases: no launches,
// no host involvement, every scalar in registers.  Same arithmetic and rounding order as the fused
// kernels of kernels.hip; reductions are wave-shuffle -> LDS -> fixed-order sum (deterministic).
#include "kernels.h"

namespace cm {

constexpr int kSmallThreads = 1024;
constexpr int kSmallWaves = kSmallThreads / 64;

struct SmallArgs {
    int n;
    const int *rp, *ci;
    const double *val, *d, *b;
    double *x, *r, *rw, *p, *v, *t;
    int maxit;
    double tol;
    int loop, no_exit, x0_ones;
    double *hist;
    int hist_cap;
    LoopState *st;
};

template <int K>
__device__ __forceinline__ void wg_sum(double (&v)[K], double *red)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[wave * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kSmallWaves; w++) s += red[w * K + k];
        v[k] = s;
    }
}

// y = A x + d.*x for the whole matrix, L lanes per row; returns partial sums of y.w and y.y
template <int L>
__device__ __forceinline__ void wg_spmv(const SmallArgs &a, const double *x, double *y, const double *w,
                                        double &acc0, double &acc1)
{
    const int lane = threadIdx.x & (L - 1), group = threadIdx.x / L;
    constexpr int RPP = kSmallThreads / L;
    for (int row = group; row < a.n; row += RPP) {
        const int s = a.rp[row], e = a.rp[row + 1];
        double sum = 0.0;
        for (int k = s + lane; k < e; k += L) sum += a.val[k] * x[a.ci[k]];
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0) {
            if (a.d) sum += a.d[row] * x[row];
            y[row] = sum;
            if (w) {
                acc0 += sum * w[row];
                acc1 += sum * sum;
            }
        }
    }
}

template <int L>
__global__ __launch_bounds__(kSmallThreads) void k_bicgstab_small(SmallArgs a)
{
    __shared__ double red[2 * kSmallWaves];
    const int tid = threadIdx.x, n = a.n;
    double *x = a.x, *r = a.r, *rw = a.rw, *p = a.p, *v = a.v, *t = a.t;
    if (a.x0_ones)
        for (int i = tid; i < n; i += kSmallThreads) x[i] = 1.0;
    __syncthreads();
    double z0 = 0.0, z1 = 0.0;
    wg_spmv<L>(a, x, r, nullptr, z0, z1);                               // r = A x0      (:67 / :645-646)
    __syncthreads();
    double acc1[1] = {0.0};
    for (int i = tid; i < n; i += kSmallThreads) {
        const double rr = a.b[i] - r[i];                                // r = f - A x   (:69-70)
        r[i] = rr; rw[i] = rr; p[i] = rr;                               // :72-73
        acc1[0] += rr * rr;
    }
    wg_sum<1>(acc1, red);
    const double nrm0 = sqrt(acc1[0]);                                  // :74
    const double tolabs = a.tol * nrm0;
    double full[2] = {acc1[0], acc1[0]};                                // (rw.r, r.r)
    double rho = 1.0, rhop = 1.0, alpha = 1.0, omega = 1.0, nrm = nrm0;
    int it = 0, state = 0;
    const bool loop1 = a.loop == CUDAMAT_LOOP_PBICGSTAB;
    for (int k = 0; k < a.maxit; k++) {
        if (it > 0) {                                                   // full-step test (:142-151 / :723-742)
            nrm = sqrt(full[1]);
            if (tid == 0 && a.hist) {
                const int slot = loop1 ? 2 * (it - 1) + 1 : it - 1;
                if (slot < a.hist_cap) a.hist[slot] = nrm;
            }
            if (!a.no_exit) {
                if (nrm < tolabs) { state = 2; break; }
                if (!loop1 && (fabs(omega) < 1e-5 || isnan(omega))) { state = 3; break; }
            }
        }
        rhop = rho;
        rho = full[0];                                                  // :81
        if (it > 0) {
            const double beta = (rho / rhop) * (alpha / omega);         // :84
            const double nomega = -omega;
            for (int i = tid; i < n; i += kSmallThreads) {
                double pp = fma(nomega, v[i], p[i]);                    // :86
                pp = beta * pp;                                         // :87
                p[i] = r[i] + pp;                                       // :88
            }
        }
        __syncthreads();
        double rv[2] = {0.0, 0.0};
        wg_spmv<L>(a, p, v, rw, rv[0], rv[1]);                          // v = A p, rw.v (:104-106)
        wg_sum<2>(rv, red);
        alpha = rho / rv[0];                                            // :107
        const double nalpha = -alpha;
        double h2[1] = {0.0};
        for (int i = tid; i < n; i += kSmallThreads) {
            const double rr = fma(nalpha, v[i], r[i]);                  // :109
            r[i] = rr;
            x[i] = fma(alpha, p[i], x[i]);                              // :110
            h2[0] += rr * rr;                                           // :111
        }
        wg_sum<1>(h2, red);
        if (loop1) {                                                    // half-step test (:116)
            nrm = sqrt(h2[0]);
            if (tid == 0 && a.hist && 2 * it < a.hist_cap) a.hist[2 * it] = nrm;
            if (!a.no_exit && nrm < tolabs) { state = 1; break; }
        }
        double tt[2] = {0.0, 0.0};
        wg_spmv<L>(a, r, t, r, tt[0], tt[1]);                           // t = A s, (t.r, t.t) (:132-136)
        wg_sum<2>(tt, red);
        omega = tt[0] / tt[1];                                          // :137
        const double nomega2 = -omega;
        full[0] = 0.0;
        full[1] = 0.0;
        for (int i = tid; i < n; i += kSmallThreads) {
            x[i] = fma(omega, r[i], x[i]);                              // :139 (s = r)
            const double rr = fma(nomega2, t[i], r[i]);                 // :140
            r[i] = rr;
            full[0] += rw[i] * rr;
            full[1] += rr * rr;                                         // :142
        }
        wg_sum<2>(full, red);
        it++;                                                           // :148 / :151
    }
    if (state == 0 && it > 0) {                                         // test of the last iteration
        nrm = sqrt(full[1]);
        if (tid == 0 && a.hist) {
            const int slot = loop1 ? 2 * (it - 1) + 1 : it - 1;
            if (slot < a.hist_cap) a.hist[slot] = nrm;
        }
        if (!a.no_exit) {
            if (nrm < tolabs) state = 2;
            else if (!loop1 && (fabs(omega) < 1e-5 || isnan(omega))) state = 3;
        }
    }
    if (tid == 0) {
        LoopState *st = a.st;
        st->state = state;
        st->it = it;
        st->rho[0] = rho;
        st->rho[1] = rhop;
        st->alpha = alpha;
        st->omega = omega;
        st->nrm0 = nrm0;
        st->tolabs = tolabs;
        st->nrm = nrm;
    }
}

int launch_bicgstab_small(hipStream_t s, int n, int64_t nnz, const int *rp, const int *ci, const double *val,
                          const double *d, const double *b, double *x, double *r, double *rw, double *p, double *v,
                          double *t, int maxit, double tol, int loop, int no_exit, int x0_ones, double *hist,
                          int hist_cap, LoopState *st)
{
    SmallArgs a{n, rp, ci, val, d, b, x, r, rw, p, v, t, maxit, tol, loop, no_exit, x0_ones, hist, hist_cap, st};
    const double mean = n > 0 ? (double)nnz / n : 1.0;
    if (mean <= 6.0) hipLaunchKernelGGL(k_bicgstab_small<4>, dim3(1), dim3(kSmallThreads), 0, s, a);
    else if (mean <= 12.0) hipLaunchKernelGGL(k_bicgstab_small<8>, dim3(1), dim3(kSmallThreads), 0, s, a);
    else if (mean <= 40.0) hipLaunchKernelGGL(k_bicgstab_small<16>, dim3(1), dim3(kSmallThreads), 0, s, a);
    else hipLaunchKernelGGL(k_bicgstab_small<32>, dim3(1), dim3(kSmallThreads), 0, s, a);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

}  // namespace cm
