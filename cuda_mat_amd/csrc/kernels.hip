// kernels.hip -- the streaming vector kernels of the BiCGSTAB inner loop (gfx950) and the BLAS-1 pieces.
//
// Everything on this path is HBM-bandwidth-bound fp64 streaming / gather work (<= 0.17 flop/byte): no MFMA.
// What matters is (a) coalesced 16-byte-per-lane loads on the streamed vectors, (b) one pass per fused update instead of
// the reference's copy/scal/axpy triplets (pbicgstab.cu:86-88,109-110,139-140,668-672,...), (c) dot products produced by
// the kernel that already streams the operands, reduced wave64-shuffle -> LDS -> per-workgroup partial -> fixed-order
// sum in the consumer's prologue (bitwise reproducible, no atomics, no host sync).
// Scalars (rho, alpha, omega, norms) never leave the device: see LoopState.
// The SpMV forms on CSR live in spmv_csr.hip, the pipelined loop's kernels in pipelined.hip, the three- and one-launch
// loops of small systems in small_loops.hip.
#include <algorithm>
#include <cstring>
#include <vector>

#include "kernels.h"
#include "device.h"

namespace cm {

__global__ __launch_bounds__(kBlock) void k_check(LoopArgs la, ScalarSrc src, int which)
{
    __shared__ double lds[8];
    if (which != CHECK_HALF && uniform_state(la.st) != 0) return;   // (check_half reads the state itself)
    if (which == CHECK_HALF) {
        check_half(la, src, lds);
    } else {
        double sc[2];
        load_scalars<2>(src, sc, lds);
        check_full(la, sc);
    }
}

int launch_check(hipStream_t s, LoopArgs la, ScalarSrc src, int which)
{
    hipLaunchKernelGGL(k_check, dim3(1), dim3(kBlock), 0, s, la, src, which);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

__global__ __launch_bounds__(kBlock) void k_reduce_parts(ScalarSrc in, int K, double *out, int sqrt_it)
{
    __shared__ double lds[8];
    for (int k = 0; k < K; k++) {
        ScalarSrc one{in.ptr + k, in.count, in.stride};
        double sc[1];
        load_scalars<1>(one, sc, lds);
        if (threadIdx.x == 0) out[k] = sqrt_it ? sqrt(sc[0]) : sc[0];
    }
}

int launch_reduce_parts(hipStream_t s, ScalarSrc in, int K, double *out, int sqrt_it)
{
    hipLaunchKernelGGL(k_reduce_parts, dim3(1), dim3(kBlock), 0, s, in, K, out, sqrt_it);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

// ------------------------------------------------------- streaming vector kernels
// 16 bytes per lane (double2) whenever every operand is 16-byte aligned; a fixed
// grid (<= kVecGridMax workgroups) walks the vector grid-stride so that the number
// of partial sums is bounded and the reduction order depends on n only.
int vec_grid(int64_t n)
{
    int64_t g = (n / 2 + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > kVecGridMax) g = kVecGridMax;
    return (int)g;
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_init(int64_t n, const double *b, double *r, double *rw,
                                                 double *p, double *parts)
{
    __shared__ double lds[8];
    double acc[1] = {0.0};
    CM_VEC_LOOP(n,
        {
            const double2 bb = ((const double2 *)b)[i];
            double2 rr = ((double2 *)r)[i];
            rr.x = bb.x - rr.x; rr.y = bb.y - rr.y;        // r = f - A x (pbicgstab.cu:67-70)
            ((double2 *)r)[i] = rr; ((double2 *)rw)[i] = rr; ((double2 *)p)[i] = rr;  // :72-73
            acc[0] += rr.x * rr.x; acc[0] += rr.y * rr.y;
        },
        {
            const double rr = b[i] - r[i];
            r[i] = rr; rw[i] = rr; p[i] = rr;
            acc[0] += rr * rr;
        })
    block_sum<1>(acc, lds);
    if (threadIdx.x == 0) {
        parts[2 * blockIdx.x] = acc[0];       // rho0 = rw.r = r.r
        parts[2 * blockIdx.x + 1] = acc[0];   // ||r0||^2
    }
}

int launch_init(hipStream_t s, int64_t n, const double *b, double *r, double *rw, double *p,
                double *parts, int *nparts)
{
    const int g = vec_grid(n);
    *nparts = g;
    if (aligned16(b) && aligned16(r) && aligned16(rw) && aligned16(p))
        hipLaunchKernelGGL(k_init<1>, dim3(g), dim3(kBlock), 0, s, n, b, r, rw, p, parts);
    else
        hipLaunchKernelGGL(k_init<0>, dim3(g), dim3(kBlock), 0, s, n, b, r, rw, p, parts);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

__global__ __launch_bounds__(kBlock) void k_init_finish(LoopState *st, ScalarSrc init, double tol, double abs_tol)
{
    __shared__ double lds[8];
    double sc[2];
    load_scalars<2>(init, sc, lds);
    if (threadIdx.x == 0) {
        const double nrm0 = sqrt(sc[1]);           // pbicgstab.cu:74 / :655
        // x0 already solves the system exactly (r0 = 0): the reference's loop would divide 0 by 0 and hand back NaNs;
        // here the loop starts frozen in the 'converged' state and x0 is returned untouched
        // abs_tol > 0 (a restart that verifies an iterate): stop at that ABSOLUTE residual, and if the residual of the
        // initial guess is already within twice of it (a recursive residual drifts by about that much) there is nothing to do
        st->state = (nrm0 == 0.0 || (abs_tol > 0.0 && nrm0 <= 2.0 * abs_tol)) ? 2 : 0;
        st->it = 0;
        st->rho[0] = 1.0;                          // pbicgstab.cu:617 (rho = 1)
        st->rho[1] = 1.0;
        st->alpha = 1.0;                           // :615
        st->omega = 1.0;                           // :614
        st->nrm0 = nrm0;
        st->tolabs = abs_tol > 0.0 ? abs_tol : tol * nrm0;
        st->nrm = nrm0;
    }
}

int launch_init_finish(hipStream_t s, LoopState *st, ScalarSrc init, double tol, double abs_tol)
{
    hipLaunchKernelGGL(k_init_finish, dim3(1), dim3(kBlock), 0, s, st, init, tol, abs_tol);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

// p = r + beta (p - omega v)          pbicgstab.cu:83-89 (axpy, scal, axpy) fused
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_update_p(LoopArgs la, ScalarSrc full, int64_t n,
                                                     const double *r, double *p, const double *v)
{
    __shared__ double lds[8];
    LoopState *st = la.st;
    if (uniform_state(st) != 0) return;
    const int it = st->it;
    double sc[2];
    load_scalars<2>(full, sc, lds);
    if (check_full(la, sc)) return;
    const double rho = sc[0];                              // :81  rho = rw.r
    const double rhop = st->rho[(it + 1) & 1];             // :80
    const double alpha = st->alpha, omega = st->omega;
    if (leader()) st->rho[it & 1] = rho;
    if (it == 0) return;                                   // :83  p = r already (:73)
    const double beta = (rho / rhop) * (alpha / omega);    // :84
    const double nomega = -omega;
    CM_VEC_LOOP(n,
        {
            const double2 rr = ((const double2 *)r)[i];
            const double2 vv = ((const double2 *)v)[i];
            double2 pp = ((double2 *)p)[i];
            pp.x = fma(nomega, vv.x, pp.x); pp.y = fma(nomega, vv.y, pp.y);   // :86
            pp.x = beta * pp.x;             pp.y = beta * pp.y;               // :87
            pp.x = rr.x + pp.x;             pp.y = rr.y + pp.y;               // :88
            ((double2 *)p)[i] = pp;
        },
        {
            double pp = fma(nomega, v[i], p[i]);
            pp = beta * pp;
            p[i] = r[i] + pp;
        })
}

int launch_update_p(hipStream_t s, LoopArgs la, ScalarSrc full, int64_t n, const double *r,
                    double *p, const double *v)
{
    const int g = vec_grid(n);
    if (aligned16(r) && aligned16(p) && aligned16(v))
        hipLaunchKernelGGL(k_update_p<1>, dim3(g), dim3(kBlock), 0, s, la, full, n, r, p, v);
    else
        hipLaunchKernelGGL(k_update_p<0>, dim3(g), dim3(kBlock), 0, s, la, full, n, r, p, v);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

// alpha = rho/(rw.v); r -= alpha v; ||r||^2     pbicgstab.cu:106-111
// The reference's x += alpha pw (:110) is carried out by k_full of the same iteration (same operation on the same
// operands, in the reference's order; an exit at the half step applies it on the way out, loops.hip): x is read and
// written once per iteration, not twice, and this kernel moves 24 B per row.
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_half(LoopArgs la, ScalarSrc rv, int64_t n, double *r,
                                                 const double *v, double *parts)
{
    __shared__ double lds[8];
    LoopState *st = la.st;
    if (st->state != 0) return;
    const int it = st->it;
    double sc[1];
    load_scalars<1>(rv, sc, lds);
    const double alpha = st->rho[it & 1] / sc[0];          // :107
    const double nalpha = -alpha;
    if (leader()) st->alpha = alpha;
    double acc[1] = {0.0};
    CM_VEC_LOOP(n,
        {
            const double2 vv = ((const double2 *)v)[i];
            double2 rr = ((double2 *)r)[i];
            rr.x = fma(nalpha, vv.x, rr.x); rr.y = fma(nalpha, vv.y, rr.y);   // :109
            ((double2 *)r)[i] = rr;
            acc[0] += rr.x * rr.x; acc[0] += rr.y * rr.y;                     // :111
        },
        {
            const double rr = fma(nalpha, v[i], r[i]);
            r[i] = rr;
            acc[0] += rr * rr;
        })
    block_sum<1>(acc, lds);
    if (threadIdx.x == 0) parts[blockIdx.x] = acc[0];
}

int launch_half(hipStream_t s, LoopArgs la, ScalarSrc rv, int64_t n, double *r, const double *v, double *parts, int *nparts)
{
    const int g = vec_grid(n);
    *nparts = g;
    if (aligned16(r) && aligned16(v))
        hipLaunchKernelGGL(k_half<1>, dim3(g), dim3(kBlock), 0, s, la, rv, n, r, v, parts);
    else
        hipLaunchKernelGGL(k_half<0>, dim3(g), dim3(kBlock), 0, s, la, rv, n, r, v, parts);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

// omega = (t.r)/(t.t); x += omega s; r -= omega t; (rw.r, r.r); it++   pbicgstab.cu:135-151
// s may alias r (no preconditioner): s is read before r is written, per element.
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_full(LoopArgs la, ScalarSrc tt, int64_t n, double *x,
                                                 const double *sv, double *r, const double *t,
                                                 const double *rw, double *parts, ScalarSrc half, const double *pw)
{
    __shared__ double lds[8];
    LoopState *st = la.st;
    // With a half-step test inside this launch (fused loop) the state is read once per workgroup, so a
    // workgroup can never split into waves that saw the leader's exit and waves that go on to update x and r.
    const int frozen = half.ptr ? uniform_state(st) : st->state;
    if (frozen != 0) {                // frozen: still tell the host this iteration's launches have drained
        publish_progress(la, frozen);
        return;
    }
    if (half.ptr && check_half(la, half, lds)) {   // fused small-system loop: the half-step test is evaluated here
        publish_progress(la, 1);
        return;
    }
    double sc[2];
    load_scalars<2>(tt, sc, lds);
    const double omega = sc[0] / sc[1];                    // :137
    const double nomega = -omega;
    double acc[2] = {0.0, 0.0};
    if (!pw) {
        CM_VEC_LOOP(n,
            {
                const double2 ss = ((const double2 *)sv)[i];
                const double2 ttv = ((const double2 *)t)[i];
                const double2 ww = ((const double2 *)rw)[i];
                double2 rr = ((double2 *)r)[i];
                double2 xx = ((double2 *)x)[i];
                xx.x = fma(omega, ss.x, xx.x);   xx.y = fma(omega, ss.y, xx.y);   // :139
                rr.x = fma(nomega, ttv.x, rr.x); rr.y = fma(nomega, ttv.y, rr.y); // :140
                ((double2 *)x)[i] = xx; ((double2 *)r)[i] = rr;
                acc[0] += ww.x * rr.x; acc[0] += ww.y * rr.y;                     // :81 of i+1
                acc[1] += rr.x * rr.x; acc[1] += rr.y * rr.y;                     // :142
            },
            {
                const double ss = sv[i];
                x[i] = fma(omega, ss, x[i]);
                const double rr = fma(nomega, t[i], r[i]);
                r[i] = rr;
                acc[0] += rw[i] * rr;
                acc[1] += rr * rr;
            })
    } else {
        // the half step's x += alpha pw (:110), left out by k_half, first -- then :139: two roundings in the reference's order
        const double alpha = st->alpha;
        CM_VEC_LOOP(n,
            {
                const double2 ss = ((const double2 *)sv)[i];
                const double2 ttv = ((const double2 *)t)[i];
                const double2 ww = ((const double2 *)rw)[i];
                const double2 pp = ((const double2 *)pw)[i];
                double2 rr = ((double2 *)r)[i];
                double2 xx = ((double2 *)x)[i];
                xx.x = fma(alpha, pp.x, xx.x);   xx.y = fma(alpha, pp.y, xx.y);   // :110
                xx.x = fma(omega, ss.x, xx.x);   xx.y = fma(omega, ss.y, xx.y);   // :139
                rr.x = fma(nomega, ttv.x, rr.x); rr.y = fma(nomega, ttv.y, rr.y); // :140
                ((double2 *)x)[i] = xx; ((double2 *)r)[i] = rr;
                acc[0] += ww.x * rr.x; acc[0] += ww.y * rr.y;                     // :81 of i+1
                acc[1] += rr.x * rr.x; acc[1] += rr.y * rr.y;                     // :142
            },
            {
                const double ss = sv[i];
                x[i] = fma(omega, ss, fma(alpha, pw[i], x[i]));
                const double rr = fma(nomega, t[i], r[i]);
                r[i] = rr;
                acc[0] += rw[i] * rr;
                acc[1] += rr * rr;
            })
    }
    block_sum<2>(acc, lds);
    if (threadIdx.x == 0) {
        parts[2 * blockIdx.x] = acc[0];
        parts[2 * blockIdx.x + 1] = acc[1];
    }
    if (leader()) {
        st->omega = omega;
        st->it = st->it + 1;                               // :148 / :151
    }
    publish_progress(la, 0);
}

int launch_full(hipStream_t s, LoopArgs la, ScalarSrc tt, int64_t n, double *x, const double *sv,
                double *r, const double *t, const double *rw, double *parts, int *nparts, ScalarSrc half, const double *pw)
{
    const int g = vec_grid(n);
    *nparts = g;
    if (aligned16(x) && aligned16(sv) && aligned16(r) && aligned16(t) && aligned16(rw) && (!pw || aligned16(pw)))
        hipLaunchKernelGGL(k_full<1>, dim3(g), dim3(kBlock), 0, s, la, tt, n, x, sv, r, t, rw, parts, half, pw);
    else
        hipLaunchKernelGGL(k_full<0>, dim3(g), dim3(kBlock), 0, s, la, tt, n, x, sv, r, t, rw, parts, half, pw);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

// ---------------------------------------------------------------- BLAS-1 pieces
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_dot(int64_t n, const double *x, const double *y,
                                                double *parts)
{
    __shared__ double lds[8];
    double acc[1] = {0.0};
    CM_VEC_LOOP(n,
        {
            const double2 a = ((const double2 *)x)[i];
            const double2 b = ((const double2 *)y)[i];
            acc[0] += a.x * b.x; acc[0] += a.y * b.y;
        },
        { acc[0] += x[i] * y[i]; })
    block_sum<1>(acc, lds);
    if (threadIdx.x == 0) parts[blockIdx.x] = acc[0];
}

int launch_dot_parts(hipStream_t s, int64_t n, const double *x, const double *y, double *parts,
                     int *nparts)
{
    const int g = vec_grid(n);
    *nparts = g;
    if (aligned16(x) && aligned16(y))
        hipLaunchKernelGGL(k_dot<1>, dim3(g), dim3(kBlock), 0, s, n, x, y, parts);
    else
        hipLaunchKernelGGL(k_dot<0>, dim3(g), dim3(kBlock), 0, s, n, x, y, parts);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_axpy(int64_t n, double alpha, const double *x, double *y)
{
    CM_VEC_LOOP(n,
        {
            const double2 a = ((const double2 *)x)[i];
            double2 b = ((double2 *)y)[i];
            b.x = fma(alpha, a.x, b.x); b.y = fma(alpha, a.y, b.y);
            ((double2 *)y)[i] = b;
        },
        { y[i] = fma(alpha, x[i], y[i]); })
}

int launch_axpy(hipStream_t s, int64_t n, double alpha, const double *x, double *y)
{
    const int g = vec_grid(n);
    if (aligned16(x) && aligned16(y))
        hipLaunchKernelGGL(k_axpy<1>, dim3(g), dim3(kBlock), 0, s, n, alpha, x, y);
    else
        hipLaunchKernelGGL(k_axpy<0>, dim3(g), dim3(kBlock), 0, s, n, alpha, x, y);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_scal(int64_t n, double alpha, double *x, int fill)
{
    CM_VEC_LOOP(n,
        {
            double2 a = ((double2 *)x)[i];
            a.x = fill ? alpha : alpha * a.x; a.y = fill ? alpha : alpha * a.y;
            ((double2 *)x)[i] = a;
        },
        { x[i] = fill ? alpha : alpha * x[i]; })
}

int launch_scal(hipStream_t s, int64_t n, double alpha, double *x)
{
    const int g = vec_grid(n);
    if (aligned16(x)) hipLaunchKernelGGL(k_scal<1>, dim3(g), dim3(kBlock), 0, s, n, alpha, x, 0);
    else hipLaunchKernelGGL(k_scal<0>, dim3(g), dim3(kBlock), 0, s, n, alpha, x, 0);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

int launch_fill(hipStream_t s, int64_t n, double value, double *x)
{
    const int g = vec_grid(n);
    if (aligned16(x)) hipLaunchKernelGGL(k_scal<1>, dim3(g), dim3(kBlock), 0, s, n, value, x, 1);
    else hipLaunchKernelGGL(k_scal<0>, dim3(g), dim3(kBlock), 0, s, n, value, x, 1);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

__global__ __launch_bounds__(kBlock) void k_rebase(int64_t n, const int *in, int shift, int *out)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        out[i] = in[i] + shift;
}

int launch_rebase(hipStream_t s, int64_t n, const int *in, int shift, int *out)
{
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_rebase, dim3((int)g), dim3(kBlock), 0, s, n, in, shift, out);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

}  // namespace cm
