// pipelined.hip -- the vector kernels of the pipelined BiCGStab loop (Solve::iterate_pipelined in loops.hip).
#include <algorithm>
#include <cstring>
#include <vector>

#include "kernels.h"
#include "device.h"

namespace cm {

// ---------------------------------------------------------------- pipelined BiCGStab
// Cools & Vanroose 2017, Alg. 4: the recurrences of the loop above re-arranged so that each of the two reduction
// phases of an iteration can run WHILE an SpMV runs (s = A p, z = A s, v = A z, w = A r, t = A w are carried by
// recurrences; only v = A z and t = A w are multiplied out).  One iteration = k_pipe_a, SpMV, k_pipe_b, SpMV:
//   k_pipe_a   full-step test of the previous iteration; beta, alpha from the five dots of k_pipe_b;
//              p = r + beta (p - omega s), s = w + beta (s - omega z), z = t + beta (z - omega v),
//              q = r - alpha s, y = w - alpha z, xh = x + alpha p;              dots (q.y, y.y, q.q)
//   SpMV       v = A z                                   [the three dots are reduced / all-reduced meanwhile]
//   k_pipe_b   half-step test on ||q||; omega = q.y / y.y; x = xh + omega q, r = q - omega y,
//              w = y - omega (t - alpha v);                    dots (rw.r, rw.w, rw.s, rw.z, r.r);  i++
//   SpMV       t = A w                                    [the five dots are reduced / all-reduced meanwhile]
// A kernel that decides an exit has no other effect under that decision (the half-step iterate x + alpha p is
// kept in xh by k_pipe_a, the host returns it when the loop leaves through the half step), so workgroups that
// start after the leader has published the exit and return at once change nothing.  alpha and rho live in two
// slots indexed by the parity of the host's iteration index la.k (the writer of iteration k never overwrites
// what a late workgroup of the same launch still reads).  Same stopping rules as LOOP_PBICGSTAB (:116, :147).
constexpr int kPipeA = 3, kPipeB = 5;

// PC = 1: the preconditioned form (M^-1 where pbicgstab.cu:92-98,121-127 apply it).  Hatted vectors are M^-1 times the
// plain ones: rh, wh, zh come in, p carries ph = M^-1 p, sh = M^-1 s is carried by its own recurrence, qh = rh - alpha sh
// goes out for k_pipe_b; x advances along the hatted directions.  With PC = 0 hatted and plain vectors coincide.
template <int VEC, int PC>
__global__ __launch_bounds__(kBlock) void k_pipe_a(LoopArgs la, ScalarSrc B, int64_t n, const double *r,
                                                   const double *w, const double *t, const double *v, double *p,
                                                   double *s, double *z, double *q, double *y, const double *x,
                                                   double *xh, double *parts, PipeHatA hat)
{
#pragma clang fp contract(off)      // one rounding per operation, like the oracle's restatement
    __shared__ double lds[4 * kPipeB];
    LoopState *st = la.st;
    if (uniform_state(st) != 0) return;
    double sc[kPipeB];
    load_scalars<kPipeB>(B, sc, lds);
    const int k = la.k;
    if (k > 0) {                                            // full-step test of iteration k-1 (:142-151)
        const double nrm = sqrt(sc[4]);
        if (leader()) {
            st->nrm = nrm;
            const int slot = 2 * (st->it - 1) + 1;
            if (la.hist && slot >= 0 && slot < la.hist_cap) la.hist[slot] = nrm;
        }
        if (!la.no_exit && nrm < st->tolabs) {
            if (leader()) st->state = 2;
            return;
        }
        if (!la.no_exit && isnan(nrm)) {                    // breakdown (0/0 somewhere): stop instead of spinning on NaNs
            if (leader()) st->state = 3;
            return;
        }
    }
    const double rho = sc[0];
    double alpha, beta = 0.0, omega = 0.0;
    if (k == 0) {
        alpha = rho / sc[1];
    } else {
        const double rhop = st->rho[(k + 1) & 1], alphap = st->alpha2[(k + 1) & 1];
        omega = st->omega;
        beta = (alphap / omega) * (rho / rhop);
        alpha = rho / (sc[1] + beta * sc[2] - beta * omega * sc[3]);
    }
    if (leader()) {
        st->rho[k & 1] = rho;
        st->alpha2[k & 1] = alpha;
        st->alpha = alpha;
    }
    double acc[kPipeA] = {0.0, 0.0, 0.0};
    const bool first = k == 0;
    // (rh, wh, zh, sh, qh are only touched when PC = 1: without a preconditioner they ARE r, w, z, s, q)
    auto elem = [&](double rr, double ww, double tt, double vv, double &pp, double &ss, double &zz, double xx,
                    double &qq, double &yy, double &xo, double rrh, double wwh, double zzh, double &ssh, double &qqh) {
        if (first) { pp = PC ? rrh : rr; ss = ww; zz = tt; if (PC) ssh = wwh; }
        else {
            pp = (PC ? rrh : rr) + beta * (pp - omega * (PC ? ssh : ss));
            if (PC) ssh = wwh + beta * (ssh - omega * zzh);
            ss = ww + beta * (ss - omega * zz);
            zz = tt + beta * (zz - omega * vv);
        }
        qq = rr - alpha * ss;
        if (PC) qqh = rrh - alpha * ssh;
        yy = ww - alpha * zz;
        xo = xx + alpha * pp;
        acc[0] += qq * yy;
        acc[1] += yy * yy;
        acc[2] += qq * qq;
    };
    CM_VEC_LOOP(n,
        {
            const double2 rr = ((const double2 *)r)[i];
            const double2 ww = ((const double2 *)w)[i];
            const double2 tt = ((const double2 *)t)[i];
            double2 vv = {0.0 COMMA 0.0};
            if (!first) vv = ((const double2 *)v)[i];
            const double2 xx = ((const double2 *)x)[i];
            double2 pp = ((double2 *)p)[i];
            double2 ss = ((double2 *)s)[i];
            double2 zz = ((double2 *)z)[i];
            double2 rrh = {0.0 COMMA 0.0}; double2 wwh = {0.0 COMMA 0.0}; double2 zzh = {0.0 COMMA 0.0};
            double2 ssh = {0.0 COMMA 0.0}; double2 qqh = {0.0 COMMA 0.0};
            if (PC) {
                rrh = ((const double2 *)hat.rh)[i]; wwh = ((const double2 *)hat.wh)[i];
                if (!first) { zzh = ((const double2 *)hat.zh)[i]; ssh = ((double2 *)hat.sh)[i]; }
            }
            double2 qq; double2 yy; double2 xo;
            elem(rr.x, ww.x, tt.x, vv.x, pp.x, ss.x, zz.x, xx.x, qq.x, yy.x, xo.x, rrh.x, wwh.x, zzh.x, ssh.x, qqh.x);
            elem(rr.y, ww.y, tt.y, vv.y, pp.y, ss.y, zz.y, xx.y, qq.y, yy.y, xo.y, rrh.y, wwh.y, zzh.y, ssh.y, qqh.y);
            ((double2 *)p)[i] = pp; ((double2 *)s)[i] = ss; ((double2 *)z)[i] = zz;
            ((double2 *)q)[i] = qq; ((double2 *)y)[i] = yy; ((double2 *)xh)[i] = xo;
            if (PC) { ((double2 *)hat.sh)[i] = ssh; ((double2 *)hat.qh)[i] = qqh; }
        },
        {
            double pp = p[i]; double ss = s[i]; double zz = z[i]; double qq; double yy; double xo;
            double ssh = 0.0; double qqh = 0.0;
            if (PC && !first) ssh = hat.sh[i];
            elem(r[i], w[i], t[i], first ? 0.0 : v[i], pp, ss, zz, x[i], qq, yy, xo, PC ? hat.rh[i] : 0.0, PC ? hat.wh[i] : 0.0,
                 (PC && !first) ? hat.zh[i] : 0.0, ssh, qqh);
            p[i] = pp; s[i] = ss; z[i] = zz; q[i] = qq; y[i] = yy; xh[i] = xo;
            if (PC) { hat.sh[i] = ssh; hat.qh[i] = qqh; }
        })
    block_sum<kPipeA>(acc, lds);
    if (threadIdx.x == 0)
        for (int j = 0; j < kPipeA; j++) parts[kPipeA * blockIdx.x + j] = acc[j];
}

// PC = 1: x advances along qh = M^-1 q, and rh' = qh - omega (wh - alpha zh) = M^-1 r' is carried along
template <int VEC, int PC>
__global__ __launch_bounds__(kBlock) void k_pipe_b(LoopArgs la, ScalarSrc A, int64_t n, const double *q,
                                                   const double *y, const double *t, const double *v,
                                                   const double *rw, const double *s, const double *z,
                                                   const double *xh, double *x, double *r, double *w, double *parts,
                                                   PipeHatB hat)
{
#pragma clang fp contract(off)
    __shared__ double lds[4 * kPipeB];
    LoopState *st = la.st;
    const int frozen = uniform_state(st);
    if (frozen != 0) {                // frozen: still tell the host this iteration's launches have drained
        publish_progress(la, frozen);
        return;
    }
    double sc[kPipeA];
    load_scalars<kPipeA>(A, sc, lds);
    const double nrm = sqrt(sc[2]);                         // ||q||: the half-step residual (:111)
    if (leader()) {
        st->nrm = nrm;
        const int slot = 2 * st->it;
        if (la.hist && slot < la.hist_cap) la.hist[slot] = nrm;
    }
    if (!la.no_exit && nrm < st->tolabs) {                  // :116 -- the iterate of this exit is xh
        if (leader()) st->state = 1;
        publish_progress(la, 1);
        return;
    }
    if (!la.no_exit && isnan(nrm)) {
        if (leader()) st->state = 3;
        publish_progress(la, 3);
        return;
    }
    const double omega = sc[0] / sc[1];
    const double alpha = st->alpha2[la.k & 1];
    double acc[kPipeB] = {0.0, 0.0, 0.0, 0.0, 0.0};
    auto elem = [&](double qq, double yy, double tt, double vv, double ww_, double ss, double zz, double xo,
                    double &xx, double &rr, double &wn, double qqh, double wwh, double zzh, double &rrh) {
        xx = xo + omega * (PC ? qqh : qq);
        rr = qq - omega * yy;
        if (PC) rrh = qqh - omega * (wwh - alpha * zzh);
        wn = yy - omega * (tt - alpha * vv);
        acc[0] += ww_ * rr;
        acc[1] += ww_ * wn;
        acc[2] += ww_ * ss;
        acc[3] += ww_ * zz;
        acc[4] += rr * rr;
    };
    CM_VEC_LOOP(n,
        {
            const double2 qq = ((const double2 *)q)[i];
            const double2 yy = ((const double2 *)y)[i];
            const double2 tt = ((const double2 *)t)[i];
            const double2 vv = ((const double2 *)v)[i];
            const double2 ww_ = ((const double2 *)rw)[i];
            const double2 ss = ((const double2 *)s)[i];
            const double2 zz = ((const double2 *)z)[i];
            const double2 xo = ((const double2 *)xh)[i];
            double2 qqh = {0.0 COMMA 0.0}; double2 wwh = {0.0 COMMA 0.0}; double2 zzh = {0.0 COMMA 0.0}; double2 rrh = {0.0 COMMA 0.0};
            if (PC) { qqh = ((const double2 *)hat.qh)[i]; wwh = ((const double2 *)hat.wh)[i]; zzh = ((const double2 *)hat.zh)[i]; }
            double2 xx; double2 rr; double2 wn;
            elem(qq.x, yy.x, tt.x, vv.x, ww_.x, ss.x, zz.x, xo.x, xx.x, rr.x, wn.x, qqh.x, wwh.x, zzh.x, rrh.x);
            elem(qq.y, yy.y, tt.y, vv.y, ww_.y, ss.y, zz.y, xo.y, xx.y, rr.y, wn.y, qqh.y, wwh.y, zzh.y, rrh.y);
            ((double2 *)x)[i] = xx; ((double2 *)r)[i] = rr; ((double2 *)w)[i] = wn;
            if (PC) ((double2 *)hat.rh)[i] = rrh;
        },
        {
            double xx; double rr; double wn; double rrh = 0.0;
            elem(q[i], y[i], t[i], v[i], rw[i], s[i], z[i], xh[i], xx, rr, wn, PC ? hat.qh[i] : 0.0, PC ? hat.wh[i] : 0.0,
                 PC ? hat.zh[i] : 0.0, rrh);
            x[i] = xx; r[i] = rr; w[i] = wn;
            if (PC) hat.rh[i] = rrh;
        })
    block_sum<kPipeB>(acc, lds);
    if (threadIdx.x == 0)
        for (int j = 0; j < kPipeB; j++) parts[kPipeB * blockIdx.x + j] = acc[j];
    if (leader()) {
        st->omega = omega;
        st->it = st->it + 1;
    }
    publish_progress(la, 0);
}

// seed of iteration 0: out = [rw.r0, rw.w0, 0, 0, r0.r0] from the partials of k_init (stride 2) and of the
// SpMV w0 = A r0 with dot = 1 (stride 2, slot 0 = sum w0 * rw)
__global__ __launch_bounds__(kBlock) void k_pipe_seed(ScalarSrc init, ScalarSrc rww, double *out)
{
    __shared__ double lds[8];
    double a[2], b[1];
    load_scalars<2>(init, a, lds);
    load_scalars<1>(rww, b, lds);
    if (threadIdx.x == 0) {
        out[0] = a[0]; out[1] = b[0]; out[2] = 0.0; out[3] = 0.0; out[4] = a[1];
    }
}

int launch_pipe_seed(hipStream_t s, ScalarSrc init, ScalarSrc rww, double *out)
{
    hipLaunchKernelGGL(k_pipe_seed, dim3(1), dim3(kBlock), 0, s, init, rww, out);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

int launch_pipe_a(hipStream_t s, LoopArgs la, ScalarSrc B, int64_t n, const double *r, const double *w, const double *t,
                  const double *v, double *p, double *sv, double *z, double *q, double *y, const double *x, double *xh,
                  double *parts, int *nparts, PipeHatA hat)
{
    const int g = vec_grid(n);
    *nparts = g;
    const bool pc = hat.rh != nullptr;
    const bool al = aligned16(r) && aligned16(w) && aligned16(t) && aligned16(v) && aligned16(p) && aligned16(sv) && aligned16(z) &&
                    aligned16(q) && aligned16(y) && aligned16(x) && aligned16(xh) &&
                    (!pc || (aligned16(hat.rh) && aligned16(hat.wh) && aligned16(hat.zh) && aligned16(hat.sh) && aligned16(hat.qh)));
#define CM_PIPE_A(V, P) hipLaunchKernelGGL((k_pipe_a<V, P>), dim3(g), dim3(kBlock), 0, s, la, B, n, r, w, t, v, p, sv, z, q, y, x, xh, parts, hat)
    if (al && pc) CM_PIPE_A(1, 1);
    else if (al) CM_PIPE_A(1, 0);
    else if (pc) CM_PIPE_A(0, 1);
    else CM_PIPE_A(0, 0);
#undef CM_PIPE_A
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

int launch_pipe_b(hipStream_t s, LoopArgs la, ScalarSrc A, int64_t n, const double *q, const double *y, const double *t,
                  const double *v, const double *rw, const double *sv, const double *z, const double *xh, double *x, double *r,
                  double *w, double *parts, int *nparts, PipeHatB hat)
{
    const int g = vec_grid(n);
    *nparts = g;
    const bool pc = hat.rh != nullptr;
    const bool al = aligned16(q) && aligned16(y) && aligned16(t) && aligned16(v) && aligned16(rw) && aligned16(sv) && aligned16(z) &&
                    aligned16(xh) && aligned16(x) && aligned16(r) && aligned16(w) &&
                    (!pc || (aligned16(hat.qh) && aligned16(hat.wh) && aligned16(hat.zh) && aligned16(hat.rh)));
#define CM_PIPE_B(V, P) hipLaunchKernelGGL((k_pipe_b<V, P>), dim3(g), dim3(kBlock), 0, s, la, A, n, q, y, t, v, rw, sv, z, xh, x, r, w, parts, hat)
    if (al && pc) CM_PIPE_B(1, 1);
    else if (al) CM_PIPE_B(1, 0);
    else if (pc) CM_PIPE_B(0, 1);
    else CM_PIPE_B(0, 0);
#undef CM_PIPE_B
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

// ---- residual replacement of the pipelined loop (solver.hip): r = f - ax, and the five dots k_pipe_b would have left
// (rw.r, rw.w, rw.s, rw.z, r.r) recomputed from the replaced vectors (same layout: stride kPipeB per workgroup)
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_residual(const LoopState *st, int64_t n, const double *f, const double *ax, double *r)
{
    if (st && st->state != 0) return;                  // frozen loop: ax is stale, r must stay the iterate's residual
    CM_VEC_LOOP(n,
        {
            const double2 ff = ((const double2 *)f)[i];
            const double2 aa = ((const double2 *)ax)[i];
            double2 rr; rr.x = ff.x - aa.x; rr.y = ff.y - aa.y;
            ((double2 *)r)[i] = rr;
        },
        { r[i] = f[i] - ax[i]; })
}

int launch_residual(hipStream_t s, const LoopArgs &la, int64_t n, const double *f, const double *ax, double *r)
{
    const int g = vec_grid(n);
    if (aligned16(f) && aligned16(ax) && aligned16(r)) hipLaunchKernelGGL(k_residual<1>, dim3(g), dim3(kBlock), 0, s, la.st, n, f, ax, r);
    else hipLaunchKernelGGL(k_residual<0>, dim3(g), dim3(kBlock), 0, s, la.st, n, f, ax, r);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_pipe_dots(const LoopState *st, int64_t n, const double *rw, const double *r, const double *w,
                                                      const double *s, const double *z, double *parts)
{
#pragma clang fp contract(off)
    __shared__ double lds[4 * kPipeB];
    if (st && st->state != 0) return;                  // frozen loop: the partials k_pipe_b left stay what they are
    double acc[kPipeB] = {0.0, 0.0, 0.0, 0.0, 0.0};
    auto elem = [&](double ww_, double rr, double wn, double ss, double zz) {
        acc[0] += ww_ * rr;
        acc[1] += ww_ * wn;
        acc[2] += ww_ * ss;
        acc[3] += ww_ * zz;
        acc[4] += rr * rr;
    };
    CM_VEC_LOOP(n,
        {
            const double2 a = ((const double2 *)rw)[i];
            const double2 b = ((const double2 *)r)[i];
            const double2 c = ((const double2 *)w)[i];
            const double2 d = ((const double2 *)s)[i];
            const double2 e = ((const double2 *)z)[i];
            elem(a.x, b.x, c.x, d.x, e.x);
            elem(a.y, b.y, c.y, d.y, e.y);
        },
        { elem(rw[i], r[i], w[i], s[i], z[i]); })
    block_sum<kPipeB>(acc, lds);
    if (threadIdx.x == 0)
        for (int j = 0; j < kPipeB; j++) parts[kPipeB * blockIdx.x + j] = acc[j];
}

int launch_pipe_dots(hipStream_t s, const LoopArgs &la, int64_t n, const double *rw, const double *r, const double *w, const double *sv,
                     const double *z, double *parts, int *nparts)
{
    const int g = vec_grid(n);
    *nparts = g;
    if (aligned16(rw) && aligned16(r) && aligned16(w) && aligned16(sv) && aligned16(z))
        hipLaunchKernelGGL(k_pipe_dots<1>, dim3(g), dim3(kBlock), 0, s, la.st, n, rw, r, w, sv, z, parts);
    else
        hipLaunchKernelGGL(k_pipe_dots<0>, dim3(g), dim3(kBlock), 0, s, la.st, n, rw, r, w, sv, z, parts);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

}  // namespace cm
