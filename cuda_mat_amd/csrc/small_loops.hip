// small_loops.hip -- loop forms for systems whose iteration is bound by launch boundaries, not bytes: the vector updates
// folded into the SpMVs (three launches per iteration) and the whole loop in one launch (grid barriers).
#include <algorithm>
#include <cstring>
#include <vector>

#include "kernels.h"
#include "device.h"

namespace cm {

// ---------------------------------------------------------------- fused loop for small (L2-resident) systems
// Below ~1e5 rows an iteration is five launches of 3-5 us each: launch boundaries, not bytes.  Here the vector
// update in front of each SpMV is folded INTO the SpMV: the kernel computes the input vector on the fly at the
// columns it gathers (three / two cheap L2 gathers instead of one) and the owner of a row also stores it, so one
// iteration is three launches:
//   k_fspmv<.., FUSE_P>     rho, beta, full-step test; p' = r + beta (p - omega v) on the fly; v' = A p';  rw.v'
//   k_fspmv<.., FUSE_HALF>  alpha; s = r - alpha v' on the fly; x += alpha p'; t = A s;  (t.s, t.t), ||s||^2
//   k_full                  half-step test, omega, x += omega s, r = s - omega t, (rw.r, ||r||^2)
// p, v and r are double-buffered (a workgroup may still gather the old vector while another stores the new one).
// Every value is produced by the same expression as in k_update_p / k_half, so iterates agree with the five-launch
// loop up to the summation order of ||s||^2 (per SpMV workgroup here, per vector chunk there).
enum { FUSE_P = 1, FUSE_HALF = 2 };

template <int MODE>
struct FusedX {
    const double *r, *b1, *b2;     // FUSE_P: b1 = p, b2 = v;  FUSE_HALF: b1 = v
    double c1, c2;                 // FUSE_P: beta, -omega;    FUSE_HALF: -alpha
    bool first;                    // FUSE_P at iteration 0: p = r already (k_init)
    // the folded vector element from its already-fetched operands (rv = r, v1 = b1, v2 = b2 at the same index)
    __device__ __forceinline__ double combine(double rv, double v1, double v2) const
    {
        if (MODE == FUSE_P) {
            if (first) return v1;
            double pp = fma(c2, v2, v1);                       // pbicgstab.cu:86
            pp = c1 * pp;                                      // :87
            return rv + pp;                                    // :88
        }
        return fma(c1, v1, rv);                                // :109
    }
    __device__ __forceinline__ double operator()(int c) const
    {
        if (MODE == FUSE_P) return first ? b1[c] : combine(r[c], b1[c], b2[c]);
        return combine(r[c], b1[c], 0.0);
    }
};

// common prologue: the scalars of the folded vector kernel; false = this launch is frozen / the loop has stopped
template <int MODE>
__device__ __forceinline__ bool fused_prologue(const LoopArgs &la, const FuseArgs &f, double *lds, FusedX<MODE> &X,
                                               double &alpha_out)
{
    LoopState *st = la.st;
    if (uniform_state(st) != 0) return false;
    const int it = st->it;
    X.r = f.r;
    if (MODE == FUSE_P) {
        double sc[2];
        load_scalars<2>(f.src, sc, lds);
        if (check_full(la, sc)) return false;
        const double rho = sc[0];
        const double rhop = st->rho[(it + 1) & 1];
        const double alpha = st->alpha, omega = st->omega;
        if (leader()) st->rho[it & 1] = rho;
        X.first = it == 0;
        X.b1 = f.p_old;
        X.b2 = f.v_old;
        X.c1 = (rho / rhop) * (alpha / omega);                 // :84 (unused at it == 0)
        X.c2 = -omega;
        alpha_out = 0.0;
    } else {
        double sc[1];
        load_scalars<1>(f.src, sc, lds);
        const double alpha = st->rho[it & 1] / sc[0];          // :107
        if (leader()) st->alpha = alpha;
        X.first = false;
        X.b1 = f.v;
        X.b2 = nullptr;
        X.c1 = -alpha;
        X.c2 = 0.0;
        alpha_out = alpha;
    }
    return true;
}

// what the owner of `row` does once its sum is known
// operands of the row owner's last step that do not depend on the loop scalars (fetched early where possible)
struct FusedRowOps {
    double w, p, x, d;
};

template <int MODE>
__device__ __forceinline__ FusedRowOps fused_row_ops(const SpmvArgs &a, const FuseArgs &f, int row)
{
    FusedRowOps o;
    o.w = 0.0; o.p = 0.0; o.x = 0.0;
    o.d = a.d ? a.d[row] : 0.0;
    if (MODE == FUSE_P) {
        o.w = a.w[row];
    } else {
        o.p = f.p[row];
        o.x = f.xsol[row];
    }
    return o;
}

template <int MODE>
__device__ __forceinline__ void fused_finish_row_x(const SpmvArgs &a, const FuseArgs &f, double alpha, int row, double sum,
                                                   const FusedRowOps &o, double xr, double (&acc)[3]);

template <int MODE>
__device__ __forceinline__ void fused_finish_row(const SpmvArgs &a, const FuseArgs &f, const FusedX<MODE> &X, double alpha,
                                                 int row, double sum, const FusedRowOps &o, double (&acc)[3])
{
    fused_finish_row_x<MODE>(a, f, alpha, row, sum, o, X(row), acc);
}

// xr: the folded vector's element of this row
template <int MODE>
__device__ __forceinline__ void fused_finish_row_x(const SpmvArgs &a, const FuseArgs &f, double alpha, int row, double sum,
                                                   const FusedRowOps &o, double xr, double (&acc)[3])
{
    if (a.d) sum += o.d * xr;
    a.y[row] = sum;                                            // alpha = 1, beta = 0 inside the loop
    if (MODE == FUSE_P) {
        f.p_out[row] = xr;
        acc[0] += sum * o.w;                                   // rw . v
    } else {
        f.s_out[row] = xr;
        f.xsol[row] = fma(alpha, o.p, o.x);                    // :110
        acc[0] += sum * xr;                                    // t . s
        acc[1] += sum * sum;                                   // t . t
        acc[2] += xr * xr;                                     // ||s||^2 (:111)
    }
}

template <int MODE>
__device__ __forceinline__ void fused_store_parts(const SpmvArgs &a, const FuseArgs &f, double (&acc)[3], double *lds)
{
    block_sum<3>(acc, lds);
    if (threadIdx.x == 0) {
        a.parts[2 * blockIdx.x] = acc[0];
        a.parts[2 * blockIdx.x + 1] = acc[1];
        if (MODE == FUSE_HALF) f.parts_half[blockIdx.x] = acc[2];
    }
}

template <int L, int MODE>
__global__ __launch_bounds__(kBlock) void k_fspmv_lanes(SpmvArgs a, int rows_per_block, FuseArgs f)
{
    __shared__ double lds[12];
    FusedX<MODE> X;
    double alpha;
    if (!fused_prologue<MODE>(a.loop, f, lds, X, alpha)) return;
    constexpr int RPB = kBlock / L;
    const int lane = threadIdx.x & (L - 1);
    const int group = threadIdx.x / L;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const int row_begin = (int)(r0 < a.n ? r0 : a.n);
    const int row_end = (int)(r0 + rows_per_block < a.n ? r0 + rows_per_block : a.n);
    double acc[3] = {0.0, 0.0, 0.0};
    for (int row = row_begin + group; row < row_end; row += RPB) {
        const int s = a.rp[row], e = a.rp[row + 1];
        double sum = 0.0;
        for (int k = s + lane; k < e; k += L) sum += a.val[k] * X(a.ci[k]);
        sum = group_sum<L>(sum);
        if (lane == 0) fused_finish_row<MODE>(a, f, X, alpha, row, sum, fused_row_ops<MODE>(a, f, row), acc);
    }
    fused_store_parts<MODE>(a, f, acc, lds);
}

template <int R, int MODE>
__global__ __launch_bounds__(kBlock) void k_fspmv_stream(SpmvArgs a, int tiles_per_block, FuseArgs f)
{
    __shared__ double prod[kStreamNnz];
    __shared__ int srp[R + 1];
    __shared__ double lds[12];
    constexpr int E = kStreamNnz / kBlock;
    const int tid = threadIdx.x;
    // The first tile's row pointers and entries do not depend on the loop scalars: fetch them BEFORE the prologue
    // (partial sums -> rho, beta / alpha), so the two dependent round trips overlap instead of adding up.
    const long long first_row = (long long)blockIdx.x * tiles_per_block * R;
    double v0[E];
    int c0[E];
    int nr0 = 0, base0 = 0, cnt0 = 0;
    if (first_row < a.n) {
        const int r0 = (int)first_row;
        nr0 = a.n - r0 < R ? a.n - r0 : R;
        for (int i = tid; i <= nr0; i += kBlock) srp[i] = a.rp[r0 + i];
        __syncthreads();
        base0 = srp[0];
        cnt0 = srp[nr0] - base0;
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int k = tid + j * kBlock;
            if (k < cnt0) {
                v0[j] = a.val[base0 + k];
                c0[j] = a.ci[base0 + k];
            }
        }
    }
    FusedRowOps ops0;
    ops0.w = ops0.p = ops0.x = ops0.d = 0.0;
    if (tid < nr0) ops0 = fused_row_ops<MODE>(a, f, (int)first_row + tid);
    FusedX<MODE> X;
    double alpha;
    if (!fused_prologue<MODE>(a.loop, f, lds, X, alpha)) return;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int t = 0; t < tiles_per_block; t++) {
        const long long r0l = first_row + (long long)t * R;
        if (r0l >= a.n) break;
        const int r0 = (int)r0l;
        int nr, base, cnt;
        if (t == 0) {
            nr = nr0; base = base0; cnt = cnt0;
#pragma unroll
            for (int j = 0; j < E; j++) {
                const int k = tid + j * kBlock;
                if (k < cnt) prod[k] = v0[j] * X(c0[j]);
            }
        } else {
            nr = a.n - r0 < R ? a.n - r0 : R;
            for (int i = tid; i <= nr; i += kBlock) srp[i] = a.rp[r0 + i];
            __syncthreads();
            base = srp[0];
            cnt = srp[nr] - base;
            for (int k = tid; k < cnt; k += kBlock) prod[k] = a.val[base + k] * X(a.ci[base + k]);
        }
        __syncthreads();
        if (tid < nr) {
            const int s = srp[tid] - base, e = srp[tid + 1] - base;
            double sum = 0.0;
            for (int j = s; j < e; j++) sum += prod[j];
            fused_finish_row<MODE>(a, f, X, alpha, r0 + tid, sum, t == 0 ? ops0 : fused_row_ops<MODE>(a, f, r0 + tid), acc);
        }
        __syncthreads();
    }
    fused_store_parts<MODE>(a, f, acc, lds);
}

bool fused_spmv_supported(const SpmvPlan &plan) { return plan.tiles == 0; }

int launch_fused_spmv(hipStream_t s, const SpmvPlan &plan, const SpmvArgs &a, const FuseArgs &f)
{
    dim3 g(plan.grid), b(kBlock);
#define CM_FS(KERNEL, PARAM)                                                                          \
    do {                                                                                              \
        if (f.mode == FUSE_P) hipLaunchKernelGGL((KERNEL<PARAM, FUSE_P>), g, b, 0, s, a, plan.rows_per_block, f);      \
        else hipLaunchKernelGGL((KERNEL<PARAM, FUSE_HALF>), g, b, 0, s, a, plan.rows_per_block, f);                    \
    } while (0)
    if (plan.stream_rows) {
        switch (plan.stream_rows) {
        case 64:  CM_FS(k_fspmv_stream, 64); break;
        case 128: CM_FS(k_fspmv_stream, 128); break;
        default:  CM_FS(k_fspmv_stream, 256); break;
        }
    } else {
        switch (plan.lanes) {
        case 2:  CM_FS(k_fspmv_lanes, 2); break;
        case 4:  CM_FS(k_fspmv_lanes, 4); break;
        case 8:  CM_FS(k_fspmv_lanes, 8); break;
        case 16: CM_FS(k_fspmv_lanes, 16); break;
        case 32: CM_FS(k_fspmv_lanes, 32); break;
        default: CM_FS(k_fspmv_lanes, 64); break;
        }
    }
#undef CM_FS
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

// ---------------------------------------------------------------- resident loop (one launch, many iterations)
// For systems of at most 128 stream tiles (<= 128 x 256 rows of <= 8 entries) even the three-launch
// loop above is bound by launch boundaries: each of its kernels spends most of its 5-7 us being dispatched and
// drained.  Here the SAME three phases run inside ONE launch: every workgroup owns one tile of R rows for the whole
// solve -- its matrix entries, row ends and rw stay in registers / LDS -- and the phases are separated by a grid
// barrier (release fence, one agent-scope atomic arrival, polling load, acquire fence) instead of a launch boundary.
// Every scalar, stopping test and vector value is produced by the expressions of the fused loop (fused_prologue,
// fused_finish_row, check_half, check_full); only the partial sums of the last phase are grouped per tile instead
// of per vector chunk.  All workgroups take every exit decision from the same partial sums, so they leave the loop
// in the same phase.  The grid is at most one workgroup per two compute units, all resident at once; should the GPU be
// shared with something that keeps some of them from starting, the barrier's bounded wait raises a flag, every
// workgroup leaves, and the host redoes the solve with the three-launch loop (cudamat_stats.loop_fallbacks).
__device__ __forceinline__ bool grid_barrier(unsigned *bar, unsigned &epoch, int *s_ok, unsigned spin_limit)
{
    // Release side: everything other workgroups read is stored with agent-scope (write-through, sc1) stores, and the
    // workgroup-scope release inside __syncthreads() has every wave wait for its stores -- so no L2 write-back here.
    // Acquire side: buffer_inv sc1, after which plain (cached) loads of the others' data are served from memory.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch += gridDim.x;
        __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool good = true;
        for (unsigned spins = 0; __hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch; spins++) {
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 255u) == 255u && __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                good = false;                                       // another workgroup gave up
                break;
            }
            if (spins >= spin_limit) {                              // seconds: this launch is not making progress
                __hip_atomic_store(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = false;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *s_ok = good ? 1 : 0;
    }
    __syncthreads();
    return *s_ok != 0;
}

// a store other workgroups (or the host) will read: agent scope = written through, never a dirty line in this XCD's L2
// (the loop's acquire side invalidates that L2)
template <typename T>
__device__ __forceinline__ void st_shared(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int R>
__global__ __launch_bounds__(kBlock) void k_resident_loop(SpmvArgs a, ResidentArgs q)
{
    __shared__ double prod[kStreamNnz];
    __shared__ int srp[R + 1];
    __shared__ double lds[12];
    __shared__ int s_ok;
    constexpr int E = kStreamNnz / kBlock;
    const int tid = threadIdx.x;
    const int r0 = (int)blockIdx.x * R;                 // one tile per workgroup (launch_resident_loop checks)
    const int nr = a.n - r0 < R ? a.n - r0 : R;
    for (int i = tid; i <= nr; i += kBlock) srp[i] = a.rp[r0 + i];
    __syncthreads();
    const int base = srp[0], cnt = srp[nr] - base;
    double v0[E];
    int c0[E];
#pragma unroll
    for (int j = 0; j < E; j++) {
        const int k = tid + j * kBlock;
        v0[j] = k < cnt ? a.val[base + k] : 0.0;
        c0[j] = k < cnt ? a.ci[base + k] : 0;
    }
    const bool own = tid < nr;
    const int row = r0 + tid;
    const int lo = own ? srp[tid] - base : 0, hi = own ? srp[tid + 1] - base : 0;
    const double w_own = own ? q.rw[row] : 0.0;
    const double d_own = own && a.d ? a.d[row] : 0.0;
    double *p_a = q.p_a, *p_b = q.p_b, *v_a = q.v_a, *v_b = q.v_b, *r = q.r, *sv = q.s;
    const LoopArgs la = a.loop;
    LoopState *st = la.st;
    const bool lead = leader();
    const int G = (int)gridDim.x;
    // The loop scalars live in registers: every workgroup derives them from the same partial sums in the same
    // order, so all hold the same values and take the same decisions; the leader mirrors them into LoopState (for
    // the host and for the launch that follows this one).
    int it = st->it;
    double rho_s[2] = {st->rho[0], st->rho[1]};
    double alpha = st->alpha, omega = st->omega;
    const double tolabs = st->tolabs;
    if (st->state != 0) return;                        // (launch-uniform: nothing in this launch has written it yet)
    double x_cur = own ? q.x[row] : 0.0;               // this row's x: a register for the whole launch
    unsigned epoch = 0;
    for (int k = 0; k < q.iters; k++) {
        // ---- rho, beta, full-step test; p' = r + beta (p - omega v) on the fly; v' = A p'; rw.v'     :80-89, :104-106
        double p_new = 0.0;
        {
            // the gathers do not depend on this phase's scalars: issue them first, combine once the scalars are known
            double gr[E], gp[E], gv[E];
#pragma unroll
            for (int j = 0; j < E; j++) {
                const int e = tid + j * kBlock;
                gr[j] = 0.0; gp[j] = 0.0; gv[j] = 0.0;
                if (e < cnt) {
                    gp[j] = p_a[c0[j]];
                    gr[j] = r[c0[j]];
                    gv[j] = v_a[c0[j]];      // (unused at iteration 0, where p = r already)
                }
            }
            const double r_own = own ? r[row] : 0.0, p_own = own ? p_a[row] : 0.0, v_own = own ? v_a[row] : 0.0;
            double sc[2];
            load_scalars<2>(ScalarSrc{q.parts_full, k == 0 ? q.first_count : G, 2}, sc, lds);
            if (it != 0) {                                         // full-step test of iteration it-1 (check_full)
                const double nrm = sqrt(sc[1]);
                if (lead) {
                    st_shared(&st->nrm, nrm);
                    if (la.hist) {
                        const int slot = (la.loop != CUDAMAT_LOOP_PBICGSTAB2) ? 2 * (it - 1) + 1 : it - 1;
                        if (slot < la.hist_cap) st_shared(&la.hist[slot], nrm);
                    }
                }
                if (!la.no_exit) {
                    int stop = 0;
                    if (nrm < tolabs) stop = 2;
                    else if (la.loop == CUDAMAT_LOOP_PBICGSTAB2 && (fabs(omega) < 1e-5 || isnan(omega))) stop = 3;
                    else if (isnan(nrm)) stop = 3;
                    if (stop) {
                        if (lead) st_shared(&st->state, stop);
                        break;
                    }
                }
            }
            const double rho = sc[0], rhop = rho_s[(it + 1) & 1];
            rho_s[it & 1] = rho;
            if (lead) st_shared(&st->rho[it & 1], rho);
            FusedX<FUSE_P> X;
            X.first = it == 0;
            X.c1 = (rho / rhop) * (alpha / omega);                 // :84 (unused at it == 0)
            X.c2 = -omega;
#pragma unroll
            for (int j = 0; j < E; j++) {
                const int e = tid + j * kBlock;
                if (e < cnt) prod[e] = v0[j] * X.combine(gr[j], gp[j], gv[j]);
            }
            __syncthreads();
            double acc[1] = {0.0};
            if (own) {
                double sum = 0.0;
                for (int j = lo; j < hi; j++) sum += prod[j];
                p_new = X.combine(r_own, p_own, v_own);
                if (a.d) sum += d_own * p_new;
                st_shared(&v_b[row], sum);
                st_shared(&p_b[row], p_new);
                acc[0] = sum * w_own;                              // rw . v
            }
            block_sum<1>(acc, lds);
            if (tid == 0) st_shared(&q.parts_rv[2 * blockIdx.x], acc[0]);
        }
        if (!grid_barrier(q.bar, epoch, &s_ok, q.spin_limit)) break;
        // ---- alpha; s = r - alpha v' on the fly; x += alpha p'; t = A s; (t.s, t.t), ||s||^2          :107-111, :132-136
        double s_new = 0.0, t_new = 0.0, x_half = 0.0;
        {
            double gr[E], gv[E];
#pragma unroll
            for (int j = 0; j < E; j++) {
                const int e = tid + j * kBlock;
                gr[j] = 0.0; gv[j] = 0.0;
                if (e < cnt) {
                    gr[j] = r[c0[j]];
                    gv[j] = v_b[c0[j]];
                }
            }
            const double r_own = own ? r[row] : 0.0, v_own = own ? v_b[row] : 0.0;
            double sc[1];
            load_scalars<1>(ScalarSrc{q.parts_rv, G, 2}, sc, lds);
            alpha = rho_s[it & 1] / sc[0];                         // :107
            if (lead) st_shared(&st->alpha, alpha);
            FusedX<FUSE_HALF> X;
            X.first = false;
            X.c1 = -alpha;
            X.c2 = 0.0;
#pragma unroll
            for (int j = 0; j < E; j++) {
                const int e = tid + j * kBlock;
                if (e < cnt) prod[e] = v0[j] * X.combine(gr[j], gv[j], 0.0);
            }
            __syncthreads();
            double acc[3] = {0.0, 0.0, 0.0};
            if (own) {
                double sum = 0.0;
                for (int j = lo; j < hi; j++) sum += prod[j];
                s_new = X.combine(r_own, v_own, 0.0);
                if (a.d) sum += d_own * s_new;
                t_new = sum;
                st_shared(&sv[row], s_new);
                x_half = fma(alpha, p_new, x_cur);                 // :110
                acc[0] = sum * s_new;                              // t . s
                acc[1] = sum * sum;                                // t . t
                acc[2] = s_new * s_new;                            // ||s||^2 (:111)
            }
            block_sum<3>(acc, lds);
            if (tid == 0) {
                st_shared(&q.parts_tt[2 * blockIdx.x], acc[0]);
                st_shared(&q.parts_tt[2 * blockIdx.x + 1], acc[1]);
                st_shared(&q.parts_half[blockIdx.x], acc[2]);
            }
        }
        if (!grid_barrier(q.bar, epoch, &s_ok, q.spin_limit)) break;
        // ---- half-step test, omega, x += omega s, r = s - omega t, (rw.r, ||r||^2), i++                :116, :137-151
        {
            double sc[3] = {0.0, 0.0, 0.0};                        // ||s||^2, t.s, t.t
            for (int j = tid; j < G; j += kBlock) {
                sc[0] += q.parts_half[j];
                sc[1] += q.parts_tt[2 * j];
                sc[2] += q.parts_tt[2 * j + 1];
            }
            block_sum<3>(sc, lds);
            if (la.loop == CUDAMAT_LOOP_PBICGSTAB) {               // half-step test (check_half)
                const double nrm = sqrt(sc[0]);
                if (lead) {
                    st_shared(&st->nrm, nrm);
                    if (la.hist && 2 * it < la.hist_cap) st_shared(&la.hist[2 * it], nrm);
                }
                if (!la.no_exit && (nrm < tolabs || isnan(nrm))) {
                    if (lead) st_shared(&st->state, nrm < tolabs ? 1 : 3);
                    x_cur = x_half;                                // x += alpha p' belongs to the half step
                    break;
                }
            }
            omega = sc[1] / sc[2];                                 // :137
            double acc[2] = {0.0, 0.0};
            if (own) {
                x_cur = fma(omega, s_new, x_half);                 // :139
                const double rr = fma(-omega, t_new, s_new);       // :140
                st_shared(&sv[row], rr);   // the new residual goes over s (as k_full does)
                acc[0] = w_own * rr;                               // :81 of i+1
                acc[1] = rr * rr;                                  // :142
            }
            block_sum<2>(acc, lds);
            if (tid == 0) {
                st_shared(&q.parts_full[2 * blockIdx.x], acc[0]);
                st_shared(&q.parts_full[2 * blockIdx.x + 1], acc[1]);
            }
            it++;
            if (lead) {
                st_shared(&st->omega, omega);
                st_shared(&st->it, it);                            // :148 / :151
            }
        }
        if (!grid_barrier(q.bar, epoch, &s_ok, q.spin_limit)) break;
        double *tp = p_a; p_a = p_b; p_b = tp;
        tp = v_a; v_a = v_b; v_b = tp;
        tp = r; r = sv; sv = tp;
    }
    if (own) q.x[row] = x_cur;
}

bool resident_loop_supported(const SpmvPlan &plan, int n)
{
    // one stream tile per workgroup; up to 128 workgroups (half the compute units of an MI355X): a grid barrier costs
    // 1.1 us with 8 workgroups, 1.4 us with 40, 2.5 us with 128 and 4.3 us with 256 (scripts/probe_barrier.hip: the
    // arrivals serialise on one counter), and beyond ~150 tiles three barriers cost more than three launch boundaries
    // (scripts/resident_sizes.sh: 1.56x at 10 tiles, 1.38x at 40, 1.29x at 78, 1.09x at 127, 1.0x at 157, 0.76x at 255)
    return plan.tiles == 0 && plan.stream_rows > 0 && plan.rows_per_block == 1 && plan.grid >= 1 && plan.grid <= 128 &&
           (long long)plan.grid * plan.stream_rows >= n;
}

int launch_resident_loop(hipStream_t s, const SpmvPlan &plan, const SpmvArgs &a, const ResidentArgs &q)
{
    dim3 g(plan.grid), b(kBlock);
    switch (plan.stream_rows) {
    case 64:  hipLaunchKernelGGL(k_resident_loop<64>, g, b, 0, s, a, q); break;
    case 128: hipLaunchKernelGGL(k_resident_loop<128>, g, b, 0, s, a, q); break;
    default:  hipLaunchKernelGGL(k_resident_loop<256>, g, b, 0, s, a, q); break;
    }
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

}  // namespace cm
