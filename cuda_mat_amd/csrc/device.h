// device.h -- device-side helpers shared by the kernels of libcudamat_hip.so: fixed-order workgroup reductions, the
// scalars that live in HBM (per-workgroup partial sums summed by every consumer), the loop state as a workgroup reads
// it, and the half-step stopping test that SpMV kernels evaluate in their prologue.  256-thread workgroups (kBlock).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace cm {

// ------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int L>
__device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// every thread of the 256-thread workgroup receives the K sums (fixed order)
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double *lds)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = wave_sum(v[k]);
    __syncthreads();  // lds may still be read from a previous use
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) lds[wave * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++)
        v[k] = ((lds[0 * K + k] + lds[1 * K + k]) + lds[2 * K + k]) + lds[3 * K + k];
}

template <int K>
__device__ __forceinline__ void load_scalars(const ScalarSrc &s, double (&out)[K], double *lds)
{
    if (s.count == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) out[k] = s.ptr[k];
        return;
    }
#pragma unroll
    for (int k = 0; k < K; k++) out[k] = 0.0;
    for (int j = threadIdx.x; j < s.count; j += kBlock) {
#pragma unroll
        for (int k = 0; k < K; k++) out[k] += s.ptr[(size_t)j * s.stride + k];
    }
    block_sum<K>(out, lds);
}

__device__ __forceinline__ bool leader() { return blockIdx.x == 0 && threadIdx.x == 0; }

// The loop state as ONE thread of the workgroup reads it, handed to the others through LDS.  A stopping
// test's leader may publish state != 0 while this very launch is still starting waves; waves of one
// workgroup must not disagree about it (those that carried on would reduce over LDS slots the others never
// wrote).  Different workgroups may still read different values: each then evaluates the same test on the
// same partial sums and reaches the same decision.  Used by every kernel that contains a stopping test.
__device__ __forceinline__ int uniform_state(const LoopState *st)
{
    __shared__ int s_state;
    if (threadIdx.x == 0) s_state = st->state;
    __syncthreads();
    const int v = s_state;
    __syncthreads();
    return v;
}

__device__ __forceinline__ void publish_progress(const LoopArgs &la, int state)
{
    if (la.snap && leader())
        __hip_atomic_store(&la.snap[la.k % la.snap_slots],
                           ((unsigned long long)(unsigned)(la.k + 1) << 32) | (unsigned long long)(unsigned)state,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------- stopping tests
// half-step test, pbicgstab.cu:111-118.  Returns true when the caller must return.
__device__ __forceinline__ bool check_half(const LoopArgs &la, const ScalarSrc &half, double *lds)
{
    LoopState *st = la.st;
    if (uniform_state(st) != 0) return true;
    double sc[1];
    load_scalars<1>(half, sc, lds);
    const double nrm = sqrt(sc[0]);
    const int it = st->it;
    if (la.loop == CUDAMAT_LOOP_PBICGSTAB) {
        if (leader()) {
            st->nrm = nrm;
            if (la.hist && 2 * it < la.hist_cap) la.hist[2 * it] = nrm;
        }
        if (!la.no_exit && nrm < st->tolabs) {
            if (leader()) st->state = 1;
            return true;
        }
        // A NaN residual never passes a test: the reference would spin to maxit on NaNs (pbicgstab.cu:116 has no guard);
        // here the loop stops and reports a breakdown, like the reference's own guard of the other loop (:735-742).
        if (!la.no_exit && isnan(nrm)) {
            if (leader()) st->state = 3;
            return true;
        }
    }
    return false;
}

// full-step test of iteration it-1, pbicgstab.cu:142-151 / :723-742.  sc = (rw.r, r.r)
__device__ __forceinline__ bool check_full(const LoopArgs &la, const double (&sc)[2])
{
    LoopState *st = la.st;
    const int it = st->it;
    if (it == 0) return false;
    const double nrm = sqrt(sc[1]);
    const double omega = st->omega;
    if (leader()) {
        st->nrm = nrm;
        if (la.hist) {
            const int slot = (la.loop != CUDAMAT_LOOP_PBICGSTAB2) ? 2 * (it - 1) + 1 : it - 1;
            if (slot < la.hist_cap) la.hist[slot] = nrm;
        }
    }
    if (la.no_exit) return false;
    if (nrm < st->tolabs) {
        if (leader()) st->state = 2;
        return true;
    }
    if (la.loop == CUDAMAT_LOOP_PBICGSTAB2 && (fabs(omega) < 1e-5 || isnan(omega))) {
        if (leader()) st->state = 3;
        return true;
    }
    if (isnan(nrm)) {                       // (see check_half)
        if (leader()) st->state = 3;
        return true;
    }
    return false;
}

// entries of one stream tile (kernels.h: SpmvPlan.stream_rows); LDS: kStreamNnz products + R+1 row pointers
constexpr int kStreamNnz = 2048;

// exclusive scan of one int per thread over the workgroup; *total = the sum
__device__ __forceinline__ int block_scan_int(int v, int *lds_waves, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();                       // lds_waves may still be read from the previous round
    if (lane == 63) lds_waves[wave] = inc;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) {
        const int t = lds_waves[w];
        if (w < wave) before += t;
        all += t;
    }
    *total = all;
    return before + inc - v;
}

// y = alpha*(sum + d.*xd) + beta*y for one row, and the row's share of the fused dots (w.y, y.y)
__device__ __forceinline__ void spmv_finish_row(const SpmvArgs &a, int row, double sum, double (&acc)[2])
{
    if (a.d) sum += a.d[row] * a.xd[row];
    double out = a.alpha * sum;
    if (a.beta != 0.0) out += a.beta * a.y[row];
    a.y[row] = out;
    if (a.dot) {
        acc[0] += out * a.w[row];
        acc[1] += out * out;
    }
}

// ---- streaming vector kernels: 16 bytes per lane (double2) whenever every operand is 16-byte aligned
static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

#define COMMA ,
#define CM_VEC_LOOP(N, BODY2, BODY1)                                                   \
    {                                                                                  \
        const int64_t n2__ = VEC ? (N) / 2 : 0;                                        \
        const int64_t stride__ = (int64_t)gridDim.x * kBlock;                          \
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2__; i += stride__) { BODY2 } \
        for (int64_t i = 2 * n2__ + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < (N); i += stride__) { BODY1 } \
    }

}  // namespace cm
