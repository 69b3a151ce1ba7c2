// trsv.hip -- application of the ILU(0) factors: t = L^-1 y (unit diagonal), U^-1 t (cusparseDcsrsv_solve x4 per
// iteration, pbicgstab.cu:92-98,:121-127).  The forms and when each is taken are described at the top of ilu.hip.
#include <algorithm>
#include <vector>

#include "ilu.h"

using namespace cm;

namespace cm {

// ---------------------------------------------------------------- triangular solves
// out[o(pr)] = (rhs[i(pr)] - sum_k val[k] out[col[k]]) * dinv   for the permuted rows [r0, r1).
// Index spaces: a permuted row pr reads rhs at rhs_of[pr] and writes out at out_of[pr] (a nullptr map = pr itself);
// the stored columns are indices into `out`.  Factors in ORIGINAL index space have rhs_of = out_of = row_of; factors
// in LEVEL-MAJOR space (hybrid, TriFactor::lm) have out_of = nullptr -- the solve writes a contiguous stream -- and
// rhs_of = nullptr (L: the right-hand side is in L's space) or the U-position -> L-position map (U).
// LANES lanes per row, exactly the SpMV inner loop; rows of one level are independent.
template <int LANES>
__device__ __forceinline__ void trsv_rows(int r0, int r1, int first, int stride, const int *frp, const int *fci,
                                          const double *fval, const int *rhs_of, const int *out_of, const double *dinv,
                                          const double *far, const double *rhs, double *out)
{
    const int lane = threadIdx.x & (LANES - 1);
    for (int pr = r0 + first; pr < r1; pr += stride) {
        const int s = frp[pr], e = frp[pr + 1];
        double sum = 0.0;
        for (int k = s + lane; k < e; k += LANES) sum += fval[k] * out[fci[k]];
#pragma unroll
        for (int o = LANES / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0) {
            double v = rhs[rhs_of ? rhs_of[pr] : pr] - sum;
            if (far) v -= far[pr];              // entries whose column lies in an earlier group
            if (dinv) v *= dinv[pr];
            out[out_of ? out_of[pr] : pr] = v;
        }
    }
}

template <int LANES>
__global__ __launch_bounds__(kBlock) void k_trsv_level(int r0, int r1, const int *frp, const int *fci,
                                                       const double *fval, const int *rhs_of, const int *out_of,
                                                       const double *dinv, const double *far, const double *rhs,
                                                       double *out)
{
    constexpr int RPB = kBlock / LANES;
    trsv_rows<LANES>(r0, r1, blockIdx.x * RPB + threadIdx.x / LANES, gridDim.x * RPB, frp, fci, fval, rhs_of, out_of,
                     dinv, far, rhs, out);
}

// several consecutive small levels in ONE workgroup: a workgroup-scope fence + barrier publishes a
// level's results (same CU, same L1) to the threads that consume them in the next level.
template <int LANES>
__global__ __launch_bounds__(kBlock) void k_trsv_small_levels(int l0, int l1, const int *level_ptr,
                                                              const int *frp, const int *fci, const double *fval,
                                                              const int *rhs_of, const int *out_of, const double *dinv,
                                                              const double *far, const double *rhs, double *out)
{
    constexpr int RPB = kBlock / LANES;
    for (int l = l0; l < l1; l++) {
        trsv_rows<LANES>(level_ptr[l], level_ptr[l + 1], threadIdx.x / LANES, RPB, frp, fci, fval, rhs_of, out_of, dinv,
                         far, rhs, out);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// ---- dependency-driven ("sync-free") solve of a whole group of levels in ONE launch
// Rows are stored level-major, so every dependency of permuted row pr sits at a smaller pr.  Workgroups CLAIM
// chunks of consecutive rows from an atomic ticket, so chunk t is only ever held by a workgroup that is already
// running, and it is claimed after every chunk < t: a waiting row waits for rows of its own wave or of chunks
// held by resident workgroups, and the lowest unfinished chunk never waits for anything unclaimed -- forward
// progress does not depend on the order or the number of workgroups the dispatcher starts (the grid may be
// smaller or larger than what fits; other kernels may hold part of the GPU).  A workgroup fetches its next
// ticket while it works on the current one (the atomic's round trip is off the chain); holding a ticket early is
// harmless: its owner is resident and working on a lower chunk.  Readiness travels with
// the data: `out` is pre-filled with a SIGNALLING-NaN bit pattern that no arithmetic result can have (every
// operation quiets a signalling NaN), a producer publishes its value with one 8-byte write-through (sc1)
// store and consumers poll the value itself with 8-byte sc1 loads -- no flags, no fences (one naturally
// aligned 8-byte granule written by one store).  Each row is still summed by its own LANES lanes in the
// level kernel's order, so the result is bit-identical to the level-by-level solve.
// Every spin is still bounded (defence in depth): a lane that gives up sets *err (pinned host word) and proceeds
// with what it read; cudamat_solver_solve then redoes the solve level by level and counts it (trsv_fallbacks).
constexpr unsigned long long kNotReady = 0x7FF4C0DEC0DEC0DEull;

__global__ __launch_bounds__(kBlock) void k_fill_not_ready(long long n, unsigned long long *out)
{
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
        out[i] = kNotReady;
}

template <int LANES, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_trsv_syncfree(int r0, int r1, const int *frp, const int *fci,
                                                          const double *fval, const int *rhs_of, const int *out_of,
                                                          const double *dinv, const double *far,
                                                          const double *rhs, double *out, int *err, int spin_limit,
                                                          int nap, unsigned *ticket, int steps)
{
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    constexpr int RPB = BLOCK / LANES;
    const int lane = threadIdx.x & (LANES - 1);
    const int team_shift = (threadIdx.x & 63) & ~(LANES - 1);
    constexpr unsigned long long team_bits = LANES == 64 ? ~0ull : ((1ull << LANES) - 1ull);
    __shared__ unsigned s_ticket[2];
    const long long nsub = ((long long)(r1 - r0) + RPB - 1) / RPB;       // sub-chunks of RPB rows
    if (threadIdx.x == 0) s_ticket[0] = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int turn = 0;; turn ^= 1) {
    __syncthreads();                                   // this turn's ticket is in LDS (the slots alternate)
    const long long first = (long long)s_ticket[turn] * steps;
    if (first >= nsub) return;
    unsigned next_ticket = 0;                          // in flight while this chunk is solved, stored at its end
    if (threadIdx.x == 0) next_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   for (int step = 0; step < steps && first + step < nsub; step++) {
    const long long prl = (long long)r0 + (first + step) * RPB + threadIdx.x / LANES;
    const bool valid = prl < r1;
    const int pr = valid ? (int)prl : r0;
    int k = 0, e = 0;
    if (valid) {
        k = frp[pr] + lane;
        e = frp[pr + 1];
    }
    bool have = k < e;
    int c = 0;
    double a = 0.0;
    if (have) {
        c = fci[k];
        a = fval[k];
    }
    // everything the row's last step needs is fetched up front: only the polled values are on the chain
    int r = 0;
    double base = 0.0, fr = 0.0, di = 1.0;
    if (valid && lane == 0) {
        r = out_of ? out_of[pr] : pr;
        base = rhs[rhs_of ? rhs_of[pr] : pr];
        if (far) fr = far[pr];
        if (dinv) di = dinv[pr];
    }
    double sum = 0.0;
    bool done = !valid;
    int spins = 0;
    for (;;) {
        if (have) {
            const unsigned long long bits = __hip_atomic_load((gu64 *)(out + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool ready = bits != kNotReady;
            // give up after kSpinLimit polls -- or at once when another row already has (checked every 1024 polls),
            // so that a broken dependency costs one timeout, not one per waiting row
            bool give_up = false;
            if (!ready && (++spins & 1023) == 0)
                give_up = spins > spin_limit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
            if (ready || give_up) {
                if (!ready) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                sum += a * __longlong_as_double((long long)bits);
                spins = 0;
                k += LANES;
                have = k < e;
                if (have) {
                    c = fci[k];
                    a = fval[k];
                }
            }
        }
        const unsigned long long pending = __ballot(have);
        if (!done && ((pending >> team_shift) & team_bits) == 0) {      // uniform over the row's lanes
            double tot = sum;
#pragma unroll
            for (int o = LANES / 2; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
            if (lane == 0) {
                double v = base - tot;
                if (far) v -= fr;
                if (dinv) v *= di;
                __hip_atomic_store((gu64 *)(out + r), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            done = true;
        }
        if (__ballot(!done) == 0) break;
        if (pending) {
            if (nap == 1) __builtin_amdgcn_s_sleep(1);
            else if (nap == 2) __builtin_amdgcn_s_sleep(2);
            else if (nap >= 4) {
                for (int q = 0; q < nap; q += 4) __builtin_amdgcn_s_sleep(8);      // nap/4 x 512 cycles
            }
        }
    }
   }   // steps of one ticket
    if (threadIdx.x == 0) s_ticket[turn ^ 1] = next_ticket;
  }    // tickets
}

// ---- small systems (n <= 16384): the whole solve in ONE workgroup with the solution vector in LDS
// The level-by-level chain is then LDS read -> multiply-add -> shuffle -> LDS write -> barrier (~0.15 us per level):
// each team's row of the NEXT level (row pointers, first entries, right-hand side, 1/diagonal) is fetched from
// global memory before the barrier, so nothing but LDS sits between two levels (mat10000: 199 levels per factor).
// Same per-row summation as k_trsv_level (lane k takes entries k, k + LANES, ...; xor tree) => bit-identical.

template <int LANES>
__global__ __launch_bounds__(kBlock) void k_trsv_lds(int n, int nlev, const int *level_ptr, const int *frp, const int *fci,
                                                     const double *fval, const int *row_of, const double *dinv,
                                                     const double *rhs, double *out)
{
    extern __shared__ __attribute__((aligned(16))) double xs[];       // n doubles, original row numbering
    constexpr int RPB = kBlock / LANES;
    const int lane = threadIdx.x & (LANES - 1), team = threadIdx.x / LANES;
    // prefetched state of this team's first row of the coming level
    int pr = 0, s = 0, e = 0, r = 0, c = 0;
    double a = 0.0, b = 0.0, di = 1.0;
    bool mine = false;
    auto fetch = [&](int l) {
        mine = false;
        if (l >= nlev) return;
        pr = level_ptr[l] + team;
        mine = pr < level_ptr[l + 1];
        if (!mine) return;
        s = frp[pr];
        e = frp[pr + 1];
        if (s + lane < e) {
            c = fci[s + lane];
            a = fval[s + lane];
        }
        if (lane == 0) {
            r = row_of[pr];
            b = rhs[r];
            if (dinv) di = dinv[pr];
        }
    };
    fetch(0);
    for (int l = 0; l < nlev; l++) {
        const int lend = level_ptr[l + 1];
        const bool have = mine;
        const int pr0 = pr, s0 = s, e0 = e, r0 = r;
        const double b0 = b, di0 = di;
        double sum = 0.0;
        if (have) {
            if (s0 + lane < e0) sum = a * xs[c];
            for (int k = s0 + lane + LANES; k < e0; k += LANES) sum += fval[k] * xs[fci[k]];
        }
#pragma unroll
        for (int o = LANES / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (have && lane == 0) {
            double v = b0 - sum;
            if (dinv) v *= di0;
            xs[r0] = v;
            out[r0] = v;
        }
        // further rows of a level wider than the workgroup's teams
        for (int q = pr0 + RPB; have && q < lend; q += RPB) {
            const int qs = frp[q], qe = frp[q + 1];
            double t = 0.0;
            for (int k = qs + lane; k < qe; k += LANES) t += fval[k] * xs[fci[k]];
#pragma unroll
            for (int o = LANES / 2; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane == 0) {
                const int rr = row_of[q];
                double v = rhs[rr] - t;
                if (dinv) v *= dinv[q];
                xs[rr] = v;
                out[rr] = v;
            }
        }
        fetch(l + 1);                      // global loads of the next level overlap the barrier
        __syncthreads();
    }
}

// the far SpMV of one group: far_buf[rows of the group] = far_g . out
static int launch_far(hipStream_t st, const TriFactor &F, const TriHost &H, int grp, const double *out)
{
    const int r0 = F.level_ptr[(size_t)H.grp_level[(size_t)grp]];
    SpmvArgs a{};
    a.n = H.far[(size_t)grp].n;
    a.x = out;
    a.alpha = 1.0;
    a.beta = 0.0;
    a.y = H.far_buf + r0;
    a.dot = 0;
    a.loop = LoopArgs{nullptr, nullptr, 0, 0, 0};
    a.check = CHECK_NONE;
    a.half = ScalarSrc{nullptr, 0, 1};
    a.pb_strict = H.pb_strict;
    return launch_spmv_pb(st, H.far[(size_t)grp], a);
}

template <int LANES>
static int launch_trsv_segments(hipStream_t st, const TriFactor &F, const TriHost &H, const double *rhs, double *out)
{
    constexpr int RPB = kBlock / LANES;
    int cur_group = -1;
    for (size_t g = 0; g < H.seg_begin.size(); g++) {
        const int grp = H.seg_group.empty() ? 0 : H.seg_group[g];
        const double *far = nullptr;
        if (H.hybrid && grp > 0 && H.far[(size_t)grp].nnz > 0) {
            if (grp != cur_group) CM_TRY(launch_far(st, F, H, grp, out));   // one blocked SpMV per group
            far = H.far_buf;
        }
        cur_group = grp;
        const int l0 = H.seg_begin[g], l1 = H.seg_end[g];
        const int r0 = F.level_ptr[(size_t)l0], r1 = F.level_ptr[(size_t)l1];
        const bool big = (l1 - l0 == 1) && (r1 - r0 > kSmallLevel);
        if (big) {
            int grid = (r1 - r0 + RPB - 1) / RPB;
            if (grid > 4096) grid = 4096;
            hipLaunchKernelGGL(k_trsv_level<LANES>, dim3(grid), dim3(kBlock), 0, st, r0, r1, F.rp, F.ci, F.val,
                               F.rhs_of, F.out_of, F.dinv, far, rhs, out);
        } else {
            hipLaunchKernelGGL(k_trsv_small_levels<LANES>, dim3(1), dim3(kBlock), 0, st, l0, l1, H.level_ptr_dev,
                               F.rp, F.ci, F.val, F.rhs_of, F.out_of, F.dinv, far, rhs, out);
        }
    }
    return CUDAMAT_OK;
}

template <int LANES, int BLOCK>
static int launch_trsv_syncfree_b(hipStream_t st, const TriFactor &F, const TriHost &H, int n, const double *rhs,
                                  double *out, int *err, int per_cu)
{
    constexpr int RPB = BLOCK / LANES;
    int fill_grid = (int)(((long long)n + kBlock - 1) / kBlock);
    if (fill_grid > kVecGridMax) fill_grid = kVecGridMax;
    hipLaunchKernelGGL(k_fill_not_ready, dim3(fill_grid ? fill_grid : 1), dim3(kBlock), 0, st, (long long)n,
                       (unsigned long long *)out);
    const int K = (int)H.grp_level.size() - 1;
    CM_HIP(hipMemsetAsync(H.tickets, 0, sizeof(unsigned) * (size_t)(K > 0 ? K : 1), st));     // one ticket counter per launch
    for (int g = 0; g < K; g++) {
        const int r0 = F.level_ptr[(size_t)H.grp_level[(size_t)g]], r1 = F.level_ptr[(size_t)H.grp_level[(size_t)g + 1]];
        if (r1 <= r0) continue;
        const double *far = nullptr;
        if (H.hybrid && g > 0 && H.far[(size_t)g].nnz > 0) {
            CM_TRY(launch_far(st, F, H, g, out));
            far = H.far_buf;
        }
        // A ticket = `steps` consecutive sub-chunks of RPB rows: at least 256 rows, so that the one counter sees an
        // atomic per 256 rows at most (far below what one address sustains).  The rows in flight are
        // (resident workgroups) x (rows per ticket): kept as narrow as a level or two, because a row far ahead of the
        // lowest unfinished one would mostly wait -- hence big workgroups (BLOCK = 256 x occ threads, one per CU)
        // rather than many small ones.
        const long long nsub = ((long long)(r1 - r0) + RPB - 1) / RPB;
        int steps = 256 / RPB > 0 ? 256 / RPB : 1;
        while (steps > 1 && nsub / steps < 1024) steps >>= 1;
        const long long ntick = (nsub + steps - 1) / steps;
        // no more workgroups than can be resident; more would only queue behind the persistent ones
        const long long cap = (long long)per_cu * 256;
        const unsigned grid = (unsigned)(ntick < cap ? ntick : cap);
        // residency is pinned with an (unused) dynamic LDS request: fewer waiting workgroups, fewer pollers
        const size_t lds_pad = (size_t)(152 * 1024) / (size_t)per_cu;
        CM_TRY(set_max_lds((const void *)k_trsv_syncfree<LANES, BLOCK>));
        hipLaunchKernelGGL((k_trsv_syncfree<LANES, BLOCK>), dim3(grid), dim3(BLOCK), lds_pad, st, r0, r1, F.rp, F.ci, F.val,
                           F.rhs_of, F.out_of, F.dinv, far, rhs, out, err, H.spin_limit, H.nap, H.tickets + g, steps);
    }
    return CUDAMAT_OK;
}

// occ = resident waves per CU / 4 (1, 2, 4, 8): one workgroup of 256 x occ threads per CU (two of 1024 at occ = 8)
template <int LANES>
static int launch_trsv_syncfree(hipStream_t st, const TriFactor &F, const TriHost &H, int n, const double *rhs,
                                double *out, int *err)
{
    const int occ = H.occ >= 8 ? 8 : H.occ >= 4 ? 4 : H.occ >= 2 ? 2 : 1;
    if (occ == 1) return launch_trsv_syncfree_b<LANES, 256>(st, F, H, n, rhs, out, err, 1);
    if (occ == 2) return launch_trsv_syncfree_b<LANES, 512>(st, F, H, n, rhs, out, err, 1);
    return launch_trsv_syncfree_b<LANES, 1024>(st, F, H, n, rhs, out, err, occ == 8 ? 2 : 1);
}

bool trsv_syncfree_active(cudamat_solver *s)
{
    IluPlans *pl = plans_of(s, false);
    return pl && s->has_ilu && (pl->L.syncfree || pl->U.syncfree);
}

int trsv_form_code(cudamat_solver *s)
{
    IluPlans *pl = plans_of(s, false);
    if (!pl || !s->has_ilu) return 0;
    if (pl->L.syncfree || pl->U.syncfree) return 1;
    if (pl->L.lds || pl->U.lds) return 2;
    return 0;
}

void trsv_group_counts(cudamat_solver *s, int *groups_l, int *groups_u)
{
    *groups_l = *groups_u = 0;
    IluPlans *pl = plans_of(s, false);
    if (!pl || !s->has_ilu) return;
    if (pl->L.hybrid) *groups_l = (int)pl->L.grp_level.size() - 1;
    if (pl->U.hybrid) *groups_u = (int)pl->U.grp_level.size() - 1;
}

void trsv_disable_syncfree(cudamat_solver *s)
{
    if (IluPlans *pl = plans_of(s, false)) pl->L.syncfree = pl->U.syncfree = false;
}

int trsv_status(cudamat_solver *s)
{
    IluPlans *pl = plans_of(s, false);
    if (pl && pl->err_host && *pl->err_host) {
        *pl->err_host = 0;
        set_error("triangular solve: a dependency never became ready (spin limit reached)");
        return CUDAMAT_ERR_HIP;
    }
    return CUDAMAT_OK;
}

int trsv_apply(cudamat_solver *s, const TriFactor &F, bool upper, const double *rhs, double *out)
{
    IluPlans *pl = plans_of(s, false);
    if (!pl || !s->has_ilu) { set_error("ILU(0) factors missing"); return CUDAMAT_ERR_ARG; }
    const TriHost &H = upper ? pl->U : pl->L;
    hipStream_t st = s->ctx->stream;
    int rc;
    if (H.lds && !H.syncfree) {
        const size_t bytes = sizeof(double) * (size_t)s->n;
#define CM_TRSV_LDS(LV)                                                                                             \
    do {                                                                                                            \
        CM_TRY(set_max_lds((const void *)k_trsv_lds<LV>));                                                          \
        hipLaunchKernelGGL(k_trsv_lds<LV>, dim3(1), dim3(kBlock), bytes, st, s->n, F.nlevels, H.level_ptr_dev, F.rp, \
                           F.ci, F.val, F.row_of, F.dinv, rhs, out);                                                \
    } while (0)
        switch (H.lanes) {
        case 2:  CM_TRSV_LDS(2); break;
        case 4:  CM_TRSV_LDS(4); break;
        case 8:  CM_TRSV_LDS(8); break;
        case 16: CM_TRSV_LDS(16); break;
        case 32: CM_TRSV_LDS(32); break;
        default: CM_TRSV_LDS(64); break;
        }
#undef CM_TRSV_LDS
        CM_HIP(hipGetLastError());
        return CUDAMAT_OK;
    }
    if (H.syncfree) {
        if (rhs == out) { set_error("triangular solve: rhs and out must not alias"); return CUDAMAT_ERR_ARG; }
        switch (H.lanes) {
        case 2:  rc = launch_trsv_syncfree<2>(st, F, H, s->n, rhs, out, pl->err_dev); break;
        case 4:  rc = launch_trsv_syncfree<4>(st, F, H, s->n, rhs, out, pl->err_dev); break;
        case 8:  rc = launch_trsv_syncfree<8>(st, F, H, s->n, rhs, out, pl->err_dev); break;
        case 16: rc = launch_trsv_syncfree<16>(st, F, H, s->n, rhs, out, pl->err_dev); break;
        case 32: rc = launch_trsv_syncfree<32>(st, F, H, s->n, rhs, out, pl->err_dev); break;
        default: rc = launch_trsv_syncfree<64>(st, F, H, s->n, rhs, out, pl->err_dev); break;
        }
        CM_TRY(rc);
        CM_HIP(hipGetLastError());
        return CUDAMAT_OK;
    }
    switch (H.lanes) {
    case 2:  rc = launch_trsv_segments<2>(st, F, H, rhs, out); break;
    case 4:  rc = launch_trsv_segments<4>(st, F, H, rhs, out); break;
    case 8:  rc = launch_trsv_segments<8>(st, F, H, rhs, out); break;
    case 16: rc = launch_trsv_segments<16>(st, F, H, rhs, out); break;
    case 32: rc = launch_trsv_segments<32>(st, F, H, rhs, out); break;
    default: rc = launch_trsv_segments<64>(st, F, H, rhs, out); break;
    }
    CM_TRY(rc);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

// ---- vectors between the original row numbering and the level-major spaces
__global__ __launch_bounds__(kBlock) void k_perm_gather(long long n, const int *map, const double *in, double *out)
{
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) out[i] = in[map[i]];
}
__global__ __launch_bounds__(kBlock) void k_perm_scatter(long long n, const int *map, const double *in, double *out)
{
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) out[map[i]] = in[i];
}

static unsigned perm_grid(int n)
{
    long long g = ((long long)n + kBlock - 1) / kBlock;
    return (unsigned)(g < 1 ? 1 : g > 4096 ? 4096 : g);
}

// out[pr] = in[row at position pr of L's (upper: U's) level-major order]
int perm_to_space(cudamat_solver *s, bool upper, const double *in, double *out)
{
    const TriFactor &F = upper ? s->U : s->L;
    hipLaunchKernelGGL(k_perm_gather, dim3(perm_grid(s->n)), dim3(kBlock), 0, s->ctx->stream, (long long)s->n, F.row_of, in, out);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

// out[row at position pr] = in[pr]
int perm_from_space(cudamat_solver *s, bool upper, const double *in, double *out)
{
    const TriFactor &F = upper ? s->U : s->L;
    hipLaunchKernelGGL(k_perm_scatter, dim3(perm_grid(s->n)), dim3(kBlock), 0, s->ctx->stream, (long long)s->n, F.row_of, in, out);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

// M^-1 on vectors in ORIGINAL numbering while the factors live in level-major spaces: permute in, solve, permute out
int precond_apply_original(cudamat_solver *s, const double *in, double *tmp, double *out)
{
    IluPlans *pl = plans_of(s, false);
    if (!pl || !pl->perm_a || !pl->perm_b) { set_error("level-major scratch vectors missing"); return CUDAMAT_ERR_ARG; }
    CM_TRY(perm_to_space(s, false, in, pl->perm_a));
    CM_TRY(trsv_apply(s, s->L, false, pl->perm_a, tmp));
    CM_TRY(trsv_apply(s, s->U, true, tmp, pl->perm_b));
    return perm_from_space(s, true, pl->perm_b, out);
}

}  // namespace cm
