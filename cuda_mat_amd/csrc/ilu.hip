// ilu.hip -- ILU(0): level analysis, factorisation, level-major storage of the factors (their application: trsv.hip).
//
// Replaces, for the ILU path of the reference (pbicgstab.cu:157-409):
//   cusparseDcsrsv_analysis x2 (:336-347)  -> level sets of the strict-lower and strict-upper
//                                            pattern (dependency depth of every row)
//   cusparseDcsrilu0 (:359)                -> in-place ILU(0) on a copy of A's values that
//                                            shares A's pattern (:316,:357-358), no pivoting
//   cusparseDcsrsv_solve x4 / iteration    -> t = L^-1 y (unit diagonal), U^-1 t (:92-98,:121-127)
//
// MI355X design: rows are grouped by dependency level (one dependency-driven pass, k_levels_dep); L and U are
// re-stored in LEVEL-MAJOR order (rows of one level contiguous, own rowptr/colidx/values, 1/diag precomputed for
// U).  A triangular solve then takes the form that fits the factor:
//   * k_trsv_syncfree   one launch per GROUP of levels; rows wait inside the launch for the values they depend on
//                       (bounded polls of a "not ready" bit pattern) -- default when a level is wider than 512 rows;
//                       residency is throttled when the work per level is small (pollers slow the hand-offs);
//   * hybrid split      big factors with scattered columns: entries whose column lies in an EARLIER group go
//                       through one blocked two-phase SpMV per group (streams), only the rest is gathered;
//   * k_trsv_lds        n <= 16384 with narrow levels: one workgroup, solution vector in LDS, barrier per level;
//   * k_trsv_level / k_trsv_small_levels   one launch per level / one single-workgroup launch per run of small
//                       levels -- the reference form the others are bit-identical to, and the fallback when a
//                       dependency-driven solve times out (another spinning kernel on the same GPU).
// Block-Jacobi (row-sharded runs): the same machinery on the rank's diagonal block (select_precond_matrix).
// Algorithmic bytes per preconditioner application: 12 nnz + 8 (n+1) + 32 n.
#include <algorithm>
#include <chrono>
#include <vector>

#include "ilu.h"

using namespace cm;

namespace cm {

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define CM_STAMP(label)                                                                       \
    do {                                                                                      \
        if (s->ctx->cfg.verbose) {                                                            \
            CM_DROP(hipStreamSynchronize(st));   /* (diagnostic timing only) */                       \
            const double t_now = now_s();                                                     \
            fprintf(stderr, "[cudamat] ilu0 %-28s %8.3f ms\n", label, (t_now - t_stamp) * 1e3); \
            t_stamp = t_now;                                                                  \
        }                                                                                     \
    } while (0)


// Temporaries of the pattern-only analysis when it runs beside an upload (dropin.hip): hipFree waits for every stream of
// the device, i.e. for the copy piece in flight (2.4 ms each at C4) -- the frees are collected and carried out after the
// last byte has landed (ilu0_flush_deferred).  nullptr: free at once.
static thread_local std::vector<void *> *t_deferred = nullptr;
static void afree(void *p)
{
    if (!p) return;
    if (t_deferred) t_deferred->push_back(p);
    else CM_DROP(hipFree(p));
}

template <typename T>
static int dalloc(T **p, size_t count)
{
    *p = nullptr;
    hipError_t e = hipMalloc((void **)p, sizeof(T) * (count ? count : 1));
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", sizeof(T) * count, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? CUDAMAT_ERR_NOMEM : CUDAMAT_ERR_HIP;
    }
    return CUDAMAT_OK;
}

// ---------------------------------------------------------------- analysis kernels
// position of the diagonal entry of every row (-1 when structurally missing: pbicgstab.h:118)
__global__ __launch_bounds__(kBlock) void k_find_diag(int n, const int *rp, const int *ci, int *diag_pos,
                                                      int *flags)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int lo = rp[i], hi = rp[i + 1];
    const int e = hi;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ci[mid] < i) lo = mid + 1; else hi = mid;
    }
    const bool ok = lo < e && ci[lo] == i;
    diag_pos[i] = ok ? lo : -1;
    if (!ok) atomicMax(&flags[1], i + 1);
}

// One relaxation sweep of lev[i] = max_{j in deps(i)} lev[j] + 1 (deps = strict lower or strict
// upper part of row i).  8 lanes per row.  Repeated until nothing changes; the fixed point is
// the dependency depth.  Updates are in place (chaotic relaxation converges to the same fixed
// point, faster than Jacobi).
__global__ __launch_bounds__(kBlock) void k_level_sweep(int n, const int *rp, const int *ci,
                                                        const int *diag_pos, int upper, int *lev, int *flags)
{
    constexpr int L = 8;
    const int lane = threadIdx.x & (L - 1);
    const long long row = ((long long)blockIdx.x * kBlock + threadIdx.x) / L;
    if (row >= n) return;
    const int i = (int)row;
    const int s = upper ? diag_pos[i] + 1 : rp[i];
    const int e = upper ? rp[i + 1] : diag_pos[i];
    int m = 0;
    for (int k = s + lane; k < e; k += L) {
        const int l = __hip_atomic_load(&lev[ci[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        m = l > m ? l : m;
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) {
        const int t = __shfl_xor(m, o, 64);
        m = t > m ? t : m;
    }
    if (lane == 0 && m != lev[i]) {
        __hip_atomic_store(&lev[i], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flags[0] = 1;
    }
}

// The same fixed point in ONE pass: rows are visited in dependency order (ascending for the lower part,
// descending for the upper part), a row waits until the levels of its dependencies have been published
// (lev preset to -1; 4-byte write-through stores, polled with sc1 loads) and publishes its own.  Workgroups
// claim chunks of consecutive positions from an atomic ticket (see k_trsv_syncfree), so a waiting row only
// waits for rows of resident workgroups whatever the dispatcher does; spins are bounded all the same and a
// timeout (err) sends the caller back to the relaxation sweeps.
template <int L>
__global__ __launch_bounds__(kBlock) void k_levels_dep(int n, const int *rp, const int *ci, const int *diag_pos,
                                                       int upper, int *lev, int *err, unsigned *ticket, int steps)
{
    const int lane = threadIdx.x & (L - 1);
    const int team_shift = (threadIdx.x & 63) & ~(L - 1);
    __shared__ unsigned s_ticket[2];
    const long long nsub = ((long long)n * L + kBlock - 1) / kBlock;
    if (threadIdx.x == 0) s_ticket[0] = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int turn = 0;; turn ^= 1) {
    __syncthreads();
    const long long first = (long long)s_ticket[turn] * steps;
    if (first >= nsub) return;
    unsigned next_ticket = 0;
    if (threadIdx.x == 0) next_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   for (int step = 0; step < steps && first + step < nsub; step++) {
    const long long t = ((first + step) * kBlock + threadIdx.x) / L;
    const bool valid = t < n;
    const int i = valid ? (upper ? n - 1 - (int)t : (int)t) : 0;
    int k = 0, e = 0;
    if (valid) {
        k = (upper ? diag_pos[i] + 1 : rp[i]) + lane;
        e = upper ? rp[i + 1] : diag_pos[i];
    }
    bool have = k < e;
    int c = have ? ci[k] : 0;
    int m = 0, spins = 0;
    bool done = !valid;
    for (;;) {
        if (have) {
            const int l = __hip_atomic_load(&lev[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool give_up = false;
            if (l < 0 && (++spins & 1023) == 0)
                give_up = spins > (1 << 21) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
            if (l >= 0 || give_up) {
                if (l < 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                m = l + 1 > m ? l + 1 : m;
                spins = 0;
                k += L;
                have = k < e;
                if (have) c = ci[k];
            }
        }
        const unsigned long long pending = __ballot(have);
        if (!done && ((pending >> team_shift) & ((1ull << L) - 1ull)) == 0) {
            int mm = m;
#pragma unroll
            for (int o = L / 2; o > 0; o >>= 1) {
                const int q = __shfl_xor(mm, o, 64);
                mm = q > mm ? q : mm;
            }
            if (lane == 0) __hip_atomic_store(&lev[i], mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            done = true;
        }
        if (__ballot(!done) == 0) break;
        if (pending) __builtin_amdgcn_s_sleep(2);
    }
   }
    if (threadIdx.x == 0) s_ticket[turn ^ 1] = next_ticket;
  }
}

// copy one triangular part of the combined LU values into level-major storage
__global__ __launch_bounds__(kBlock) void k_fill_factor(int n, const int *rp, const int *ci, const double *lu,
                                                        const int *diag_pos, int upper, const int *row_of,
                                                        const int *frp, int *fci, double *fval, double *dinv)
{
    constexpr int L = 8;
    const int lane = threadIdx.x & (L - 1);
    const long long pr = ((long long)blockIdx.x * kBlock + threadIdx.x) / L;
    if (pr >= n) return;
    const int r = row_of[pr];
    const int s = upper ? diag_pos[r] + 1 : rp[r];
    const int e = upper ? rp[r + 1] : diag_pos[r];
    const int o = frp[pr];
    for (int k = lane; k < e - s; k += L) {
        fci[o + k] = ci[s + k];
        fval[o + k] = lu[s + k];
    }
    if (upper && lane == 0) dinv[pr] = 1.0 / lu[diag_pos[r]];
}

// ---------------------------------------------------------------- numeric ILU(0)
// One wavefront per row of the current level (IKJ ordering).  The row's values are staged in LDS
// (this wave's slice), where the wave's own in-order DS pipeline makes every update visible to the
// next elimination step without any global-memory round trip; the pivot rows k < i belong to
// earlier levels and are final in HBM.  Row k's entries right of its diagonal are spread over the
// 64 lanes, each finds its column in row i by binary search (columns are sorted).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_ilu0_level(int row_begin, int row_end, const int *row_of,
                                                          const int *rp, const int *ci, const int *diag_pos,
                                                          double *lu, int cap, int *flags)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pr = row_begin + blockIdx.x * WAVES + wave;
    if (pr >= row_end) return;
    double *sv = smem + (size_t)wave * cap;
    const int i = row_of[pr];
    const int rs = rp[i], len = rp[i + 1] - rs;
    const int nlow = diag_pos[i] - rs;
    for (int q = lane; q < len; q += 64) sv[q] = lu[rs + q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int kk = 0; kk < nlow; kk++) {
        const int k = ci[rs + kk];
        const int dk = diag_pos[k];
        const double piv = lu[dk];
        if (piv == 0.0 && lane == 0) atomicMax(&flags[1], k + 1);
        const double lik = sv[kk] / piv;                 // every lane reads the same LDS word
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) sv[kk] = lik;
        const int ke = rp[k + 1];
        for (int pp = dk + 1 + lane; pp < ke; pp += 64) {
            const int j = ci[pp];
            int lo = kk + 1, hi = len;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (ci[rs + mid] < j) lo = mid + 1; else hi = mid;
            }
            if (lo < len && ci[rs + lo] == j) sv[lo] -= lik * lu[pp];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (nlow >= 0 && sv[nlow] == 0.0 && lane == 0) atomicMax(&flags[1], i + 1);   // zero pivot of row i
    for (int q = lane; q < len; q += 64) lu[rs + q] = sv[q];
}

// The same elimination with every value-independent global access hoisted out of the dependent chain (rows of up
// to 1024 entries): the row's column ids and, for every pivot k of the row, (position of k's diagonal, end of
// row k, pivot value) are staged in LDS up front by all lanes at once, the binary search runs on the LDS copy of
// the columns, and the entries of pivot row kk+1 are fetched while pivot kk is being applied.  Per pivot the
// chain is then one LDS round trip instead of ~10 dependent global loads (C5: 169 -> 77 ms for the 137 levels).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_ilu0_level_fast(int row_begin, int row_end, const int *row_of,
                                                               const int *rp, const int *ci, const int *diag_pos,
                                                               double *lu, int cap, int *flags)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pr = row_begin + blockIdx.x * WAVES + wave;
    if (pr >= row_end) return;
    // per wave: sv[cap] values, mp[cap] pivots (doubles), then sc[cap] columns, mk[cap], me[cap] (ints)
    double *sv = smem + (size_t)wave * cap * 2;
    double *mp = sv + cap;
    int *sc = (int *)(smem + (size_t)WAVES * cap * 2) + (size_t)wave * cap * 3;
    int *mk = sc + cap;
    int *me = mk + cap;
    const int i = row_of[pr];
    const int rs = rp[i], len = rp[i + 1] - rs;
    const int nlow = diag_pos[i] - rs;
    for (int q = lane; q < len; q += 64) {
        sv[q] = lu[rs + q];
        sc[q] = ci[rs + q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int q = lane; q < nlow; q += 64) {
        const int k = sc[q];
        const int dk = diag_pos[k];
        mk[q] = dk;
        me[q] = rp[k + 1];
        mp[q] = lu[dk];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bool have = false;
    int j = 0;
    double u = 0.0;
    if (nlow > 0) {
        const int pp = mk[0] + 1 + lane;
        have = pp < me[0];
        if (have) {
            j = ci[pp];
            u = lu[pp];
        }
    }
    for (int kk = 0; kk < nlow; kk++) {
        const int dk = mk[kk], ke = me[kk];
        const double piv = mp[kk];
        bool have_n = false;                    // entries of the next pivot row: in flight while this one is applied
        int jn = 0;
        double un = 0.0;
        if (kk + 1 < nlow) {
            const int ppn = mk[kk + 1] + 1 + lane;
            have_n = ppn < me[kk + 1];
            if (have_n) {
                jn = ci[ppn];
                un = lu[ppn];
            }
        }
        if (piv == 0.0 && lane == 0) atomicMax(&flags[1], sc[kk] + 1);
        const double lik = sv[kk] / piv;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) sv[kk] = lik;
        if (have) {
            int lo = kk + 1, hi = len;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sc[mid] < j) lo = mid + 1; else hi = mid;
            }
            if (lo < len && sc[lo] == j) sv[lo] -= lik * u;
        }
        for (int pp = dk + 1 + 64 + lane; pp < ke; pp += 64) {       // pivot rows with more than 64 upper entries
            const int j2 = ci[pp];
            int lo = kk + 1, hi = len;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sc[mid] < j2) lo = mid + 1; else hi = mid;
            }
            if (lo < len && sc[lo] == j2) sv[lo] -= lik * lu[pp];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        have = have_n;
        j = jn;
        u = un;
    }
    if (nlow >= 0 && sv[nlow] == 0.0 && lane == 0) atomicMax(&flags[1], i + 1);   // zero pivot of row i
    for (int q = lane; q < len; q += 64) lu[rs + q] = sv[q];
}


static void free_factor(TriFactor &F)
{
    void *ptrs[] = {F.rp, F.ci, F.val, F.row_of, F.dinv};
    for (void *p : ptrs)
        if (p) CM_DROP(hipFree(p));
    if (F.lm && F.rhs_of) CM_DROP(hipFree(F.rhs_of));       // (in original space the maps alias row_of)
    F = TriFactor();
}

static void free_levels(TriHost &H)
{
    if (H.lev_dev) CM_DROP(hipFree(H.lev_dev));
    H.lev_dev = nullptr;
}

int ilu0_release(cudamat_solver *s)
{
    free_factor(s->L);
    free_factor(s->U);
    pb_free(&s->pb_perm);
    valdict_free(&s->vd_perm);
    if (s->x_perm) { CM_DROP(hipFree(s->x_perm)); s->x_perm = nullptr; }
    if (s->b_perm) { CM_DROP(hipFree(s->b_perm)); s->b_perm = nullptr; }
    s->perm_ready = false;
    s->perm_failed = false;
    if (s->lu) CM_DROP(hipFree(s->lu));
    if (s->diag_pos) CM_DROP(hipFree(s->diag_pos));
    if (s->pm_owned) {
        if (s->pm_rp) CM_DROP(hipFree(s->pm_rp));
        if (s->pm_ci) CM_DROP(hipFree(s->pm_ci));
        if (s->pm_val) CM_DROP(hipFree(s->pm_val));
    }
    s->pm_rp = s->pm_ci = nullptr;
    s->pm_val = nullptr;
    s->pm_nnz = 0;
    s->pm_owned = false;
    s->ilu_block = false;
    s->lu = nullptr;
    s->diag_pos = nullptr;
    s->has_ilu = false;
    if (IluPlans *pl = (IluPlans *)s->ilu_plans) {
        if (pl->L.level_ptr_dev) CM_DROP(hipFree(pl->L.level_ptr_dev));
        if (pl->U.level_ptr_dev) CM_DROP(hipFree(pl->U.level_ptr_dev));
        for (TriHost *h : {&pl->L, &pl->U}) {
            for (PbPlan &fp : h->far) pb_free(&fp);
            if (h->far_buf) CM_DROP(hipFree(h->far_buf));
            if (h->tickets) CM_DROP(hipFree(h->tickets));
            free_levels(*h);
        }
        if (pl->err_host) CM_DROP(hipHostFree(pl->err_host));
        if (pl->d_flags) CM_DROP(hipFree(pl->d_flags));
        if (pl->d_lev) CM_DROP(hipFree(pl->d_lev));
        for (void *q : pl->deferred) CM_DROP(hipFree(q));
        if (pl->posU) CM_DROP(hipFree(pl->posU));
        if (pl->perm_a) CM_DROP(hipFree(pl->perm_a));
        if (pl->perm_b) CM_DROP(hipFree(pl->perm_b));
        delete pl;
        s->ilu_plans = nullptr;
    }
    return CUDAMAT_OK;
}

// ---- exclusive prefix sums of n ints on the device (out has n + 1 entries, out[n] = total): tiles of 4096 elements
// scanned by one workgroup each, one workgroup over the tile totals, tile offsets added.  (Round 3: the host loops over
// 1e7 counts -- two device-to-host copies, a serial scan, two uploads per factor -- were ~40 ms each at C5.)
constexpr int kScanTile = 4096;

__device__ __forceinline__ int wg_exclusive_scan(int v, int *lds_waves, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) lds_waves[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < kBlock / 64; w++) {
        const int t = lds_waves[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(kBlock) void k_scan_tiles(int n, const int *in, int *out, int *tile_sum)
{
    __shared__ int lds_waves[kBlock / 64];
    const long long t0 = (long long)blockIdx.x * kScanTile;
    int run = 0;
    for (int c = 0; c < kScanTile; c += kBlock) {
        const long long i = t0 + c + threadIdx.x;
        const int v = i < n ? in[i] : 0;
        int total;
        const int ex = wg_exclusive_scan(v, lds_waves, &total);
        if (i < n) out[i] = run + ex;
        run += total;
    }
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = run;
}

__global__ __launch_bounds__(kBlock) void k_scan_tile_sums(int ntiles, int *tile_sum, int n, int *out)
{
    __shared__ int lds_waves[kBlock / 64];
    int run = 0;
    for (int c = 0; c < ntiles; c += kBlock) {
        const int i = c + threadIdx.x;
        const int v = i < ntiles ? tile_sum[i] : 0;
        int total;
        const int ex = wg_exclusive_scan(v, lds_waves, &total);
        if (i < ntiles) tile_sum[i] = run + ex;
        run += total;
    }
    if (threadIdx.x == 0) out[n] = run;
}

__global__ __launch_bounds__(kBlock) void k_scan_add(int n, const int *tile_sum, int *out)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] += tile_sum[i / kScanTile];
}

// out[0..n] = exclusive prefix sums of in[0..n); in and out may not alias
static int device_exclusive_scan(hipStream_t st, int n, const int *in, int *out)
{
    if (n <= 0) { CM_HIP(hipMemsetAsync(out, 0, sizeof(int), st)); return CUDAMAT_OK; }
    const int ntiles = (n + kScanTile - 1) / kScanTile;
    int *tile_sum = nullptr;
    CM_TRY(dalloc(&tile_sum, (size_t)ntiles));
    hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(kBlock), 0, st, n, in, out, tile_sum);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(kBlock), 0, st, ntiles, tile_sum, n, out);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)(((long long)n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, n, tile_sum, out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    afree(tile_sum);
    CM_HIP(e);
    return CUDAMAT_OK;
}

// ---- stable sort of the rows by level, on the device (round 4; the host's counting sort over 1e7 levels took 44 ms per
// factor on 16 threads, plus the read-back of the levels and of the pattern's row pointers and the upload of the result).
// LSD radix passes of 8 bits (one pass up to 256 levels: C5 has 137; two up to 65 536), each a stable counting sort:
// a wavefront takes kLsChunk consecutive positions -- histogram of its digits (k_lsort_hist), exclusive offsets in
// (digit, chunk) order (k_lsort_scan_digit / k_lsort_scan_tot), then it walks its positions IN ORDER and places every
// element behind the earlier ones of its digit (k_lsort_scatter: the lanes of one step that share a digit are ranked by
// lane number) -- so rows of one level stay in increasing order, as the level kernels and the bit-identity of the
// triangular-solve forms require.
constexpr int kLsChunk = 2048;
constexpr int kLsWaves = kBlock / 64;

__global__ __launch_bounds__(kBlock) void k_lsort_hist(int n, const int *key, int shift, int nchunks, int *cnt)
{
    __shared__ int h[kLsWaves][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * kLsWaves + wave;
    for (int d = lane; d < 256; d += 64) h[wave][d] = 0;
    __syncthreads();
    if (chunk < nchunks) {
        const long long i0 = (long long)chunk * kLsChunk;
        for (int c = 0; c < kLsChunk; c += 64) {
            const long long i = i0 + c + lane;
            if (i < n) atomicAdd(&h[wave][(key[i] >> shift) & 255], 1);
        }
    }
    __syncthreads();
    if (chunk < nchunks)
        for (int d = lane; d < 256; d += 64) cnt[(size_t)d * nchunks + chunk] = h[wave][d];
}

// one workgroup per digit: cnt[d][.] -> exclusive offsets inside the digit, tot[d] = elements with that digit
__global__ __launch_bounds__(kBlock) void k_lsort_scan_digit(int nchunks, int *cnt, int *tot)
{
    __shared__ int lds_waves[kBlock / 64];
    int *c = cnt + (size_t)blockIdx.x * nchunks;
    int run = 0;
    for (int c0 = 0; c0 < nchunks; c0 += kBlock) {
        const int i = c0 + threadIdx.x;
        const int v = i < nchunks ? c[i] : 0;
        int total;
        const int ex = wg_exclusive_scan(v, lds_waves, &total);
        if (i < nchunks) c[i] = run + ex;
        run += total;
    }
    if (threadIdx.x == 0) tot[blockIdx.x] = run;
}

__global__ __launch_bounds__(kBlock) void k_lsort_scan_tot(const int *tot, int *base)
{
    static_assert(kBlock == 256, "one digit per thread");
    __shared__ int lds_waves[kBlock / 64];
    int total;
    base[threadIdx.x] = wg_exclusive_scan(tot[threadIdx.x], lds_waves, &total);
}

__global__ __launch_bounds__(kBlock) void k_lsort_scatter(int n, const int *key_in, const int *row_in, int shift, int nchunks,
                                                          const int *cnt, const int *base, int *key_out, int *row_out)
{
    __shared__ int cur[kLsWaves][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * kLsWaves + wave;
    if (chunk < nchunks)
        for (int d = lane; d < 256; d += 64) cur[wave][d] = base[d] + cnt[(size_t)d * nchunks + chunk];
    __syncthreads();
    if (chunk >= nchunks) return;
    const long long i0 = (long long)chunk * kLsChunk;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int c = 0; c < kLsChunk; c += 64) {
        const long long i = i0 + c + lane;
        const bool active = i < n;
        const int k = active ? key_in[i] : 0;
        const int d = (k >> shift) & 255;
        const int r = active ? (row_in ? row_in[i] : (int)i) : 0;
        unsigned long long todo = __ballot(active);              // (wave-uniform: the loop below is, too)
        int dest = 0;
        while (todo) {
            const int leader = __ffsll(todo) - 1;
            const int dl = __shfl(d, leader, 64);
            const bool mine = active && d == dl;
            const unsigned long long m = __ballot(mine);
            // (one wavefront's LDS accesses execute in program order: every lane of the digit reads the cursor, then the
            // leader moves it on)
            if (mine) dest = cur[wave][dl] + __popcll(m & below);
            if (lane == leader) cur[wave][dl] += __popcll(m);
            todo &= ~m;
        }
        if (active) {
            key_out[dest] = k;
            row_out[dest] = r;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_max_int(int n, const int *v, int *out)
{
    __shared__ int red[kBlock / 64];
    int m = 0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) m = max(m, v[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) m = max(m, red[w]);
        atomicMax(out, m);
    }
}

// longest row of a CSR pattern (same reduction over rp[i + 1] - rp[i])
__global__ __launch_bounds__(kBlock) void k_max_row_len(int n, const int *rp, int *out)
{
    __shared__ int red[kBlock / 64];
    int m = 0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) m = max(m, rp[i + 1] - rp[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) m = max(m, red[w]);
        atomicMax(out, m);
    }
}

// level_ptr[l] = first position of level l in the sorted keys, level_ptr[last level + 1] = n
__global__ __launch_bounds__(kBlock) void k_level_ptr(int n, const int *skey, int *level_ptr)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int k = skey[i], kp = i ? skey[i - 1] : -1;
    for (int l = kp + 1; l <= k; l++) level_ptr[l] = (int)i;
    if (i == n - 1) level_ptr[k + 1] = n;
}

// entries of the factor's row pr = the strict lower / strict upper part of pattern row row_of[pr]
__global__ __launch_bounds__(kBlock) void k_factor_len(int n, const int *row_of, const int *rp, const int *diag, int upper, int *len)
{
    const long long pr = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (pr >= n) return;
    const int r = row_of[pr];
    len[pr] = upper ? rp[r + 1] - diag[r] - 1 : diag[r] - rp[r];
}

// rows sorted by (level, row): *row_of_out (n ints) and *level_ptr_dev_out (levels + 1 ints) are the caller's afterwards;
// level_ptr: the same table on the host.  n > 0.
static int sort_rows_by_level(hipStream_t st, int n, const int *d_lev, int **row_of_out, std::vector<int> &level_ptr, int **level_ptr_dev_out)
{
    *row_of_out = nullptr;
    *level_ptr_dev_out = nullptr;
    int *kb[2] = {nullptr, nullptr}, *rb[2] = {nullptr, nullptr}, *cnt = nullptr, *tot = nullptr, *maxv = nullptr, *lp = nullptr;
    int rc = CUDAMAT_OK;
    do {
        if ((rc = dalloc(&maxv, 1))) break;
        if (hipMemsetAsync(maxv, 0, sizeof(int), st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        const unsigned gmax = (unsigned)std::min<long long>(((long long)n + kBlock - 1) / kBlock, 4096);
        hipLaunchKernelGGL(k_max_int, dim3(gmax), dim3(kBlock), 0, st, n, d_lev, maxv);
        int maxlev = 0;
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&maxlev, maxv, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("level maximum failed"); break; }
        const int nlev = maxlev + 1;
        const int passes = nlev <= (1 << 8) ? 1 : nlev <= (1 << 16) ? 2 : nlev <= (1 << 24) ? 3 : 4;
        const int nchunks = (int)(((long long)n + kLsChunk - 1) / kLsChunk);
        const unsigned grid = (unsigned)((nchunks + kLsWaves - 1) / kLsWaves);
        if ((rc = dalloc(&cnt, (size_t)256 * nchunks))) break;
        if ((rc = dalloc(&tot, 512))) break;                    // totals, then bases
        for (int q = 0; q < (passes > 1 ? 2 : 1) && !rc; q++) {
            rc = dalloc(&kb[q], (size_t)n);
            if (!rc) rc = dalloc(&rb[q], (size_t)n);
        }
        if (rc) break;
        const int *key_in = d_lev, *row_in = nullptr;
        int last = 0;
        for (int p = 0; p < passes; p++) {
            const int o = p & 1;
            hipLaunchKernelGGL(k_lsort_hist, dim3(grid), dim3(kBlock), 0, st, n, key_in, 8 * p, nchunks, cnt);
            hipLaunchKernelGGL(k_lsort_scan_digit, dim3(256), dim3(kBlock), 0, st, nchunks, cnt, tot);
            hipLaunchKernelGGL(k_lsort_scan_tot, dim3(1), dim3(kBlock), 0, st, tot, tot + 256);
            hipLaunchKernelGGL(k_lsort_scatter, dim3(grid), dim3(kBlock), 0, st, n, key_in, row_in, 8 * p, nchunks, cnt, tot + 256, kb[o], rb[o]);
            key_in = kb[o];
            row_in = rb[o];
            last = o;
        }
        if (hipGetLastError() != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("level sort launch failed"); break; }
        if ((rc = dalloc(&lp, (size_t)nlev + 1))) break;
        hipLaunchKernelGGL(k_level_ptr, dim3((unsigned)(((long long)n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, n, kb[last], lp);
        level_ptr.assign((size_t)nlev + 1, 0);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(level_ptr.data(), lp, sizeof(int) * ((size_t)nlev + 1), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("level sort failed"); break; }
        if (level_ptr[0] != 0 || level_ptr[(size_t)nlev] != n) { rc = CUDAMAT_ERR_HIP; set_error("level sort: inconsistent level table"); break; }
        *row_of_out = rb[last];
        rb[last] = nullptr;
        *level_ptr_dev_out = lp;
        lp = nullptr;
    } while (0);
    void *tmp[] = {kb[0], kb[1], rb[0], rb[1], cnt, tot, maxv, lp};
    for (void *q : tmp) afree(q);
    return rc;
}

// levels -> level-major permutation (stable: rows of a level stay in increasing order)
static int build_levels(cudamat_solver *s, bool upper, int *d_lev, int *d_flags, TriFactor &F, TriHost &H, int *err_host, int *err_dev)
{
    hipStream_t st = s->ctx->stream;
    const Config &cfg = s->ctx->cfg;
    const int n = s->n;
    double t_stamp = now_s();
    const long long threads = (long long)n * 8;
    const unsigned grid = (unsigned)((threads + kBlock - 1) / kBlock);
    // one dependency-driven pass; the relaxation sweeps below remain as the fallback (option LEVELS_SWEEP = 1
    // or a timed-out wait)
    bool have_levels = false;
    if (n > 0 && err_host && !s->ctx->cfg.levels_sweep) {
        CM_HIP(hipMemsetAsync(d_lev, 0xFF, sizeof(int) * (size_t)n, st));
        // lanes per row from the mean number of dependencies: with 2 per row (stencils) one lane per row keeps 8x
        // more rows in flight, and rows in flight are what a chain-like dependency graph needs (unlike the
        // triangular solves, fewer resident workgroups only slow this pass down: Poisson 0.9 / 2.2 / 4.1 s at 8 / 2 / 1)
        const double mean_dep = (double)(s->pm_nnz - n) / 2.0 / n;
        const int L = mean_dep <= 3.0 ? 1 : mean_dep <= 6.0 ? 2 : 8;
        const long long nsub = ((long long)n * L + kBlock - 1) / kBlock;
        int steps = 4;                                           // a ticket = 4 x 256 threads' worth of rows
        while (steps > 1 && nsub / steps < 1024) steps >>= 1;
        const long long ntick = (nsub + steps - 1) / steps;
        const unsigned gridL = (unsigned)(ntick < 2048 ? ntick : 2048);       // 8 resident workgroups per CU
        unsigned *d_ticket = (unsigned *)(d_flags + 2);
        CM_HIP(hipMemsetAsync(d_ticket, 0, sizeof(unsigned), st));
        if (L == 1)
            hipLaunchKernelGGL(k_levels_dep<1>, dim3(gridL), dim3(kBlock), 0, st, n, s->pm_rp, s->pm_ci, s->diag_pos, upper ? 1 : 0, d_lev, err_dev, d_ticket, steps);
        else if (L == 2)
            hipLaunchKernelGGL(k_levels_dep<2>, dim3(gridL), dim3(kBlock), 0, st, n, s->pm_rp, s->pm_ci, s->diag_pos, upper ? 1 : 0, d_lev, err_dev, d_ticket, steps);
        else
            hipLaunchKernelGGL(k_levels_dep<8>, dim3(gridL), dim3(kBlock), 0, st, n, s->pm_rp, s->pm_ci, s->diag_pos, upper ? 1 : 0, d_lev, err_dev, d_ticket, steps);
        CM_HIP(hipGetLastError());
        CM_HIP(hipStreamSynchronize(st));
        have_levels = *err_host == 0;
        *err_host = 0;
    }
    if (!have_levels) CM_HIP(hipMemsetAsync(d_lev, 0, sizeof(int) * (size_t)(n ? n : 1), st));
    int sweeps = 0;
    while (n > 0 && !have_levels) {
        CM_HIP(hipMemsetAsync(d_flags, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_level_sweep, dim3(grid ? grid : 1), dim3(kBlock), 0, st, n, s->pm_rp, s->pm_ci,
                           s->diag_pos, upper ? 1 : 0, d_lev, d_flags);
        CM_HIP(hipGetLastError());
        int changed = 0;
        CM_HIP(hipMemcpyAsync(&changed, d_flags, sizeof(int), hipMemcpyDeviceToHost, st));
        CM_HIP(hipStreamSynchronize(st));
        sweeps++;
        if (!changed) break;
        if (sweeps > n + 1) { set_error("level analysis did not converge"); return CUDAMAT_ERR_HIP; }
    }
    CM_STAMP(upper ? "U levels (device)" : "L levels (device)");
    // the level of every original row stays on the device until the far / near split (split_factor); d_lev is the
    // caller's scratch and serves the other factor next
    CM_TRY(dalloc(&H.lev_dev, (size_t)n));
    if (n) CM_HIP(hipMemcpyAsync(H.lev_dev, d_lev, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, st));
    // stable sort of the rows by level, the level table, the factor's row pointers in that order: all on the device
    if (n > 0) {
        CM_TRY(sort_rows_by_level(st, n, H.lev_dev, &F.row_of, F.level_ptr, &H.level_ptr_dev));
        F.nlevels = (int)F.level_ptr.size() - 1;
    } else {
        F.nlevels = 0;
        F.level_ptr.assign(1, 0);
        CM_TRY(dalloc(&F.row_of, 0));
        CM_TRY(dalloc(&H.level_ptr_dev, 1));
        CM_HIP(hipMemsetAsync(H.level_ptr_dev, 0, sizeof(int), st));
    }
    const int nlev = F.nlevels;
    CM_STAMP("level sort (device)");
    CM_TRY(dalloc(&F.rp, (size_t)n + 1));
    {
        int *d_len = nullptr;
        CM_TRY(dalloc(&d_len, (size_t)n));
        if (n) hipLaunchKernelGGL(k_factor_len, dim3((unsigned)(((long long)n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, n, F.row_of, s->pm_rp,
                                  s->diag_pos, upper ? 1 : 0, d_len);
        int rcs = hipGetLastError() == hipSuccess ? device_exclusive_scan(st, n, d_len, F.rp) : CUDAMAT_ERR_HIP;
        int total = 0;
        if (!rcs && hipMemcpy(&total, F.rp + n, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rcs = CUDAMAT_ERR_HIP;
        afree(d_len);
        if (rcs) { set_error("factor row pointers failed"); return rcs; }
        F.nnz = total;
    }
    // (F.ci / F.val / F.dinv -- 3 GB per factor at C5 -- are allocated by fill_factor: this function may run beside an upload,
    // which multi-GB allocations slow down)
    F.rhs_of = F.out_of = F.row_of;            // original index space (until both factors go level-major, ilu0_setup)
    F.lm = false;
    CM_STAMP("factor row pointers + arrays");
    H.lanes = pick_lanes(n ? (double)F.nnz / n : 1.0);
    if (cfg.trsv_lanes) H.lanes = cfg.trsv_lanes;
    // the hybrid solve is for big factors with many wide levels whose columns are scattered (the gather-bound case);
    // the two factors decide together (ilu0_setup): they share the level-major index spaces
    H.want_hybrid = nlev >= 4 && (cfg.trsv_hybrid >= 0 ? cfg.trsv_hybrid == 1 : (n >= 500000 && F.nnz >= (8 << 20) && nlev >= 16 && n / nlev >= 16384));
    return CUDAMAT_OK;
}

// groups of consecutive levels (hybrid: ~17 levels per group leaves ~15 % of the entries near; measured at C5, round 3:
// 5 / 8 / 12 / 16 / 24 groups -> 4.85 / 4.79 / 5.03 / 5.54 / 6.57 ms per L^-1 U^-1 -- the near launches shrink with more
// groups, the far SpMVs lose more on their shorter segments) and the launch plan
static void plan_groups(const Config &cfg, const TriFactor &F, TriHost &H, bool hybrid)
{
    const int nlev = F.nlevels;
    int K = 1;
    if (hybrid) {
        K = nlev / 17;
        if (K < 2) K = 2;
        if (K > 16) K = 16;
        if (cfg.trsv_groups >= 2 && cfg.trsv_groups <= nlev) K = cfg.trsv_groups;
    }
    H.hybrid = K > 1;
    H.grp_level.assign((size_t)K + 1, 0);
    for (int g = 0; g <= K; g++) H.grp_level[(size_t)g] = (int)((long long)nlev * g / K);
    if (cfg.verbose && K > 1) {
        fprintf(stderr, "cudamat: trsv groups (levels:rows)");
        for (int g = 0; g < K; g++)
            fprintf(stderr, " %d:%d", H.grp_level[(size_t)g + 1] - H.grp_level[(size_t)g],
                    F.level_ptr[(size_t)H.grp_level[(size_t)g + 1]] - F.level_ptr[(size_t)H.grp_level[(size_t)g]]);
        fprintf(stderr, "\n");
    }
    // launch plan: a big level is its own segment; consecutive small levels are merged (inside a group)
    H.seg_begin.clear();
    H.seg_end.clear();
    H.seg_group.clear();
    for (int g = 0; g < K; g++) {
        int l = H.grp_level[(size_t)g];
        const int lend = H.grp_level[(size_t)g + 1];
        while (l < lend) {
            const int rows = F.level_ptr[(size_t)l + 1] - F.level_ptr[(size_t)l];
            int e = l + 1;
            if (rows <= kSmallLevel)
                while (e < lend && F.level_ptr[(size_t)e + 1] - F.level_ptr[(size_t)e] <= kSmallLevel) e++;
            H.seg_begin.push_back(l);
            H.seg_end.push_back(e);
            H.seg_group.push_back(g);
            l = e;
        }
    }
}

// ---- hybrid split of a level-major factor into near (same group) and far (earlier groups) entries
// Rows are in level-major order, a group is a range of positions: the group of a row or of a (relabelled) column follows
// from its position by comparison with the group boundaries (<= 17 by default) -- no per-row group table, no gather for it.
struct GroupCuts {
    int K;
    int start[130];         // start[g] = first level-major position of group g; start[K] = n  (TRSV_GROUPS <= 128)
};

__device__ __forceinline__ void group_range(const GroupCuts &gc, int pr, int *gs, int *ge)
{
    int g = 0;
#pragma unroll 1
    for (int q = 1; q < gc.K; q++) g += gc.start[q] <= pr ? 1 : 0;
    *gs = gc.start[g];
    *ge = gc.start[g + 1];
}

// Pass 1, one wavefront per row (coalesced reads of the row's columns): relabel the columns to level-major positions
// -- the ONE random gather per entry of the whole split, written out so that pass 2 streams -- and count the row's near
// entries (position inside the row's own group) and far ones.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_split_count(int n, const int *frp, const int *fci, const int *pos, GroupCuts gc,
                                                           int *ci_new, int *cnt_near, int *cnt_far)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long pr = (long long)blockIdx.x * WAVES + wave;
    if (pr >= n) return;
    int gs, ge;
    group_range(gc, (int)pr, &gs, &ge);
    const int s0 = frp[pr], len = frp[pr + 1] - s0;
    int nn = 0;
    for (int k = lane; k < len; k += 64) {
        const int p = pos[fci[s0 + k]];
        ci_new[s0 + k] = p;
        nn += (p >= gs && p < ge) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nn += __shfl_xor(nn, o, 64);
    if (lane == 0) {
        cnt_near[pr] = nn;
        cnt_far[pr] = len - nn;
    }
}

// Pass 2, one wavefront per row: near entries keep their order (a row is still summed in its original column order:
// ballot-ranked compaction), far entries go straight to their place in increasing order of the new column (the blocked
// builder needs increasing columns): the row's far columns are staged in LDS and every far entry counts the smaller
// ones -- columns are distinct; <= kSortRowMax entries per row (the hybrid form is not taken otherwise).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_split_fill(int n, const int *frp, const int *ci_new, const double *fval, GroupCuts gc,
                                                          const int *nrp, int *nci, double *nval, const int *qrp, int *qci,
                                                          double *qval)
{
    __shared__ int key[WAVES][kSortRowMax];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long pr = (long long)blockIdx.x * WAVES + wave;
    if (pr >= n) return;
    int gs, ge;
    group_range(gc, (int)pr, &gs, &ge);
    const int s0 = frp[pr], len = frp[pr + 1] - s0;
    int *kk = key[wave];
    for (int k = lane; k < len; k += 64) {
        const int p = ci_new[s0 + k];
        kk[k] = (p >= gs && p < ge) ? 0x7fffffff : p;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int on = nrp[pr], of = qrp[pr];
    int run_near = 0;
    for (int k0 = 0; k0 < len; k0 += 64) {
        const int k = k0 + lane;
        const bool active = k < len;
        const int p = active ? ci_new[s0 + k] : 0;
        const bool near = active && p >= gs && p < ge;
        const unsigned long long m = __ballot(near);
        if (near) {
            const int dst = on + run_near + __popcll(m & ((1ULL << lane) - 1ULL));
            nci[dst] = p;
            nval[dst] = fval[s0 + k];
        }
        run_near += __popcll(m);
        if (active && !near) {
            int rank = 0;
            for (int j = 0; j < len; j++) rank += kk[j] < p ? 1 : 0;
            qci[of + rank] = p;
            qval[of + rank] = fval[s0 + k];
        }
    }
}

// pos[row_of[pr]] = pr
__global__ __launch_bounds__(kBlock) void k_invert_perm(int n, const int *row_of, int *pos)
{
    const int pr = blockIdx.x * kBlock + threadIdx.x;
    if (pr < n) pos[row_of[pr]] = pr;
}

// map[pr] = pos[row_of[pr]]   (position in one ordering of the row at position pr of another)
__global__ __launch_bounds__(kBlock) void k_compose_perm(int n, const int *row_of, const int *pos, int *map)
{
    const int pr = blockIdx.x * kBlock + threadIdx.x;
    if (pr < n) map[pr] = pos[row_of[pr]];
}

// Rows re-ordered by a new column numbering: destination row pr takes the entries of source row src_of[pr] (nullptr: pr),
// columns relabelled through colmap (nullptr: kept), sorted by the new column.  One wavefront per row; a row's new
// columns are staged in LDS (<= kSortRowMax entries) and every entry finds its rank by counting the smaller ones
// (columns are distinct) -- O(len^2 / 64) wave steps, ~50 for the 50-entry rows this is for.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_sort_rows(int nrows, const int *src_rp, const int *src_of, const int *dst_rp,
                                                         const int *ci, const double *val, const int *colmap, int *out_ci,
                                                         double *out_val)
{
    __shared__ int cols[WAVES][kSortRowMax];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long pr = (long long)blockIdx.x * WAVES + wave;
    if (pr >= nrows) return;
    const int r = src_of ? src_of[pr] : (int)pr;
    const int s0 = src_rp[r], len = src_rp[r + 1] - s0, d0 = dst_rp[pr];
    int *c = cols[wave];
    for (int k = lane; k < len; k += 64) {
        const int col = ci[s0 + k];
        c[k] = colmap ? colmap[col] : col;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < len; k += 64) {
        const int mine = c[k];
        int rank = 0;
        for (int j = 0; j < len; j++) rank += c[j] < mine ? 1 : 0;
        out_ci[d0 + rank] = mine;
        out_val[d0 + rank] = val[s0 + k];
    }
}

int launch_sort_rows(hipStream_t st, int nrows, const int *src_rp, const int *src_of, const int *dst_rp, const int *ci,
                     const double *val, const int *colmap, int *out_ci, double *out_val)
{
    if (nrows <= 0) return CUDAMAT_OK;
    hipLaunchKernelGGL(k_sort_rows<4>, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, st, nrows, src_rp, src_of, dst_rp, ci, val,
                       colmap, out_ci, out_val);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

static int split_factor(cudamat_solver *s, TriFactor &F, TriHost &H, const int *pos)
{
    if (!H.hybrid) return CUDAMAT_OK;
    hipStream_t st = s->ctx->stream;
    const int n = s->n;
    const int K = (int)H.grp_level.size() - 1;
    double t_stamp = now_s();
    // the groups as ranges of level-major positions (rows are stored level by level)
    GroupCuts gc{};
    if (K > 128) { set_error("factor split: more than 128 groups"); return CUDAMAT_ERR_ARG; }
    gc.K = K;
    for (int g = 0; g <= K; g++) gc.start[g] = F.level_ptr[(size_t)H.grp_level[(size_t)g]];
    int *d_cn = nullptr, *d_cf = nullptr, *nrp = nullptr, *qrp = nullptr, *nci = nullptr, *qci = nullptr, *ci_new = nullptr;
    double *nval = nullptr, *qval = nullptr;
    int rc = CUDAMAT_OK;
    do {
        if ((rc = dalloc(&d_cn, (size_t)n))) break;
        if ((rc = dalloc(&d_cf, (size_t)n))) break;
        if ((rc = dalloc(&ci_new, (size_t)(F.nnz > 0 ? F.nnz : 1)))) break;
        constexpr int kSplitWaves = 4;
        const unsigned grid = (unsigned)((n + kSplitWaves - 1) / kSplitWaves);
        hipLaunchKernelGGL(k_split_count<kSplitWaves>, dim3(grid), dim3(64 * kSplitWaves), 0, st, n, F.rp, F.ci, pos, gc, ci_new, d_cn, d_cf);
        // row pointers of the near and the far part: prefix sums on the device; the host needs the two totals and the far
        // pointer at the group boundaries only
        if ((rc = dalloc(&nrp, (size_t)n + 1))) break;
        if ((rc = dalloc(&qrp, (size_t)n + 1))) break;
        if ((rc = device_exclusive_scan(st, n, d_cn, nrp))) break;
        if ((rc = device_exclusive_scan(st, n, d_cf, qrp))) break;
        int tot_near = 0, tot_far = 0;
        std::vector<int> hf_grp((size_t)K + 1, 0);
        bool copied = hipMemcpy(&tot_near, nrp + n, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess &&
                      hipMemcpy(&tot_far, qrp + n, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
        for (int g = 0; g <= K && copied; g++)
            copied = hipMemcpy(&hf_grp[(size_t)g], qrp + F.level_ptr[(size_t)H.grp_level[(size_t)g]], sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
        if (!copied) { rc = CUDAMAT_ERR_HIP; set_error("factor split scan failed"); break; }
        const int64_t nnz_near = tot_near, nnz_far = tot_far;
        CM_STAMP("split count + device scan");
        if ((rc = dalloc(&nci, (size_t)nnz_near))) break;
        if ((rc = dalloc(&nval, (size_t)nnz_near))) break;
        if ((rc = dalloc(&qci, (size_t)nnz_far))) break;
        if ((rc = dalloc(&qval, (size_t)nnz_far))) break;
        // (far rows come out in increasing level-major column order, as the blocked builder needs them)
        hipLaunchKernelGGL(k_split_fill<kSplitWaves>, dim3(grid), dim3(64 * kSplitWaves), 0, st, n, F.rp, ci_new, F.val, gc, nrp, nci,
                           nval, qrp, qci, qval);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("factor split failed"); break; }
        CM_STAMP("split fill (far rows sorted)");
        // one blocked SpMV plan per group: rows of the group (level-major, contiguous) x the columns of the EARLIER
        // groups -- in level-major space those are the positions [0, r0), so only that prefix of `out` is tiled
        H.far.assign((size_t)K, PbPlan());
        for (int g = 1; g < K && !rc; g++) {
            const int r0 = F.level_ptr[(size_t)H.grp_level[(size_t)g]], r1 = F.level_ptr[(size_t)H.grp_level[(size_t)g + 1]];
            const int64_t cnt = (int64_t)hf_grp[(size_t)g + 1] - hf_grp[(size_t)g];
            if (r1 <= r0 || cnt <= 0) continue;
            rc = pb_build(st, s->ctx->cfg, r1 - r0, r0, cnt, qrp + r0, qci, qval, &H.far[(size_t)g]);
        }
        CM_STAMP("far plans (pb_build)");
        if (rc) break;
        if ((rc = dalloc(&H.far_buf, (size_t)n))) break;
        if (hipMemsetAsync(H.far_buf, 0, sizeof(double) * (size_t)n, st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        // the level kernels keep only the near entries
        CM_DROP(hipFree(F.rp)); CM_DROP(hipFree(F.ci)); CM_DROP(hipFree(F.val));
        F.rp = nrp; F.ci = nci; F.val = nval;
        F.nnz = nnz_near;
        nrp = nullptr; nci = nullptr; nval = nullptr;
        H.lanes = pick_lanes(n ? (double)F.nnz / n : 1.0);
    } while (0);
    void *tmp[] = {d_cn, d_cf, nrp, qrp, nci, qci, nval, qval, ci_new};
    for (void *q : tmp)
        if (q) CM_DROP(hipFree(q));
    free_levels(H);
    if (rc) {   // e.g. not enough memory for the blocked copies: keep the pure level solve (F is intact)
        for (PbPlan &pl : H.far) pb_free(&pl);
        H.far.clear();
        if (H.far_buf) { CM_DROP(hipFree(H.far_buf)); H.far_buf = nullptr; }
        H.hybrid = false;
        return rc == CUDAMAT_ERR_NOMEM ? CUDAMAT_OK : rc;
    }
    return CUDAMAT_OK;
}

static int fill_factor(cudamat_solver *s, bool upper, TriFactor &F)
{
    const int n = s->n;
    if (!F.ci) CM_TRY(dalloc(&F.ci, (size_t)F.nnz));
    if (!F.val) CM_TRY(dalloc(&F.val, (size_t)F.nnz));
    if (upper && !F.dinv) CM_TRY(dalloc(&F.dinv, (size_t)n));
    if (!n) return CUDAMAT_OK;
    const long long threads = (long long)n * 8;
    const unsigned grid = (unsigned)((threads + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_fill_factor, dim3(grid), dim3(kBlock), 0, s->ctx->stream, n, s->pm_rp, s->pm_ci, s->lu,
                       s->diag_pos, upper ? 1 : 0, F.row_of, F.rp, F.ci, F.val, F.dinv);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

// ---- block-Jacobi: the rank's diagonal block (columns [c0, c0 + n) of its rows) with local column ids
__device__ __forceinline__ int lower_bound_col(const int *ci, int lo, int hi, long long key)
{
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ci[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kBlock) void k_block_count(int n, const int *rp, const int *ci, long long c0, int *first,
                                                        int *cnt)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int a = lower_bound_col(ci, rp[i], rp[i + 1], c0);
    const int b = lower_bound_col(ci, a, rp[i + 1], c0 + n);
    first[i] = a;
    cnt[i] = b - a;
}

__global__ __launch_bounds__(kBlock) void k_block_fill(int n, const int *ci, const double *val, long long c0,
                                                       const int *first, const int *brp, int *bci, double *bval)
{
    constexpr int L = 8;
    const long long t = ((long long)blockIdx.x * kBlock + threadIdx.x) / L;
    if (t >= n) return;
    const int i = (int)t, lane = threadIdx.x & (L - 1);
    const int src = first[i], dst = brp[i], cnt = brp[i + 1] - dst;
    for (int k = lane; k < cnt; k += L) {
        bci[dst + k] = (int)(ci[src + k] - c0);
        bval[dst + k] = val[src + k];
    }
}

static int select_precond_matrix(cudamat_solver *s)
{
    if (!s->sharded) {
        s->pm_rp = s->rp;
        s->pm_ci = s->ci;
        s->pm_val = s->val;
        s->pm_nnz = s->nnz;
        s->pm_owned = false;
        return CUDAMAT_OK;
    }
    hipStream_t st = s->ctx->stream;
    const int n = s->n;
    const long long c0 = (long long)s->comm.rank * s->n_pad;
    int *first = nullptr, *cnt = nullptr;
    int rc = CUDAMAT_OK;
    s->pm_owned = true;
    do {
        if ((rc = dalloc(&first, (size_t)n))) break;
        if ((rc = dalloc(&cnt, (size_t)n))) break;
        const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
        if (n) hipLaunchKernelGGL(k_block_count, dim3(grid), dim3(kBlock), 0, st, n, s->rp, s->ci, c0, first, cnt);
        std::vector<int> h((size_t)n), brp((size_t)n + 1, 0);
        if (n && (hipMemcpyAsync(h.data(), cnt, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess ||
                  hipStreamSynchronize(st) != hipSuccess)) { rc = CUDAMAT_ERR_HIP; set_error("block count failed"); break; }
        for (int i = 0; i < n; i++) brp[(size_t)i + 1] = brp[(size_t)i] + h[(size_t)i];
        s->pm_nnz = brp[(size_t)n];
        if ((rc = dalloc(&s->pm_rp, (size_t)n + 1))) break;
        if ((rc = dalloc(&s->pm_ci, (size_t)s->pm_nnz))) break;
        if ((rc = dalloc(&s->pm_val, (size_t)s->pm_nnz))) break;
        if (hipMemcpy(s->pm_rp, brp.data(), sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice) != hipSuccess) {
            rc = CUDAMAT_ERR_HIP; set_error("block row pointers upload failed"); break;
        }
        const unsigned grid8 = (unsigned)(((long long)n * 8 + kBlock - 1) / kBlock);
        if (n) hipLaunchKernelGGL(k_block_fill, dim3(grid8), dim3(kBlock), 0, st, n, s->ci, s->val, c0, first, s->pm_rp,
                                  s->pm_ci, s->pm_val);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            rc = CUDAMAT_ERR_HIP; set_error("block extraction failed"); break;
        }
    } while (0);
    if (first) CM_DROP(hipFree(first));
    if (cnt) CM_DROP(hipFree(cnt));
    return rc;       // on failure ilu0_setup's error path releases the partial copy
}

// The PATTERN-ONLY part of the set-up (pbicgstab.cu:336-347: the two csrsv analyses; plus the diagonal positions and the
// longest row): needs rp / ci only, so the drop-in entry point runs it while the values are still being uploaded
// (ilu0_analyse_early); ilu0_setup runs it itself when nobody has.  Leaves its scratch (flags, levels) in the plans.
static int ilu0_analysis(cudamat_solver *s, bool block)
{
    ilu0_release(s);
    if (int rc0 = select_precond_matrix(s)) {
        char saved[512];
        snprintf(saved, sizeof(saved), "%s", cudamat_last_error());
        ilu0_release(s);
        set_error("%s", saved);
        return rc0;
    }
    s->ilu_block = block;
    hipStream_t st = s->ctx->stream;
    const int n = s->n;
    IluPlans *pl = plans_of(s, true);
    int *&d_flags = pl->d_flags, *&d_lev = pl->d_lev;
    int rc = CUDAMAT_OK;
    const double t0 = now_s();
    double t_stamp = t0;
    do {
        if ((rc = dalloc(&d_flags, 5))) break;     // [0..1] flags, [2] ticket counter of k_levels_dep, [3] longest row
        if ((rc = dalloc(&d_lev, (size_t)n))) break;
        if (hipHostMalloc((void **)&pl->err_host, sizeof(int), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&pl->err_dev, pl->err_host, 0) != hipSuccess) {
            rc = CUDAMAT_ERR_HIP; set_error("pinned status word allocation failed"); break;
        }
        *pl->err_host = 0;
        if ((rc = dalloc(&s->diag_pos, (size_t)n))) break;
        if (hipMemsetAsync(d_flags, 0, 2 * sizeof(int), st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        if (n) {
            hipLaunchKernelGGL(k_find_diag, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, s->pm_rp, s->pm_ci,
                               s->diag_pos, d_flags);
        }
        int hflags[2] = {0, 0};
        if (hipMemcpyAsync(hflags, d_flags, sizeof(hflags), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("find_diag failed"); break; }
        if (hflags[1]) {
            set_error("ILU(0): row %d has no diagonal entry (pbicgstab.h:118 requires A[i,i] != 0)", hflags[1] - 1);
            rc = CUDAMAT_ERR_ZERO_PIVOT;
            break;
        }
        CM_STAMP("find diag");
        // ---- analysis (pbicgstab.cu:336-347)
        if ((rc = build_levels(s, false, d_lev, d_flags, s->L, pl->L, pl->err_host, pl->err_dev))) break;
        s->t_analysis_l = now_s() - t0;
        const double tu = now_s();
        if ((rc = build_levels(s, true, d_lev, d_flags, s->U, pl->U, pl->err_host, pl->err_dev))) break;
        int maxrow_all = 0;
        if (n) {           // the longest row of the pattern (LDS staging of the factorisation, sortable far rows)
            if (hipMemsetAsync(d_flags + 3, 0, sizeof(int), st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
            hipLaunchKernelGGL(k_max_row_len, dim3((unsigned)std::min<long long>(((long long)n + kBlock - 1) / kBlock, 4096)), dim3(kBlock), 0, st,
                               n, s->pm_rp, d_flags + 3);
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&maxrow_all, d_flags + 3, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("row length maximum failed"); break; }
        }
        {
            // the two factors take the hybrid solve TOGETHER: they then share the level-major index spaces (L's output
            // feeds U's right-hand side, U's output the permuted matrix of the preconditioned loop)
            const bool hybrid = pl->L.want_hybrid && pl->U.want_hybrid && maxrow_all <= kSortRowMax;
            plan_groups(s->ctx->cfg, s->L, pl->L, hybrid);
            plan_groups(s->ctx->cfg, s->U, pl->U, hybrid);
        }
        s->t_analysis_u = now_s() - tu;
        s->t_analysis = now_s() - t0;
        pl->maxrow_all = maxrow_all;
        pl->analysed = true;
    } while (0);
    if (rc) {
        char saved[512];
        snprintf(saved, sizeof(saved), "%s", cudamat_last_error());
        ilu0_release(s);
        set_error("%s", saved);
    }
    return rc;
}

int ilu0_analyse_early(cudamat_solver *s)
{
    if (s->sharded || !s->cols_sorted) return CUDAMAT_OK;      // (ilu0_setup will say what is wrong, or run the block variant)
    CM_HIP(hipSetDevice(s->ctx->device));
    Range range_ilu("cudamat: ILU(0) level analysis (beside the upload)");
    IluPlans *pl = plans_of(s, true);
    std::vector<void *> keep;
    t_deferred = &keep;
    const int rc = ilu0_analysis(s, false);                     // (releases and re-creates the plans: `pl` is stale after this)
    t_deferred = nullptr;
    (void)pl;
    if (IluPlans *p2 = plans_of(s, false)) p2->deferred.insert(p2->deferred.end(), keep.begin(), keep.end());
    else for (void *q : keep) CM_DROP(hipFree(q));
    return rc;
}

// after ilu0_analyse_early: will both factors take the hybrid solve, i.e. will the preconditioned reference loop run in the
// level-major index spaces on the permuted copy of the matrix (loops.hip)?
bool ilu0_will_use_level_major(cudamat_solver *s)
{
    IluPlans *pl = plans_of(s, false);
    return pl && pl->analysed && pl->L.hybrid && pl->U.hybrid;
}

// the temporaries of an analysis that ran beside an upload: free them now (the upload is over)
void ilu0_flush_deferred(cudamat_solver *s)
{
    if (IluPlans *pl = plans_of(s, false)) {
        for (void *q : pl->deferred) CM_DROP(hipFree(q));
        pl->deferred.clear();
    }
}

int ilu0_setup(cudamat_solver *s, bool block)
{
    CM_ARG(block || !s->sharded, "ILU(0) of the whole matrix is single-GPU only (use the block variant)");
    CM_ARG(s->cols_sorted, "ILU(0) needs every row's column indices strictly increasing (mmio_wrapper.h:123 delivers that)");
    CM_HIP(hipSetDevice(s->ctx->device));
    Range range_ilu("cudamat: ILU(0) analysis + factorisation + factor layout");
    {
        IluPlans *p0 = plans_of(s, false);
        const bool have = p0 && p0->analysed && !s->has_ilu && s->ilu_block == block;     // the drop-in call ran it beside its upload
        if (!have) CM_TRY(ilu0_analysis(s, block));
    }
    hipStream_t st = s->ctx->stream;
    const int n = s->n;
    IluPlans *pl = plans_of(s, true);
    pl->analysed = false;                  // (consumed: a second ilu0_setup on this solver starts over)
    int *&d_flags = pl->d_flags, *&d_lev = pl->d_lev;
    const int maxrow_all = pl->maxrow_all;
    int hflags[2] = {0, 0};
    int rc = CUDAMAT_OK;
    double t_stamp = now_s();
    do {
        // ---- factorisation on a copy of A's values (pbicgstab.cu:316, :356-363)
        const double t1 = now_s();
        t_stamp = t1;
        if ((rc = dalloc(&s->lu, (size_t)s->pm_nnz))) break;
        if (hipMemcpyAsync(s->lu, s->pm_val, sizeof(double) * (size_t)s->pm_nnz, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            rc = CUDAMAT_ERR_HIP; set_error("copy of A values failed"); break;
        }
        const int maxrow = maxrow_all;
        const int cap = ((maxrow + 63) / 64) * 64 + 64;
        const bool one_wave = cap > 2048;
        const bool fast = cap <= 1024 && !s->ctx->cfg.ilu0_simple;
        if ((size_t)cap * sizeof(double) > 150 * 1024) {
            set_error("ILU(0): a row with %d entries exceeds the %d-entry LDS staging limit", maxrow, 150 * 1024 / 8);
            rc = CUDAMAT_ERR_ARG;
            break;
        }
        if (hipMemsetAsync(d_flags, 0, 2 * sizeof(int), st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        if ((rc = set_max_lds((const void *)k_ilu0_level<1>))) break;
        if ((rc = set_max_lds((const void *)k_ilu0_level<4>))) break;
        if ((rc = set_max_lds((const void *)k_ilu0_level_fast<4>))) break;
        for (int l = 0; l < s->L.nlevels; l++) {
            const int r0 = s->L.level_ptr[(size_t)l], r1 = s->L.level_ptr[(size_t)l + 1];
            const int rows = r1 - r0;
            if (fast) {     // 28 bytes of LDS per entry slot and wave: values, pivots, columns, two pivot tables
                hipLaunchKernelGGL(k_ilu0_level_fast<4>, dim3((rows + 3) / 4), dim3(256), (size_t)28 * 4 * (size_t)cap, st,
                                   r0, r1, s->L.row_of, s->pm_rp, s->pm_ci, s->diag_pos, s->lu, cap, d_flags);
            } else if (one_wave) {
                hipLaunchKernelGGL(k_ilu0_level<1>, dim3(rows), dim3(64), sizeof(double) * (size_t)cap, st, r0, r1,
                                   s->L.row_of, s->pm_rp, s->pm_ci, s->diag_pos, s->lu, cap, d_flags);
            } else {
                hipLaunchKernelGGL(k_ilu0_level<4>, dim3((rows + 3) / 4), dim3(256), sizeof(double) * 4 * (size_t)cap, st,
                                   r0, r1, s->L.row_of, s->pm_rp, s->pm_ci, s->diag_pos, s->lu, cap, d_flags);
            }
        }
        if (hipGetLastError() != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("ilu0 launch failed"); break; }
        if (hipMemcpyAsync(hflags, d_flags, sizeof(hflags), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("ilu0 failed"); break; }
        if (hflags[1]) {
            set_error("ILU(0): zero pivot in row %d", hflags[1] - 1);
            rc = CUDAMAT_ERR_ZERO_PIVOT;
            break;
        }
        CM_STAMP("numeric ILU(0)");
        if ((rc = fill_factor(s, false, s->L))) break;
        if ((rc = fill_factor(s, true, s->U))) break;
        if (hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("factor fill failed"); break; }
        CM_STAMP("factor fill");
        if (pl->L.hybrid && pl->U.hybrid) {
            // level-major index spaces: position of every original row in L's and in U's order
            int *posL = nullptr;
            const unsigned gp = (unsigned)((n + kBlock - 1) / kBlock);
            if ((rc = dalloc(&posL, (size_t)n))) break;
            if ((rc = dalloc(&pl->posU, (size_t)n))) { CM_DROP(hipFree(posL)); break; }
            hipLaunchKernelGGL(k_invert_perm, dim3(gp), dim3(kBlock), 0, st, n, s->L.row_of, posL);
            hipLaunchKernelGGL(k_invert_perm, dim3(gp), dim3(kBlock), 0, st, n, s->U.row_of, pl->posU);
            rc = split_factor(s, s->L, pl->L, posL);
            if (!rc) rc = split_factor(s, s->U, pl->U, pl->posU);
            int *mapUL = nullptr;
            if (!rc && pl->L.hybrid && pl->U.hybrid) rc = dalloc(&mapUL, (size_t)n);
            if (!rc && mapUL) {
                hipLaunchKernelGGL(k_compose_perm, dim3(gp), dim3(kBlock), 0, st, n, s->U.row_of, posL, mapUL);
                if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("permutation maps failed"); }
            }
            CM_DROP(hipFree(posL));
            if (rc) { if (mapUL) CM_DROP(hipFree(mapUL)); break; }
            if (!(pl->L.hybrid && pl->U.hybrid)) {
                // (split_factor fell back on one side: out of memory for the blocked copies) -- a half-split pair has no
                // consistent index space
                set_error("ILU(0): not enough device memory for the far/near split of the factors");
                rc = CUDAMAT_ERR_NOMEM;
                break;
            }
            s->L.lm = s->U.lm = true;
            s->L.rhs_of = s->L.out_of = nullptr;       // L: right-hand side and solution in L's level-major space
            s->U.rhs_of = mapUL;                       // U: right-hand side in L's space, solution in U's
            s->U.out_of = nullptr;
            if ((rc = dalloc(&pl->perm_a, (size_t)n))) break;
            if ((rc = dalloc(&pl->perm_b, (size_t)n))) break;
        } else {
            free_levels(pl->L);
        }
        free_levels(pl->U);
        s->t_factor = now_s() - t1;
        // solve form: one dependency-driven launch per group (default whenever there is more than one level
        // to chain), or one launch per level / run of small levels (option TRSV_SYNCFREE = 0)
        {
            const Config &cfg = s->ctx->cfg;
            const bool on = cfg.trsv_syncfree != 0;
            // narrow levels (mat10000: <= 100 rows each): the whole factor runs as ONE single-workgroup launch with a
            // barrier per level (0.75 us per level, measured), which beats hand-offs through memory (1.0 us); from a few
            // hundred rows per level on, the dependency-driven form wins (Poisson 4000x2500: 6x)
            auto widest = [](const TriFactor &F) {
                int w = 0;
                for (int l = 0; l < F.nlevels; l++) w = std::max(w, F.level_ptr[(size_t)l + 1] - F.level_ptr[(size_t)l]);
                return w;
            };
            const bool forced = cfg.trsv_syncfree == 1;
            pl->L.syncfree = on && s->L.nlevels > 1 && (forced || widest(s->L) > 512);
            pl->U.syncfree = on && s->U.nlevels > 1 && (forced || widest(s->U) > 512);
            const bool lds_on = cfg.trsv_lds && n > 0 && n <= kLdsTrsvRows;
            pl->L.lds = lds_on && !pl->L.hybrid && widest(s->L) <= 512;
            pl->U.lds = lds_on && !pl->U.hybrid && widest(s->U) <= 512;
            if (cfg.trsv_spin_limit) pl->L.spin_limit = pl->U.spin_limit = cfg.trsv_spin_limit;
            int strict = 0;
            if ((rc = pb_strict_for(s->ctx, &strict))) break;
            pl->L.pb_strict = pl->U.pb_strict = strict;
            // Resident workgroups per CU of the dependency-driven launch.  Every waiting row polls memory, and pollers
            // slow the very hand-offs they wait for: with little work per level the chain of hand-offs is the whole
            // cost and FEWER resident workgroups are faster (Poisson 4000x2500, 5 K entries per level: 27.6 ms per
            // application at 8 per CU, 15.0 ms at 1); with much work per level the gathers need the occupancy
            // (1e6 x 50 random, 190 K entries per level: 1.75 ms at 1, 0.82 ms at 4, 0.89 ms at 8).
            auto pick_occ = [](const TriFactor &F) {
                const double per_level = F.nlevels > 0 ? (double)F.nnz / F.nlevels : 0.0;
                return per_level < 16384.0 ? 1 : per_level < 131072.0 ? 2 : 4;
            };
            pl->L.occ = pick_occ(s->L);
            pl->U.occ = pick_occ(s->U);
            for (TriHost *h : {&pl->L, &pl->U}) {
                const size_t k = h->grp_level.size() > 1 ? h->grp_level.size() - 1 : 1;
                if (!h->tickets && (rc = dalloc(&h->tickets, k))) break;
            }
            if (rc) break;
            // (The early column blocks of a group's far phase 1 on a side stream beside the previous group's near launch were
            // built and measured in round 3, alternating A/B at C5: 4.95 ms per L^-1 U^-1 with the overlap against 4.78
            // without -- the streaming phase 1 and the gathering near launch compete for the same fabric requests, and the
            // near launch loses its one-workgroup-per-CU residency.  Removed in round 4; HISTORY.md.)
        }
        s->has_ilu = true;
    } while (0);
    if (d_flags) CM_DROP(hipFree(d_flags));
    if (d_lev) CM_DROP(hipFree(d_lev));
    d_flags = nullptr;
    d_lev = nullptr;
    if (rc) {
        char saved[512];
        snprintf(saved, sizeof(saved), "%s", cudamat_last_error());
        ilu0_release(s);
        set_error("%s", saved);
    }
    return rc;
}

__global__ __launch_bounds__(kBlock) void k_perm_row_len(int n, const int *rp, const int *row_of, int *len)
{
    const int pr = blockIdx.x * kBlock + threadIdx.x;
    if (pr < n) { const int r = row_of[pr]; len[pr] = rp[r + 1] - rp[r]; }
}

// The solver's matrix for the loop that runs IN the level-major spaces (solver.hip, "permuted loop"): rows in L's
// order (the residual-side vectors r, p, v, t live there), columns in U's positions (the SpMV inputs M^-1 p, M^-1 r and
// the iterate x live there), as a blocked two-phase copy.  Built once, at the first preconditioned solve.
int ilu_perm_matrix(cudamat_solver *s)
{
    if (s->perm_ready) return CUDAMAT_OK;
    IluPlans *pl = plans_of(s, false);
    if (!pl || !s->L.lm || !s->U.lm || !pl->posU) {
        set_error("level-major index spaces are not available");
        return CUDAMAT_ERR_ARG;
    }
    hipStream_t st = s->ctx->stream;
    const int n = s->n;
    const int64_t nnz = s->pm_nnz;
    const double t0 = now_s();
    double t_stamp = t0;
    int *d_rp = nullptr, *d_ci = nullptr, *d_len = nullptr;
    double *d_val = nullptr;
    int rc = CUDAMAT_OK;
    do {
        if ((rc = dalloc(&d_rp, (size_t)n + 1))) break;
        if ((rc = dalloc(&d_len, (size_t)n))) break;
        if ((rc = dalloc(&d_ci, (size_t)nnz))) break;
        if ((rc = dalloc(&d_val, (size_t)nnz))) break;
        CM_STAMP("perm: allocations");
        // row pointers of the permuted matrix: lengths of the rows in L's order, prefix sums on the device
        hipLaunchKernelGGL(k_perm_row_len, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, n, s->pm_rp, s->L.row_of, d_len);
        if ((rc = device_exclusive_scan(st, n, d_len, d_rp))) break;
        if ((rc = launch_sort_rows(st, n, s->pm_rp, s->L.row_of, d_rp, s->pm_ci, s->pm_val, pl->posU, d_ci, d_val))) break;
        CM_STAMP("perm: rows permuted and sorted");
        if (nnz >= (1 << 20)) {
            if ((rc = valdict_build(st, s->ctx->cfg, nnz, d_val, &s->vd_perm))) break;       // (n == 0 afterwards: no dictionary, fp64 values)
        }
        CM_STAMP("perm: value dictionary");
        if ((rc = pb_build(st, s->ctx->cfg, n, n, nnz, d_rp, d_ci, d_val, &s->pb_perm, nullptr, &s->vd_perm))) break;
        CM_STAMP("perm: blocked copy");
        if (!s->x_perm && (rc = dalloc(&s->x_perm, (size_t)(s->n_pad > n ? s->n_pad : n)))) break;
        if (!s->b_perm && (rc = dalloc(&s->b_perm, (size_t)(s->n_pad > n ? s->n_pad : n)))) break;
        if (hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("permuted matrix build failed"); break; }
    } while (0);
    void *tmp[] = {d_rp, d_ci, d_val, d_len};
    for (void *q : tmp)
        if (q) CM_DROP(hipFree(q));
    CM_STAMP("perm: temporaries freed");
    if (rc) {
        pb_free(&s->pb_perm);
        valdict_free(&s->vd_perm);
        return rc;
    }
    s->perm_ready = true;
    CM_DROP(hipFree(pl->posU));
    pl->posU = nullptr;
    s->t_perm_matrix = now_s() - t0;
    s->t_factor += s->t_perm_matrix;       // (cudamat_stats.t_factor: everything between the level analysis and the first iteration)
    if (s->ctx->cfg.verbose) fprintf(stderr, "[cudamat] ilu0 permuted matrix (rows in L order, columns in U positions) %8.3f ms\n", s->t_perm_matrix * 1e3);
    return CUDAMAT_OK;
}

}  // namespace cm

extern "C" int cudamat_solver_ilu0(cudamat_solver *s)
{
    CM_ARG(s, "solver is NULL");
    return ilu0_setup(s, false);
}

extern "C" int cudamat_solver_ilu0_nnz(cudamat_solver *s, int64_t *count)
{
    CM_ARG(s && count, "null pointer");
    CM_ARG(s->has_ilu, "call cudamat_solver_ilu0 first");
    *count = s->pm_nnz;
    return CUDAMAT_OK;
}

extern "C" int cudamat_solver_trsv_form(cudamat_solver *s, int *form)
{
    CM_ARG(s && form, "null pointer");
    CM_ARG(s->has_ilu, "call cudamat_solver_ilu0 first");
    *form = trsv_syncfree_active(s) ? 1 : 0;
    return CUDAMAT_OK;
}

extern "C" int cudamat_solver_block_ilu0(cudamat_solver *s)
{
    CM_ARG(s, "solver is NULL");
    return ilu0_setup(s, true);
}

extern "C" int cudamat_solver_ilu0_values(cudamat_solver *s, double *out_dev)
{
    CM_ARG(s && out_dev, "null pointer");
    CM_ARG(s->has_ilu, "call cudamat_solver_ilu0 first");
    CM_HIP(hipMemcpyAsync(out_dev, s->lu, sizeof(double) * (size_t)s->pm_nnz, hipMemcpyDeviceToDevice, s->ctx->stream));
    return CUDAMAT_OK;
}

