// spmv_pat.hip -- SpMV on a ROW-PATTERN DICTIONARY (round 4), for matrices whose rows repeat a few shapes: stencils
// and other banded operators (the reference's own fixtures mat900 / mat10000 are 9- and 5-point Laplacians; BASELINE
// configs[2] is the 5-point Laplacian on a 4000 x 2500 grid: 9 row shapes for 1e7 rows).
//
// A row's pattern = its length and its column offsets (column - row) in column order.  When the whole matrix shows
// at most 255 distinct patterns of at most 15 entries, a row is stored as ONE BYTE (the pattern id) plus its fp64
// values -- no column indices at all: 8 B per entry + 1 B per row instead of CSR's 12 B per entry + 4 B per row, or
// the compressed stream kernel's 10 B per entry + 1 B per row.  At C3 one launch then moves 0.58 GB instead of 0.77 GB
// for the same 0.80 GB of algorithmic bytes.  Exact: the kernel multiplies the same values by the same x entries.
//
// Layout: chunks of 64 consecutive rows (natural order, nothing is sorted), values slot-major inside a chunk, padded
// to the matrix's longest row W: value j of the chunk's lane-th row sits at (chunk * W + j) * 64 + lane, so the 64
// lanes of a wave load 64 consecutive doubles per slot.  One lane owns one row: the pattern table (<= 16 KB) sits in
// LDS, a lane's column for slot j is row + table[pattern][j] -- no index is fetched from memory, the x gathers depend
// on nothing but the one-byte pattern id -- and the row's products are added by that lane in column order with one
// rounding per product and per sum: the rounding sequence of the reference CPU loop `b[i] += A.Value[j] * x[A.Col[j]]`
// (bicstab_omp/bicstab.cpp:72-77), bit-identical to the oracle on real-valued data.  Padding slots are never
// multiplied.  Workgroups take tiles of 4 chunks dealt cyclically inside an XCD's contiguous share (as the stream
// kernels do), so the uses of an x entry by neighbouring grid rows meet in one L2.
//
// Detection (pat_build): one pass hashes every row's pattern into a 4096-slot table of 64-bit keys (as valdict.hip
// does for values) and remembers the smallest row of each key; the distinct keys are sorted (deterministic ids), the
// table rows are extracted from those representative rows, and a second pass assigns the ids and COMPARES every
// row with its table entry -- a hash collision or a row the table cannot describe makes the builder give up, it never
// produces a wrong copy.  Chosen per matrix by timing it against the other forms (solver.hip, ensure_spmv_mode).
#include <algorithm>
#include <chrono>
#include <vector>

#include "device.h"
#include "spmv_pat.h"

namespace cm {

constexpr int kPatChunk = 64;
constexpr int kPatSlots = 4096;                                  // hash table slots (>= 16 x the patterns it may hold)
constexpr unsigned long long kPatEmpty = 0xFFFFFFFFFFFFFFFFull;
constexpr int kPatRow = kPatMaxLen + 1;                          // ints per table row

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
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

void pat_free(PatPlan *p)
{
    void *ptrs[] = {p->val, p->vidx, p->pid, p->tab};
    for (void *q : ptrs)
        if (q) CM_DROP(hipFree(q));
    *p = PatPlan();
}

__device__ __forceinline__ unsigned long long pat_mix(unsigned long long h, unsigned long long v)
{
    h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    return h;
}

// key of row `row`: its length and column offsets; kPatEmpty is never returned.  len > kPatMaxLen: *too_long = true
__device__ __forceinline__ unsigned long long pat_key(const int *rp, const int *ci, int row, bool *too_long)
{
    const int s = rp[row], e = rp[row + 1];
    *too_long = e - s > kPatMaxLen;
    unsigned long long h = pat_mix(0x5EEDull, (unsigned long long)(unsigned)(e - s));
    if (!*too_long)
        for (int k = s; k < e; k++) h = pat_mix(h, (unsigned long long)(unsigned)(ci[k] - row));
    return h == kPatEmpty ? 0x1234567ull : h;
}

// flags[0] = distinct keys claimed, flags[1] = give up (a row too long, or too many patterns)
__global__ __launch_bounds__(kBlock) void k_pat_probe(int n, const int *rp, const int *ci, unsigned long long *table, int *rep, int *flags)
{
    unsigned long long last = kPatEmpty;
    int last_slot = -1;
    for (long long row = (long long)blockIdx.x * kBlock + threadIdx.x; row < n; row += (long long)gridDim.x * kBlock) {
        if (__hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        bool too_long;
        const unsigned long long key = pat_key(rp, ci, (int)row, &too_long);
        if (too_long) { __hip_atomic_store(&flags[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        int slot = last_slot;
        if (key != last) {
            unsigned h = (unsigned)(key >> 17) & (kPatSlots - 1);
            slot = -1;
            for (int probe = 0; probe < kPatSlots; probe++) {
                unsigned long long cur = table[h];       // (a slot only goes EMPTY -> key: a stale EMPTY just sends us into the CAS)
                if (cur == kPatEmpty) {
                    cur = atomicCAS(&table[h], kPatEmpty, key);
                    if (cur == kPatEmpty) {
                        if (atomicAdd(&flags[0], 1) + 1 > kPatMax - 1) __hip_atomic_store(&flags[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        cur = key;
                    }
                }
                if (cur == key) { slot = (int)h; break; }
                h = (h + 1) & (kPatSlots - 1);
            }
            if (slot < 0) { __hip_atomic_store(&flags[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
            last = key;
            last_slot = slot;
        }
        if (rep[slot] > (int)row) atomicMin(&rep[slot], (int)row);     // the smallest row of every pattern: deterministic
    }
}

// tab[id] = (length, offsets) of the representative row reps[id]
__global__ __launch_bounds__(kBlock) void k_pat_extract(int npat, const int *reps, const int *rp, const int *ci, int *tab)
{
    const int id = blockIdx.x * kBlock + threadIdx.x;
    if (id >= kPatMax) return;
    int *t = tab + (size_t)id * kPatRow;
    for (int j = 0; j < kPatRow; j++) t[j] = 0;
    if (id >= npat || reps[id] < 0) return;              // (reps[id] < 0: the empty pattern added for the padding rows)
    const int row = reps[id], s = rp[row], len = rp[row + 1] - s;
    t[0] = len;
    for (int j = 0; j < len; j++) t[1 + j] = ci[s + j] - row;
}

// pid[row] = id of the row's pattern; every row is compared with its table entry (flags[0] = 1: a mismatch)
__global__ __launch_bounds__(kBlock) void k_pat_assign(int n, int nrows_padded, int npat, int empty_id, const int *rp, const int *ci,
                                                       const unsigned long long *keys, const int *tab, unsigned char *pid, int *flags)
{
    __shared__ unsigned long long skey[kPatMax];
    __shared__ int stab[kPatMax * kPatRow];
    for (int i = threadIdx.x; i < kPatMax; i += kBlock) skey[i] = i < npat ? keys[i] : kPatEmpty;
    for (int i = threadIdx.x; i < kPatMax * kPatRow; i += kBlock) stab[i] = tab[i];
    __syncthreads();
    for (long long row = (long long)blockIdx.x * kBlock + threadIdx.x; row < nrows_padded; row += (long long)gridDim.x * kBlock) {
        if (row >= n) { pid[row] = (unsigned char)empty_id; continue; }
        bool too_long;
        const unsigned long long key = pat_key(rp, ci, (int)row, &too_long);
        int lo = 0, hi = npat - 1;                       // keys are sorted ascending (the empty pattern's slot holds its own key)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (skey[mid] < key) lo = mid + 1; else hi = mid;
        }
        const int *t = stab + lo * kPatRow;
        const int s = rp[row], len = rp[row + 1] - s;
        bool ok = !too_long && skey[lo] == key && t[0] == len;
        for (int j = 0; ok && j < len; j++) ok = t[1 + j] == ci[s + j] - (int)row;
        if (!ok) flags[0] = 1;
        pid[row] = (unsigned char)lo;
    }
}

// values into the slot-major chunks; one lane per row
__global__ __launch_bounds__(kBlock) void k_pat_fill(int n, int nchunks, int W, const int *rp, const double *val, double *pval)
{
    const long long row = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (row >= (long long)nchunks * kPatChunk) return;
    const int c = (int)(row >> 6), lane = (int)(row & 63);
    const int s = row < n ? rp[row] : 0, len = row < n ? rp[row + 1] - s : 0;
    double *dst = pval + ((size_t)c * W) * kPatChunk + lane;
    for (int j = 0; j < W; j++) dst[(size_t)j * kPatChunk] = j < len ? val[s + j] : 0.0;
}

// dictionary indices into one word per row (bytes in column order, the rest 0); one lane per row
__global__ __launch_bounds__(kBlock) void k_pat_fill_idx(int n, long long padded, int vword, const int *rp, const unsigned char *idx,
                                                         unsigned char *vidx)
{
    const long long row = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (row >= padded) return;
    const int s = row < n ? rp[row] : 0, len = row < n ? rp[row + 1] - s : 0;
    unsigned long long w[2] = {0ull, 0ull};
    for (int j = 0; j < len; j++) w[j >> 3] |= (unsigned long long)idx[s + j] << (8 * (j & 7));
    unsigned long long *dst = (unsigned long long *)(vidx + (size_t)row * vword);
    dst[0] = w[0];
    if (vword == 16) dst[1] = w[1];
}

int pat_build(hipStream_t st, int n, int64_t nnz, const int *rp, const int *ci, const double *val, PatPlan *out, double max_fill,
              const ValDict *vd)
{
    const double t0 = now_s();
    PatPlan p;
    p.n = n;
    p.nnz = nnz;
    p.nchunks = (n + kPatChunk - 1) / kPatChunk;
    const long long padded = (long long)p.nchunks * kPatChunk;
    unsigned long long *table = nullptr, *keys_dev = nullptr;
    int *rep = nullptr, *flags = nullptr, *reps_dev = nullptr;
    int rc = CUDAMAT_OK;
    do {
        if (n <= 0 || nnz <= 0) { rc = CUDAMAT_ERR_ARG; set_error("pat_build: empty matrix"); break; }
        if ((rc = dalloc(&table, (size_t)kPatSlots))) break;
        if ((rc = dalloc(&rep, (size_t)kPatSlots))) break;
        if ((rc = dalloc(&flags, 2))) break;
        if ((rc = CM_RC(hipMemsetAsync(table, 0xFF, sizeof(unsigned long long) * kPatSlots, st)))) break;
        if ((rc = CM_RC(hipMemsetAsync(rep, 0x7F, sizeof(int) * kPatSlots, st)))) break;
        if ((rc = CM_RC(hipMemsetAsync(flags, 0, 2 * sizeof(int), st)))) break;
        int grid = (int)(((long long)n + kBlock - 1) / kBlock);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(k_pat_probe, dim3(grid), dim3(kBlock), 0, st, n, rp, ci, table, rep, flags);
        int h[2] = {0, 0};
        std::vector<unsigned long long> hk((size_t)kPatSlots);
        std::vector<int> hr((size_t)kPatSlots);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(h, flags, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(hk.data(), table, sizeof(unsigned long long) * kPatSlots, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(hr.data(), rep, sizeof(int) * kPatSlots, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("pattern probe failed"); break; }
        if (h[1] || h[0] < 1 || h[0] > kPatMax - 1) {
            rc = CUDAMAT_ERR_ARG;
            set_error("pat_build: rows longer than %d entries or more than %d distinct row patterns", kPatMaxLen, kPatMax - 1);
            break;
        }
        // distinct patterns sorted by key (deterministic ids) + the empty pattern for the padding rows of the last chunk
        std::vector<std::pair<unsigned long long, int>> pats;
        for (int i = 0; i < kPatSlots; i++)
            if (hk[(size_t)i] != kPatEmpty) pats.push_back({hk[(size_t)i], hr[(size_t)i]});
        if ((int)pats.size() != h[0]) { rc = CUDAMAT_ERR_HIP; set_error("pat_build: table count mismatch"); break; }
        std::sort(pats.begin(), pats.end());
        std::vector<unsigned long long> keys;
        std::vector<int> reps;
        for (auto &q : pats) { keys.push_back(q.first); reps.push_back(q.second); }
        p.npat = (int)keys.size();
        if ((rc = dalloc(&keys_dev, (size_t)kPatMax))) break;
        if ((rc = dalloc(&reps_dev, (size_t)kPatMax))) break;
        if ((rc = dalloc(&p.tab, (size_t)kPatMax * kPatRow))) break;
        if (hipMemcpyAsync(keys_dev, keys.data(), sizeof(unsigned long long) * keys.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(reps_dev, reps.data(), sizeof(int) * reps.size(), hipMemcpyHostToDevice, st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        hipLaunchKernelGGL(k_pat_extract, dim3(1), dim3(kBlock), 0, st, p.npat, reps_dev, rp, ci, p.tab);
        std::vector<int> htab((size_t)kPatMax * kPatRow);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(htab.data(), p.tab, sizeof(int) * htab.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("pattern table failed"); break; }
        // the id the padding rows take: a pattern of length 0 (present already when the matrix has empty rows, else the
        // first unused table row, which k_pat_extract left all zero)
        int empty_id = -1;
        for (int i = 0; i < p.npat; i++) {
            if (htab[(size_t)i * kPatRow] == 0) empty_id = i;
            p.W = std::max(p.W, htab[(size_t)i * kPatRow]);
        }
        if (empty_id < 0) empty_id = p.npat;             // (< kPatMax: at most kPatMax - 1 patterns came from the matrix)
        p.fill = (double)padded * p.W / (double)nnz;
        if (p.W < 1 || (max_fill > 0.0 && p.fill > max_fill)) {
            rc = CUDAMAT_ERR_ARG;
            set_error("pat_build: padded copy would hold %.2f x the entries", p.fill);
            break;
        }
        if ((rc = dalloc(&p.pid, (size_t)padded))) break;
        if ((rc = CM_RC(hipMemsetAsync(flags, 0, 2 * sizeof(int), st)))) break;
        hipLaunchKernelGGL(k_pat_assign, dim3(grid), dim3(kBlock), 0, st, n, (int)padded, p.npat, empty_id, rp, ci, keys_dev, p.tab, p.pid, flags);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(h, flags, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("pattern assignment failed"); break; }
        if (h[0]) { rc = CUDAMAT_ERR_ARG; set_error("pat_build: two row patterns share a hash key; this matrix keeps its indices"); break; }
        if (vd && vd->n > 0) {
            p.vword = p.W <= 8 ? 8 : 16;
            p.dict = vd->dict;
            p.ndict = vd->n;
            if ((rc = dalloc(&p.vidx, (size_t)padded * p.vword))) break;
            hipLaunchKernelGGL(k_pat_fill_idx, dim3((unsigned)((padded + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, n, padded, p.vword, rp, vd->idx, p.vidx);
        } else {
            if ((rc = dalloc(&p.val, (size_t)padded * p.W))) break;
            hipLaunchKernelGGL(k_pat_fill, dim3((unsigned)((padded + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, n, p.nchunks, p.W, rp, val, p.val);
        }
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("pattern fill failed"); break; }
    } while (0);
    void *tmp[] = {table, rep, flags, keys_dev, reps_dev};
    for (void *q : tmp)
        if (q) CM_DROP(hipFree(q));
    if (rc) {
        pat_free(&p);
        return rc;
    }
    // a workgroup takes tiles of 4 chunks (256 rows); tiles_per_block so that the grid stays within kSpmvGridMax and is a
    // multiple of 8 (the XCD-aware dealing below)
    const long long tiles = ((long long)p.nchunks + 3) / 4;
    long long tpb = (tiles + kSpmvGridMax - 1) / kSpmvGridMax;
    if (tpb < 1) tpb = 1;
    long long g = (tiles + tpb - 1) / tpb;
    if (g >= 8) g = (g + 7) / 8 * 8;
    if (g > kSpmvGridMax) { tpb++; g = ((tiles + tpb - 1) / tpb + 7) / 8 * 8; }
    p.grid = (int)g;
    p.tiles_per_block = (int)((tiles + g - 1) / g);
    p.build_seconds = now_s() - t0;
    *out = p;
    return CUDAMAT_OK;
}

template <int WMAX>
__global__ __launch_bounds__(kBlock) void k_spmv_pat(SpmvArgs a, int nchunks, int W, int tiles_per_block, const unsigned char *pid,
                                                     const int *tab, int npat_rows, const double *pval)
{
#pragma clang fp contract(off)      // one rounding per product and per sum, in column order (bicstab.cpp:72-77)
    __shared__ int stab[kPatMax * kPatRow];
    __shared__ double lds[8];
    if (a.loop.st) {
        if (a.check == CHECK_HALF) {
            if (check_half(a.loop, a.half, lds)) return;
        } else if (a.loop.st->state != 0) {
            return;
        }
    }
    for (int i = threadIdx.x; i < npat_rows * kPatRow; i += kBlock) stab[i] = tab[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nb = gridDim.x, b = blockIdx.x;
    // tiles dealt CYCLICALLY inside an XCD's contiguous share (see k_spmv_stream): the workgroups of one XCD sit on
    // neighbouring tiles at any moment, so the uses of an x entry by rows i - nx, i, i + nx meet in that XCD's L2
    const bool xcd_split = (nb & 7) == 0;
    const int wg_per_set = xcd_split ? nb >> 3 : nb;
    const int set = xcd_split ? (b & 7) : 0;
    const int w = xcd_split ? (b >> 3) : b;
    const long long set_tile0 = (long long)set * wg_per_set * tiles_per_block;
    double acc[2] = {0.0, 0.0};
    for (int t = 0; t < tiles_per_block; t++) {
        const long long tile = set_tile0 + (long long)t * wg_per_set + w;
        const long long c = tile * 4 + wave;
        if (c >= nchunks) continue;
        const long long rowl = c * kPatChunk + lane;
        const int *e = stab + (int)pid[rowl] * kPatRow;
        const int len = e[0];
        const double *src = pval + ((size_t)c * W) * kPatChunk + lane;
        double v[WMAX], xv[WMAX];
#pragma unroll
        for (int j = 0; j < WMAX; j++)
            if (j < W) v[j] = __builtin_nontemporal_load(src + (size_t)j * kPatChunk);
#pragma unroll
        for (int j = 0; j < WMAX; j++)
            if (j < len) xv[j] = a.x[rowl + e[1 + j]];
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < WMAX; j++)
            if (j < len) {
                const double prod = v[j] * xv[j];
                sum = sum + prod;
            }
        if (rowl < a.n) {
            const int row = (int)rowl;
            if (a.d) {
                const double dx = a.d[row] * a.xd[row];
                sum = sum + dx;
            }
            double out = a.alpha * sum;
            if (a.beta != 0.0) {
                const double by = a.beta * a.y[row];
                out = out + by;
            }
            a.y[row] = out;
            if (a.dot) {
                acc[0] += out * a.w[row];
                acc[1] += out * out;
            }
        }
    }
    if (a.dot) {
        block_sum<2>(acc, lds);
        if (threadIdx.x == 0) {
            a.parts[2 * b] = acc[0];
            a.parts[2 * b + 1] = acc[1];
        }
    }
}

// The same with a VALUE DICTIONARY: a row's values are 8-bit indices packed into one 8- or 16-byte word (one coalesced
// load per wave and chunk), the dictionary sits beside the pattern table in LDS; dict[index] IS the stored double, so the
// products and their order are those of k_spmv_pat: bit-identical.  A stencil row then costs 8 + 1 bytes of matrix data.
template <int WORDS>
__global__ __launch_bounds__(kBlock) void k_spmv_pat_d(SpmvArgs a, int nchunks, int tiles_per_block, const unsigned char *pid,
                                                       const int *tab, int npat_rows, const unsigned char *vidx, const double *dict)
{
#pragma clang fp contract(off)
    constexpr int WMAX = 8 * WORDS;
    __shared__ int stab[kPatMax * kPatRow];
    __shared__ double dv[kDictMax];
    __shared__ double lds[8];
    if (a.loop.st) {
        if (a.check == CHECK_HALF) {
            if (check_half(a.loop, a.half, lds)) return;
        } else if (a.loop.st->state != 0) {
            return;
        }
    }
    for (int i = threadIdx.x; i < npat_rows * kPatRow; i += kBlock) stab[i] = tab[i];
    dv[threadIdx.x] = dict[threadIdx.x];                 // (kBlock == kDictMax == 256)
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nb = gridDim.x, b = blockIdx.x;
    const bool xcd_split = (nb & 7) == 0;
    const int wg_per_set = xcd_split ? nb >> 3 : nb;
    const int set = xcd_split ? (b & 7) : 0;
    const int w = xcd_split ? (b >> 3) : b;
    const long long set_tile0 = (long long)set * wg_per_set * tiles_per_block;
    double acc[2] = {0.0, 0.0};
    for (int t = 0; t < tiles_per_block; t++) {
        const long long tile = set_tile0 + (long long)t * wg_per_set + w;
        const long long c = tile * 4 + wave;
        if (c >= nchunks) continue;
        const long long rowl = c * kPatChunk + lane;
        const int *e = stab + (int)pid[rowl] * kPatRow;
        const int len = e[0];
        unsigned long long word[WORDS];
#pragma unroll
        for (int q = 0; q < WORDS; q++) word[q] = __builtin_nontemporal_load((const unsigned long long *)(vidx + (size_t)rowl * (8 * WORDS)) + q);
        double xv[WMAX];
#pragma unroll
        for (int j = 0; j < WMAX; j++)
            if (j < len) xv[j] = a.x[rowl + e[1 + j]];
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < WMAX; j++)
            if (j < len) {
                const double prod = dv[(word[j >> 3] >> (8 * (j & 7))) & 0xffull] * xv[j];
                sum = sum + prod;
            }
        if (rowl < a.n) {
            const int row = (int)rowl;
            if (a.d) {
                const double dx = a.d[row] * a.xd[row];
                sum = sum + dx;
            }
            double out = a.alpha * sum;
            if (a.beta != 0.0) {
                const double by = a.beta * a.y[row];
                out = out + by;
            }
            a.y[row] = out;
            if (a.dot) {
                acc[0] += out * a.w[row];
                acc[1] += out * out;
            }
        }
    }
    if (a.dot) {
        block_sum<2>(acc, lds);
        if (threadIdx.x == 0) {
            a.parts[2 * b] = acc[0];
            a.parts[2 * b + 1] = acc[1];
        }
    }
}

int launch_spmv_pat(hipStream_t st, const PatPlan &p, const SpmvArgs &a)
{
    const int rows = p.npat + 1 < kPatMax ? p.npat + 1 : kPatMax;      // (+ the empty pattern of the padding rows)
    if (p.vidx) {
        static_assert(kBlock == kDictMax, "one dictionary entry per thread");
        if (p.vword == 8)
            hipLaunchKernelGGL(k_spmv_pat_d<1>, dim3(p.grid), dim3(kBlock), 0, st, a, p.nchunks, p.tiles_per_block, p.pid, p.tab, rows, p.vidx, p.dict);
        else
            hipLaunchKernelGGL(k_spmv_pat_d<2>, dim3(p.grid), dim3(kBlock), 0, st, a, p.nchunks, p.tiles_per_block, p.pid, p.tab, rows, p.vidx, p.dict);
        CM_HIP(hipGetLastError());
        return CUDAMAT_OK;
    }
    if (p.W <= 8)
        hipLaunchKernelGGL(k_spmv_pat<8>, dim3(p.grid), dim3(kBlock), 0, st, a, p.nchunks, p.W, p.tiles_per_block, p.pid, p.tab, rows, p.val);
    else
        hipLaunchKernelGGL(k_spmv_pat<16>, dim3(p.grid), dim3(kBlock), 0, st, a, p.nchunks, p.W, p.tiles_per_block, p.pid, p.tab, rows, p.val);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

}  // namespace cm
