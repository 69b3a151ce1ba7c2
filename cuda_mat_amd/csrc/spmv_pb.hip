// spmv_pb.hip -- two-phase ("propagation blocking") CSR SpMV for scattered columns.
//
// Why: with 50 random columns per row over an 80 MB x (C4), the wave-per-row CSR kernel is bound
// by the fabric, not by HBM: every 8-byte gather of x misses the 4 MiB XCD L2 and drags a 128-byte
// line (measured 69 GB moved per launch for 6.2 GB of algorithmic bytes).
// Here both the gather side and the scatter side live in LDS and HBM sees only streams:
//
//   phase 1  one workgroup per COLUMN block: the x tile (<= 13 K doubles) is staged in LDS; the
//            block's entries (value fp64 + local column u16, stored contiguously) are streamed and
//            the products P[k] = val[k] * x[col[k]] are streamed back out (8 B each);
//   phase 2  one workgroup per ROW block, one wavefront per sub-block of rows whose y tile sits
//            in LDS: the wave walks the column blocks in order and adds the products of its
//            (sub-block, column block) segment into the tile with ds_add_f64, then writes y
//            (+ diagonal term, alpha/beta, fused dot partials).
//
// Entries are stored once, at analysis time, in (column block, row block, row, column) order.
// A row's products therefore reach its accumulator in increasing column order, each product rounded
// once and added once -- the same sequence of roundings as the reference CPU loop
// `b[i] += A.Value[j] * x[A.Col[j]]` (bicstab_omp/bicstab.cpp:72-77) -- and a row is owned by exactly
// one wave, so the result is deterministic (no inter-wave races, no global atomics).
// HBM traffic per SpMV: 10 B/nnz read + 8 B/nnz written (phase 1), 10 B/nnz read (phase 2).
#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "spmv_pb.h"

namespace cm {

typedef unsigned short u16;
constexpr int kPbBuildWaves = 4;
constexpr int kP1Threads = 1024;
constexpr int kP1Unroll = 8;          // steps of phase 1 whose loads are in flight together (k_pb_phase1_dict)
constexpr int kTileMax = 13312;        // doubles per y tile of a row block / x tile of a column block (104 KB): 768 column blocks at
                                       // 1e7 columns = three whole rounds over 256 CUs (512 / 768 / 1024 / 1280 blocks at C4: 2.66 / 2.53 /
                                       // 2.70 / 2.89 ms per SpMV pair, HISTORY.md round 3)
constexpr int kPbAlign = 64;           // every column block's entries start on a 64-entry boundary: the 1 KB product stores, the 1 KB value
                                       // loads and the 256-byte column loads of a phase-1 wave then cover whole 128-byte lines (<= 63 idle
                                       // slots per block: value 0 x column 0, never read by phase 2; -4.5 % per C4 SpMV against packed blocks)

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

void pb_free(PbPlan *p)
{
    void *ptrs[] = {p->pv, p->pvi, p->pc, p->pr, p->P, p->cstart, p->col0, p->sstart, p->slen, p->order};
    for (void *q : ptrs)
        if (q) CM_DROP(hipFree(q));
    *p = PbPlan();
}

// ------------------------------------------------------------------ candidate estimate
bool pb_candidate(hipStream_t st, int n, int64_t n_cols, int64_t nnz, const int *rp, const int *ci)
{
    // Candidates are chosen by size only and then TIMED against the CSR kernel (ensure_spmv_mode): even
    // when x fits an XCD's L2 the lanes-per-row kernel is held to the L2-hit gather rate (~1.5 TB/s
    // algorithmic at 50 nnz/row), so the blocked form can win for banded matrices with long rows too
    // (scripts/size_probe.py).  Below ~1 MB of x or ~1 M entries the fixed costs dominate.
    (void)st; (void)rp; (void)ci;
    if (n < 65536 || nnz < (1 << 20) || n_cols * 8 < (1 << 20)) return false;
    if ((double)nnz / n < 4.0) return false;
    if ((int64_t)n > (int64_t)kMaxParts * kTileMax) return false;
    return true;
}

// ------------------------------------------------------------------ analysis (one-off)
// One wavefront per sub-block walks its rows in order.  COUNT: entries per column block.
// FILL: place every entry at cursor[cb]++ (cursor in LDS, seeded with the global start of the
// (cb, sub) segment) -- deterministic, no global atomics.  Equal-cb lanes of one load are adjacent
// (columns are sorted), so a lane's rank inside its run is lane - (first lane of the run).
__device__ __forceinline__ int col_to_cb(const PbCut &c, int col)
{
    const int q = col / c.per, w = col - q * c.per;
    const int ch = w / c.chunk_len, off = w - ch * c.chunk_len;
    return (q * c.chunks + ch) * c.bpc + off / c.CB;
}

// the same for the builders' inner loops: one slice, one piece (an unsharded copy) is a single division, done in double
// precision (exact for 0 <= col < 2^31: the quotient of two integers below 2^31 differs from the next integer by more than
// 2^-31, a double's relative error is 2^-53) -- three 32-bit integer divisions cost ~100 instructions of a wave otherwise
__device__ __forceinline__ int col_to_cb_fast(const PbCut &c, double inv_cb, int col)
{
    if (c.chunks == 1 && col < c.per) {
        int cb = (int)((double)col * inv_cb);
        cb -= (long long)cb * c.CB > (long long)col ? 1 : 0;
        cb += (long long)(cb + 1) * c.CB <= (long long)col ? 1 : 0;
        return cb;
    }
    return col_to_cb(c, col);
}

// Rank of every active lane among the EARLIER active lanes holding the same key (0 <= key < nkeys <= 64), and for lane q <
// nkeys the number of lanes holding key q.  One 64-bit word of LDS per key, owned by this wave: lanes OR their bit into
// their key's word (integer ORs: any order of service gives the same word), read it back and count the bits below their own.
// ~15 instructions for what a ballot per key costs ~10 each.
__device__ __forceinline__ int wave_match_rank(unsigned long long *tab, int key, bool active, int lane, int nkeys, int *count_of_key_lane)
{
    if (lane < nkeys) tab[lane] = 0ULL;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (active) atomicOr(&tab[key], 1ULL << lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long mine = active ? tab[key] : 0ULL;
    *count_of_key_lane = lane < nkeys ? __popcll(tab[lane]) : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return __popcll(mine & ((1ULL << lane) - 1ULL));
}

// The launch covers the sub-blocks [sub0, sub1) (the drop-in entry point fills a blocked copy piece by piece while the
// values are still being uploaded: rows are independent here).
template <bool FILL>
__global__ __launch_bounds__(64 * kPbBuildWaves) void k_pb_rows(int n, const int *rp, const int *ci,
                                                               const double *val, PbCut cut, const int *col0, int NCB,
                                                               int SR, int NSUB, int *bins, double *pv, u16 *pc, u16 *pr,
                                                               const unsigned char *vidx, unsigned char *pvi, int sub0, int sub1)
{
    extern __shared__ int lds_i[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub = sub0 + blockIdx.x * kPbBuildWaves + wave;
    if (sub >= sub1) return;
    int *cur = lds_i + (size_t)wave * NCB;
    for (int c = lane; c < NCB; c += 64) cur[c] = FILL ? bins[(size_t)c * NSUB + sub] : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const long long row0 = (long long)sub * SR;
    const int row1 = (int)(row0 + SR < n ? row0 + SR : n);
    for (int row = (int)row0; row < row1; row++) {
        const int rs = rp[row], re = rp[row + 1];
        for (int k0 = rs; k0 < re; k0 += 64) {
            const int k = k0 + lane;
            const bool active = k < re;
            const int col = active ? ci[k] : -1;
            const int cb = active ? col_to_cb(cut, col) : -1;
            const int prev = __shfl_up(cb, 1, 64);
            const bool head = lane == 0 || cb != prev;
            int hs = head ? lane : 0;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(hs, o, 64);
                if (lane >= o) hs = t > hs ? t : hs;
            }
            const int next = __shfl_down(cb, 1, 64);
            const bool tail = lane == 63 || next != cb;
            if (active) {
                const int base = cur[cb];
                const int rank = lane - hs;
                if (FILL) {
                    const int dest = base + rank;
                    if (pvi) pvi[dest] = vidx[k];
                    else pv[dest] = val[k];
                    pc[dest] = (u16)(col - col0[cb]);
                    pr[dest] = (u16)(row - (int)row0);
                }
                if (tail) cur[cb] = base + rank + 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!FILL)
        for (int c = lane; c < NCB; c += 64) bins[(size_t)c * NSUB + sub] = cur[c];
}

// ------------------------------------------------------------------ fill in two passes (round 5)
// k_pb_rows<true> scatters every entry straight to its (column block, sub-block) segment: a wave keeps NCB (768 at C4)
// write streams open for as long as it walks its sub-block, each fed 8 + 2 + 2 bytes at a time, and the partly written
// 128-byte lines leave the L2 long before they are complete (55 GB moved for a 16 GB job, 41 ms at C4).  The same
// permutation as a two-level partition whose streams stay few and whose runs are whole segments:
//   pass A (k_pb_group)    a wave walks its sub-block as before but sorts the entries only by GROUP of GB consecutive
//                          column blocks (<= 32 groups): <= 32 streams per wave, each fed ~1.5 entries per row, all inside
//                          the wave's own contiguous scratch region [rp[row0], rp[row1]) -- value (8 B, or the 8-bit
//                          dictionary index) into the product stream P (not needed before the first SpMV), packed
//                          (column block, local column, local row) into an 8-byte scratch word;
//   pass B (k_pb_scatter)  one wave per (sub-block, group) bucket (~1300 entries at C4, in (row, column) order): a stable
//                          counting sort by column block -- ranks by ballot per block, no atomics, no order left to the
//                          hardware -- gives every entry its place; the inverse permutation is kept in LDS (2 bytes per
//                          entry) and the bucket is then written in DESTINATION order: GB whole (sub-block, column block)
//                          segments, consecutive lanes to consecutive addresses.
// The result is the array k_pb_rows<true> builds, bit for bit (tests: every blocked-SpMV test compares with the oracle).
constexpr int kPbGroups = 32;            // pass-A streams per wave at most (one lane per group: <= 64).  48 groups x 1024-entry buckets
                                         // measured 8.6 + 4.9 ms per C4 copy against 7.0 + 6.0 ms with 32 x 2048: more streams cost pass A
                                         // what the shorter buckets save pass B
static_assert(kPbGroups <= 64, "one lane per group");
constexpr int kPbRpMax = 2047;           // rows of a sub-block whose row pointers pass A stages in LDS (longer sub-blocks read them from L2)
constexpr int kPbBucketMax = 2048;       // entries of a bucket that is sorted in the wave's LDS slice (7 bytes each)

__device__ __forceinline__ unsigned long long pb_pack(int cb, int pc, int pr)
{
    return ((unsigned long long)(unsigned)cb << 32) | ((unsigned long long)(unsigned)(pc & 0xffff) << 16) | (unsigned long long)(unsigned)(pr & 0xffff);
}

// Count pass (the pattern only): entries of every (column block, sub-block) segment.  One wave per sub-block streams its
// entries [rp[row0], rp[row1]) 64 at a time -- no per-row round trips -- and counts into LDS (integer adds: any order).
__global__ __launch_bounds__(64 * kPbBuildWaves) void k_pb_count(int n, const int *rp, const int *ci, PbCut cut, int NCB, int SR, int NSUB,
                                                                int *bins, int sub0, int sub1)
{
    extern __shared__ int lds_i[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub = sub0 + blockIdx.x * kPbBuildWaves + wave;
    if (sub >= sub1) return;
    int *cur = lds_i + (size_t)wave * NCB;
    for (int c = lane; c < NCB; c += 64) cur[c] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const long long row0 = (long long)sub * SR;
    const int row1 = (int)(row0 + SR < n ? row0 + SR : n);
    const int e0 = row0 < n ? rp[row0] : rp[n], e1 = rp[row1];
    const double inv_cb = 1.0 / (double)cut.CB;
    for (int k0 = e0; k0 < e1; k0 += 256) {
        int col[4];
#pragma unroll
        for (int u = 0; u < 4; u++) col[u] = k0 + 64 * u + lane < e1 ? ci[k0 + 64 * u + lane] : -1;
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (col[u] >= 0) atomicAdd(&cur[col_to_cb_fast(cut, inv_cb, col[u])], 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < NCB; c += 64) bins[(size_t)c * NSUB + sub] = cur[c];
}

// gstart: [NSUB][NG + 1] first scratch entry of every (sub-block, group) bucket (written here, read by pass B)
// The wave streams its sub-block's entries 64 at a time in CSR order (= (row, column) order); an entry of group g goes
// behind the earlier entries of g: ranks by one ballot per group present, the groups' cursors live in lane g's register.
__global__ __launch_bounds__(64 * kPbBuildWaves) void k_pb_group(int n, const int *rp, const int *ci, const double *val,
                                                                const unsigned char *vidx, PbCut cut, const int *col0, int NCB,
                                                                int SR, const int *slen, int GB, int NG, int *gstart,
                                                                double *sval, unsigned char *sval8, unsigned long long *smeta,
                                                                int sub0, int sub1)
{
    __shared__ unsigned long long tab_all[kPbBuildWaves][kPbGroups];
    __shared__ int lrp_all[kPbBuildWaves][kPbRpMax + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub = sub0 + blockIdx.x * kPbBuildWaves + wave;
    if (sub >= sub1) return;
    unsigned long long *tab = tab_all[wave];
    int *lrp = lrp_all[wave];
    const int gshift = 31 - __builtin_clz(GB);                  // GB is a power of two
    const double inv_cb = 1.0 / (double)cut.CB;
    const long long row0 = (long long)sub * SR;
    const int row1 = (int)(row0 + SR < n ? row0 + SR : n);
    // entries of this sub-block per group, from the count pass's segment lengths: lane g sums its GB blocks
    int cnt = 0;
    if (lane < NG) {
        const int *sl = slen + (size_t)sub * NCB;
        for (int c = lane * GB; c < (lane + 1) * GB && c < NCB; c++) cnt += sl[c];
    }
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    const int e0 = row0 < n ? rp[row0] : rp[n], e1 = rp[row1];
    const int base = e0 - rp[0];
    const int total = __builtin_amdgcn_readlane(inc, 63);      // (lanes >= NG add nothing)
    int run = base + inc - cnt;                                 // lane g: next scratch slot of group g
    if (lane < NG) gstart[(size_t)sub * (NG + 1) + lane] = run;
    if (lane == 0) gstart[(size_t)sub * (NG + 1) + NG] = base + total;      // (NG may be 64: there is no lane NG)
    // Nothing a chunk needs may wait for a load issued in that chunk (a wave walks ~640 chunks one after the other: every
    // exposed round trip costs the launch ~1 us x 640): the sub-block's row pointers are staged in LDS once, the columns AND
    // values of the next chunk are in flight while this one is placed, the local column follows from the block by arithmetic.
    const int nrows = row1 - (int)row0;
    const bool rp_lds = nrows <= kPbRpMax;
    if (rp_lds)
        for (int i = lane; i <= nrows; i += 64) lrp[i] = rp[row0 + i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool one_piece = cut.chunks == 1 && cut.bpc >= NCB;   // one slice, one piece: block cb starts at column cb * CB
    int rbase = 0;                                              // (row - row0) of the current chunk's first entry, or an empty row before it
    int col_n = e0 + lane < e1 ? ci[e0 + lane] : -1;
    double val_n = 0.0;
    unsigned char vidx_n = 0;
    if (e0 + lane < e1) { if (sval8) vidx_n = vidx[e0 + lane]; else val_n = val[e0 + lane]; }
    for (int k0 = e0; k0 < e1; k0 += 64) {
        const int k = k0 + lane;
        const bool active = k < e1;
        const int col = col_n;
        const double v = val_n;
        const unsigned char vi = vidx_n;
        col_n = k + 64 < e1 ? ci[k + 64] : -1;
        if (k + 64 < e1) { if (sval8) vidx_n = vidx[k + 64]; else val_n = val[k + 64]; }
        int r = rbase;
        if (active) {
            if (rp_lds) { while (k >= lrp[r + 1]) r++; }        // (rows may be empty)
            else { while (k >= rp[row0 + r + 1]) r++; }
        }
        const int cb = active ? col_to_cb_fast(cut, inv_cb, col) : 0;
        const int g = cb >> gshift;
        int added;
        const int rank = wave_match_rank(tab, g, active, lane, NG, &added);
        const int dest = __shfl(run, g, 64) + rank;
        run += added;
        if (active) {
            if (sval8) sval8[dest] = vi;
            else sval[dest] = v;
            smeta[dest] = pb_pack(cb, col - (one_piece ? cb * cut.CB : col0[cb]), r);
        }
        // the last active lane's row starts the next chunk's search
        const unsigned long long am = __ballot(active);
        rbase = __builtin_amdgcn_readlane(r, 63 - __builtin_clzll(am));
    }
}

constexpr int kPbScatterWaves = 4;
__global__ __launch_bounds__(64 * kPbScatterWaves) void k_pb_scatter(int NCB, int NSUB, const int *slen, const int *bins, int GB, int NG,
                                                                    const int *gstart, const double *sval, const unsigned char *sval8,
                                                                    const unsigned long long *smeta, double *pv, unsigned char *pvi,
                                                                    u16 *pc, u16 *pr, int sub0, int sub1)
{
    // 7 bytes per entry and wave (local column | local row, block inside the group, inverse permutation): 14.5 KB per wave,
    // two workgroups (8 waves) per CU -- the first version kept the 8-byte word (20.5 KB per wave: ONE workgroup per CU,
    // 7.8 ms at C4 for work that is a chain of ~40 short steps per wave)
    __shared__ unsigned raw_all[kPbScatterWaves][kPbBucketMax];
    __shared__ unsigned char cbl_all[kPbScatterWaves][kPbBucketMax];
    __shared__ u16 inv_all[kPbScatterWaves][kPbBucketMax];
    __shared__ unsigned long long tab_all[kPbScatterWaves][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long bucket = (long long)blockIdx.x * kPbScatterWaves + wave;
    const int sub = sub0 + (int)(bucket / NG), g = (int)(bucket % NG);
    if (sub >= sub1) return;
    u16 *inv = inv_all[wave];
    unsigned *raw = raw_all[wave];
    unsigned char *cbs = cbl_all[wave];
    const int start = gstart[(size_t)sub * (NG + 1) + g], len = gstart[(size_t)sub * (NG + 1) + g + 1] - start;
    if (len <= 0) return;
    // lane c < GB: column block g * GB + c -- its entries in this bucket, their first position inside the bucket (blocks
    // in order), and the first slot of its (column block, sub-block) segment in the copy
    const int cbid = g * GB + lane;
    const bool owner = lane < GB && cbid < NCB;
    const int cnt = owner ? slen[(size_t)sub * NCB + cbid] : 0;
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    const int lbase = inc - cnt;
    const int gdst = owner ? bins[(size_t)cbid * NSUB + sub] : 0;
    const bool staged = len <= kPbBucketMax;
    // stable ranks: the bucket is in (row, column) order; an entry of block c goes behind the earlier entries of c
    int run = staged ? lbase : gdst;
    unsigned long long m_next = lane < len ? smeta[start + lane] : 0ULL;       // the next chunk's words are in flight while this one is ranked
    for (int k0 = 0; k0 < len; k0 += 64) {
        const int k = k0 + lane;
        const bool active = k < len;
        const unsigned long long m = m_next;
        m_next = k + 64 < len ? smeta[start + k + 64] : 0ULL;
        const int cbl = active ? (int)(m >> 32) - g * GB : 0;
        int added;
        const int rank = wave_match_rank(tab_all[wave], cbl, active, lane, GB, &added);
        const int r = __shfl(run, cbl, 64) + rank;
        run += added;
        if (active) {
            if (staged) { inv[r] = (u16)k; raw[k] = (unsigned)(m & 0xffffffffULL); cbs[k] = (unsigned char)cbl; }
            else {           // a bucket too long for the LDS slice: straight to its place
                if (pvi) pvi[r] = sval8[start + k];
                else pv[r] = sval[start + k];
                pc[r] = (u16)((m >> 16) & 0xffff);
                pr[r] = (u16)(m & 0xffff);
            }
        }
    }
    if (!staged) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // destination order: position t of the sorted bucket is entry inv[t]; consecutive t of one block are consecutive slots
    for (int t0 = 0; t0 < len; t0 += 256) {        // (uniform trip count: the lane shuffles below need every lane; four
        unsigned m[4];                              // chunks' value gathers in flight together)
        double v[4];
        unsigned char vi[4];
        int dest[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = t0 + 64 * u + lane;
            const bool active = t < len;
            const int loc = active ? (int)inv[t] : 0;
            m[u] = raw[loc];
            v[u] = 0.0;
            vi[u] = 0;
            if (active) { if (pvi) vi[u] = sval8[start + loc]; else v[u] = sval[start + loc]; }
            const int cbl = active ? (int)cbs[loc] : 0;
            dest[u] = __shfl(gdst, cbl, 64) + (t - __shfl(lbase, cbl, 64));
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (t0 + 64 * u + lane < len) {
                if (pvi) pvi[dest[u]] = vi[u];
                else pv[dest[u]] = v[u];
                pc[dest[u]] = (u16)((m[u] >> 16) & 0xffff);
                pr[dest[u]] = (u16)(m[u] & 0xffff);
            }
        }
    }
}

// ---- exclusive scan of the segment counts in storage order [column block][sub-block], on the device
// k_pb_colsum: entries per column block; k_pb_colscan (one workgroup): their exclusive scan -> cstart;
// k_pb_segscan (one workgroup per column block): running offsets of its NSUB segments -> bins (fill
// cursors) and the sub-block-major copies phase 2 reads (sstart/slen [sub][cb]).
__global__ __launch_bounds__(kBlock) void k_pb_colsum(int NSUB, const int *bins, int *colsum)
{
    __shared__ int red[kBlock / 64];
    const int *b = bins + (size_t)blockIdx.x * NSUB;
    int acc = 0;
    for (int i = threadIdx.x; i < NSUB; i += kBlock) acc += b[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < kBlock / 64; w++) t += red[w];
        colsum[blockIdx.x] = t;
    }
}

// inclusive scan of one value per thread over the workgroup; returns the exclusive prefix, *total = sum
__device__ __forceinline__ int block_exclusive_scan(int v, int *lds_waves, int *total)
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
    for (int w = 0; w < kBlock / 64; w++) {
        const int t = lds_waves[w];
        if (w < wave) before += t;
        all += t;
    }
    *total = all;
    return before + inc - v;
}

__global__ __launch_bounds__(kBlock) void k_pb_colscan(int NCB, const int *colsum, int *cstart, int align)
{
    __shared__ int lds_waves[kBlock / 64];
    int run = 0;
    for (int c0 = 0; c0 < NCB; c0 += kBlock) {
        const int c = c0 + threadIdx.x;
        const int v = c < NCB ? (colsum[c] + align - 1) / align * align : 0;     // every block starts on an `align` boundary
        int total;
        const int ex = block_exclusive_scan(v, lds_waves, &total);
        if (c < NCB) cstart[c] = run + ex;
        run += total;
    }
    if (threadIdx.x == 0) cstart[NCB] = run;
}

__global__ __launch_bounds__(kBlock) void k_pb_segscan(int NCB, int NSUB, const int *cstart, int *bins, int *sstart,
                                                       int *slen)
{
    __shared__ int lds_waves[kBlock / 64];
    const int c = blockIdx.x;
    int *b = bins + (size_t)c * NSUB;
    int run = cstart[c];
    for (int s0 = 0; s0 < NSUB; s0 += kBlock) {
        const int sidx = s0 + threadIdx.x;
        const int cnt = sidx < NSUB ? b[sidx] : 0;
        int total;
        const int ex = block_exclusive_scan(cnt, lds_waves, &total);
        if (sidx < NSUB) {
            const int start = run + ex;
            b[sidx] = start;
            sstart[(size_t)sidx * NCB + c] = start;
            slen[(size_t)sidx * NCB + c] = cnt;
        }
        run += total;
    }
}

static int round_blocks(int64_t n, int tile_max)
{
    // number of blocks: a multiple of the CU count (whole rounds of workgroups) with tiles <= tile_max
    int64_t m = (n + (int64_t)256 * tile_max - 1) / ((int64_t)256 * tile_max);
    if (m < 1) m = 1;
    return (int)(256 * m);
}

// ---- staged construction: pb_build_begin (geometry, tables, count pass, scans: the PATTERN only) -> pb_build_values (the
// value array: fp64 or 8-bit dictionary indices) -> pb_build_fill over sub-block ranges (any partition of [0, NSUB)) ->
// pb_build_end.  pb_build is the four in a row; the drop-in entry point runs them while the matrix is still being uploaded.
static void pb_scratch_free(PbBuild *b)
{
    if (b->smeta) CM_DROP(hipFree(b->smeta));
    if (b->gstart) CM_DROP(hipFree(b->gstart));
    b->smeta = nullptr;
    b->gstart = nullptr;
}

void pb_build_abort(PbBuild *b)
{
    pb_scratch_free(b);
    if (b->bins) CM_DROP(hipFree(b->bins));
    b->bins = nullptr;
    pb_free(&b->p);
}

int pb_build_begin(hipStream_t st, const Config &cfg, int n, int64_t n_cols, int64_t nnz, const int *rp, const int *ci,
                   const PbCols *cols, PbBuild *b)
{
    CM_TRY(pb_build_alloc(st, cfg, n, n_cols, nnz, cols, b));
    return pb_build_count(st, cfg, rp, ci, b);
}

// geometry and EVERY allocation of the copy except its value array: depends on the sizes only, not on the matrix
int pb_build_alloc(hipStream_t st, const Config &cfg, int n, int64_t n_cols, int64_t nnz, const PbCols *cols, PbBuild *b)
{
    b->t0 = now_s();
    PbPlan &p = b->p;
    p = PbPlan();
    p.n = n;
    p.n_cols = n_cols;
    p.nnz = nnz;
    // the column cut: slices of `per` columns, `chunks` pieces per slice, `bpc` blocks of <= kTileMax columns per piece
    int local_slice = 0;
    if (cols && cols->per > 0 && cols->per < n_cols) {
        p.per = cols->per;
        p.chunks = cols->chunks < 1 ? 1 : cols->chunks > kPbMaxChunks ? kPbMaxChunks : cols->chunks;
        p.chunk_len = (p.per + p.chunks - 1) / p.chunks;
        p.bpc = (int)((p.chunk_len + kTileMax - 1) / kTileMax);
        p.CB = (int)((p.chunk_len + p.bpc - 1) / p.bpc);
        const int64_t slices = (n_cols + p.per - 1) / p.per;
        p.NCB = (int)(slices * p.chunks * p.bpc);
        local_slice = cols->rank;
    } else {
        p.per = n_cols;
        p.chunks = 1;
        p.chunk_len = n_cols;
        p.NCB = round_blocks(n_cols, kTileMax);
        p.CB = (int)((n_cols + p.NCB - 1) / p.NCB);
        p.NCB = (int)((n_cols + p.CB - 1) / p.CB);
        p.bpc = p.NCB;
    }
    // Sub-blocks (one wave each): ~48 entries per (sub-block, column block) segment keeps a wave's lanes busy.  When
    // that gives fewer than 2560 waves (10 per CU: a shard of a row-partitioned matrix, the far part of a triangular
    // factor) the waves are kept to >= 2048 with longer segments, and each wave keeps 16 segment loads in flight
    // instead of 4 -- phase 2 is bound by fabric REQUESTS (~55 G/s: 3.4 per 20-entry segment, 5.8 per 48-entry one), so
    // longer segments are cheaper per entry as long as the loads in flight make up for the fewer waves.  Measured in
    // alternation on one box (scripts/ab_env.sh, G = 8 shard of C4, ms per iteration of one rank): 4096 waves x 4 loads
    // 0.892 / 0.906 / 0.936; 4096 x 8 the same; 2048 x 8 0.852 / 0.874 / 0.873; 2048 x 16 0.808 / 0.847 / 0.848;
    // 1024 x 16 0.827 / 0.841 / 0.855.  On the full matrix (13.5 K waves) 4 loads in flight stay the best (2.58-2.67 ms
    // per SpMV against 2.69-2.70 with 8 and 2.68-2.75 with 16; 32 / 48 / 72 / 96 entries per segment: 2.87-3.02 / best / equal /
    // 3.3-3.5 ms).  The options PB_MIN_WAVES / PB_DEPTH override (tests force the shard-shaped plans on small matrices).
    const double seg_target = 48.0;
    double nsub_t = (double)nnz / ((double)p.NCB * seg_target);
    // (the 2048-wave plan needs 8 waves per row block whose y tiles fit the LDS: up to 4.7 M rows; a 1e7-row matrix with few
    // entries per row keeps the 4096-wave rule -- with 4-wave row blocks its phase 2 took 0.50 ms instead of 0.23)
    const bool few = nsub_t < 2560.0 && (double)n / 2048.0 * 8.0 * 8.0 <= 144.0 * 1024.0;          // (a G = 4 shard, 3390 natural waves, is better off with 4096 x 4: 1.50 / 1.57 / 1.50
                                               // against 1.53 / 1.60 / 1.59 ms per iteration with 3390 x 16)
    double min_waves = few ? 2048.0 : 4096.0;
    if (cfg.pb_min_waves) min_waves = (double)cfg.pb_min_waves;
    if (nsub_t < min_waves) nsub_t = min_waves;
    if (nsub_t > (double)n / 16.0) nsub_t = (double)n / 16.0;
    if (nsub_t < 4.0) nsub_t = 4.0;
    p.NW = nsub_t >= 4096.0 ? 16 : nsub_t >= 2048.0 ? 8 : 4;
    int nrb = (int)(nsub_t / p.NW / 256.0 + 0.5) * 256;          // whole rounds of workgroups
    if (nrb < 256) nrb = 256;
    const int nrb_min = (int)(((int64_t)n + kTileMax - 1) / kTileMax);
    while (nrb < nrb_min) nrb += 256;
    if (nrb > kMaxParts) nrb = kMaxParts;
    p.NRB = nrb;
    p.RB = (int)(((int64_t)n + p.NRB - 1) / p.NRB);
    p.SR = (p.RB + p.NW - 1) / p.NW;
    if (p.SR < 1) p.SR = 1;
    p.RB = p.SR * p.NW;
    p.NRB = (int)(((int64_t)n + p.RB - 1) / p.RB);
    p.NSUB = p.NRB * p.NW;
    const double seg = (double)nnz / ((double)p.NCB * p.NSUB);      // mean entries per segment
    p.LPS = seg <= 6.0 ? 16 : seg <= 22.0 ? 32 : 64;
    p.depth = few ? 16 : 4;
    if (cfg.pb_depth) p.depth = cfg.pb_depth;
    if (p.NRB > kMaxParts || p.CB > 65536 || p.SR > 65536 || (size_t)p.RB * 8 > 150 * 1024 ||
        sizeof(int) * (size_t)kPbBuildWaves * p.NCB > 150 * 1024) {     // the analysis keeps one cursor per column block in LDS
        set_error("pb_build: matrix shape outside the blocked kernel's limits");
        return CUDAMAT_ERR_ARG;
    }
    const size_t nbins = (size_t)p.NCB * p.NSUB;
    int *&bins = b->bins;
    bins = nullptr;
    int rc = CUDAMAT_OK;
    do {
        if ((rc = dalloc(&bins, nbins))) break;
        constexpr int align = kPbAlign;
        const size_t cap = (size_t)nnz + (size_t)(align - 1) * (size_t)p.NCB + 16;
        b->cap = cap;
        // (pc, pr, P and the value array: pb_build_values, which places them)
        if ((rc = dalloc(&p.cstart, (size_t)p.NCB + 1))) break;
        if ((rc = dalloc(&p.col0, (size_t)p.NCB + 1))) break;
        if ((rc = dalloc(&p.order, (size_t)p.NCB))) break;
        {
            // first column of every block (blocks tile [0, n_cols) in order; blocks past the end are empty) and the
            // phase-1 launch parts: the local slice first, then piece after piece of the other slices
            std::vector<int> h0((size_t)p.NCB + 1), ord;
            const int per_slice = p.chunks * p.bpc;
            for (int cb = 0; cb <= p.NCB; cb++) {
                const int64_t q = cb / per_slice, ch = (cb % per_slice) / p.bpc, k = cb % p.bpc;
                int64_t in_piece = k * (int64_t)p.CB;
                if (in_piece > p.chunk_len) in_piece = p.chunk_len;
                int64_t in_slice = ch * p.chunk_len + in_piece;
                if (in_slice > p.per) in_slice = p.per;
                int64_t c0 = q * p.per + in_slice;
                if (c0 > n_cols) c0 = n_cols;
                h0[(size_t)cb] = (int)c0;
            }
            ord.reserve((size_t)p.NCB);
            const int slices = p.NCB / per_slice;
            if (slices > 1) {
                for (int k = 0; k < per_slice; k++) ord.push_back(local_slice * per_slice + k);
                p.part_off[0] = 0;
                p.part_off[1] = (int)ord.size();
                for (int ch = 0; ch < p.chunks; ch++) {
                    for (int q = 0; q < slices; q++)
                        if (q != local_slice)
                            for (int k = 0; k < p.bpc; k++) ord.push_back((q * p.chunks + ch) * p.bpc + k);
                    p.part_off[ch + 2] = (int)ord.size();
                }
            } else {
                for (int cb = 0; cb < p.NCB; cb++) ord.push_back(cb);
                p.part_off[0] = 0;
                for (int ch = 0; ch <= p.chunks; ch++) p.part_off[ch + 1] = p.NCB;     // everything is "local"
            }
            if (hipMemcpyAsync(p.col0, h0.data(), sizeof(int) * h0.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemcpyAsync(p.order, ord.data(), sizeof(int) * ord.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("pb column cut upload failed"); break; }
        }
        b->cut = PbCut{(int)p.per, p.chunks, (int)p.chunk_len, p.bpc, p.CB};
        if ((rc = dalloc(&p.sstart, nbins))) break;
        if ((rc = dalloc(&p.slen, nbins))) break;
        if ((rc = set_max_lds((const void *)k_pb_count))) break;
        if ((rc = set_max_lds((const void *)k_pb_rows<true>))) break;
        // the two-pass fill's scratch (see k_pb_group): one packed word per entry, the buckets' first entries
        b->GB = 1;
        while (b->GB * kPbGroups < p.NCB) b->GB *= 2;            // a power of two (group = block >> shift), <= kPbGroups groups
        b->NG = (p.NCB + b->GB - 1) / b->GB;
        b->fill_occ = cfg.pb_fill_occ;
        b->place = cfg.pb_place;
        b->place_max_seconds = 1e-3 * cfg.pb_place_max_ms;
        b->verbose = cfg.verbose != 0;
        b->report_classes = cfg.verbose >= 2;
        b->two_pass = cfg.pb_fill2 != 0 && nnz > 0 && nnz < 0x7fffffffLL && p.SR <= 65536 && b->GB <= 64;
        if (b->two_pass) {
            if (dalloc(&b->smeta, (size_t)nnz) != CUDAMAT_OK || dalloc(&b->gstart, (size_t)p.NSUB * (size_t)(b->NG + 1)) != CUDAMAT_OK) {
                // (no room for the scratch: the single-pass fill needs none)
                if (b->smeta) CM_DROP(hipFree(b->smeta));
                b->smeta = nullptr;
                b->two_pass = false;
            }
        }
    } while (0);
    if (rc) pb_build_abort(b);
    return rc;
}

// count pass and scans: the pattern (rp, ci) is in place.  No allocation, no free.
int pb_build_count(hipStream_t st, const Config &cfg, const int *rp, const int *ci, PbBuild *b)
{
    PbPlan &p = b->p;
    const int n = p.n;
    const int64_t nnz = p.nnz;
    int *bins = b->bins;
    constexpr int align = kPbAlign;
    int rc = CUDAMAT_OK;
    do {
        const PbCut cut = b->cut;
        const unsigned grid = (unsigned)((p.NSUB + kPbBuildWaves - 1) / kPbBuildWaves);
        const size_t lds = sizeof(int) * (size_t)kPbBuildWaves * p.NCB;
        hipLaunchKernelGGL(k_pb_count, dim3(grid), dim3(64 * kPbBuildWaves), lds, st, n, rp, ci, cut, p.NCB, p.SR, p.NSUB, bins, 0, p.NSUB);
        if (hipGetLastError() != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("pb count launch failed"); break; }
        hipLaunchKernelGGL(k_pb_colsum, dim3(p.NCB), dim3(kBlock), 0, st, p.NSUB, bins, p.cstart);   // cstart doubles as scratch
        hipLaunchKernelGGL(k_pb_colscan, dim3(1), dim3(kBlock), 0, st, p.NCB, p.cstart, p.cstart, align);
        hipLaunchKernelGGL(k_pb_segscan, dim3(p.NCB), dim3(kBlock), 0, st, p.NCB, p.NSUB, p.cstart, bins, p.sstart, p.slen);
        int counted = -1;
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(&counted, p.cstart + p.NCB, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("pb scan failed"); break; }
        if ((int64_t)counted < nnz || (int64_t)counted > nnz + (int64_t)(align - 1) * p.NCB) {
            rc = CUDAMAT_ERR_ARG; set_error("pb_build: counted %d entries, expected %lld", counted, (long long)nnz); break;
        }
        b->verbose = cfg.verbose != 0;
        b->report_classes = cfg.verbose >= 2;
    } while (0);
    if (rc) pb_build_abort(b);
    return rc;
}

// the value array of the copy under construction: 8-bit indices into vd's dictionary when it has one, else fp64 values
// ---- where the big arrays of a copy go (round 5) ---------------------------------------------------------------------
// Device memory is not uniform for a kernel that reads one array while it writes another.  scripts/probe_classes.hip (48 chunks
// of 2 GB, chunk i read while chunk j is written, every pair): 5.05 TB/s for some pairs, 5.52 for the others, nothing in
// between -- and "slow together" is an equivalence relation with three classes (the stack ids of the 12-high HBM stacks,
// presumably), each 2 GB block of the driver's allocator wholly in one, the classes following each other in runs of 1, 2, 4,
// 8 ... blocks along the allocation order.  A phase-1-shaped kernel (8 B + 2 B read, 8 B written per entry) loses 5.5 % when
// values and products share a class, 1.7 % when indices and products do; a phase-2-shaped one (8 B + 2 B read) 2.3 % when
// products and indices do.  A process that allocates its arrays one after the other gets them from one class or from
// whatever mixture the allocation order crosses: the judged SpMV pair took 2.50 ... 2.73 ms with identical code, from box
// to box, from process to process, and alternating with every re-creation of a solver inside one process
// (scripts/placement_probe*.py) -- the "+- 5 % box variance" of rounds 1-4.
// So the four big arrays of a large copy are PLACED: slabs of 16 GB are taken from the allocator, cut into the driver's
// 2 GB blocks (from the segment's start), every block is classified by timing (read a known block of class k, write this
// one: slow = same class; <= 2 launches of 0.4 ms per block, remembered per address), and the arrays are cut out of runs
// of blocks: the product stream in a class of its own, values and indices in another.  What is not used goes back to the
// pool.  Bounded (<= 96 GB held, <= PB_PLACE_MAX_MS = 0.3 s); if no such arrangement turns up the arrays are allocated as before round 5.
__global__ __launch_bounds__(1024) void k_place_probe(const double2 *rd, double2 *wr, long n2)
{
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long lo = blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    for (long i = lo + threadIdx.x; i < hi; i += 1024) {
        const double2 v = rd[i];
        double2 o;
        o.x = v.x * 1.5;
        o.y = v.y + 1.0;
        wr[i] = o;
    }
}

struct PlaceTimer {
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = false;
    int launches = 0;
    explicit PlaceTimer(hipStream_t s) : st(s) { ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess; }
    ~PlaceTimer()
    {
        if (e0) CM_DROP(hipEventDestroy(e0));
        if (e1) CM_DROP(hipEventDestroy(e1));
    }
    // one timing: launches are repeated until three in a row agree to 1.5 % (a GPU that idled through a slow allocation comes
    // back at low clocks: the first launches after the pause read 3-8 % long -- the size of the effect to be measured), at most
    // twelve; the fastest of the last three.  < 0 on error
    float ms(const void *rd, void *wr, size_t bytes)
    {
        float t[12];
        int n = 0;
        while (ok && n < 12) {
            if (hipEventRecord(e0, st) != hipSuccess) { ok = false; break; }
            hipLaunchKernelGGL(k_place_probe, dim3(256), dim3(1024), 0, st, (const double2 *)rd, (double2 *)wr, (long)(bytes / 16));
            float x = 0.f;
            if (hipGetLastError() != hipSuccess || hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                hipEventElapsedTime(&x, e0, e1) != hipSuccess) { ok = false; break; }
            launches++;
            t[n++] = x;
            if (n >= 3) {
                const float lo = fminf(t[n - 1], fminf(t[n - 2], t[n - 3])), hi = fmaxf(t[n - 1], fmaxf(t[n - 2], t[n - 3]));
                if (hi <= 1.015f * lo) return lo;
            }
        }
        if (!ok || n < 3) return -1.f;
        return fminf(t[n - 1], fminf(t[n - 2], t[n - 3]));
    }
};

constexpr size_t kPlaceBlock = (size_t)2 << 30;           // the driver hands out device memory in blocks of at most 2 GB
constexpr size_t kPlaceSample = (size_t)896 << 20;        // bytes read and bytes written by one timed launch: well beyond the 256 MB
                                                          // Infinity Cache (256 MB samples read 0.101 against 0.097 ms whatever the classes)
constexpr size_t kPlaceSlab = (size_t)16 << 30;
constexpr size_t kPlaceMinBytes = (size_t)2 << 30;        // copies whose product stream is smaller are not placed
constexpr size_t kPlaceMaxHeld = (size_t)96 << 30;
constexpr float kPlaceGap = 1.035f;                       // two groups of cross timings: slowest / fastest beyond this (measured: 1.06-1.07)

// what is known about pooled addresses: the class of every 2 GB block met so far, one reference block per class, the time of
// a sample read and written inside one block.  Forgotten when the pool hands a segment back to the driver.
struct PlaceMemo {
    unsigned gen = ~0u;
    std::map<char *, int> cls;
    std::vector<char *> ref;
    float t_own = 0.f;                                 // read + write inside the first block (reported only)
    float t_lo = 0.f, t_hi = 0.f;                      // fastest / slowest cross timing met so far
    std::vector<std::pair<char *, float>> provisional; // blocks labelled before both groups of timings had been seen
    bool relabel = false;                              // ... some of them have just been forgotten: the caller classifies its slabs again
    void reset() { cls.clear(); ref.clear(); provisional.clear(); t_own = t_lo = t_hi = 0.f; relabel = false; }
};
static std::mutex g_place_mu;
static std::map<int, PlaceMemo> g_place;          // per device (a reference block is read by the probe: it has to be local)

struct PlaceRun { size_t slab; size_t first, count; int cls; };      // blocks [first, first + count) of one slab, all of one class

struct PlaceSlab {
    char *base = nullptr;        // the pooled block
    size_t bytes = 0;
    char *b0 = nullptr;          // first whole 2 GB block inside it (from the segment's start)
    std::vector<int> cls;        // class of every whole block
    std::vector<char> taken;
};

// class of the 2 GB block at `blk` (memo held by the caller's lock); -1: no verdict.  Decided by CROSS timings alone (read a
// reference block of class k, write into this one): same class = slow.  "Slow" is relative: the timings met so far fall into
// two groups 6-7 % apart (0.365 / 0.341 ms here); until both have been seen every block counts as the first block's class
// (what a fresh process's first 14 GB are), and when the faster group first shows up the provisional labels are re-examined.
// (Round 5's first version compared with a read + write INSIDE one block, which is itself up to 4 % slower than a same-class
// cross pair: on one box in five a same-class pair passed for "another class" and the placement put all four arrays in one.)
static int place_class_of(PlaceTimer &tm, PlaceMemo &m, char *blk)
{
    auto it = m.cls.find(blk);
    if (it != m.cls.end()) return it->second;
    if (m.ref.empty()) {
        m.t_own = tm.ms(blk, blk + kPlaceBlock - kPlaceSample, kPlaceSample);          // (reported only)
        if (m.t_own <= 0.f) return -1;
        m.ref.push_back(blk);
        m.cls[blk] = 0;
        return 0;
    }
    float t[3] = {0.f, 0.f, 0.f};
    const bool was_separated = m.t_hi >= kPlaceGap * m.t_lo && m.t_lo > 0.f;
    for (size_t k = 0; k < m.ref.size() && k < 3; k++) {
        t[k] = tm.ms(m.ref[k], blk + kPlaceBlock - kPlaceSample, kPlaceSample);
        if (t[k] <= 0.f) return -1;
        if (m.t_lo <= 0.f || t[k] < m.t_lo) m.t_lo = t[k];
        if (t[k] > m.t_hi) m.t_hi = t[k];
    }
    const bool separated = m.t_hi >= kPlaceGap * m.t_lo;
    int c = 0;
    if (separated) {
        const float mid = sqrtf(m.t_lo * m.t_hi);
        if (!was_separated) {
            // the first fast pair: whoever was labelled "class 0" on the strength of a timing that now counts as fast is forgotten
            for (auto p = m.provisional.begin(); p != m.provisional.end(); ++p)
                if (p->second <= mid) { m.cls.erase(p->first); m.relabel = true; }
            m.provisional.clear();
        }
        int slow = 0, arg = 0;
        for (size_t k = 0; k < m.ref.size() && k < 3; k++) {
            if (t[k] > mid) slow++;
            if (t[k] > t[arg]) arg = (int)k;
        }
        if (slow == 0 && m.ref.size() < 3) {
            c = (int)m.ref.size();
            m.ref.push_back(blk);
        } else {
            c = arg;
        }
    } else {
        m.provisional.push_back({blk, t[0]});
    }
    m.cls[blk] = c;
    return c;
}

// the first run of >= `need` free blocks of class `want` (>= 0) or of any class but `avoid`; false if none
static bool place_find(std::vector<PlaceSlab> &slabs, size_t need, int want, int avoid, PlaceRun *out)
{
    for (size_t si = 0; si < slabs.size(); si++) {
        PlaceSlab &sl = slabs[si];
        for (size_t i = 0; i + need <= sl.cls.size(); i++) {
            const int c = sl.cls[i];
            if (c < 0 || (want >= 0 && c != want) || c == avoid) continue;
            bool ok = true;
            for (size_t j = i; j < i + need && ok; j++) ok = !sl.taken[j] && sl.cls[j] == c;
            if (!ok) continue;
            for (size_t j = i; j < i + need; j++) sl.taken[j] = 1;
            *out = PlaceRun{si, i, need, c};
            return true;
        }
    }
    return false;
}

// the four big arrays of a copy, placed (see above).  sizes in bytes; vals / idx / prod receive pointers that are pooled blocks
// of their own (freed one by one as ever).  CUDAMAT_OK with *placed = false: nothing allocated, allocate as before.
static int place_copy(hipStream_t st, bool verbose, double max_seconds, size_t b_vals, size_t b_idx, size_t b_prod, void **vals, void **pc, void **pr, void **prod, bool *placed,
                      PbPlan *report)
{
    *placed = false;
    if (!pool_enabled()) return CUDAMAT_OK;
    const double t0 = now_s();
    PlaceTimer tm(st);
    if (!tm.ok) return CUDAMAT_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return CUDAMAT_OK;
    std::lock_guard<std::mutex> lock(g_place_mu);
    PlaceMemo &m = g_place[dev];
    if (m.gen != pool_generation()) {
        m.gen = pool_generation();
        m.reset();
    }
    const size_t gran = (size_t)2 << 20;
    auto up = [&](size_t x) { return (x + gran - 1) / gran * gran; };
    const size_t a_vals = up(b_vals), a_idx = up(b_idx), a_prod = up(b_prod);
    // [values | column indices | row indices] out of one run of blocks, the product stream out of a run of another class.  (Measured
    // at C4, ms per launch of phase 1 / phase 2: this arrangement 1.645-1.658 / 0.934-0.964; the indices in a third class
    // 1.662-1.683 / 0.935-0.971; the indices in the products' class 1.674-1.715 / 0.957-0.987; unplaced, same box: 1.715-1.752 /
    // 0.964-1.003.)
    const size_t n_prod = (a_prod + kPlaceBlock - 1) / kPlaceBlock;
    const size_t n_vals = (a_vals + 2 * a_idx + kPlaceBlock - 1) / kPlaceBlock;
    const size_t slab_bytes = std::max(kPlaceSlab, (n_prod + n_vals + 1) * kPlaceBlock);
    std::vector<PlaceSlab> slabs;
    PlaceRun r_prod{}, r_vals{};
    float check_ms = 0.f;
    bool found = false, verified = false;
    size_t held = 0;
    while (!found) {
        if (held + slab_bytes > kPlaceMaxHeld || now_s() - t0 > max_seconds) break;
        PlaceSlab sl;
        if (hipMalloc(&sl.base, slab_bytes) != hipSuccess) { CM_DROP(hipGetLastError()); break; }
        sl.bytes = slab_bytes;
        held += slab_bytes;
        char *seg = nullptr;
        size_t blk_bytes = 0;
        if (!pool_segment_of(sl.base, &seg, &blk_bytes)) { slabs.push_back(sl); break; }
        sl.bytes = blk_bytes;
        sl.b0 = seg + ((size_t)(sl.base - seg) + kPlaceBlock - 1) / kPlaceBlock * kPlaceBlock;
        const size_t whole = sl.b0 + kPlaceBlock <= sl.base + sl.bytes ? (size_t)(sl.base + sl.bytes - sl.b0) / kPlaceBlock : 0;
        bool verdicts = true;
        for (size_t i = 0; i < whole && verdicts; i++) {
            const int c = place_class_of(tm, m, sl.b0 + i * kPlaceBlock);
            verdicts = c >= 0;
            sl.cls.push_back(c);
        }
        sl.taken.assign(sl.cls.size(), 0);
        slabs.push_back(sl);
        if (verdicts && m.relabel) {          // provisional labels were withdrawn: every held block once more (the memo answers for most)
            m.relabel = false;
            for (auto &q : slabs)
                for (size_t i = 0; i < q.cls.size() && verdicts; i++) {
                    q.cls[i] = place_class_of(tm, m, q.b0 + i * kPlaceBlock);
                    verdicts = q.cls[i] >= 0;
                }
        }
        if (!verdicts) break;
        // the product stream in a class of its own: try every class met so far for it
        for (size_t a = 0; a < m.ref.size() && !found; a++) {
            for (auto &q : slabs) std::fill(q.taken.begin(), q.taken.end(), 0);
            found = place_find(slabs, n_prod, (int)a, -1, &r_prod) && place_find(slabs, n_vals, -1, (int)a, &r_vals);
        }
    }
    if (found) {
        // cut the slabs: every array becomes a block of the pool (freed on its own as ever), the pieces not wanted go back to it
        struct Keep { size_t slab; char *p; size_t bytes; void **out; };
        char *p_vals = slabs[r_vals.slab].b0 + r_vals.first * kPlaceBlock, *p_prod = slabs[r_prod.slab].b0 + r_prod.first * kPlaceBlock;
        char *p_idx = p_vals + a_vals;
        const size_t s_idx = r_vals.slab;
        const Keep keeps[4] = {{r_vals.slab, p_vals, a_vals, vals}, {s_idx, p_idx, a_idx, pc}, {s_idx, p_idx + a_idx, a_idx, pr}, {r_prod.slab, p_prod, a_prod, prod}};
        for (size_t si = 0; si < slabs.size(); si++) {
            PlaceSlab &sl = slabs[si];
            std::vector<std::pair<char *, bool>> marks, cuts;          // (address, the block that starts there is kept)
            bool base_kept = false;
            for (const Keep &k : keeps) {
                if (k.slab != si) continue;
                marks.push_back({k.p, true});
                marks.push_back({k.p + k.bytes, false});
                base_kept = base_kept || k.p == sl.base;
            }
            std::sort(marks.begin(), marks.end());
            for (size_t k = 0; k < marks.size(); k++) {
                if (k + 1 < marks.size() && marks[k + 1].first == marks[k].first) { marks[k + 1].second = marks[k].second || marks[k + 1].second; continue; }
                if (marks[k].first > sl.base && marks[k].first < sl.base + sl.bytes) cuts.push_back(marks[k]);
            }
            // from the highest address down: every split then acts on the block that still starts at sl.base
            bool ok = true;
            for (size_t k = cuts.size(); k-- > 0 && ok;) ok = pool_split(sl.base, (size_t)(cuts[k].first - sl.base));
            if (!ok) { found = false; break; }           // (cannot happen: every cut is granule-aligned inside the block)
            if (!base_kept) CM_DROP(hipFree(sl.base));
            for (auto &c : cuts)
                if (!c.second) CM_DROP(hipFree(c.first));
            sl.base = nullptr;
        }
        if (found) {
            for (const Keep &k : keeps) *k.out = k.p;
            // the arrangement once more, directly: the values read beside writes into the product stream must time with the fast group
            const float t_check = tm.ms(p_vals, p_prod + kPlaceBlock - kPlaceSample, kPlaceSample);
            *placed = true;          // (the arrays are the caller's now, whatever the check says)
            verified = t_check > 0.f && m.t_hi >= kPlaceGap * m.t_lo && t_check < sqrtf(m.t_lo * m.t_hi);
            check_ms = t_check;
        }
    }
    for (auto &sl : slabs)
        if (sl.base) CM_DROP(hipFree(sl.base));
    std::string seen;
    for (auto &sl : slabs) {
        for (int c : sl.cls) seen += c < 0 ? '?' : (char)('0' + c % 10);
        seen += ' ';
    }
    report->placed = *placed && verified ? 1 : 0;
    report->place_check_ms = check_ms;
    report->place_fast_ms = m.t_lo;
    report->place_slow_ms = m.t_hi;
    report->place_slabs = (int)slabs.size();
    report->place_seconds = now_s() - t0;
    snprintf(report->place_classes, sizeof(report->place_classes), "%s| check %.3f ms, groups %.3f / %.3f ms", seen.c_str(), check_ms, m.t_lo, m.t_hi);
    if (verbose) {
        fprintf(stderr, "[cudamat] pb placement: %s; %zu slab(s) of %.0f GB, blocks by class: %s, %d timed launches, %.1f ms (one block read + written: %.3f ms)\n",
                *placed && verified ? "product stream in a memory class of its own, values and indices in another" : *placed ? "an arrangement was cut, but its own check timed slow (classes misjudged?)" : "no arrangement found: arrays allocated one after the other",
                slabs.size(), (double)slab_bytes / (double)((size_t)1 << 30), seen.c_str(), tm.launches, (now_s() - t0) * 1e3, m.t_own);
    }
    return CUDAMAT_OK;
}

int pb_build_values(hipStream_t st, PbBuild *b, const ValDict *vd)
{
    PbPlan &p = b->p;
    const size_t cap = b->cap;
    const bool dict = vd && vd->n > 0;
    int rc = CUDAMAT_OK;
    do {
        if (!p.P && b->place > 0 && sizeof(double) * cap >= kPlaceMinBytes) {
            void *v = nullptr, *c = nullptr, *r = nullptr, *pp = nullptr;
            bool placed = false;
            if ((rc = place_copy(st, b->verbose, b->place_max_seconds, dict ? cap : sizeof(double) * cap, sizeof(u16) * cap, sizeof(double) * cap, &v, &c, &r, &pp, &placed, &p))) break;
            if (placed) {
                if (dict) p.pvi = (unsigned char *)v; else p.pv = (double *)v;
                p.pc = (u16 *)c;
                p.pr = (u16 *)r;
                p.P = (double *)pp;
            }
        }
        if (dict) {
            if (!p.pvi && (rc = dalloc(&p.pvi, cap))) break;
            if (hipMemsetAsync(p.pvi, 0, cap, st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
            p.dict = vd->dict;
            p.ndict = vd->n;
        } else {
            if (!p.pv && (rc = dalloc(&p.pv, cap))) break;
            if (hipMemsetAsync(p.pv, 0, sizeof(double) * cap, st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        }
        if (p.P && p.pc && p.pr && p.pc_zeroed) break;          // (the drop-in path swaps fp64 values for dictionary indices afterwards: the rest is in place)
        if (!p.pc && (rc = dalloc(&p.pc, cap))) break;
        if (!p.pr && (rc = dalloc(&p.pr, cap))) break;
        if (!p.P && (rc = dalloc(&p.P, cap))) break;
        if (kPbAlign > 1 && hipMemsetAsync(p.pc, 0, sizeof(u16) * cap, st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        p.pc_zeroed = true;
    } while (0);
    if (rc) pb_build_abort(b);
    return rc;
}

// place the entries of the sub-blocks [sub0, sub1) (rows [sub0 * SR, sub1 * SR)): needs rp, ci and val (or vd->idx) of those rows only
int pb_build_fill(hipStream_t st, PbBuild *b, const int *rp, const int *ci, const double *val, const ValDict *vd, int sub0, int sub1)
{
    PbPlan &p = b->p;
    if (sub1 > p.NSUB) sub1 = p.NSUB;
    if (sub1 <= sub0) return CUDAMAT_OK;
    const unsigned grid = (unsigned)((sub1 - sub0 + kPbBuildWaves - 1) / kPbBuildWaves);
    const unsigned char *vidx = p.pvi ? vd->idx : (const unsigned char *)nullptr;
    const int cfg_fill_occ = b->fill_occ;
    if (b->two_pass) {
        // the product stream is free until the first SpMV: it carries the values between the passes
        // (dynamic LDS the kernel never touches caps the resident waves per CU)
        // (PB_FILL_OCC, probing only: measured flat between 4 and 16 waves per CU)
        size_t pad = 0;
        if (cfg_fill_occ > 0) {
            const size_t per_wg = (size_t)(160 * 1024) / (size_t)((cfg_fill_occ + kPbBuildWaves - 1) / kPbBuildWaves);
            pad = per_wg > 40 * 1024 ? per_wg - 40 * 1024 : 0;
            if (pad > 64 * 1024) pad = 64 * 1024;
        }
        hipLaunchKernelGGL(k_pb_group, dim3(grid), dim3(64 * kPbBuildWaves), pad, st, p.n, rp, ci, val, vidx, b->cut, p.col0, p.NCB, p.SR,
                           p.slen, b->GB, b->NG, b->gstart, p.P, p.pvi ? (unsigned char *)p.P : (unsigned char *)nullptr, b->smeta, sub0, sub1);
        const long long buckets = (long long)(sub1 - sub0) * b->NG;
        hipLaunchKernelGGL(k_pb_scatter, dim3((unsigned)((buckets + kPbScatterWaves - 1) / kPbScatterWaves)), dim3(64 * kPbScatterWaves), 0, st,
                           p.NCB, p.NSUB, p.slen, b->bins, b->GB, b->NG, b->gstart, p.P, p.pvi ? (const unsigned char *)p.P : (const unsigned char *)nullptr,
                           b->smeta, p.pv, p.pvi, p.pc, p.pr, sub0, sub1);
    } else {
        const size_t lds = sizeof(int) * (size_t)kPbBuildWaves * p.NCB;
        hipLaunchKernelGGL(k_pb_rows<true>, dim3(grid), dim3(64 * kPbBuildWaves), lds, st, p.n, rp, ci, val, b->cut, p.col0,
                           p.NCB, p.SR, p.NSUB, b->bins, p.pv, p.pc, p.pr, vidx, p.pvi, sub0, sub1);
    }
    if (hipGetLastError() != hipSuccess) { set_error("pb fill launch failed"); pb_build_abort(b); return CUDAMAT_ERR_HIP; }
    return CUDAMAT_OK;
}

// CUDAMAT_VERBOSE: the plan and how evenly the entries fall into its (sub-block, column block) segments: phase 2 is bound by
// fabric requests per segment, so a layout whose segments are much shorter or longer than the 48-entry target pays for it.
// (Printed when the copy is complete, not from the count pass: the 38 MB read-back at C4 would sit in the way of an upload
// that the drop-in call runs beside the build, and of the stage stamps that measure it.)
static void pb_print_plan(const PbPlan &p)
{
    const size_t nbins = (size_t)p.NCB * p.NSUB;
    std::vector<int> hl(nbins);
    if (hipMemcpy(hl.data(), p.slen, sizeof(int) * nbins, hipMemcpyDeviceToHost) != hipSuccess) return;
    long long hist[7] = {0, 0, 0, 0, 0, 0, 0}, in_long = 0;
    for (int v : hl) {
        hist[v == 0 ? 0 : v <= 8 ? 1 : v <= 24 ? 2 : v <= 48 ? 3 : v <= 64 ? 4 : v <= 128 ? 5 : 6]++;
        if (v > 64) in_long += v;
    }
    fprintf(stderr, "[cudamat] pb arrays: pv %p pvi %p pc %p pr %p P %p sstart %p slen %p cstart %p\n", (void *)p.pv, (void *)p.pvi, (void *)p.pc,
            (void *)p.pr, (void *)p.P, (void *)p.sstart, (void *)p.slen, (void *)p.cstart);
    fprintf(stderr, "[cudamat] pb plan: %d x %lld, nnz %lld, NCB %d (CB %d), NSUB %d (NW %d, SR %d), LPS %d, depth %d; segments: "
                    "empty %lld, 1-8 %lld, 9-24 %lld, 25-48 %lld, 49-64 %lld, 65-128 %lld, >128 %lld; %.1f %% of the entries in "
                    "segments longer than a wave\n",
            p.n, (long long)p.n_cols, (long long)p.nnz, p.NCB, p.CB, p.NSUB, p.NW, p.SR, p.LPS, p.depth, hist[0], hist[1], hist[2], hist[3],
            hist[4], hist[5], hist[6], 100.0 * (double)in_long / (double)(p.nnz > 0 ? p.nnz : 1));
}

// VERBOSE >= 2 (diagnosis): the memory class of every ~1 GB of the copy's four big arrays, found from the READ side (the arrays
// are in use): the region is read while a free block of a known class is written; slow = that class.  Needs one free block per
// class: slabs are taken from the pool until three classes (or 64 GB) have been met, and given back.
static void place_report_classes(hipStream_t st, const PbPlan &p, size_t cap)
{
    if (!pool_enabled()) return;
    PlaceTimer tm(st);
    if (!tm.ok) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_place_mu);
    PlaceMemo &m = g_place[dev];
    if (m.gen != pool_generation()) { m.gen = pool_generation(); m.reset(); }
    std::vector<char *> slabs;
    std::vector<char *> scratch(3, nullptr);          // a writable block per class
    for (int k = 0; k < 4 && !(scratch[0] && scratch[1] && scratch[2]); k++) {
        char *base = nullptr;
        if (hipMalloc(&base, kPlaceSlab) != hipSuccess) { CM_DROP(hipGetLastError()); break; }
        slabs.push_back(base);
        char *seg = nullptr;
        size_t bytes = 0;
        if (!pool_segment_of(base, &seg, &bytes)) break;
        char *b0 = seg + ((size_t)(base - seg) + kPlaceBlock - 1) / kPlaceBlock * kPlaceBlock;
        for (char *q = b0; q + kPlaceBlock <= base + bytes; q += kPlaceBlock) {
            const int c = place_class_of(tm, m, q);
            if (c >= 0 && c < 3 && !scratch[c]) scratch[c] = q;
        }
    }
    struct Arr { const char *name; const char *ptr; size_t bytes; };
    const Arr arrs[4] = {{"values", p.pv ? (const char *)p.pv : (const char *)p.pvi, p.pv ? sizeof(double) * cap : cap},
                         {"col idx", (const char *)p.pc, sizeof(u16) * cap}, {"row idx", (const char *)p.pr, sizeof(u16) * cap}, {"products", (const char *)p.P, sizeof(double) * cap}};
    for (const Arr &a : arrs) {
        std::string line;
        for (size_t off = 0; off + kPlaceSample <= a.bytes; off += kPlaceSample) {
            int cls = -1;
            float worst = 0.f;
            for (int k = 0; k < 3; k++) {
                if (!scratch[k]) continue;
                const float t = tm.ms(a.ptr + off, scratch[k] + kPlaceBlock - kPlaceSample, kPlaceSample);
                if (t > worst) { worst = t; cls = k; }
            }
            line += cls < 0 || !(m.t_hi >= kPlaceGap * m.t_lo) || worst <= sqrtf(m.t_lo * m.t_hi) ? '?' : (char)('0' + cls);
        }
        fprintf(stderr, "[cudamat] pb classes: %-8s at %p, per %zu MB: %s\n", a.name, (const void *)a.ptr, kPlaceSample >> 20, line.c_str());
    }
    for (char *q : slabs) CM_DROP(hipFree(q));
}

int pb_build_end(hipStream_t st, PbBuild *b, PbPlan *out)
{
    if (hipStreamSynchronize(st) != hipSuccess) { set_error("pb fill failed"); pb_build_abort(b); return CUDAMAT_ERR_HIP; }
    if (b->verbose) pb_print_plan(b->p);
    if (b->report_classes && sizeof(double) * b->cap >= kPlaceMinBytes) place_report_classes(st, b->p, b->cap);
    pb_scratch_free(b);
    CM_DROP(hipFree(b->bins));
    b->bins = nullptr;
    b->p.build_seconds = now_s() - b->t0;
    *out = b->p;
    b->p = PbPlan();
    return CUDAMAT_OK;
}

int pb_build(hipStream_t st, const Config &cfg, int n, int64_t n_cols, int64_t nnz, const int *rp, const int *ci,
             const double *val, PbPlan *out, const PbCols *cols, const ValDict *vd)
{
    PbBuild b;
    CM_TRY(pb_build_begin(st, cfg, n, n_cols, nnz, rp, ci, cols, &b));
    CM_TRY(pb_build_values(st, &b, vd));
    CM_TRY(pb_build_fill(st, &b, rp, ci, val, vd, 0, b.p.NSUB));
    return pb_build_end(st, &b, out);
}


// ------------------------------------------------------------------ phase 1
// The streaming loop: two entries per lane per step (16-byte value loads), four steps in flight, every bound checked.
// gfx950 has ONE counter (vmcnt) for loads and stores, and the compiler drains it before every use of a loaded value
// while a store is pending, so this loop's queue empties four times per iteration.  The remedy -- loads and waits as
// inline assembly, `s_waitcnt vmcnt(4)`, 8 loads AND 4 stores in flight per wave at all times -- was built in round 3
// and measured 7 % SLOWER (as the chunked body of the dictionary kernel is on fp64 values, 10 %): the launch is not bound
// by its waves' round trips; its time is close to reading its input (5.3 GB at ~6 TB/s) plus writing its output (4 GB
// at ~6 TB/s) one after the other, which is how the memory system treats a 10-read : 8-write mix (HISTORY.md round 3;
// the variant is in the history of this file, commit 5f8463c and before).
__global__ __launch_bounds__(kP1Threads) void k_pb_phase1(const double *x, const int *col0, const int *list,
                                                          const int *cstart, const double *pv, const u16 *pc,
                                                          double *P, const LoopState *st, int cb0)
{
    extern __shared__ __attribute__((aligned(16))) double xs[];
    if (st && st->state != 0) return;
    const int cb = list ? list[blockIdx.x] : (int)blockIdx.x + cb0;
    const int s = cstart[cb], e = cstart[cb + 1];
    if (s >= e) return;                                    // (block-uniform) nothing stored in this block
    const int c0 = col0[cb], cn = col0[cb + 1] - c0;
    for (int i = threadIdx.x; i < cn; i += kP1Threads) xs[i] = x[c0 + i];
    __syncthreads();
    constexpr int STEP = 2 * kP1Threads;
    const int k0 = (s & ~1) + 2 * (int)threadIdx.x;        // pairs at even entries (16-byte aligned)
    for (int k = k0; k < e; k += 4 * STEP) {
        double2 v[4];
        ushort2 c[4];
        bool full[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int kk = k + u * STEP;
            full[u] = kk >= s && kk + 1 < e;
            if (full[u]) {
                v[u] = *(const double2 *)(pv + kk);
                c[u] = *(const ushort2 *)(pc + kk);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int kk = k + u * STEP;
            if (full[u]) {
                double2 o;
                o.x = v[u].x * xs[c[u].x];
                o.y = v[u].y * xs[c[u].y];
                *(double2 *)(P + kk) = o;
            } else {
                if (kk >= s && kk < e) P[kk] = pv[kk] * xs[pc[kk]];
                if (kk + 1 >= s && kk + 1 < e) P[kk + 1] = pv[kk + 1] * xs[pc[kk + 1]];
            }
        }
    }
}

// phase 1 for a matrix with a value dictionary: 8-bit value indices (1 byte per entry instead of 8), the dictionary
// beside the x tile in LDS; same products bit for bit: dict[idx] IS the stored value.
//
// Shape of the streaming loop (round 2, from the ISA of its predecessor): on gfx9-family parts loads and stores share
// ONE counter (vmcnt) and complete out of order with respect to each other, so the compiler waits for vmcnt(0) -- all of
// a wave's earlier stores included -- before it uses any loaded value while stores are pending, and a loop whose steps
// sit behind per-step bounds checks is compiled into load / wait / load / wait.  With one 1024-thread workgroup per
// compute unit (the x tile fills the LDS) that left every wave with ~0.5 KB of loads and 4 KB of stores in flight per
// memory round trip, and the launch bound by LATENCY (1.45-1.8 ms with 5.5 GB to move; stores removed: 0.6 ms).  So:
// a branch-free body over whole chunks of U = 8 steps -- 2U loads issued back to back, ONE wait (which the previous
// chunk's stores overlap), U product pairs, U 16-byte stores per lane -- and the bounds checks confined to the head and
// the tail: 1.20-1.29 ms (U = 4 / 8 / 16 within 5 %), i.e. 5.5 GB at the 4.3-4.6 TB/s a write-heavy mix reaches.
// (The same loop shape makes k_pb_phase1, which reads 10 B per entry, 10 % SLOWER -- 2.05 ms against 1.87 on one box
// for U = 2, 4, 8 alike: bursts of reads followed by bursts of writes suit the memory system less than its
// interleaved steps -- so that kernel keeps its loop.)
__global__ __launch_bounds__(kP1Threads) void k_pb_phase1_dict(const double *x, const int *col0, const int *list,
                                                               const int *cstart, const unsigned char *pvi, const u16 *pc,
                                                               const double *dict, int cb_doubles, double *P,
                                                               const LoopState *st, int cb0)
{
    extern __shared__ __attribute__((aligned(16))) double xs[];
    if (st && st->state != 0) return;
    const int cb = list ? list[blockIdx.x] : (int)blockIdx.x + cb0;
    const int s = cstart[cb], e = cstart[cb + 1];
    if (s == e) return;                                    // (block-uniform) nothing stored in this block
    double *dv = xs + cb_doubles;
    const int c0 = col0[cb], cn = col0[cb + 1] - c0;
    for (int i = threadIdx.x; i < cn; i += kP1Threads) xs[i] = x[c0 + i];
    if (threadIdx.x < kDictMax) dv[threadIdx.x] = dict[threadIdx.x];
    __syncthreads();
    constexpr int U = kP1Unroll, STEP = 2 * kP1Threads, CH = U * STEP;
    const int a0 = (s + 1) & ~1;                           // pairs start at an even entry (16-byte aligned products)
    if (threadIdx.x == 0 && s < a0) P[s] = dv[pvi[s]] * xs[pc[s]];
    int kb = a0;                                           // (workgroup-uniform) first entry of the current chunk
    const int t2 = 2 * (int)threadIdx.x;
    for (; kb + CH <= e; kb += CH) {                       // whole chunks: no bounds checks
        const int k = kb + t2;
        u16 iv[U];
        unsigned cv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            iv[u] = *(const u16 *)(pvi + k + u * STEP);
            cv[u] = *(const unsigned *)(pc + k + u * STEP);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            double2 o;
            o.x = dv[iv[u] & 0xffu] * xs[cv[u] & 0xffffu];
            o.y = dv[iv[u] >> 8] * xs[cv[u] >> 16];
            *(double2 *)(P + k + u * STEP) = o;
        }
    }
    {                                                      // the last, partial chunk: m whole steps (uniform) ...
        const int m = e > kb ? (e - kb) / STEP : 0;
        const int k = kb + t2;
        u16 iv[U - 1];
        unsigned cv[U - 1];
#pragma unroll
        for (int u = 0; u < U - 1; u++) {
            if (u < m) {
                iv[u] = *(const u16 *)(pvi + k + u * STEP);
                cv[u] = *(const unsigned *)(pc + k + u * STEP);
            }
        }
#pragma unroll
        for (int u = 0; u < U - 1; u++) {
            if (u < m) {
                double2 o;
                o.x = dv[iv[u] & 0xffu] * xs[cv[u] & 0xffffu];
                o.y = dv[iv[u] >> 8] * xs[cv[u] >> 16];
                *(double2 *)(P + k + u * STEP) = o;
            }
        }
        const int kr = k + m * STEP;                       // ... and less than one step, checked per lane
        if (kr + 1 < e) {
            const u16 iw = *(const u16 *)(pvi + kr);
            const unsigned cw = *(const unsigned *)(pc + kr);
            double2 o;
            o.x = dv[iw & 0xffu] * xs[cw & 0xffffu];
            o.y = dv[iw >> 8] * xs[cw >> 16];
            *(double2 *)(P + kr) = o;
        } else if (kr < e) {
            P[kr] = dv[pvi[kr]] * xs[pc[kr]];
        }
    }
}

// ------------------------------------------------------------------ phase 2
struct Pb2Args {
    int n, NCB, SR;
    const int *sstart, *slen;
    const double *P;
    const u16 *pr;
    const double *d, *xd;
    double alpha, beta;
    double *y;
    int dot;
    const double *w;
    double *parts;
    const LoopState *st;
    int strict;      // 1: a row's products of one wave instruction are added by SEPARATE instructions, rank by rank
};

// One wave instruction's worth of products into the wave's y tile.  Equal rows sit in ADJACENT active lanes (a segment is
// sorted by row, then column), and a row must receive its products in column order, one rounding per addition.
//  * default: one ds_add_f64 instruction, relying on the OBSERVED property that the LDS serves equal addresses of one
//    instruction in lane order.
//  * strict (option PB_STRICT = 1; round 3): lanes are ranked inside their run of equal rows and every rank is its own
//    instruction -- within an instruction all addresses are distinct, and a wave's LDS instructions execute in issue
//    order, so the order of the additions is ARCHITECTED.  Bit-identical results on gfx950 (the guard test runs both
//    against the oracle) at a price: +10 % per C4 SpMV (2.70 -> 2.97 ms), +25 % per iteration of a G = 8 rank -- phase 2
//    feels every extra LDS-pipe instruction (two lane shifts, a ballot and a second add per segment step).  Hence an
//    option: the form to switch to should a future part serve equal addresses in another order.
// `group_first`: first lane of this lane's segment inside the instruction (0 when one segment fills the wave).
__device__ __forceinline__ void pb_seg_add(double *my, int r, double p, bool on, int lane, int group_first, bool strict)
{
    if (!strict) {
        if (on) unsafeAtomicAdd(&my[r], p);
        return;
    }
    // (the lane below through ds_bpermute; a DPP wave_shr:1 move was tried instead: wrong neighbours across the 16-lane
    // rows on gfx950 and no faster)
    const int r_prev = __shfl_up(r, 1, 64);
    const int on_prev = __shfl_up((int)on, 1, 64);
    const bool d1 = on && lane > group_first && on_prev && r_prev == r;      // same row as the lane before
    if (on && !d1) unsafeAtomicAdd(&my[r], p);                               // rank 0: distinct addresses
    if (__ballot(d1) == 0) return;
    const bool d2 = d1 && __shfl_up((int)d1, 1, 64) != 0;                    // ... and as the lane before that: rank >= 2
    if (d1 && !d2) unsafeAtomicAdd(&my[r], p);                               // rank 1
    if (__ballot(d2) == 0) return;
    // longer runs (few column blocks, or clustered columns): rank = lane - first lane of the run (max-scan of the run
    // heads), then one instruction per rank
    int head = d1 ? 0 : lane;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(head, o, 64);
        if (lane >= o) head = t > head ? t : head;
    }
    const int rank = on ? lane - head : 0;
    for (int q = 2; __ballot(on && rank >= q); q++)
        if (on && rank == q) unsafeAtomicAdd(&my[r], p);
}

// ---- run-time guard of the default form's assumption.  One wave, four address patterns (all lanes on one address; runs
// of 8 adjacent lanes; lanes l and l + 32 -- two 32-lane segments in one instruction; l mod 4 -- 16-lane segments), four
// value sets each: every address starts from a non-zero partial sum and receives its addends through ONE ds_add_f64
// instruction; one lane per address then adds the same addends sequentially in lane order and compares the bits.  Addends
// have full 52-bit mantissas around 1, so another order of service changes the rounding of some partial sum with
// overwhelming probability (64 addresses x 16 rounds).  Any mismatch: the context uses the architected-order form.
__global__ __launch_bounds__(64) void k_lds_order_probe(int *mismatch)
{
    __shared__ double acc[64], val[64], start[64];
    const int lane = threadIdx.x;
    unsigned long long z = 0x9E3779B97F4A7C15ULL * (unsigned long long)(lane + 1);
    int bad = 0;
    for (int pat = 0; pat < 4; pat++) {
        const int addr = pat == 0 ? 0 : pat == 1 ? lane >> 3 : pat == 2 ? (lane & 31) : (lane & 3);
        const int naddr = pat == 0 ? 1 : pat == 1 ? 8 : pat == 2 ? 32 : 4;
        for (int rep = 0; rep < 4; rep++) {
            z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 27; z *= 0x94D049BB133111EBULL; z ^= z >> 31;
            const double v = 0.5 + (double)(z >> 11) * 0x1.0p-53;                  // [0.5, 1.5), all mantissa bits in play
            const double c = 3.0 + (double)((z * 0x9E3779B97F4A7C15ULL) >> 11) * 0x1.0p-53;
            val[lane] = v;
            start[lane] = c;
            acc[lane] = c;
            __syncthreads();
            unsafeAtomicAdd(&acc[addr], v);                                        // the instruction under test
            __syncthreads();
            if (lane < naddr) {
#pragma clang fp contract(off)
                double e = start[lane];
                for (int l = 0; l < 64; l++) {
                    const int al = pat == 0 ? 0 : pat == 1 ? l >> 3 : pat == 2 ? (l & 31) : (l & 3);
                    if (al == lane) e = e + val[l];
                }
                if (__double_as_longlong(e) != __double_as_longlong(acc[lane])) bad = 1;
            }
            __syncthreads();
        }
    }
    if (bad) atomicOr(mismatch, 1);
}

int pb_strict_for(cudamat_ctx *ctx, int *strict)
{
    *strict = 1;
    if (ctx->cfg.pb_strict) return CUDAMAT_OK;
    if (ctx->lds_lane_order < 0) {
        int *flag = (int *)ctx->scratch, h = 1;
        CM_HIP(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
        hipLaunchKernelGGL(k_lds_order_probe, dim3(1), dim3(64), 0, ctx->stream, flag);
        CM_HIP(hipGetLastError());
        CM_HIP(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        CM_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->cfg.pb_probe_fail) h = 1;
        ctx->lds_lane_order = h ? 0 : 1;
        if (h || ctx->cfg.verbose)
            fprintf(stderr, "[cudamat] LDS order probe: equal addresses of one ds_add_f64 are %s -> blocked SpMV phase 2 in its %s form\n",
                    h ? (ctx->cfg.pb_probe_fail ? "reported NOT in lane order (PB_PROBE_FAIL)" : "NOT served in lane order")
                      : "served in lane order", h ? "architected-order (PB_STRICT)" : "default");
    }
    *strict = ctx->lds_lane_order ? 0 : 1;
    return CUDAMAT_OK;
}

// DEPTH: segment loads a wave issues before it consumes the first (4; 8 / 16 when few waves are resident: shards)
template <int NW, int LPS, int DEPTH>
__global__ __launch_bounds__(64 * NW) void k_pb_phase2(Pb2Args a)
{
    extern __shared__ __attribute__((aligned(16))) double yt[];   // NW * SR, then 2 * NW for the reduction
    if (a.st && a.st->state != 0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub = blockIdx.x * NW + wave;
    double *my = yt + (size_t)wave * a.SR;
    for (int i = lane; i < a.SR; i += 64) my[i] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int *ss = a.sstart + (size_t)sub * a.NCB;
    const int *sl = a.slen + (size_t)sub * a.NCB;
    for (int cb0 = 0; cb0 < a.NCB; cb0 += 64) {
        const int c = cb0 + lane;
        const int mys = c < a.NCB ? ss[c] : 0;
        const int myl = c < a.NCB ? sl[c] : 0;
        const int lim = a.NCB - cb0 < 64 ? a.NCB - cb0 : 64;
        // LPS < 64: 64/LPS segments share one wave instruction (lane group g takes column block j+g).
        // Groups are ordered by column block and ds_add_f64 serves equal addresses in lane order, so a
        // row still receives its products in increasing column order -- unless a segment is longer than
        // LPS (its tail would land after the next block's head): such chunks take the 64-lane path.
        const bool grouped = LPS < 64 && !__any(myl > LPS);
        if (!grouped) {
            for (int j = 0; j < lim; j += DEPTH) {
                int s[DEPTH], l[DEPTH];
                double pv4[DEPTH];
                int r4[DEPTH];
#pragma unroll
                for (int u = 0; u < DEPTH; u++) {
                    // lanes past `lim` carry length 0, so j + u may safely run to 63
                    s[u] = __builtin_amdgcn_readlane(mys, (j + u) & 63);
                    l[u] = __builtin_amdgcn_readlane(myl, (j + u) & 63);
                }
#pragma unroll
                for (int u = 0; u < DEPTH; u++) {
                    const bool on = lane < l[u];
                    pv4[u] = on ? a.P[s[u] + lane] : 0.0;
                    r4[u] = on ? (int)a.pr[s[u] + lane] : 0;
                }
#pragma unroll
                for (int u = 0; u < DEPTH; u++) {
                    pb_seg_add(my, r4[u], pv4[u], lane < l[u], lane, 0, a.strict != 0);
                    for (int off0 = 64; off0 < l[u]; off0 += 64) {         // segments longer than a wave (uniform trip count)
                        const int off = off0 + lane;
                        const bool on = off < l[u];
                        // (a run of equal rows that crosses the 64-entry cut continues in the next instruction: later in
                        // program order, hence after it)
                        pb_seg_add(my, on ? (int)a.pr[s[u] + off] : 0, on ? a.P[s[u] + off] : 0.0, on, lane, 0, a.strict != 0);
                    }
                }
            }
        } else {
            constexpr int G = 64 / LPS;
            const int g = lane / LPS, li = lane % LPS;
            for (int j = 0; j < lim; j += DEPTH * G) {
                int s[DEPTH], l[DEPTH];
                double pv4[DEPTH];
                int r4[DEPTH];
#pragma unroll
                for (int u = 0; u < DEPTH; u++) {
                    const int cbi = j + u * G + g;
                    s[u] = __shfl(mys, cbi & 63, 64);
                    l[u] = cbi < 64 ? __shfl(myl, cbi & 63, 64) : 0;
                }
#pragma unroll
                for (int u = 0; u < DEPTH; u++) {
                    const bool on = li < l[u];
                    pv4[u] = on ? a.P[s[u] + li] : 0.0;
                    r4[u] = on ? (int)a.pr[s[u] + li] : 0;
                }
#pragma unroll
                for (int u = 0; u < DEPTH; u++) {
                    if (!a.strict) {
                        if (li < l[u]) unsafeAtomicAdd(&my[r4[u]], pv4[u]);
                    } else {
                        // strict: the lane groups (= consecutive column blocks) one after the other, each ranked by itself --
                        // a row that occurs in two groups gets the lower column block's product first by program order
#pragma unroll
                        for (int gg = 0; gg < G; gg++)
                            pb_seg_add(my, r4[u], pv4[u], g == gg && li < l[u], lane, gg * LPS, true);
                    }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // epilogue: this wave's rows
    double acc0 = 0.0, acc1 = 0.0;
    const long long row0 = (long long)sub * a.SR;
    for (int i = lane; i < a.SR && row0 + i < a.n; i += 64) {
#pragma clang fp contract(off)   // one rounding per product and per sum, like cusparse's mult_spec + csrmv(beta=1)
        const int row = (int)(row0 + i);
        double sum = my[i];
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
            acc0 += out * a.w[row];
            acc1 += out * out;
        }
    }
    if (a.dot) {
        double *red = yt + (size_t)NW * a.SR;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            acc0 += __shfl_xor(acc0, o, 64);
            acc1 += __shfl_xor(acc1, o, 64);
        }
        if (lane == 0) {
            red[2 * wave] = acc0;
            red[2 * wave + 1] = acc1;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t0 = 0.0, t1 = 0.0;
            for (int q = 0; q < NW; q++) {
                t0 += red[2 * q];
                t1 += red[2 * q + 1];
            }
            a.parts[2 * blockIdx.x] = t0;
            a.parts[2 * blockIdx.x + 1] = t1;
        }
    }
}

int launch_pb_check(hipStream_t st, const SpmvArgs &a)
{
    if (a.loop.st && a.check == CHECK_HALF) CM_TRY(launch_check(st, a.loop, a.half, CHECK_HALF));
    return CUDAMAT_OK;
}

// phase 1 of the column blocks [first, last) in index order (list == nullptr) or of order[first .. last)
static int launch_pb_phase1_blocks(hipStream_t st, const PbPlan &p, const SpmvArgs &a, const int *list, int first, int last)
{
    if (last <= first) return CUDAMAT_OK;
    if (p.pvi) {
        CM_TRY(set_max_lds((const void *)k_pb_phase1_dict));
        const int cbd = (p.CB + 1) & ~1;                     // the dictionary starts 16-byte aligned behind the x tile
        hipLaunchKernelGGL(k_pb_phase1_dict, dim3(last - first), dim3(kP1Threads), sizeof(double) * (size_t)(cbd + kDictMax), st, a.x,
                           p.col0, list ? list + first : (const int *)nullptr, p.cstart, p.pvi, p.pc, p.dict, cbd, p.P, a.loop.st,
                           list ? 0 : first);
        CM_HIP(hipGetLastError());
        return CUDAMAT_OK;
    }
    CM_TRY(set_max_lds((const void *)k_pb_phase1));
    hipLaunchKernelGGL(k_pb_phase1, dim3(last - first), dim3(kP1Threads), sizeof(double) * (size_t)p.CB, st, a.x, p.col0,
                       list ? list + first : (const int *)nullptr, p.cstart, p.pv, p.pc, p.P, a.loop.st, list ? 0 : first);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

int launch_pb_phase1(hipStream_t st, const PbPlan &p, const SpmvArgs &a, int part)
{
    // part < 0: every block in index order; otherwise the blocks of one launch part (PbPlan::order)
    if (part < 0) return launch_pb_phase1_blocks(st, p, a, nullptr, 0, p.NCB);
    return launch_pb_phase1_blocks(st, p, a, p.order, p.part_off[part], p.part_off[part + 1]);
}

int launch_spmv_pb(hipStream_t st, const PbPlan &p, const SpmvArgs &a)
{
    CM_TRY(launch_pb_check(st, a));
    CM_TRY(launch_pb_phase1(st, p, a, -1));
    return launch_pb_phase2(st, p, a);
}

int launch_pb_phase2(hipStream_t st, const PbPlan &p, const SpmvArgs &a)
{
    Pb2Args b;
    b.n = p.n; b.NCB = p.NCB; b.SR = p.SR;
    b.sstart = p.sstart; b.slen = p.slen;
    b.P = p.P; b.pr = p.pr;
    b.d = a.d; b.xd = a.xd;
    b.alpha = a.alpha; b.beta = a.beta;
    b.y = a.y; b.dot = a.dot; b.w = a.w; b.parts = a.parts;
    b.st = a.loop.st;
    b.strict = a.pb_strict;
    const size_t lds = sizeof(double) * ((size_t)p.NW * p.SR + 2 * (size_t)p.NW);
#define CM_P2D(NWV, LPSV, DV)                                                                             \
    do {                                                                                                  \
        CM_TRY(set_max_lds((const void *)k_pb_phase2<NWV, LPSV, DV>));                                   \
        hipLaunchKernelGGL((k_pb_phase2<NWV, LPSV, DV>), dim3(p.NRB), dim3(64 * NWV), lds, st, b);       \
    } while (0)
#define CM_P2(NWV, LPSV)                                                                                  \
    do {                                                                                                  \
        if (p.depth >= 16) CM_P2D(NWV, LPSV, 16);                                                         \
        else if (p.depth >= 8) CM_P2D(NWV, LPSV, 8);                                                      \
        else CM_P2D(NWV, LPSV, 4);                                                                        \
    } while (0)
#define CM_P2_LPS(NWV)                                           \
    do {                                                         \
        if (p.LPS == 16) CM_P2(NWV, 16);                         \
        else if (p.LPS == 32) CM_P2(NWV, 32);                    \
        else CM_P2(NWV, 64);                                     \
    } while (0)
    switch (p.NW) {
    case 4:  CM_P2_LPS(4); break;
    case 8:  CM_P2_LPS(8); break;
    default: CM_P2_LPS(16); break;
    }
#undef CM_P2_LPS
#undef CM_P2
#undef CM_P2D
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

}  // namespace cm
