// spmv_sell.hip -- SELL-C-sigma SpMV, C = 64 = one wavefront per chunk.
//
// For matrices whose row lengths vary moderately (Matrix Market inputs): rows are sorted by length inside
// windows of sigma = 1024 consecutive rows (stable, so equal lengths keep their order and a uniform matrix
// keeps its row order), cut into chunks of 64 rows, and each chunk is stored COLUMN-major, padded to its
// longest row: entry j of the chunk's lane-th row sits at chunk_off * 64 + j * 64 + lane.  One lane owns one
// row: the 64 lanes of a wave load 64 consecutive values / column ids per step (coalesced, no LDS, no
// cross-lane reduction) and a row's products are added by ONE lane in increasing column order, one rounding per
// product and per sum -- the rounding sequence of the reference CPU loop `b[i] += A.Value[j] * x[A.Col[j]]`
// (bicstab_omp/bicstab.cpp:72-77): bit-identical to the oracle on real-valued data.  Padding slots are never
// multiplied (a lane stops at its own length), so non-finite x entries of other rows cannot leak in.
// The window sort bounds both the padding and the distance between a row and its output (good for banded x).
// Chosen per matrix by timing it against the other forms (solver.hip, ensure_spmv_mode).
#include <algorithm>
#include <chrono>
#include <vector>

#include "spmv_sell.h"

namespace cm {

constexpr int kSigma = 1024;          // rows per sorting window (16 chunks)
constexpr int kChunk = 64;

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

void sell_free(SellPlan *p)
{
    void *ptrs[] = {p->val, p->col, p->perm, p->len, p->chunk_off};
    for (void *q : ptrs)
        if (q) CM_DROP(hipFree(q));
    *p = SellPlan();
}

// one workgroup per window: sort (length descending, row ascending) with a bitonic network in LDS
__global__ __launch_bounds__(kBlock) void k_sell_sort(int n, const int *rp, int *perm, int *len, int *width)
{
    __shared__ unsigned long long key[kSigma];
    const int w0 = blockIdx.x * kSigma;
    for (int i = threadIdx.x; i < kSigma; i += kBlock) {
        const int row = w0 + i;
        // larger key first: length in the high word, (sigma - 1 - index) in the low word; rows past the end sort last
        key[i] = row < n ? ((unsigned long long)(unsigned)(rp[row + 1] - rp[row]) << 32) | (unsigned)(kSigma - 1 - i) | (1ull << 63)
                         : (unsigned long long)(unsigned)(kSigma - 1 - i);
    }
    __syncthreads();
    for (int k = 2; k <= kSigma; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < kSigma; t += kBlock) {
                const int partner = t ^ j;
                if (partner > t) {
                    const bool desc = (t & k) == 0;
                    const unsigned long long a = key[t], b = key[partner];
                    if (desc ? a < b : a > b) { key[t] = b; key[partner] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < kSigma; i += kBlock) {
        const unsigned long long kk = key[i];
        const bool real = (kk >> 63) != 0;
        const int idx = kSigma - 1 - (int)(unsigned)(kk & 0xffffffffull);
        const int l = real ? (int)((kk >> 32) & 0x7fffffffull) : 0;
        perm[w0 + i] = real ? w0 + idx : -1;
        len[w0 + i] = l;
        if ((i & (kChunk - 1)) == 0) width[(w0 + i) / kChunk] = l;       // the chunk's longest row comes first
    }
}

// one wave per chunk: copy the rows' entries into the column-major chunk
__global__ __launch_bounds__(kBlock) void k_sell_fill(int nchunks, const int *rp, const int *ci, const double *val,
                                                      const int *perm, const int *len, const long long *chunk_off,
                                                      double *sval, int *scol)
{
    const int c = blockIdx.x * (kBlock / kChunk) + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= nchunks) return;
    const int row = perm[(size_t)c * kChunk + lane], l = len[(size_t)c * kChunk + lane];
    const long long base = chunk_off[c] * kChunk + lane;
    const int width = (int)(chunk_off[c + 1] - chunk_off[c]);
    const int s = row >= 0 ? rp[row] : 0;
    for (int j = 0; j < width; j++) {
        const bool have = j < l;
        sval[base + (long long)j * kChunk] = have ? val[s + j] : 0.0;
        scol[base + (long long)j * kChunk] = have ? ci[s + j] : 0;
    }
}

int sell_build(hipStream_t st, int n, int64_t nnz, const int *rp, const int *ci, const double *val, SellPlan *out, double max_fill)
{
    const double t0 = now_s();
    SellPlan p;
    p.n = n;
    p.nnz = nnz;
    const int nwin = (n + kSigma - 1) / kSigma;
    p.nchunks = nwin * (kSigma / kChunk);
    int *width = nullptr;
    int rc = CUDAMAT_OK;
    do {
        if (n <= 0) { rc = CUDAMAT_ERR_ARG; set_error("sell_build: empty matrix"); break; }
        if ((rc = dalloc(&p.perm, (size_t)p.nchunks * kChunk))) break;
        if ((rc = dalloc(&p.len, (size_t)p.nchunks * kChunk))) break;
        if ((rc = dalloc(&p.chunk_off, (size_t)p.nchunks + 1))) break;
        if ((rc = dalloc(&width, (size_t)p.nchunks))) break;
        hipLaunchKernelGGL(k_sell_sort, dim3(nwin), dim3(kBlock), 0, st, n, rp, p.perm, p.len, width);
        std::vector<int> hw((size_t)p.nchunks);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(hw.data(), width, sizeof(int) * hw.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("sell sort failed"); break; }
        std::vector<long long> off((size_t)p.nchunks + 1, 0);
        for (int c = 0; c < p.nchunks; c++) off[(size_t)c + 1] = off[(size_t)c] + hw[(size_t)c];
        p.slots = off[(size_t)p.nchunks] * kChunk;
        p.fill = nnz > 0 ? (double)p.slots / (double)nnz : 1.0;
        if (max_fill > 0.0 && p.fill > max_fill) {            // too much padding: not worth a copy (not an error)
            rc = CUDAMAT_ERR_ARG;
            set_error("sell_build: padded copy would hold %.2f x the entries", p.fill);
            break;
        }
        if (hipMemcpyAsync(p.chunk_off, off.data(), sizeof(long long) * off.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        if ((rc = dalloc(&p.val, (size_t)p.slots))) break;
        if ((rc = dalloc(&p.col, (size_t)p.slots))) break;
        hipLaunchKernelGGL(k_sell_fill, dim3((unsigned)((p.nchunks + 3) / 4)), dim3(kBlock), 0, st, p.nchunks, rp, ci, val,
                           p.perm, p.len, p.chunk_off, p.val, p.col);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = CUDAMAT_ERR_HIP; set_error("sell fill failed"); break; }
    } while (0);
    if (width) CM_DROP(hipFree(width));
    if (rc) {
        sell_free(&p);
        return rc;
    }
    // workgroups own contiguous runs of chunks; at most kSpmvGridMax of them (one pair of dot partials each)
    const int per_block = kBlock / kChunk;
    long long blocks = ((long long)p.nchunks + per_block - 1) / per_block;
    p.chunks_per_block = per_block;
    while (blocks > kSpmvGridMax) {
        p.chunks_per_block += per_block;
        blocks = ((long long)p.nchunks + p.chunks_per_block - 1) / p.chunks_per_block;
    }
    p.grid = (int)blocks;
    p.build_seconds = now_s() - t0;
    *out = p;
    return CUDAMAT_OK;
}

__global__ __launch_bounds__(kBlock) void k_spmv_sell(SpmvArgs a, int nchunks, int chunks_per_block, const int *perm,
                                                      const int *len, const long long *chunk_off, const double *sval,
                                                      const int *scol)
{
#pragma clang fp contract(off)      // one rounding per product and per sum, in column order (bicstab.cpp:72-77)
    __shared__ double red[2 * (kBlock / kChunk)];
    if (a.loop.st && a.loop.st->state != 0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c0 = blockIdx.x * chunks_per_block;
    const int c1 = c0 + chunks_per_block < nchunks ? c0 + chunks_per_block : nchunks;
    double acc0 = 0.0, acc1 = 0.0;
    for (int c = c0 + wave; c < c1; c += kBlock / kChunk) {
        const int row = perm[(size_t)c * kChunk + lane], l = len[(size_t)c * kChunk + lane];
        const long long base = chunk_off[c] * kChunk + lane;
        const int width = (int)(chunk_off[c + 1] - chunk_off[c]);      // wave-uniform
        double sum = 0.0;
        int j = 0;
        for (; j + 4 <= width; j += 4) {                                // four steps in flight
            double v[4];
            int cc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                v[u] = __builtin_nontemporal_load(sval + base + (long long)(j + u) * kChunk);
                cc[u] = __builtin_nontemporal_load(scol + base + (long long)(j + u) * kChunk);
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (j + u < l) {
                    const double prod = v[u] * a.x[cc[u]];
                    sum = sum + prod;
                }
        }
        for (; j < width; j++)
            if (j < l) {
                const double prod = sval[base + (long long)j * kChunk] * a.x[scol[base + (long long)j * kChunk]];
                sum = sum + prod;
            }
        if (row >= 0) {
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
    }
    if (a.dot) {
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
            for (int q = 0; q < kBlock / kChunk; q++) {
                t0 += red[2 * q];
                t1 += red[2 * q + 1];
            }
            a.parts[2 * blockIdx.x] = t0;
            a.parts[2 * blockIdx.x + 1] = t1;
        }
    }
}

int launch_spmv_sell(hipStream_t st, const SellPlan &p, const SpmvArgs &a)
{
    if (a.loop.st && a.check == CHECK_HALF) CM_TRY(launch_check(st, a.loop, a.half, CHECK_HALF));
    hipLaunchKernelGGL(k_spmv_sell, dim3(p.grid), dim3(kBlock), 0, st, a, p.nchunks, p.chunks_per_block, p.perm, p.len,
                       p.chunk_off, p.val, p.col);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

}  // namespace cm
