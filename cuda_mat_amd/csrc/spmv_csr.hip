// spmv_csr.hip -- the SpMV forms that run on the CSR arrays as handed over (cusparseDcsrmv, pbicgstab.cu:67,104,132,...):
// lanes per row, stream tiles (plain / compressed / dictionary), nnz-balanced tiles; their plans; launch_spmv.
#include <algorithm>
#include <cstring>
#include <vector>

#include "kernels.h"
#include "device.h"

namespace cm {

// ------------------------------------------------------------------------ SpMV
// One group of L lanes per row (L = 64: one wavefront per row), a workgroup owns a
// contiguous chunk of rows so that its 4 waves stream one contiguous piece of
// val/colidx; chunks are dealt to XCDs in contiguous eighths so neighbouring rows
// (which share x entries for banded matrices) meet in the same 4 MiB L2.
constexpr int kLongRow = 4096;        // entries: rows beyond this are swept by the whole workgroup
constexpr int kLongRowSlots = 32;

template <int L>
__global__ __launch_bounds__(kBlock) void k_spmv(SpmvArgs a, int rows_per_block)
{
    __shared__ double lds[8];
    if (a.loop.st) {
        if (a.check == CHECK_HALF) {
            if (check_half(a.loop, a.half, lds)) return;
        } else if (a.loop.st->state != 0) {
            return;
        }
    }
    constexpr int RPB = kBlock / L;
    const int lane = threadIdx.x & (L - 1);
    const int group = threadIdx.x / L;
    const int nb = gridDim.x, b = blockIdx.x;
    const int cid = ((nb & 7) == 0) ? (b & 7) * (nb >> 3) + (b >> 3) : b;
    const long long r0 = (long long)cid * rows_per_block;
    const int row_begin = (int)(r0 < a.n ? r0 : a.n);
    const int row_end = (int)(r0 + rows_per_block < a.n ? r0 + rows_per_block : a.n);

    // Rows far longer than the rest (skewed .mtx inputs) would leave one group of L lanes walking
    // tens of thousands of entries while the chip idles: a group only NOTES such a row; afterwards
    // the whole workgroup sweeps each noted row with all 256 lanes (fixed reduction tree, rows taken
    // in increasing order => deterministic).
    __shared__ int long_rows[kLongRowSlots];
    __shared__ int n_long;
    if (threadIdx.x == 0) n_long = 0;
    __syncthreads();

    double acc[2] = {0.0, 0.0};
    for (int row = row_begin + group; row < row_end; row += RPB) {
        const int s = a.rp[row], e = a.rp[row + 1];
        if (e - s > kLongRow) {
            int took = 0;
            if (lane == 0) {
                const int slot = atomicAdd(&n_long, 1);
                if (slot < kLongRowSlots) { long_rows[slot] = row; took = 1; }
            }
            took = __shfl(took, (int)(threadIdx.x & 63) & ~(L - 1), 64);   // from the group's first lane
            if (took) continue;            // (a full table leaves the row to the group itself)
        }
        double sum = 0.0;
        for (int k = s + lane; k < e; k += L)
            sum += __builtin_nontemporal_load(a.val + k) * a.x[__builtin_nontemporal_load(a.ci + k)];
        sum = group_sum<L>(sum);
        if (lane == 0) {
            if (a.d) sum += a.d[row] * a.xd[row];
            double out = a.alpha * sum;
            if (a.beta != 0.0) out += a.beta * a.y[row];
            a.y[row] = out;
            if (a.dot) {
                acc[0] += out * a.w[row];
                acc[1] += out * out;
            }
        }
    }
    __syncthreads();
    const int nl = n_long < kLongRowSlots ? n_long : kLongRowSlots;
    if (nl > 0) {
        if (threadIdx.x == 0) {            // increasing row order (insertion sort of a handful of ids)
            for (int i = 1; i < nl; i++) {
                const int r = long_rows[i];
                int j = i - 1;
                while (j >= 0 && long_rows[j] > r) { long_rows[j + 1] = long_rows[j]; j--; }
                long_rows[j + 1] = r;
            }
        }
        __syncthreads();
        for (int i = 0; i < nl; i++) {
            const int row = long_rows[i];
            const int s = a.rp[row], e = a.rp[row + 1];
            double part[1] = {0.0};
            for (int k = s + (int)threadIdx.x; k < e; k += kBlock)
                part[0] += __builtin_nontemporal_load(a.val + k) * a.x[__builtin_nontemporal_load(a.ci + k)];
            block_sum<1>(part, lds);
            if (threadIdx.x == 0) {
                double sum = part[0];
                if (a.d) sum += a.d[row] * a.xd[row];
                double out = a.alpha * sum;
                if (a.beta != 0.0) out += a.beta * a.y[row];
                a.y[row] = out;
                if (a.dot) {
                    acc[0] += out * a.w[row];
                    acc[1] += out * out;
                }
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

SpmvPlan plan_spmv(const Config &cfg, int n_rows, int64_t nnz)
{
    SpmvPlan p;
    p.stream_rows = 0;
    const double mean = n_rows > 0 ? (double)nnz / n_rows : 1.0;
    int L = 64;
    if (mean <= 3.0) L = 2;
    else if (mean <= 6.0) L = 4;
    else if (mean <= 12.0) L = 8;
    else if (mean <= 40.0) L = 16;
    else if (mean <= 96.0) L = 32;
    if (cfg.spmv_lanes) L = cfg.spmv_lanes;
    p.lanes = L;
    const int rpb = kBlock / L;
    long long groups = ((long long)n_rows + rpb - 1) / rpb;
    int grid = (int)(groups < kSpmvGridMax ? groups : kSpmvGridMax);
    if (grid < 1) grid = 1;
    long long per = ((long long)n_rows + grid - 1) / grid;
    per = (per + rpb - 1) / rpb * rpb;
    if (per < rpb) per = rpb;
    p.rows_per_block = (int)per;
    p.grid = (int)(((long long)n_rows + per - 1) / per);
    if (p.grid < 1) p.grid = 1;
    return p;
}

// ------------------------------------------------------------ SpMV, short rows
// Rows of ~5 entries (stencils) leave most of a lanes-per-row group idle and pay a shuffle tree
// per row.  Here a workgroup takes a tile of R consecutive rows: their entries are one contiguous
// piece of val/colidx, streamed with every lane busy; the products land in LDS; one thread per row
// then adds its products in column order (the rounding sequence of the CPU loop,
// bicstab.cpp:72-77 => bit-exact).  LDS: kStreamNnz products + R+1 row pointers.

template <int R>
__global__ __launch_bounds__(kBlock) void k_spmv_stream(SpmvArgs a, int tiles_per_block)
{
    __shared__ double prod[kStreamNnz];
    __shared__ int srp[R + 1];
    __shared__ double lds[8];
    if (a.loop.st) {
        if (a.check == CHECK_HALF) {
            if (check_half(a.loop, a.half, lds)) return;
        } else if (a.loop.st->state != 0) {
            return;
        }
    }
    const int tid = threadIdx.x;
    const int nb = gridDim.x, b = blockIdx.x;
    // Tiles are dealt CYCLICALLY inside an XCD's contiguous share: at any moment the workgroups of one XCD sit on
    // neighbouring tiles, so the three uses of an x entry by a stencil row (rows i - nx, i, i + nx) fall into the
    // same few microseconds and hit the XCD's L2 instead of being re-fetched after 20 MB of streamed entries.
    const bool xcd_split = (nb & 7) == 0;
    const int wg_per_set = xcd_split ? nb >> 3 : nb;
    const int set = xcd_split ? (b & 7) : 0;
    const int w = xcd_split ? (b >> 3) : b;
    const long long set_tile0 = (long long)set * wg_per_set * tiles_per_block;
    double acc[2] = {0.0, 0.0};
    for (int t = 0; t < tiles_per_block; t++) {
        const long long r0l = (set_tile0 + (long long)t * wg_per_set + w) * R;
        if (r0l >= a.n) continue;
        const int r0 = (int)r0l;
        const int nr = a.n - r0 < R ? a.n - r0 : R;
        for (int i = tid; i <= nr; i += kBlock) srp[i] = a.rp[r0 + i];
        __syncthreads();
        const int base = srp[0];
        const int cnt = srp[nr] - base;
        for (int k = tid; k < cnt; k += kBlock)
            prod[k] = __builtin_nontemporal_load(a.val + base + k) * a.x[__builtin_nontemporal_load(a.ci + base + k)];
        __syncthreads();
        if (tid < nr) {
            const int row = r0 + tid;
            const int s = srp[tid] - base, e = srp[tid + 1] - base;
            double sum = 0.0;
            for (int j = s; j < e; j++) sum += prod[j];
            if (a.d) sum += a.d[row] * a.xd[row];
            double out = a.alpha * sum;
            if (a.beta != 0.0) out += a.beta * a.y[row];
            a.y[row] = out;
            if (a.dot) {
                acc[0] += out * a.w[row];
                acc[1] += out * out;
            }
        }
        __syncthreads();
    }
    if (a.dot) {
        block_sum<2>(acc, lds);
        if (tid == 0) {
            a.parts[2 * b] = acc[0];
            a.parts[2 * b + 1] = acc[1];
        }
    }
}

// ---- the same with compressed indices (banded matrices): per entry a 16-bit column offset from the tile's first
// row instead of a 32-bit column id, per row an 8-bit length instead of a 32-bit row pointer (+ one entry offset per
// tile).  The stream kernel sits at the mixed-traffic HBM ceiling, so bytes are the only lever: C3 moves 0.81 GB
// instead of 0.94 GB.  Built once per system by plan_spmv_compress when every offset fits; same arithmetic, same
// summation order => bit-identical to k_spmv_stream.

template <int R>
__global__ __launch_bounds__(kBlock) void k_spmv_stream_c(SpmvArgs a, int tiles_per_block, const int *tile_base,
                                                          const short *off16, const unsigned char *len8, const double *vals)
{
    __shared__ double prod[kStreamNnz];
    __shared__ int srp[R + 1];
    __shared__ int scan_w[kBlock / 64];
    __shared__ double lds[8];
    if (a.loop.st) {
        if (a.check == CHECK_HALF) {
            if (check_half(a.loop, a.half, lds)) return;
        } else if (a.loop.st->state != 0) {
            return;
        }
    }
    const int tid = threadIdx.x;
    const int nb = gridDim.x, b = blockIdx.x;
    const bool xcd_split = (nb & 7) == 0;
    const int wg_per_set = xcd_split ? nb >> 3 : nb;
    const int set = xcd_split ? (b & 7) : 0;
    const int w = xcd_split ? (b >> 3) : b;
    const long long set_tile0 = (long long)set * wg_per_set * tiles_per_block;
    double acc[2] = {0.0, 0.0};
    for (int t = 0; t < tiles_per_block; t++) {
        const long long tile = set_tile0 + (long long)t * wg_per_set + w;
        const long long r0l = tile * R;
        if (r0l >= a.n) continue;
        const int r0 = (int)r0l;
        const int nr = a.n - r0 < R ? a.n - r0 : R;
        const int base = tile_base[tile];                 // (tiles may be padded: the count comes from the row lengths)
        const int len = tid < nr ? (int)len8[r0 + tid] : 0;
        int cnt;
        const int start = block_scan_int(len, scan_w, &cnt);
        if (tid < nr) srp[tid] = start;
        if (tid == 0) srp[nr] = cnt;
        for (int k = tid; k < cnt; k += kBlock)
            prod[k] = __builtin_nontemporal_load(vals + base + k) * a.x[r0 + (int)__builtin_nontemporal_load(off16 + base + k)];
        __syncthreads();
        if (tid < nr) {
            const int row = r0 + tid;
            const int s = srp[tid], e = srp[tid + 1];
            double sum = 0.0;
            for (int j = s; j < e; j++) sum += prod[j];
            if (a.d) sum += a.d[row] * a.xd[row];
            double out = a.alpha * sum;
            if (a.beta != 0.0) out += a.beta * a.y[row];
            a.y[row] = out;
            if (a.dot) {
                acc[0] += out * a.w[row];
                acc[1] += out * out;
            }
        }
        __syncthreads();
    }
    if (a.dot) {
        block_sum<2>(acc, lds);
        if (tid == 0) {
            a.parts[2 * b] = acc[0];
            a.parts[2 * b + 1] = acc[1];
        }
    }
}

// The same kernel for a matrix with a VALUE DICTIONARY (valdict.h; at most 256 distinct fp64 bit patterns): the plan
// holds per-tile copies of the 16-bit offsets and of 8-bit value indices, each tile padded to a multiple of 8 entries
// (plan_spmv_dict), so a thread fetches its 8 consecutive entries with one 16-byte and one 8-byte load -- 3 bytes per
// entry instead of 10 -- and multiplies dict[index], the very same double, by x: bit-identical results.  (Requesting
// the next tile's operands while this one is summed was tried: 0.159 ms against 0.118 ms for this plain loop.)
template <int R>
__global__ __launch_bounds__(kBlock) void k_spmv_stream_d(SpmvArgs a, int tiles_per_block, const int *pbase,
                                                          const short *off16p, const unsigned char *val8p,
                                                          const unsigned char *len8, const double *dict)
{
    __shared__ double prod[kStreamNnz];
    __shared__ int srp[R + 1];
    __shared__ int scan_w[kBlock / 64];
    __shared__ double lds[8];
    __shared__ double dv[kBlock];                     // the dictionary, one entry per thread (kBlock == 256)
    dv[threadIdx.x] = dict[threadIdx.x];              // (visible after the first __syncthreads below)
    if (a.loop.st) {
        if (a.check == CHECK_HALF) {
            if (check_half(a.loop, a.half, lds)) return;
        } else if (a.loop.st->state != 0) {
            return;
        }
    }
    const int tid = threadIdx.x;
    const int nb = gridDim.x, b = blockIdx.x;
    const bool xcd_split = (nb & 7) == 0;
    const int wg_per_set = xcd_split ? nb >> 3 : nb;
    const int set = xcd_split ? (b & 7) : 0;
    const int w = xcd_split ? (b >> 3) : b;
    const long long set_tile0 = (long long)set * wg_per_set * tiles_per_block;
    double acc[2] = {0.0, 0.0};
    for (int t = 0; t < tiles_per_block; t++) {
        const long long tile = set_tile0 + (long long)t * wg_per_set + w;
        const long long r0l = tile * R;
        if (r0l >= a.n) continue;
        const int r0 = (int)r0l;
        const int nr = a.n - r0 < R ? a.n - r0 : R;
        const int base = pbase[tile], cnt = pbase[tile + 1] - base;    // a multiple of 8, at most kStreamNnz = 8 * kBlock
        const int len = tid < nr ? (int)len8[r0 + tid] : 0;
        const bool mine = 8 * tid < cnt;
        double xv[8];
        unsigned iw[2] = {0u, 0u};
        if (mine) {
            const uint2 iv = *(const uint2 *)(val8p + base + 8 * tid);
            const uint4 ov = *(const uint4 *)(off16p + base + 8 * tid);
            const unsigned ow[4] = {ov.x, ov.y, ov.z, ov.w};
            iw[0] = iv.x; iw[1] = iv.y;
#pragma unroll
            for (int q = 0; q < 8; q++) xv[q] = a.x[r0 + (int)(short)((ow[q >> 1] >> (16 * (q & 1))) & 0xffffu)];
        }
        int total;
        const int start = block_scan_int(len, scan_w, &total);
        if (tid < nr) srp[tid] = start;
        if (tid == 0) srp[nr] = total;
        if (mine) {
#pragma unroll
            for (int q = 0; q < 8; q++) prod[8 * tid + q] = dv[(iw[q >> 2] >> (8 * (q & 3))) & 0xffu] * xv[q];
        }
        __syncthreads();
        if (tid < nr) {
            const int row = r0 + tid;
            const int s = srp[tid], e = srp[tid + 1];
            double sum = 0.0;
            for (int j = s; j < e; j++) sum += prod[j];
            if (a.d) sum += a.d[row] * a.xd[row];
            double out = a.alpha * sum;
            if (a.beta != 0.0) out += a.beta * a.y[row];
            a.y[row] = out;
            if (a.dot) {
                acc[0] += out * a.w[row];
                acc[1] += out * out;
            }
        }
        __syncthreads();
    }
    if (a.dot) {
        block_sum<2>(acc, lds);
        if (tid == 0) {
            a.parts[2 * b] = acc[0];
            a.parts[2 * b + 1] = acc[1];
        }
    }
}

// one 8-lane team per row: 8-bit length, 16-bit offsets from the first row of the row's tile; flags[0] = does not fit
__global__ __launch_bounds__(kBlock) void k_stream_compress(int n, int R, const int *rp, const int *ci, short *off16,
                                                            unsigned char *len8, int *tile_base, int *flags)
{
    constexpr int L = 8;
    const long long row = ((long long)blockIdx.x * kBlock + threadIdx.x) / L;
    if (row > n) return;
    const int lane = threadIdx.x & (L - 1);
    if (row == n) {                                   // closing entry of the tile table
        if (lane == 0) tile_base[(n + R - 1) / R] = rp[n] - rp[0];
        return;
    }
    const int s = rp[row], e = rp[row + 1];
    const int r0 = (int)(row / R) * R;
    if (lane == 0) {
        if (e - s > 255) flags[0] = 1;
        len8[row] = (unsigned char)(e - s);
        if (row == r0) tile_base[row / R] = s - rp[0];
    }
    for (int k = s + lane; k < e; k += L) {
        const int off = ci[k] - r0;
        if (off < -32768 || off > 32767) flags[0] = 1;
        off16[k - rp[0]] = (short)off;
    }
}

int plan_spmv_compress(hipStream_t s, const Config &cfg, int n_rows, int64_t nnz, const int *rp, const int *ci, SpmvPlan *plan)
{
    if (!plan->stream_rows || nnz <= 0 || !cfg.spmv_compress) return CUDAMAT_OK;
    const int R = plan->stream_rows;
    const size_t ntiles = ((size_t)n_rows + R - 1) / R;
    int *flags = nullptr, h = 0;
    int rc = CUDAMAT_OK;
    do {
        if (hipMalloc((void **)&plan->c_off16, sizeof(short) * (size_t)nnz) != hipSuccess ||
            hipMalloc((void **)&plan->c_len8, (size_t)n_rows) != hipSuccess ||
            hipMalloc((void **)&plan->c_tile_base, sizeof(int) * (ntiles + 1)) != hipSuccess ||
            hipMalloc((void **)&flags, sizeof(int)) != hipSuccess) { rc = CUDAMAT_ERR_NOMEM; break; }
        if ((rc = CM_RC(hipMemsetAsync(flags, 0, sizeof(int), s)))) break;
        const long long threads = ((long long)n_rows + 1) * 8;
        hipLaunchKernelGGL(k_stream_compress, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows, R,
                           rp, ci, plan->c_off16, plan->c_len8, plan->c_tile_base, flags);
        if (hipMemcpyAsync(&h, flags, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
    } while (0);
    if (flags) CM_DROP(hipFree(flags));
    if (rc || h) {                                    // does not fit (or no memory): the plain stream kernel stays
        void *ptrs[] = {plan->c_off16, plan->c_len8, plan->c_tile_base};
        for (void *q : ptrs)
            if (q) CM_DROP(hipFree(q));
        plan->c_off16 = nullptr;
        plan->c_len8 = nullptr;
        plan->c_tile_base = nullptr;
        if (rc == CUDAMAT_ERR_HIP) return fail_hip(hipGetLastError(), "index compression", __FILE__, __LINE__);
    }
    return CUDAMAT_OK;
}

// ---- padded per-tile copies for the dictionary form
// one workgroup: exclusive scan of the tiles' entry counts rounded up to 8 -> pbase[0..ntiles]
__global__ __launch_bounds__(kBlock) void k_tile_pad_scan(int n, int R, int ntiles, const int *rp, int *pbase, int round_to)
{
    __shared__ int scan_w[kBlock / 64];
    int run = 0;
    for (int t0 = 0; t0 < ntiles; t0 += kBlock) {
        const int t = t0 + threadIdx.x;
        int padded = 0;
        if (t < ntiles) {
            const long long r1 = (long long)(t + 1) * R;
            const int cnt = rp[r1 < n ? r1 : n] - rp[(long long)t * R];
            padded = (cnt + round_to - 1) / round_to * round_to;
        }
        int total;
        const int ex = block_scan_int(padded, scan_w, &total);
        if (t < ntiles) pbase[t] = run + ex;
        run += total;
    }
    if (threadIdx.x == 0) pbase[ntiles] = run;
}

// an 8-lane team per row copies the row's offsets and value indices to their padded places
__global__ __launch_bounds__(kBlock) void k_tile_pad_fill(int n, int R, const int *rp, const short *off16, const unsigned char *vidx,
                                                          const int *pbase, short *off16p, unsigned char *val8p)
{
    constexpr int L = 8;
    const long long row = ((long long)blockIdx.x * kBlock + threadIdx.x) / L;
    if (row >= n) return;
    const int lane = threadIdx.x & (L - 1);
    const int t = (int)(row / R);
    const int s = rp[row], e = rp[row + 1], first = rp[(long long)t * R];
    const int dst = pbase[t] + (s - first);
    for (int k = s + lane; k < e; k += L) {
        off16p[dst + (k - s)] = off16[k - rp[0]];
        val8p[dst + (k - s)] = vidx[k - rp[0]];
    }
}

// an 8-lane team per row copies the row's offsets and values to their places in the aligned copies
__global__ __launch_bounds__(kBlock) void k_tile_align_fill(int n, int R, const int *rp, const short *off16, const double *val,
                                                            const int *abase, short *off16a, double *vala)
{
    constexpr int L = 8;
    const long long row = ((long long)blockIdx.x * kBlock + threadIdx.x) / L;
    if (row >= n) return;
    const int lane = threadIdx.x & (L - 1);
    const int t = (int)(row / R);
    const int s = rp[row], e = rp[row + 1], first = rp[(long long)t * R];
    const int dst = abase[t] + (s - first);
    for (int k = s + lane; k < e; k += L) {
        off16a[dst + (k - s)] = off16[k - rp[0]];
        vala[dst + (k - s)] = val[k - rp[0]];
    }
}

// Line-aligned copies of the compressed stream kernel's two entry streams (round 3).  The kernel is held by the rate of
// L1 -> L2 requests, and a request moves at most one 128-byte line: with a tile's entries starting
// anywhere, a wave's 512-byte value load touches five lines and its 128-byte offset load two; with every tile starting on
// a 64-entry boundary they touch four and one.  Costs <= 63 idle slots per tile (2.5 % at 5 entries per row) and a second
// copy of the values in HBM.  The option SPMV_ALIGN = 0 keeps the packed arrays.
int plan_spmv_align(hipStream_t s, const Config &cfg, int n_rows, int64_t nnz, const int *rp, const double *val, SpmvPlan *plan)
{
    if (!plan->stream_rows || !plan->c_off16 || nnz <= 0 || !cfg.spmv_align) return CUDAMAT_OK;
    const int R = plan->stream_rows;
    const int ntiles = (int)(((long long)n_rows + R - 1) / R);
    if (nnz + 63LL * ntiles > 0x7fffffffLL) return CUDAMAT_OK;
    int total = 0, rc = CUDAMAT_OK;
    do {
        if (hipMalloc((void **)&plan->a_base, sizeof(int) * ((size_t)ntiles + 1)) != hipSuccess) { rc = CUDAMAT_ERR_NOMEM; break; }
        hipLaunchKernelGGL(k_tile_pad_scan, dim3(1), dim3(kBlock), 0, s, n_rows, R, ntiles, rp, plan->a_base, 64);
        if (hipMemcpyAsync(&total, plan->a_base + ntiles, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        if (total < nnz || (int64_t)total > nnz + 64LL * ntiles) { rc = CUDAMAT_ERR_HIP; break; }
        if (hipMalloc((void **)&plan->a_off16, sizeof(short) * (size_t)total + 256) != hipSuccess ||
            hipMalloc((void **)&plan->a_val, sizeof(double) * (size_t)total + 256) != hipSuccess) { rc = CUDAMAT_ERR_NOMEM; break; }
        if ((rc = CM_RC(hipMemsetAsync(plan->a_off16, 0, sizeof(short) * (size_t)total + 256, s)))) break;      // idle slots: offset 0, value 0
        if ((rc = CM_RC(hipMemsetAsync(plan->a_val, 0, sizeof(double) * (size_t)total + 256, s)))) break;
        const long long threads = (long long)n_rows * 8;
        hipLaunchKernelGGL(k_tile_align_fill, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows, R, rp,
                           plan->c_off16, val, plan->a_base, plan->a_off16, plan->a_val);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
    } while (0);
    if (rc) {                                         // no memory / failure: the packed arrays stay
        void *ptrs[] = {plan->a_base, plan->a_off16, plan->a_val};
        for (void *q : ptrs)
            if (q) CM_DROP(hipFree(q));
        plan->a_base = nullptr;
        plan->a_off16 = nullptr;
        plan->a_val = nullptr;
        if (rc == CUDAMAT_ERR_HIP) return fail_hip(hipGetLastError(), "aligned stream copies", __FILE__, __LINE__);
    }
    return CUDAMAT_OK;
}

int plan_spmv_dict(hipStream_t s, int n_rows, int64_t nnz, const int *rp, const unsigned char *vidx, const double *dict,
                   SpmvPlan *plan)
{
    if (!plan->stream_rows || !plan->c_off16 || !vidx || !dict || nnz <= 0) return CUDAMAT_OK;
    const int R = plan->stream_rows;
    const int ntiles = (int)(((long long)n_rows + R - 1) / R);
    int total = 0, rc = CUDAMAT_OK;
    do {
        if (hipMalloc((void **)&plan->d_pbase, sizeof(int) * ((size_t)ntiles + 1)) != hipSuccess) { rc = CUDAMAT_ERR_NOMEM; break; }
        hipLaunchKernelGGL(k_tile_pad_scan, dim3(1), dim3(kBlock), 0, s, n_rows, R, ntiles, rp, plan->d_pbase, 8);
        if (hipMemcpyAsync(&total, plan->d_pbase + ntiles, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        if (total < nnz || (int64_t)total > nnz + 8LL * ntiles) { rc = CUDAMAT_ERR_HIP; break; }
        if (hipMalloc((void **)&plan->d_off16, sizeof(short) * (size_t)total + 16) != hipSuccess ||
            hipMalloc((void **)&plan->d_val8, (size_t)total + 16) != hipSuccess) { rc = CUDAMAT_ERR_NOMEM; break; }
        if ((rc = CM_RC(hipMemsetAsync(plan->d_off16, 0, sizeof(short) * (size_t)total + 16, s)))) break;      // padding: offset 0, value index 0
        if ((rc = CM_RC(hipMemsetAsync(plan->d_val8, 0, (size_t)total + 16, s)))) break;
        const long long threads = (long long)n_rows * 8;
        hipLaunchKernelGGL(k_tile_pad_fill, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows, R, rp,
                           plan->c_off16, vidx, plan->d_pbase, plan->d_off16, plan->d_val8);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        plan->c_dict = dict;
    } while (0);
    if (rc) {                                         // no memory / failure: the plain compressed kernel stays
        void *ptrs[] = {plan->d_pbase, plan->d_off16, plan->d_val8};
        for (void *q : ptrs)
            if (q) CM_DROP(hipFree(q));
        plan->d_pbase = nullptr;
        plan->d_off16 = nullptr;
        plan->d_val8 = nullptr;
        plan->c_dict = nullptr;
        if (rc == CUDAMAT_ERR_HIP) return fail_hip(hipGetLastError(), "dictionary tiles", __FILE__, __LINE__);
    }
    return CUDAMAT_OK;
}

// ------------------------------------------------------------ SpMV, skewed row lengths
// Tiles of kTileNnz consecutive ENTRIES (not rows): every workgroup streams the same number of entries
// whatever the row-length distribution (a few rows of 1e5 entries among rows of 8 leave the
// lanes-per-row kernel at 0.3-0.5 TB/s).  S[t] = first row that STARTS at or after the tile's first entry
// (lower bound in rowptr, found once per matrix).  Per tile: products -> LDS; rows that start here and have
// at most kTileShort entries in the tile are summed by one thread in column order; longer ones, the piece of
// a row that began in an earlier tile ("head") and the piece of a row that continues into the next one
// ("tail") are summed by one wavefront each.  Rows confined to one tile are finished here; a row spanning
// tiles is finished by k_spmv_tiles_fix from tails[t] + heads[t+1..] in tile order.  Work lists are built
// with a prefix sum (no atomics), so every row's summation tree and the dot partials are reproducible.
constexpr int kTileNnz = 2048;
constexpr int kTileShort = 32;
constexpr int kTileItems = kTileNnz / (kTileShort + 1) + 4;

struct TileItem {
    int j0, j1, row, kind;     // kind 0: whole row, 1: head piece, 2: tail piece
};

// exclusive prefix of one flag per thread over the 256-thread workgroup (+ the total)
__device__ __forceinline__ int block_scan_flag(int v, int *lds_waves, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(v != 0);
    const int before_in_wave = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();                       // lds_waves may still be read from the previous round
    if (lane == 0) lds_waves[wave] = __popcll(m);
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) {
        const int c = lds_waves[w];
        if (w < wave) before += c;
        all += c;
    }
    *total = all;
    return before + before_in_wave;
}


__global__ __launch_bounds__(kBlock) void k_spmv_tiles(SpmvArgs a, const int *S, int ntiles, int tiles_per_block,
                                                       double *heads, double *tails)
{
    __shared__ double prod[kTileNnz];
    __shared__ TileItem items[kTileItems];
    __shared__ int scan_w[kBlock / 64];
    __shared__ int n_items;
    __shared__ double lds[8];
    if (a.loop.st) {
        if (a.check == CHECK_HALF) {
            if (check_half(a.loop, a.half, lds)) return;
        } else if (a.loop.st->state != 0) {
            return;
        }
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nb = gridDim.x, b = blockIdx.x;
    const int cid = ((nb & 7) == 0) ? (b & 7) * (nb >> 3) + (b >> 3) : b;
    const int kbase = a.rp[0], kend = a.rp[a.n];
    double acc[2] = {0.0, 0.0};
    for (int tt = 0; tt < tiles_per_block; tt++) {
        const long long tl = (long long)cid * tiles_per_block + tt;
        if (tl >= ntiles) break;
        const int t = (int)tl;
        const int k0 = kbase + t * kTileNnz;
        const int k1 = kend - k0 < kTileNnz ? kend : k0 + kTileNnz;
        const int cnt = k1 - k0;
        const int s0 = S[t], s1 = S[t + 1];
        for (int j = tid; j < cnt; j += kBlock)
            prod[j] = __builtin_nontemporal_load(a.val + k0 + j) * a.x[__builtin_nontemporal_load(a.ci + k0 + j)];
        if (tid == 0) {
            int m = 0;
            const int first_start = a.rp[s0];          // s0 == n: rp[n] = kend > k0
            if (first_start > k0) {                    // entry k0 belongs to row s0 - 1, which began earlier
                items[m].j0 = 0;
                items[m].j1 = (first_start < k1 ? first_start : k1) - k0;
                items[m].row = s0 - 1;
                items[m].kind = 1;
                m++;
            }
            n_items = m;
        }
        __syncthreads();
        for (int r0 = s0; r0 < s1; r0 += kBlock) {
            const int base_items = n_items;
            const int r = r0 + tid;
            const bool isrow = r < s1;
            int rb = 0, re = 0;
            if (isrow) {
                rb = a.rp[r];
                re = a.rp[r + 1];
            }
            const bool spans = isrow && re > k1;       // only the last row that starts here can
            const int rend = re < k1 ? re : k1;
            const bool coop = isrow && (spans || rend - rb > kTileShort);
            if (isrow && !coop) {
                double sum = 0.0;
                for (int j = rb - k0; j < rend - k0; j++) sum += prod[j];
                spmv_finish_row(a, r, sum, acc);
            }
            int total;
            const int pos = block_scan_flag(coop ? 1 : 0, scan_w, &total);
            if (coop) {
                TileItem it;
                it.j0 = rb - k0;
                it.j1 = rend - k0;
                it.row = r;
                it.kind = spans ? 2 : 0;
                items[base_items + pos] = it;
            }
            __syncthreads();
            if (tid == 0) n_items = base_items + total;
            __syncthreads();
        }
        const int m = n_items;
        for (int i = wave; i < m; i += kBlock / 64) {
            const TileItem it = items[i];
            double sum = 0.0;
            for (int j = it.j0 + lane; j < it.j1; j += 64) sum += prod[j];
            sum = wave_sum(sum);
            if (lane == 0) {
                if (it.kind == 0) spmv_finish_row(a, it.row, sum, acc);
                else if (it.kind == 1) heads[t] = sum;
                else tails[t] = sum;
            }
        }
        __syncthreads();       // prod and items are reused by the next tile
    }
    if (a.dot) {
        block_sum<2>(acc, lds);
        if (tid == 0) {
            a.parts[2 * b] = acc[0];
            a.parts[2 * b + 1] = acc[1];
        }
    }
}

// rows spanning several tiles: one wavefront per row adds tails[t] + heads[t+1 .. last] (fixed tree)
__global__ __launch_bounds__(kBlock) void k_spmv_tiles_fix(SpmvArgs a, const int *S, const int *span, int nspan,
                                                           const double *heads, const double *tails, int parts_off)
{
    __shared__ double lds[8];
    if (a.loop.st && a.loop.st->state != 0) return;      // (a half-step test was evaluated by k_spmv_tiles)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kbase = a.rp[0];
    double acc[2] = {0.0, 0.0};
    for (int g = blockIdx.x * (kBlock / 64) + wave; g < nspan; g += gridDim.x * (kBlock / 64)) {
        const int t = span[g];
        const int row = S[t + 1] - 1;
        const int last = (a.rp[row + 1] - 1 - kbase) / kTileNnz;      // tile holding the row's last entry
        double sum = 0.0;
        for (int q = t + 1 + lane; q <= last; q += 64) sum += heads[q];
        sum = wave_sum(sum);
        if (lane == 0) spmv_finish_row(a, row, tails[t] + sum, acc);
    }
    if (a.dot) {
        block_sum<2>(acc, lds);
        if (threadIdx.x == 0) {
            a.parts[2 * (parts_off + blockIdx.x)] = acc[0];
            a.parts[2 * (parts_off + blockIdx.x) + 1] = acc[1];
        }
    }
}

// S[t] = first row r with rp[r] >= first entry of tile t (rows are rp[0..n]); flag[t] = the last row that
// starts in tile t continues beyond it
__global__ __launch_bounds__(kBlock) void k_tiles_rows(int n, const int *rp, int ntiles, int *S)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t > ntiles) return;
    if (t == ntiles) { S[t] = n; return; }
    const long long key = (long long)rp[0] + (long long)t * kTileNnz;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (rp[mid] < key) lo = mid + 1; else hi = mid;
    }
    S[t] = lo;
}

__global__ __launch_bounds__(kBlock) void k_tiles_span(int n, const int *rp, int ntiles, const int *S, int *flag)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= ntiles) return;
    const long long k1 = (long long)rp[0] + (long long)(t + 1) * kTileNnz;
    const int s0 = S[t], s1 = S[t + 1];
    flag[t] = (s1 > s0 && rp[s1] > k1) ? 1 : 0;         // row s1 - 1 ends at rp[s1]
}

// lane-iterations the lanes-per-row kernel spends: sum over groups of 64/L consecutive rows (one wave
// instruction stream) of max ceil(len / L), times 64 -- compared with nnz this is its imbalance
__global__ __launch_bounds__(kBlock) void k_lane_cost(int n, const int *rp, int L, unsigned long long *out)
{
    const int G = 64 / L;
    const long long g = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long r0 = g * G;
    unsigned long long it = 0;
    if (r0 < n) {
        int m = 0;
        for (int q = 0; q < G && r0 + q < n; q++) {
            const int len = rp[r0 + q + 1] - rp[r0 + q];
            const int c = (len + L - 1) / L;
            m = c > m ? c : m;
        }
        it = (unsigned long long)m;
    }
    it = (unsigned long long)wave_sum((double)it);      // exact below 2^53
    if ((threadIdx.x & 63) == 0 && it) atomicAdd(out, it);
}

void plan_spmv_free(SpmvPlan *plan)
{
    void *ptrs[] = {plan->tile_S, plan->tile_span, plan->tile_heads, plan->tile_tails, plan->c_off16, plan->c_len8,
                    plan->c_tile_base, plan->d_pbase, plan->d_off16, plan->d_val8, plan->a_base, plan->a_off16, plan->a_val};
    plan->a_base = nullptr;
    plan->a_off16 = nullptr;
    plan->a_val = nullptr;
    plan->d_pbase = nullptr;
    plan->d_off16 = nullptr;
    plan->d_val8 = nullptr;
    for (void *q : ptrs)
        if (q) CM_DROP(hipFree(q));
    plan->c_off16 = nullptr;
    plan->c_len8 = nullptr;
    plan->c_tile_base = nullptr;
    plan->tile_S = plan->tile_span = nullptr;
    plan->tile_heads = plan->tile_tails = nullptr;
    plan->tiles = 0;
}

static int plan_spmv_tiles(hipStream_t s, int n_rows, int64_t nnz, const int *rp, SpmvPlan *plan)
{
    const int64_t nt64 = (nnz + kTileNnz - 1) / kTileNnz;
    if (nt64 < 1 || nt64 > (1 << 24)) return CUDAMAT_OK;                 // keep the lanes-per-row plan
    const int ntiles = (int)nt64;
    int *flag = nullptr;
    int rc = CUDAMAT_OK;
    do {
        if (hipMalloc((void **)&plan->tile_S, sizeof(int) * ((size_t)ntiles + 1)) != hipSuccess ||
            hipMalloc((void **)&plan->tile_heads, sizeof(double) * (size_t)ntiles) != hipSuccess ||
            hipMalloc((void **)&plan->tile_tails, sizeof(double) * (size_t)ntiles) != hipSuccess ||
            hipMalloc((void **)&flag, sizeof(int) * (size_t)ntiles) != hipSuccess) { rc = CUDAMAT_ERR_NOMEM; break; }
        if ((rc = CM_RC(hipMemsetAsync(plan->tile_heads, 0, sizeof(double) * (size_t)ntiles, s)))) break;
        if ((rc = CM_RC(hipMemsetAsync(plan->tile_tails, 0, sizeof(double) * (size_t)ntiles, s)))) break;
        hipLaunchKernelGGL(k_tiles_rows, dim3((unsigned)((ntiles + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows, rp,
                           ntiles, plan->tile_S);
        hipLaunchKernelGGL(k_tiles_span, dim3((unsigned)((ntiles + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows, rp,
                           ntiles, plan->tile_S, flag);
        std::vector<int> h((size_t)ntiles), span;
        if (hipMemcpyAsync(h.data(), flag, sizeof(int) * (size_t)ntiles, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        for (int t = 0; t < ntiles; t++)
            if (h[(size_t)t]) span.push_back(t);
        plan->tile_nspan = (int)span.size();
        if (!span.empty()) {
            if (hipMalloc((void **)&plan->tile_span, sizeof(int) * span.size()) != hipSuccess) { rc = CUDAMAT_ERR_NOMEM; break; }
            if (hipMemcpy(plan->tile_span, span.data(), sizeof(int) * span.size(), hipMemcpyHostToDevice) != hipSuccess) {
                rc = CUDAMAT_ERR_HIP; break;
            }
        }
        const int fix_grid = span.empty() ? 0 : (int)std::min<size_t>(64, (span.size() + 3) / 4);
        const int main_max = kSpmvGridMax - 64;
        int grid = ntiles < main_max ? ntiles : main_max;
        const int per = (ntiles + grid - 1) / grid;
        grid = (ntiles + per - 1) / per;
        plan->tiles = ntiles;
        plan->rows_per_block = per;          // tiles per workgroup
        plan->grid = grid;
        plan->tile_fix_grid = fix_grid;
    } while (0);
    if (flag) CM_DROP(hipFree(flag));
    if (rc) {
        plan_spmv_free(plan);
        if (rc == CUDAMAT_ERR_NOMEM) return CUDAMAT_OK;                  // no room for the tables: lanes-per-row plan stays
        return fail_hip(hipGetLastError(), "tile plan", __FILE__, __LINE__);
    }
    return CUDAMAT_OK;
}

// max over tiles of R rows of the number of entries in the tile, for R = 64, 128, 256
__global__ __launch_bounds__(kBlock) void k_tile_nnz_max(int n, const int *rp, int *out)
{
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;   // 64-row tile index
    const long long r0 = t * 64;
    if (r0 >= n) return;
    const int b0 = rp[r0];
    auto at = [&](long long r) { return rp[r < n ? r : n]; };
    atomicMax(&out[0], at(r0 + 64) - b0);
    if ((t & 1) == 0) atomicMax(&out[1], at(r0 + 128) - b0);
    if ((t & 3) == 0) atomicMax(&out[2], at(r0 + 256) - b0);
}

// skewed row lengths: measure what the lanes-per-row plan would cost and switch to tiles when it is unbalanced
static int plan_spmv_balance(hipStream_t s, const Config &cfg, int n_rows, int64_t nnz, const int *rp, SpmvPlan *plan, void *scratch)
{
    if (cfg.spmv_form == 1) return CUDAMAT_OK;          // lanes
    const bool force = cfg.spmv_form == 2;              // tiles
    if (nnz <= 0 || n_rows <= 0) return CUDAMAT_OK;
    if (!force) {
        if (nnz < 65536) return CUDAMAT_OK;
        unsigned long long *d = (unsigned long long *)scratch, h = 0;
        if (!scratch) CM_HIP(hipMalloc((void **)&d, sizeof(h)));
        hipError_t e = hipMemsetAsync(d, 0, sizeof(h), s);
        const long long groups = ((long long)n_rows + (64 / plan->lanes) - 1) / (64 / plan->lanes);
        hipLaunchKernelGGL(k_lane_cost, dim3((unsigned)((groups + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows, rp,
                           plan->lanes, d);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (!scratch) CM_DROP(hipFree(d));
        if (e != hipSuccess) return fail_hip(e, "lane cost", __FILE__, __LINE__);
        plan->lane_cost = (double)h * 64.0 / (double)nnz;
        // measured (scripts/skew_probe.py): at 2.0 (rows of 2 and 62 alternating) the lanes kernel is still memory-bound
        // and 25 % faster than the tiles; at 3.7-5 (hub rows, Pareto lengths) the tiles win 1.4-9x
        if (plan->lane_cost <= 2.5) return CUDAMAT_OK;
    }
    return plan_spmv_tiles(s, n_rows, nnz, rp, plan);
}

int plan_spmv_refine(hipStream_t s, const Config &cfg, int n_rows, int64_t nnz, const int *rp, int base, SpmvPlan *plan, void *scratch)
{
    (void)base;
    plan->stream_rows = 0;
    if (cfg.spmv_lanes) return CUDAMAT_OK;          // explicit lanes-per-row request
    const double mean = n_rows > 0 ? (double)nnz / n_rows : 0.0;
    if (n_rows < 64) return CUDAMAT_OK;
    if (mean > 12.0) return plan_spmv_balance(s, cfg, n_rows, nnz, rp, plan, scratch);
    int *d = (int *)scratch, h[3] = {0, 0, 0};
    if (!scratch) CM_HIP(hipMalloc((void **)&d, 3 * sizeof(int)));
    hipError_t e = hipMemsetAsync(d, 0, 3 * sizeof(int), s);
    const long long tiles = ((long long)n_rows + 63) / 64;
    hipLaunchKernelGGL(k_tile_nnz_max, dim3((unsigned)((tiles + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows, rp, d);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (!scratch) CM_DROP(hipFree(d));
    if (e != hipSuccess) return fail_hip(e, "tile scan", __FILE__, __LINE__);
    int R = 0;
    if (h[2] <= kStreamNnz) R = 256;
    else if (h[1] <= kStreamNnz) R = 128;
    else if (h[0] <= kStreamNnz) R = 64;
    if (!R) return plan_spmv_balance(s, cfg, n_rows, nnz, rp, plan, scratch);
    const long long tiles_r = ((long long)n_rows + R - 1) / R;
    int grid = (int)(tiles_r < kSpmvGridMax ? tiles_r : kSpmvGridMax);
    const long long per = (tiles_r + grid - 1) / grid;
    plan->stream_rows = R;
    plan->rows_per_block = (int)per;                               // tiles per workgroup
    plan->grid = (int)((tiles_r + per - 1) / per);
    return CUDAMAT_OK;
}

int launch_spmv(hipStream_t s, const SpmvPlan &plan, const SpmvArgs &a)
{
    dim3 g(plan.grid), b(kBlock);
    if (plan.tiles) {
        hipLaunchKernelGGL(k_spmv_tiles, g, b, 0, s, a, plan.tile_S, plan.tiles, plan.rows_per_block, plan.tile_heads,
                           plan.tile_tails);
        if (plan.tile_fix_grid)
            hipLaunchKernelGGL(k_spmv_tiles_fix, dim3(plan.tile_fix_grid), b, 0, s, a, plan.tile_S, plan.tile_span,
                               plan.tile_nspan, plan.tile_heads, plan.tile_tails, plan.grid);
        CM_HIP(hipGetLastError());
        return CUDAMAT_OK;
    }
    if (plan.stream_rows && plan.c_off16) {
        switch (plan.stream_rows) {
#define CM_SC(RV)                                                                                                          \
    do {                                                                                                                   \
        if (plan.d_pbase)                                                                                                  \
            hipLaunchKernelGGL(k_spmv_stream_d<RV>, g, b, 0, s, a, plan.rows_per_block, plan.d_pbase, plan.d_off16, plan.d_val8, \
                               plan.c_len8, plan.c_dict);                                                                 \
        else                                                                                                               \
            hipLaunchKernelGGL(k_spmv_stream_c<RV>, g, b, 0, s, a, plan.rows_per_block,                                    \
                               plan.a_base ? plan.a_base : plan.c_tile_base, plan.a_base ? plan.a_off16 : plan.c_off16,   \
                               plan.c_len8, plan.a_base ? plan.a_val : a.val);                                             \
    } while (0)
        case 64:  CM_SC(64); break;
        case 128: CM_SC(128); break;
        default:  CM_SC(256); break;
        }
#undef CM_SC
        CM_HIP(hipGetLastError());
        return CUDAMAT_OK;
    }
    if (plan.stream_rows) {
        switch (plan.stream_rows) {
        case 64:  hipLaunchKernelGGL(k_spmv_stream<64>, g, b, 0, s, a, plan.rows_per_block); break;
        case 128: hipLaunchKernelGGL(k_spmv_stream<128>, g, b, 0, s, a, plan.rows_per_block); break;
        default:  hipLaunchKernelGGL(k_spmv_stream<256>, g, b, 0, s, a, plan.rows_per_block); break;
        }
        CM_HIP(hipGetLastError());
        return CUDAMAT_OK;
    }
    switch (plan.lanes) {
    case 2:  hipLaunchKernelGGL(k_spmv<2>, g, b, 0, s, a, plan.rows_per_block); break;
    case 4:  hipLaunchKernelGGL(k_spmv<4>, g, b, 0, s, a, plan.rows_per_block); break;
    case 8:  hipLaunchKernelGGL(k_spmv<8>, g, b, 0, s, a, plan.rows_per_block); break;
    case 16: hipLaunchKernelGGL(k_spmv<16>, g, b, 0, s, a, plan.rows_per_block); break;
    case 32: hipLaunchKernelGGL(k_spmv<32>, g, b, 0, s, a, plan.rows_per_block); break;
    default: hipLaunchKernelGGL(k_spmv<64>, g, b, 0, s, a, plan.rows_per_block); break;
    }
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

}  // namespace cm
