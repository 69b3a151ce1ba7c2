// kernels.hip -- hand-written CDNA4 (gfx950) kernels of the BiCGSTAB inner loop.
//
// Everything here is HBM-bandwidth-bound fp64 streaming / gather work (<= 0.17
// flop/byte): no MFMA.  What matters is (a) coalesced 16-byte-per-lane loads on
// the streamed vectors and contiguous row chunks per workgroup on the CSR
// arrays, (b) one pass per fused update instead of the reference's
// copy/scal/axpy triplets (pbicgstab.cu:86-88,109-110,139-140,668-672,...),
// (c) dot products produced by the kernel that already streams the operands,
// reduced wave64-shuffle -> LDS -> per-workgroup partial -> fixed-order sum in
// the consumer's prologue (bitwise reproducible, no atomics, no host sync).
//
// Scalars (rho, alpha, omega, norms) never leave the device: see LoopState.
#include <algorithm>
#include <cstring>
#include <vector>

#include "kernels.h"
#include "device.h"

namespace cm {

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
constexpr int kStreamNnz = 2048;

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
        hipMemsetAsync(flags, 0, sizeof(int), s);
        const long long threads = ((long long)n_rows + 1) * 8;
        hipLaunchKernelGGL(k_stream_compress, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows, R,
                           rp, ci, plan->c_off16, plan->c_len8, plan->c_tile_base, flags);
        if (hipMemcpyAsync(&h, flags, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
    } while (0);
    if (flags) hipFree(flags);
    if (rc || h) {                                    // does not fit (or no memory): the plain stream kernel stays
        void *ptrs[] = {plan->c_off16, plan->c_len8, plan->c_tile_base};
        for (void *q : ptrs)
            if (q) hipFree(q);
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
        hipMemsetAsync(plan->a_off16, 0, sizeof(short) * (size_t)total + 256, s);      // idle slots: offset 0, value 0
        hipMemsetAsync(plan->a_val, 0, sizeof(double) * (size_t)total + 256, s);
        const long long threads = (long long)n_rows * 8;
        hipLaunchKernelGGL(k_tile_align_fill, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows, R, rp,
                           plan->c_off16, val, plan->a_base, plan->a_off16, plan->a_val);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
    } while (0);
    if (rc) {                                         // no memory / failure: the packed arrays stay
        void *ptrs[] = {plan->a_base, plan->a_off16, plan->a_val};
        for (void *q : ptrs)
            if (q) hipFree(q);
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
        hipMemsetAsync(plan->d_off16, 0, sizeof(short) * (size_t)total + 16, s);      // padding: offset 0, value index 0
        hipMemsetAsync(plan->d_val8, 0, (size_t)total + 16, s);
        const long long threads = (long long)n_rows * 8;
        hipLaunchKernelGGL(k_tile_pad_fill, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows, R, rp,
                           plan->c_off16, vidx, plan->d_pbase, plan->d_off16, plan->d_val8);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        plan->c_dict = dict;
    } while (0);
    if (rc) {                                         // no memory / failure: the plain compressed kernel stays
        void *ptrs[] = {plan->d_pbase, plan->d_off16, plan->d_val8};
        for (void *q : ptrs)
            if (q) hipFree(q);
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
        if (q) hipFree(q);
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
        hipMemsetAsync(plan->tile_heads, 0, sizeof(double) * (size_t)ntiles, s);
        hipMemsetAsync(plan->tile_tails, 0, sizeof(double) * (size_t)ntiles, s);
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
    if (flag) hipFree(flag);
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
        hipMemsetAsync(d, 0, sizeof(h), s);
        const long long groups = ((long long)n_rows + (64 / plan->lanes) - 1) / (64 / plan->lanes);
        hipLaunchKernelGGL(k_lane_cost, dim3((unsigned)((groups + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows, rp,
                           plan->lanes, d);
        hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, s);
        const hipError_t e = hipStreamSynchronize(s);
        if (!scratch) hipFree(d);
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
    hipMemsetAsync(d, 0, 3 * sizeof(int), s);
    const long long tiles = ((long long)n_rows + 63) / 64;
    hipLaunchKernelGGL(k_tile_nnz_max, dim3((unsigned)((tiles + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows, rp, d);
    hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s);
    hipError_t e = hipStreamSynchronize(s);
    if (!scratch) hipFree(d);
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

static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

#define COMMA ,
#define CM_VEC_LOOP(N, BODY2, BODY1)                                                   \
    {                                                                                  \
        const int64_t n2__ = VEC ? (N) / 2 : 0;                                        \
        const int64_t stride__ = (int64_t)gridDim.x * kBlock;                          \
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2__; i += stride__) { BODY2 } \
        for (int64_t i = 2 * n2__ + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < (N); i += stride__) { BODY1 } \
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
