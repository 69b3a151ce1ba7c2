// gen.hip -- synthetic inputs generated directly in HBM.
// Bit-for-bit the same matrices as oracle/oracle_gen.c defines on the CPU
// (tests/test_gpu_parity.py compares them); each rank generates only its own
// row block, so a 1e7 x 50 matrix never crosses PCIe.
#include "kernels.h"

namespace cm {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

constexpr int kMaxPerRow = 128;

__global__ __launch_bounds__(kBlock) void k_gen_rand_rows(int64_t n, int rn, uint64_t seed,
                                                          int64_t row0, int64_t row1, int base,
                                                          int *rp, int *ci, double *val)
{
    const int64_t li = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t i = row0 + li;
    if (li == 0) rp[0] = base;
    if (i >= row1) return;
    rp[li + 1] = base + (int)((li + 1) * rn);
    const int noff = rn - 1;
    int cols[kMaxPerRow];
    signed char cv[kMaxPerRow];
    const uint64_t key = mix64(seed + (uint64_t)i);
    int cnt = 0;
    int absum = 0;
    for (uint64_t a = 0; cnt < noff; a++) {
        const uint64_t h = mix64(key + a);
        const int64_t c = (int64_t)__umul64hi(h, (uint64_t)n);
        if (c == i) continue;
        int pos = cnt;
        while (pos > 0 && cols[pos - 1] > c) pos--;
        if (pos > 0 && cols[pos - 1] == c) continue;
        for (int q = cnt; q > pos; q--) { cols[q] = cols[q - 1]; cv[q] = cv[q - 1]; }
        const int sel = (int)(h & 3);
        const int v = sel == 0 ? -2 : sel == 1 ? -1 : sel == 2 ? 1 : 2;
        cols[pos] = (int)c;
        cv[pos] = (signed char)v;
        absum += v < 0 ? -v : v;
        cnt++;
    }
    int *oc = ci + li * rn;
    double *ov = val + li * rn;
    int k = 0;
    bool placed = false;
    for (int q = 0; q < noff; q++) {
        if (!placed && cols[q] > i) { oc[k] = (int)i + base; ov[k++] = 1.0 + absum; placed = true; }
        oc[k] = cols[q] + base;
        ov[k++] = (double)cv[q];
    }
    if (!placed) { oc[k] = (int)i + base; ov[k++] = 1.0 + absum; }
}

// 5-point Laplacian rows [row0,row1) of the nx x ny grid; rowptr is local
// (rp[0] = base), column ids global.  One thread per row; the row offset is a
// closed form so no scan is needed.
__device__ __forceinline__ int64_t poisson_prefix(int64_t i, int nx, int ny)
{
    // number of entries in rows [0, i)
    if (i <= 0) return 0;
    const int64_t y = i / nx, x = i % nx;   // full grid rows y, plus x entries of row y
    // per full grid row: 5*nx - 2 (left/right edges) ; minus nx for y==0 (no up) and y==ny-1 (no down)
    int64_t cnt = y * (5LL * nx - 2);
    if (y > 0) cnt -= nx;                    // grid row 0 has no "up" entries
    if (y > ny - 1) cnt -= nx;               // (only when i == n) last grid row has no "down"
    // partial grid row y (if y < ny): x cells
    if (x > 0) {
        int64_t per = 5 * x;
        per -= 1;                            // cell 0 has no left neighbour
        if (y == 0) per -= x;
        if (y == ny - 1) per -= x;
        cnt += per;                          // right edge cell (x == nx-1) is never inside a partial row
    }
    return cnt;
}

__global__ __launch_bounds__(kBlock) void k_gen_poisson5(int nx, int ny, int64_t row0, int64_t row1,
                                                         int base, int *rp, int *ci, double *val)
{
    const int64_t li = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t i = row0 + li;
    const int64_t off0 = poisson_prefix(row0, nx, ny);
    if (li == 0) rp[0] = base;
    if (i >= row1) return;
    const int x = (int)(i % nx), y = (int)(i / nx);
    int64_t k = poisson_prefix(i, nx, ny) - off0;
    if (y > 0)      { ci[k] = (int)(i - nx) + base; val[k++] = -1.0; }
    if (x > 0)      { ci[k] = (int)(i - 1) + base;  val[k++] = -1.0; }
    ci[k] = (int)i + base; val[k++] = 4.0;
    if (x < nx - 1) { ci[k] = (int)(i + 1) + base;  val[k++] = -1.0; }
    if (y < ny - 1) { ci[k] = (int)(i + nx) + base; val[k++] = -1.0; }
    rp[li + 1] = (int)k + base;
}

__global__ __launch_bounds__(kBlock) void k_gen_xstar(int64_t i0, int64_t i1, uint64_t seed, double *x)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = i0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < i1; i += stride)
        x[i - i0] = 1.0 + (double)(mix64(seed + (uint64_t)i) & 7) * 0.125;
}

}  // namespace cm

using namespace cm;

extern "C" int64_t cudamat_poisson5_nnz(int nx, int ny)
{
    return 5LL * nx * ny - 2LL * nx - 2LL * ny;
}

extern "C" int cudamat_rand_row_nnz(int64_t n, int per_row)
{
    int64_t off = per_row - 1;
    if (off > n - 1) off = n - 1;
    if (off < 0) off = 0;
    return (int)off + 1;
}

extern "C" int cudamat_gen_rand_rows(cudamat_ctx *ctx, int64_t n, int per_row, uint64_t seed,
                                     int64_t row0, int64_t row1, int base, int *rowptr,
                                     int *colidx, double *val)
{
    CM_ARG(ctx && rowptr && colidx && val, "null pointer");
    CM_ARG(n > 0 && per_row >= 1 && per_row <= kMaxPerRow, "per_row must be in [1,128]");
    CM_ARG(0 <= row0 && row0 <= row1 && row1 <= n, "row range");
    const int rn = cudamat_rand_row_nnz(n, per_row);
    CM_ARG((row1 - row0) * (int64_t)rn < (1LL << 31), "local nnz must fit int32");
    const int64_t rows = row1 - row0;
    const int64_t g = rows > 0 ? (rows + kBlock - 1) / kBlock : 1;
    hipLaunchKernelGGL(k_gen_rand_rows, dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, rn, seed,
                       row0, row1, base, rowptr, colidx, val);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

extern "C" int cudamat_gen_poisson5(cudamat_ctx *ctx, int nx, int ny, int64_t row0, int64_t row1,
                                    int base, int *rowptr, int *colidx, double *val)
{
    CM_ARG(ctx && rowptr && colidx && val, "null pointer");
    CM_ARG(nx >= 1 && ny >= 1, "grid");
    const int64_t n = (int64_t)nx * ny;
    CM_ARG(n < (1LL << 31), "dimension must fit int32");
    CM_ARG(0 <= row0 && row0 <= row1 && row1 <= n, "row range");
    const int64_t rows = row1 - row0;
    const int64_t g = rows > 0 ? (rows + kBlock - 1) / kBlock : 1;
    hipLaunchKernelGGL(k_gen_poisson5, dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, nx, ny, row0,
                       row1, base, rowptr, colidx, val);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}

extern "C" int cudamat_gen_xstar(cudamat_ctx *ctx, int64_t i0, int64_t i1, uint64_t seed, double *x)
{
    CM_ARG(ctx && x && i0 <= i1, "bad range");
    int64_t g = (i1 - i0 + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_gen_xstar, dim3((int)g), dim3(kBlock), 0, ctx->stream, i0, i1, seed, x);
    CM_HIP(hipGetLastError());
    return CUDAMAT_OK;
}
