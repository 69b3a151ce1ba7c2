// valdict.hip -- value dictionary ("CSR-VI": Kourtis, Goumas, Koziris, "Optimizing sparse matrix-vector
// multiplication using index and value compression", CF'08).
//
// The SpMV kernels of this library run at the HBM ceiling for the bytes they move, and 8 of the 12 bytes of a CSR
// entry are its fp64 value.  Matrices assembled from a few stencil / element coefficients -- the reference's own
// fixtures (mat900, mat10000: 5-point Laplacians) and both synthetic BASELINE systems -- hold a handful of distinct
// values: storing an 8-bit index into a dictionary of the distinct BIT PATTERNS instead moves 1 byte per entry where
// the value moved 8, and the kernels multiply exactly the same doubles (dict[idx[k]] == val[k] bit for bit, signed
// zeros and NaN payloads included), so every result stays bit-identical.  A matrix with more than 256 distinct
// values keeps its fp64 values; nothing else changes for it.
//
// Detection is one pass over the values with a 4096-slot open-addressing table of bit patterns in global memory
// (claimed with 64-bit compare-and-swap; lookups of values already present are plain loads served from L1/L2) and an
// overflow flag that ends the pass early; a 2^20-entry sample goes first, so a matrix of arbitrary values costs
// microseconds.  The dictionary is sorted by bit pattern (deterministic), the indices come from a binary search in LDS.
#include <algorithm>
#include <cstring>
#include <vector>

#include "valdict.h"

namespace cm {

constexpr int kTableSlots = 4096;
constexpr unsigned long long kEmpty = 0xFFFFFFFFFFFFFFFFull;    // (a NaN with every payload bit set: reported as overflow)

__device__ __forceinline__ unsigned slot_of(unsigned long long bits)
{
    bits ^= bits >> 33;
    bits *= 0xff51afd7ed558ccdull;
    bits ^= bits >> 29;
    return (unsigned)bits & (kTableSlots - 1);
}

// flags[0] = distinct values claimed so far, flags[1] = overflow
__global__ __launch_bounds__(kBlock) void k_dict_probe(int64_t first, int64_t count, const double *val,
                                                       unsigned long long *table, int *flags)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    unsigned long long last = kEmpty;
    int since_check = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(val[first + i]);
        if (bits == last) continue;
        if (++since_check >= 64) {
            since_check = 0;
            if (__hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        }
        if (bits == kEmpty) { __hip_atomic_store(&flags[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        unsigned h = slot_of(bits);
        for (int probe = 0; probe < kTableSlots; probe++) {
            // plain (cached) load on purpose: a slot only ever goes EMPTY -> key, so a stale EMPTY merely sends this
            // thread into the compare-and-swap below, which returns the key that is really there
            unsigned long long key = table[h];
            if (key == bits) break;
            if (key == kEmpty) {
                key = atomicCAS(&table[h], kEmpty, bits);
                if (key == kEmpty) {                                   // a new distinct value
                    if (atomicAdd(&flags[0], 1) + 1 > kDictMax) __hip_atomic_store(&flags[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                if (key == bits) break;
            }
            h = (h + 1) & (kTableSlots - 1);
        }
        last = bits;
    }
}

__global__ __launch_bounds__(kBlock) void k_dict_index(int64_t nnz, const double *val, const double *dict, int n,
                                                       unsigned char *idx, int *flags)
{
    __shared__ unsigned long long d[kDictMax];
    for (int i = threadIdx.x; i < kDictMax; i += kBlock)
        d[i] = i < n ? (unsigned long long)__double_as_longlong(dict[i]) : kEmpty;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += stride) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(val[i]);
        int lo = 0, hi = n - 1;                                        // last position with d[pos] <= bits
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (d[mid] <= bits) lo = mid; else hi = mid - 1;
        }
        if (d[lo] != bits) flags[1] = 1;                               // cannot happen; checked by the host all the same
        idx[i] = (unsigned char)lo;
    }
}

int valdict_sample_overflows(hipStream_t st, int64_t count, const double *val, void *scratch, bool *many)
{
    *many = false;
    if (count <= 0) return CUDAMAT_OK;
    unsigned long long *table = (unsigned long long *)scratch;
    int *flags = (int *)((char *)scratch + sizeof(unsigned long long) * kTableSlots);
    int h[2] = {0, 0};
    CM_HIP(hipMemsetAsync(table, 0xFF, sizeof(unsigned long long) * kTableSlots, st));
    CM_HIP(hipMemsetAsync(flags, 0, 2 * sizeof(int), st));
    int64_t g = (count + kBlock - 1) / kBlock;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_dict_probe, dim3((unsigned)g), dim3(kBlock), 0, st, (int64_t)0, count, val, table, flags);
    if (hipMemcpyAsync(h, flags, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        set_error("value dictionary: sample probe failed (%s)", hipGetErrorString(hipGetLastError()));
        return CUDAMAT_ERR_HIP;
    }
    *many = h[1] != 0 || h[0] > kDictMax;
    return CUDAMAT_OK;
}

void valdict_free(ValDict *v)
{
    if (v->dict) CM_DROP(hipFree(v->dict));
    if (v->idx) CM_DROP(hipFree(v->idx));
    *v = ValDict();
}

int valdict_build(hipStream_t st, const Config &cfg, int64_t nnz, const double *val, ValDict *out)
{
    *out = ValDict();
    if (!cfg.value_dict || nnz < 4096) return CUDAMAT_OK;      // (tiny matrices live in caches anyway)
    unsigned long long *table = nullptr;
    int *flags = nullptr;
    int h[2] = {0, 0};
    int rc = CUDAMAT_OK;
    ValDict v;
    do {
        if (hipMalloc((void **)&table, sizeof(unsigned long long) * kTableSlots) != hipSuccess ||
            hipMalloc((void **)&flags, 2 * sizeof(int)) != hipSuccess) { rc = CUDAMAT_ERR_NOMEM; set_error("value dictionary: out of memory"); break; }
        if ((rc = CM_RC(hipMemsetAsync(table, 0xFF, sizeof(unsigned long long) * kTableSlots, st)))) break;
        if ((rc = CM_RC(hipMemsetAsync(flags, 0, 2 * sizeof(int), st)))) break;
        // a sample first: arbitrary values overflow the table within the first few thousand entries
        const int64_t sample = nnz < (1 << 20) ? nnz : (1 << 20);
        const int64_t pieces[2][2] = {{0, sample}, {sample, nnz - sample}};
        bool over = false;
        for (int p = 0; p < 2 && !over; p++) {
            if (pieces[p][1] <= 0) continue;
            int64_t g = (pieces[p][1] + kBlock - 1) / kBlock;
            if (g > 8192) g = 8192;
            hipLaunchKernelGGL(k_dict_probe, dim3((unsigned)g), dim3(kBlock), 0, st, pieces[p][0], pieces[p][1], val, table, flags);
            if (hipMemcpyAsync(h, flags, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                rc = CUDAMAT_ERR_HIP; set_error("value dictionary: probe failed (%s)", hipGetErrorString(hipGetLastError())); break;
            }
            over = h[1] != 0 || h[0] > kDictMax;
        }
        if (rc || over || h[0] < 1) break;
        std::vector<unsigned long long> keys(kTableSlots);
        if (hipMemcpy(keys.data(), table, sizeof(unsigned long long) * kTableSlots, hipMemcpyDeviceToHost) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        keys.erase(std::remove(keys.begin(), keys.end(), kEmpty), keys.end());
        std::sort(keys.begin(), keys.end());
        if ((int)keys.size() != h[0] || keys.size() > (size_t)kDictMax) { rc = CUDAMAT_ERR_HIP; set_error("value dictionary: table count mismatch"); break; }
        std::vector<double> dict((size_t)kDictMax, 0.0);
        for (size_t i = 0; i < keys.size(); i++) std::memcpy(&dict[i], &keys[i], sizeof(double));
        if (hipMalloc((void **)&v.dict, sizeof(double) * kDictMax) != hipSuccess ||
            hipMalloc((void **)&v.idx, (size_t)nnz + 16) != hipSuccess) { valdict_free(&v); break; }      // no memory: no dictionary
        if (hipMemcpy(v.dict, dict.data(), sizeof(double) * kDictMax, hipMemcpyHostToDevice) != hipSuccess) { rc = CUDAMAT_ERR_HIP; break; }
        if ((rc = CM_RC(hipMemsetAsync(flags, 0, 2 * sizeof(int), st)))) break;
        int64_t g = (nnz + kBlock - 1) / kBlock;
        if (g > 16384) g = 16384;
        hipLaunchKernelGGL(k_dict_index, dim3((unsigned)g), dim3(kBlock), 0, st, nnz, val, v.dict, (int)keys.size(), v.idx, flags);
        if (hipMemcpyAsync(h, flags, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            rc = CUDAMAT_ERR_HIP; set_error("value dictionary: index pass failed (%s)", hipGetErrorString(hipGetLastError())); break;
        }
        if (h[1]) { rc = CUDAMAT_ERR_HIP; set_error("value dictionary: a value is missing from its own dictionary"); break; }
        v.n = (int)keys.size();
    } while (0);
    if (table) CM_DROP(hipFree(table));
    if (flags) CM_DROP(hipFree(flags));
    if (rc != CUDAMAT_OK || v.n == 0) valdict_free(&v);
    else *out = v;
    return rc;
}

}  // namespace cm
