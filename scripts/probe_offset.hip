// ad-hoc probe (round 5): does the time of a phase-1-shaped kernel -- 768 workgroups of 1024 threads, one per CU at a time,
// each streaming ITS contiguous piece of three arrays in step (8 B read, 2 B read, 8 B written per entry) -- depend on the
// RELATIVE placement of the three arrays?  One allocation, the arrays cut out of it at chosen offsets.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probe_offset scripts/probe_offset.hip ; scripts/probe_offset
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} } while(0)

__global__ __launch_bounds__(1024) void k_stream3(const double2 *a, const ushort2 *c, double2 *P, long n2, int parts)
{
    extern __shared__ double tile[];                 // 104 KB: one workgroup per CU, as k_pb_phase1
    if (threadIdx.x == 0) tile[0] = 1.0;
    __syncthreads();
    const double f = tile[0];
    const long per = (n2 + parts - 1) / parts;
    for (int part = blockIdx.x; part < parts; part += gridDim.x) {
        const long lo = part * per, hi = lo + per < n2 ? lo + per : n2;
        for (long i = lo + threadIdx.x; i < hi; i += 1024) {
            const double2 v = a[i];
            const ushort2 k = c[i];
            double2 o;
            o.x = v.x * f + (double)k.x;
            o.y = v.y * f + (double)k.y;
            P[i] = o;
        }
    }
}

int main(int argc, char **argv)
{
    const long N = 500000000L;                      // entries
    const size_t A = (size_t)N * 8, C = (size_t)N * 2;
    const size_t SLACK = (size_t)64 << 20;
    char *base;
    CK(hipMalloc(&base, 2 * A + C + 4 * SLACK));
    CK(hipMemset(base, 0, 2 * A + C + 4 * SLACK));
    CK(hipFuncSetAttribute((const void *)k_stream3, hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t A2 = (A + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);       // arrays start on 2 MB boundaries ...
    const size_t C2 = (C + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);
    std::vector<long> deltas = {0, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288,
                                1 << 20, 2 << 20, 3 << 20, 4 << 20, 5 << 20, 6 << 20, 8 << 20, 12 << 20, 16 << 20, 24 << 20, 32 << 20, 48 << 20};
    printf("base %p\n", (void *)base);
    for (int which = 0; which < 2; which++) {       // 0: move P against a;  1: move c against a
        for (long d : deltas) {
            char *a = base;
            char *c = base + A2 + SLACK + (which == 1 ? d : 0);
            char *P = c + C2 + SLACK + (which == 0 ? d : 0) - (which == 1 ? d : 0);
            float best = 1e9f, sum = 0;
            for (int rep = 0; rep < 6; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_stream3, dim3(768), dim3(1024), 104 * 1024, 0, (const double2 *)a, (const ushort2 *)c, (double2 *)P, N / 2, 768);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0) { sum += ms; if (ms < best) best = ms; }
            }
            printf("%s by %9ld B: best %.3f ms  mean %.3f ms  (%.0f GB/s)\n", which == 0 ? "P moved" : "c moved", d, best, sum / 5, 18.0 * N / (sum / 5) * 1e-6);
        }
    }
    return 0;
}
