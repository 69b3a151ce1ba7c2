// ad-hoc probe (round 5): is device memory uniform?  260 chunks of 1 GB are allocated one after the other; (1) every chunk is
// read alone; (2) chunk 0 is read while chunk j is written, for every j (a phase-1-shaped kernel: 256 workgroups, each on its
// own contiguous piece of both arrays).  If placement did not matter both series would be flat.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probe_regions scripts/probe_regions.hip ; scripts/probe_regions [chunks]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} } while(0)

__global__ __launch_bounds__(1024) void k_rw(const double2 *a, double2 *P, long n2, int write)
{
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long lo = blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    double acc = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 1024) {
        const double2 v = a[i];
        if (write) {
            double2 o;
            o.x = v.x * 1.5;
            o.y = v.y + 1.0;
            P[i] = o;
        } else {
            acc += v.x + v.y;
        }
    }
    if (!write && acc == 1.2345) P[0].x = acc;
}

static float run(const void *a, void *P, long n2, int write, hipEvent_t e0, hipEvent_t e1)
{
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rw, dim3(256), dim3(1024), 0, 0, (const double2 *)a, (double2 *)P, n2, write);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    const int chunks = argc > 1 ? atoi(argv[1]) : 260;
    const size_t G = (size_t)1 << 30;
    std::vector<char *> c((size_t)chunks);
    for (int i = 0; i < chunks; i++) {
        CK(hipMalloc(&c[(size_t)i], G));
        CK(hipMemsetAsync(c[(size_t)i], 0, G, 0));
    }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long n2 = (long)(G / 16);
    printf("chunk  address          read alone (GB/s)   read chunk 0 + write this one (GB/s)   read this + write chunk 1 (GB/s)\n");
    for (int j = 0; j < chunks; j++) {
        const float r = run(c[(size_t)j], c[(size_t)j], n2, 0, e0, e1);
        const float w = j == 0 ? 0.f : run(c[0], c[(size_t)j], n2, 1, e0, e1);
        const float v = j == 1 ? 0.f : run(c[(size_t)j], c[1], n2, 1, e0, e1);
        printf("%4d  %p  %8.0f  %8.0f  %8.0f\n", j, (void *)c[(size_t)j], G / (r * 1e-3) * 1e-9, w > 0 ? 2.0 * G / (w * 1e-3) * 1e-9 : 0.0, v > 0 ? 2.0 * G / (v * 1e-3) * 1e-9 : 0.0);
    }
    return 0;
}
