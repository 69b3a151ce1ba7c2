// ad-hoc probe (round 5): can arrays be ASSEMBLED from physical chunks of a chosen memory class?  2 GB physical handles
// (hipMemCreate), each mapped and classified by the R/W test of probe_classes.hip; then 4 GB arrays mapped from two chunks of one
// class each, and the phase-1-shaped kernel on (values, products) in equal / different classes.  Also: what the calls cost.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("err %s (%d) line %d\n",hipGetErrorString(e),(int)e,__LINE__); exit(1);} } while(0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ __launch_bounds__(1024) void k_rw(const double2 *a, double2 *P, long n2)
{
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long lo = blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    for (long i = lo + threadIdx.x; i < hi; i += 1024) { const double2 v = a[i]; double2 o; o.x = v.x * 1.5; o.y = v.y + 1.0; P[i] = o; }
}
static hipEvent_t e0, e1;
static float rw_ms(const void *a, void *b, size_t bytes)
{
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rw, dim3(256), dim3(1024), 0, 0, (const double2 *)a, (double2 *)b, (long)(bytes / 16));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 24;
    const size_t G = (size_t)2 << 30;
    int dev = 0; CK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu\n", gran);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<hipMemGenericAllocationHandle_t> h((size_t)N);
    std::vector<char *> va((size_t)N);
    double t_create = 0, t_map = 0;
    for (int i = 0; i < N; i++) {
        double t = now();
        CK(hipMemCreate(&h[i], G, &prop, 0));
        t_create += now() - t;
        t = now();
        CK(hipMemAddressReserve((void **)&va[i], G, G, nullptr, 0));
        CK(hipMemMap(va[i], G, 0, h[i], 0));
        CK(hipMemSetAccess(va[i], G, &acc, 1));
        t_map += now() - t;
    }
    printf("hipMemCreate %.2f ms each, reserve + map + access %.2f ms each; first va %p\n", t_create / N * 1e3, t_map / N * 1e3, (void *)va[0]);
    for (int i = 0; i < N; i++) CK(hipMemsetAsync(va[i], 0, G, 0));
    CK(hipDeviceSynchronize());
    // classes by the R/W test against the first chunk of every class found so far
    std::vector<int> cls((size_t)N, -1), rep;
    const float t_self = rw_ms(va[0], va[0] + G / 2, G / 2);       // inside one chunk: the same-class figure
    printf("inside one chunk (1 GB read, 1 GB written): %.3f ms\n", t_self);
    for (int i = 0; i < N; i++) {
        for (size_t k = 0; k < rep.size() && cls[i] < 0; k++) {
            const float t = rw_ms(va[rep[k]], va[i] + G / 2, G / 2);
            if (t > 0.97f * t_self) cls[i] = (int)k;
        }
        if (cls[i] < 0) { cls[i] = (int)rep.size(); rep.push_back(i); }
    }
    printf("classes: ");
    for (int i = 0; i < N; i++) printf("%d", cls[i]);
    printf("\n");
    // two 4 GB arrays per class, assembled from two chunks of that class each
    std::vector<std::vector<int>> by(rep.size());
    for (int i = 0; i < N; i++) by[cls[i]].push_back(i);
    std::vector<char *> arr;          // arr[2 * k], arr[2 * k + 1]: two arrays of class k
    std::vector<int> arr_cls;
    for (size_t k = 0; k < by.size(); k++) {
        for (int a = 0; a + 1 < (int)by[k].size() && a < 4; a += 2) {
            const int c0 = by[k][a], c1 = by[k][a + 1];
            double t = now();
            CK(hipMemUnmap(va[c0], G)); CK(hipMemUnmap(va[c1], G));
            char *v = nullptr;
            CK(hipMemAddressReserve((void **)&v, 2 * G, G, nullptr, 0));
            CK(hipMemMap(v, G, 0, h[c0], 0)); CK(hipMemMap(v + G, G, 0, h[c1], 0));
            CK(hipMemSetAccess(v, 2 * G, &acc, 1));
            if (k == 0 && a == 0) printf("unmap x2 + reserve + map x2 + access: %.2f ms\n", (now() - t) * 1e3);
            arr.push_back(v);
            arr_cls.push_back((int)k);
        }
    }
    printf("4 GB arrays: ");
    for (int c : arr_cls) printf("%d", c);
    printf("\nR(values 4 GB) + W(products 4 GB):\n");
    for (size_t x = 0; x < arr.size(); x++)
        for (size_t z = 0; z < arr.size(); z++) {
            if (x == z) continue;
            const float ms = rw_ms(arr[x], arr[z], 2 * G);
            printf("  values class %d, products class %d: %.3f ms  %.0f GB/s\n", arr_cls[x], arr_cls[z], ms, 4.0 * G / ms * 1e-6);
        }
    // the same through plain hipMalloc, for reference
    char *p0, *p1;
    CK(hipMalloc(&p0, 2 * G)); CK(hipMalloc(&p1, 2 * G));
    printf("  two hipMalloc'd 4 GB arrays: %.3f ms\n", rw_ms(p0, p1, 2 * G));
    return 0;
}
