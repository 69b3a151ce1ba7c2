// ad-hoc probe (round 5): the classes of device memory.  N chunks of 2 GB; (1) R(i)+W(j) for every pair -> which pairs are slow
// -> classes (connected components of "slow"); (2) with one representative per class: a phase-1-shaped kernel (read 8 B from X,
// read 2 B from Y, write 8 B to Z per entry) for every (X, Y, Z), and a phase-2-shaped one (read 8 B from Z, 2 B from Y).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probe_classes scripts/probe_classes.hip ; scripts/probe_classes [chunks]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} } while(0)

__global__ __launch_bounds__(1024) void k_rw(const double2 *a, double2 *P, long n2)
{
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long lo = blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    for (long i = lo + threadIdx.x; i < hi; i += 1024) {
        const double2 v = a[i];
        double2 o;
        o.x = v.x * 1.5;
        o.y = v.y + 1.0;
        P[i] = o;
    }
}
__global__ __launch_bounds__(1024) void k_p1(const double2 *a, const ushort2 *c, double2 *P, long n2)      // 8 + 2 read, 8 written
{
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long lo = blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    for (long i = lo + threadIdx.x; i < hi; i += 1024) {
        const double2 v = a[i];
        const ushort2 k = c[i];
        double2 o;
        o.x = v.x * 1.5 + (double)k.x;
        o.y = v.y + (double)k.y;
        P[i] = o;
    }
}
__global__ __launch_bounds__(1024) void k_p2(const double2 *a, const ushort2 *c, double *out, long n2)      // 8 + 2 read
{
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long lo = blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    double acc = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 1024) {
        const double2 v = a[i];
        const ushort2 k = c[i];
        acc += v.x * (double)k.x + v.y * (double)k.y;
    }
    if (acc == 1.2345) out[0] = acc;
}

static hipEvent_t e0, e1;
template <typename F> static float best_of(F launch, int reps = 3)
{
    float best = 1e9f;
    for (int rep = 0; rep <= reps; rep++) {
        CK(hipEventRecord(e0));
        launch();
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
    const int N = argc > 1 ? atoi(argv[1]) : 48;
    const size_t G = (size_t)2 << 30;
    std::vector<char *> c((size_t)N);
    for (int i = 0; i < N; i++) {
        CK(hipMalloc(&c[(size_t)i], G));
        CK(hipMemsetAsync(c[(size_t)i], 0, G, 0));
    }
    double *out;
    CK(hipMalloc(&out, 8));
    CK(hipDeviceSynchronize());
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long n2 = (long)(G / 16);
    std::vector<float> t((size_t)N * N, 0.f);
    std::vector<float> all;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            if (i == j) continue;
            const float ms = best_of([&] { hipLaunchKernelGGL(k_rw, dim3(256), dim3(1024), 0, 0, (const double2 *)c[i], (double2 *)c[j], n2); }, 2);
            t[(size_t)i * N + j] = ms;
            all.push_back(ms);
        }
    std::sort(all.begin(), all.end());
    const float lo = all[all.size() / 20], hi = all[all.size() * 19 / 20], cut = 0.5f * (lo + hi);
    printf("R(i)+W(j), 2 GB each: 5%% %.3f ms (%.0f GB/s)  95%% %.3f ms (%.0f GB/s); cut at %.3f ms.  Rows: i (read), columns: j (written); S = slow\n", lo,
           2.0 * G / lo * 1e-6, hi, 2.0 * G / hi * 1e-6, cut);
    for (int i = 0; i < N; i++) {
        printf("%3d ", i);
        for (int j = 0; j < N; j++) putchar(i == j ? '\\' : t[(size_t)i * N + j] > cut ? 'S' : '.');
        putchar('\n');
    }
    // classes: connected components of the symmetric "slow" relation
    std::vector<int> cls((size_t)N, -1);
    int ncls = 0;
    for (int i = 0; i < N; i++) {
        if (cls[i] >= 0) continue;
        std::vector<int> stack{i};
        cls[i] = ncls;
        while (!stack.empty()) {
            const int u = stack.back();
            stack.pop_back();
            for (int v = 0; v < N; v++)
                if (v != u && cls[v] < 0 && (t[(size_t)u * N + v] > cut || t[(size_t)v * N + u] > cut)) { cls[v] = ncls; stack.push_back(v); }
        }
        ncls++;
    }
    printf("classes (%d): ", ncls);
    for (int i = 0; i < N; i++) printf("%d", cls[i] % 10);
    printf("\n");
    std::vector<int> rep, rep2;
    for (int k = 0; k < ncls && k < 4; k++) {
        int first = -1, second = -1;
        for (int i = 0; i < N; i++)
            if (cls[i] == k) { if (first < 0) first = i; else if (second < 0) second = i; }
        rep.push_back(first);
        rep2.push_back(second < 0 ? first : second);
    }
    const int K = (int)rep.size();
    printf("phase-1-shaped kernel (values X, indices Y, products Z; two chunks of one class are distinct chunks), ms and GB/s of 18 B per entry:\n");
    for (int x = 0; x < K; x++)
        for (int y = 0; y < K; y++)
            for (int z = 0; z < K; z++) {
                // distinct chunks even when the class is the same: X = rep[x]; Y = rep2[y] unless equal class handled; Z from a third
                char *X = c[rep[x]], *Y = c[rep2[y]], *Z = nullptr;
                for (int i = 0; i < N && !Z; i++)
                    if (cls[i] == z && c[i] != X && c[i] != Y) Z = c[i];
                if (!Z || X == Y) continue;
                const float ms = best_of([&] { hipLaunchKernelGGL(k_p1, dim3(256), dim3(1024), 0, 0, (const double2 *)X, (const ushort2 *)Y, (double2 *)Z, n2); });
                printf("  X%d Y%d Z%d  %.3f ms  %.0f GB/s%s\n", x, y, z, ms, 18.0 * (G / 8) / ms * 1e-6, (x == z) ? "   (values and products share a class)" : "");
            }
    printf("phase-2-shaped kernel (products Z, row indices Y):\n");
    for (int z = 0; z < K; z++)
        for (int y = 0; y < K; y++) {
            char *Z = c[rep[z]], *Y = c[rep2[y]];
            if (Z == Y) continue;
            const float ms = best_of([&] { hipLaunchKernelGGL(k_p2, dim3(256), dim3(1024), 0, 0, (const double2 *)Z, (const ushort2 *)Y, out, n2); });
            printf("  Z%d Y%d  %.3f ms  %.0f GB/s\n", z, y, ms, 10.0 * (G / 8) / ms * 1e-6);
        }
    return 0;
}
