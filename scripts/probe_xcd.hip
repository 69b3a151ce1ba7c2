// probe_xcd.hip -- ad-hoc probe: can the product stream of the blocked SpMV be handed from the
// workgroups that hold x tiles to the workgroups that hold y tiles THROUGH THE XCD's L2, without going
// to HBM?  One persistent launch, one 1024-thread workgroup per CU; the 32 workgroups of an XCD form a
// team.  8 producer waves per workgroup stream (value fp64, local column u16) entries, multiply with
// the x tile in LDS and write the products into a small ring (NS slots of E products) that is meant to
// stay in L2; 8 consumer waves read their pieces of every team member's products back (sc1 loads: L2
// served) and add them into the y tile in LDS.  Flow control: per-slot monotonic counters, one atomic
// per workgroup per chunk.  Every spin is bounded.
//
// build: hipcc --offload-arch=gfx950 -O3 -o probe_xcd probe_xcd.hip
// run:   ./probe_xcd <ESW 8|16|32> <NS> <XT> <nt 0|1> [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned short u16;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

constexpr int kTeam = 32;          // workgroups per XCD
constexpr int kPW = 8, kCW = 8;    // producer / consumer waves per workgroup
constexpr int kSpin = 1 << 21;

struct Params {
    const double *ent_v;
    const u16 *ent_c;
    const u16 *pr;
    const double *xg;
    double *ring;        // [8][NS][E]
    unsigned *ctr;       // [8][2][8] counters, each on its own 128-byte line (32 uints)
    unsigned *team;      // [0..7]*32: team tickets; [8*32]: arrived; [9*32]: error word
    double *out;         // [8][32] checksums
    int T, NS, XT, CX, RY, nt;
};

__device__ __forceinline__ unsigned ld_u32(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// lane 0 polls a global counter until it reaches target; false on timeout / error
__device__ __forceinline__ bool wait_ge(const unsigned *p, unsigned target, unsigned *err)
{
    int ok = 1;
    if ((threadIdx.x & 63) == 0) {
        ok = 0;
        for (int spin = 0; spin < kSpin; spin++) {
            if (ld_u32(p) >= target) { ok = 1; break; }
            if ((spin & 255) == 255 && ld_u32(err)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return __builtin_amdgcn_readfirstlane(ok) != 0;
}

__device__ __forceinline__ bool lds_wait_ge(volatile unsigned *p, unsigned target, unsigned *err)
{
    int ok = 1;
    if ((threadIdx.x & 63) == 0) {
        ok = 0;
        for (int spin = 0; spin < kSpin; spin++) {
            if (*p >= target) { ok = 1; break; }
            if ((spin & 1023) == 1023 && ld_u32(err)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return __builtin_amdgcn_readfirstlane(ok) != 0;
}

template <int ESW>
__global__ __launch_bounds__(1024) void k_xcd(Params a)
{
    constexpr int E = 8192 * ESW;        // products per chunk per XCD
    constexpr int Ep = E / kTeam;        // per producing workgroup
    constexpr int Epw = Ep / kPW;        // per producer wave (= 32 * ESW)
    constexpr int Es = Ep / kTeam;       // per (producer, consumer) pair
    constexpr int NLD = Epw / 128;       // 16-byte loads per lane per chunk (producer and consumer alike)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *xs = lds;
    double *ys = lds + a.CX;
    __shared__ unsigned lprod[8], lcons[8], lbar, s_team, s_rank, s_ok;
    unsigned *err = a.team + 9 * 32;

    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7;
        s_team = xcc;
        s_rank = __hip_atomic_fetch_add(a.team + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.team + 8 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 0; i < 8; i++) lprod[i] = lcons[i] = 0;
        lbar = 0;
    }
    __syncthreads();
    // everybody resident?  all teams complete?
    bool ok = true;
    if (threadIdx.x < 64) {
        ok = wait_ge(a.team + 8 * 32, gridDim.x, err);
        if (ok && threadIdx.x == 0)
            for (int i = 0; i < 8; i++)
                if (ld_u32(a.team + i * 32) != kTeam) { ok = false; __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        ok = __builtin_amdgcn_readfirstlane(ok ? 1 : 0) != 0;
        if (threadIdx.x == 0) s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    if (!s_ok) return;
    const int t = s_team, k = s_rank;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < a.CX; i += 1024) xs[i] = a.xg[(size_t)k * a.CX + i];
    for (int i = threadIdx.x; i < a.RY; i += 1024) ys[i] = 0.0;
    __syncthreads();
    unsigned *produced = a.ctr + (size_t)t * 2 * 8 * 32;
    unsigned *consumed = produced + 8 * 32;
    double *ring = a.ring + (size_t)t * a.NS * E;

    if (wave < kPW) {
        // ---------------------------------------------------------------- producer
        const int pw = wave;
        d2 v[NLD], vn[NLD];
        us2 c[NLD], cn[NLD];
        auto issue = [&](int ch, d2 *vv, us2 *cc) {
            const size_t base = (((size_t)t * a.T + ch) * kTeam + k) * Ep + (size_t)pw * Epw;
#pragma unroll
            for (int u = 0; u < NLD; u++) {
                const size_t e = base + (size_t)u * 128 + 2 * lane;
                if (a.nt) {
                    vv[u] = __builtin_nontemporal_load((const d2 *)(a.ent_v + e));
                    cc[u] = __builtin_nontemporal_load((const us2 *)(a.ent_c + e));
                } else {
                    vv[u] = *(const d2 *)(a.ent_v + e);
                    cc[u] = *(const us2 *)(a.ent_c + e);
                }
            }
        };
        issue(0, v, c);
        unsigned bar_phase = 0;
        for (int ch = 0; ch < a.T; ch++) {
            const int slot = ch % a.NS, gen = ch / a.NS;
            if (a.XT > 0 && ch > 0 && ch % a.XT == 0) {
                // new x tile: all producer waves have finished gathering from the old one
                bar_phase++;
                if (lane == 0) atomicAdd(&lbar, 1u);
                if (!lds_wait_ge(&lbar, kPW * bar_phase, err)) return;
                for (int i = pw * 64 + lane; i < a.CX; i += kPW * 64) xs[i] = a.xg[(size_t)k * a.CX + i];
                bar_phase++;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) atomicAdd(&lbar, 1u);
                if (!lds_wait_ge(&lbar, kPW * bar_phase, err)) return;
            }
            if (ch + 1 < a.T) issue(ch + 1, vn, cn);
            if (gen > 0 && !wait_ge(consumed + slot * 32, (unsigned)(kTeam * gen), err)) return;
            double *dst = ring + (size_t)slot * E + (size_t)k * Ep + (size_t)pw * Epw;
#pragma unroll
            for (int u = 0; u < NLD; u++) {
                d2 o;
                o.x = v[u].x * xs[c[u].x];
                o.y = v[u].y * xs[c[u].y];
                *(d2 *)(dst + u * 128 + 2 * lane) = o;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                const unsigned old = atomicAdd(&lprod[slot], 1u);
                if (old == (unsigned)(kPW * (gen + 1) - 1))
                    __hip_atomic_fetch_add(produced + slot * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < NLD; u++) { v[u] = vn[u]; c[u] = cn[u]; }
        }
    } else {
        // ---------------------------------------------------------------- consumer
        const int cw = wave - kPW;
        constexpr int LPP = ESW / 2;            // lanes per piece
        constexpr int PPI = 64 / LPP;           // pieces per wave instruction
        for (int ch = 0; ch < a.T; ch++) {
            const int slot = ch % a.NS, gen = ch / a.NS;
            // static row ids of this consumer wave's products, consumer-major
            const size_t rbase = (((size_t)t * a.T + ch) * kTeam + k) * Ep + (size_t)cw * Epw;
            us2 rr[NLD];
#pragma unroll
            for (int u = 0; u < NLD; u++) rr[u] = *(const us2 *)(a.pr + rbase + (size_t)u * 128 + 2 * lane);
            if (!wait_ge(produced + slot * 32, (unsigned)(kTeam * (gen + 1)), err)) return;
            const double *src = ring + (size_t)slot * E + (size_t)k * Es + (size_t)cw * ESW + 2 * (lane % LPP);
            d2 p[NLD];
#pragma unroll
            for (int u = 0; u < NLD; u++) {
                const int piece = u * PPI + lane / LPP;          // producing workgroup
                const double *q = src + (size_t)piece * Ep;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(p[u]) : "v"(q) : "memory");
            }
            if constexpr (NLD == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(p[0]) :: "memory");
            if constexpr (NLD == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(p[0]), "+v"(p[1]) :: "memory");
            if constexpr (NLD == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) :: "memory");
            if constexpr (NLD == 8)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) :: "memory");
            if (lane == 0) {
                const unsigned old = atomicAdd(&lcons[slot], 1u);
                if (old == (unsigned)(kCW * (gen + 1) - 1))
                    __hip_atomic_fetch_add(consumed + slot * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < NLD; u++) {
                unsafeAtomicAdd(&ys[rr[u].x], p[u].x);
                unsafeAtomicAdd(&ys[rr[u].y], p[u].y);
            }
        }
    }
    __syncthreads();
    // checksum of the y tile
    double acc = 0.0;
    for (int i = threadIdx.x; i < a.RY; i += 1024) acc += ys[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ double red[16];
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; w++) s += red[w];
        a.out[t * kTeam + k] = s;
    }
}

// ------------------------------------------------------------------ data + reference
__device__ __forceinline__ unsigned long long mix(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void k_fill(double *v, u16 *c, u16 *r, size_t n, int CX, int RY)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        const unsigned long long h = mix(i);
        v[i] = (double)((int)(h & 7) - 3);
        c[i] = (u16)((h >> 8) % CX);
        r[i] = (u16)((h >> 32) % RY);
    }
}
__global__ void k_fillx(double *x, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = (double)(i % 5 + 1);
}
// expected checksum of consumer (t, j): the products at positions [j*Es, (j+1)*Es) of every producer's piece
__global__ __launch_bounds__(1024) void k_ref(const double *v, const u16 *c, const double *xg, int T, int E, int CX, double *ref)
{
    const int t = blockIdx.x / kTeam, j = blockIdx.x % kTeam;
    const int Ep = E / kTeam, Es = Ep / kTeam;
    double acc = 0.0;
    const size_t total = (size_t)T * kTeam * Es;
    for (size_t q = threadIdx.x; q < total; q += 1024) {
        const size_t ch = q / ((size_t)kTeam * Es), rem = q % ((size_t)kTeam * Es);
        const int kk = (int)(rem / Es), pos = (int)(rem % Es);
        const size_t e = (((size_t)t * T + ch) * kTeam + kk) * Ep + (size_t)j * Es + pos;
        acc += v[e] * xg[(size_t)kk * CX + c[e]];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ double red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; w++) s += red[w];
        ref[blockIdx.x] = s;
    }
}

int main(int argc, char **argv)
{
    const int ESW = argc > 1 ? atoi(argv[1]) : 16;
    const int NS = argc > 2 ? atoi(argv[2]) : 2;
    const int XT = argc > 3 ? atoi(argv[3]) : 3;
    const int nt = argc > 4 ? atoi(argv[4]) : 0;
    const int reps = argc > 5 ? atoi(argv[5]) : 3;
    const int CX = 6144, RY = 13312;
    const int E = 8192 * ESW;
    const size_t TOTAL = 500000000;
    const int T = (int)(TOTAL / 8 / E);
    const size_t n = (size_t)8 * T * E;
    if (NS < 2 || NS > 8 || (ESW != 8 && ESW != 16 && ESW != 32)) { printf("bad args\n"); return 1; }
    double *v, *xg, *ring, *out, *ref;
    u16 *c, *r;
    unsigned *ctr, *team;
    CK(hipMalloc(&v, n * 8)); CK(hipMalloc(&c, n * 2 + 64)); CK(hipMalloc(&r, n * 2 + 64));
    CK(hipMalloc(&xg, (size_t)kTeam * CX * 8));
    CK(hipMalloc(&ring, (size_t)8 * NS * E * 8));
    CK(hipMalloc(&ctr, 8 * 2 * 8 * 32 * 4)); CK(hipMalloc(&team, 10 * 32 * 4));
    CK(hipMalloc(&out, 256 * 8)); CK(hipMalloc(&ref, 256 * 8));
    k_fill<<<2048, 256>>>(v, c, r, n, CX, RY);
    k_fillx<<<(kTeam * CX + 255) / 256, 256>>>(xg, kTeam * CX);
    k_ref<<<256, 1024>>>(v, c, xg, T, E, CX, ref);
    CK(hipDeviceSynchronize());
    Params a{v, c, r, xg, ring, ctr, team, out, T, NS, XT, CX, RY, nt};
    const size_t lds = (size_t)(CX + RY) * 8;
    auto kern = ESW == 8 ? k_xcd<8> : ESW == 16 ? k_xcd<16> : k_xcd<32>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("ESW %d  E %d (%.2f MB of products per chunk per XCD)  NS %d  ring %.2f MB/XCD  T %d  XT %d  nt %d  entries %zu\n", ESW, E,
           E * 8e-6, NS, NS * E * 8e-6, T, XT, nt, n);
    for (int rep = 0; rep < reps; rep++) {
        CK(hipMemsetAsync(ctr, 0, 8 * 2 * 8 * 32 * 4)); CK(hipMemsetAsync(team, 0, 10 * 32 * 4)); CK(hipMemsetAsync(out, 0, 256 * 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), lds, 0, a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned herr[1];
        CK(hipMemcpy(herr, team + 9 * 32, 4, hipMemcpyDeviceToHost));
        std::vector<double> ho(256), hr(256);
        CK(hipMemcpy(ho.data(), out, 256 * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), ref, 256 * 8, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 256; i++) bad += ho[i] != hr[i];
        const double xre = XT > 0 ? (double)(T / XT) * 256 * CX * 8 : 0.0;
        printf("rep %d: %.3f ms  err %u  checksum mismatches %d/256  | stream 12 B/entry = %.2f GB (+%.2f GB x tiles) -> %.0f GB/s\n", rep, ms,
               herr[0], bad, n * 12e-9, xre * 1e-9, (n * 12.0 + xre) / ms * 1e-6);
        fflush(stdout);
        if (herr[0]) break;
    }
    return 0;
}
