// probe_barrier.hip -- what does one grid-wide synchronisation cost on MI355X, and how much cheaper is it when all
// participating workgroups sit on ONE XCD (their L2 is then the point of coherence: no write-back / invalidate, no trip
// through the fabric)?  Each variant runs N rounds of: write a value, synchronise, read the neighbour's value (checked).
//   variant 0  all XCDs, agent-scope fences (buffer_wbl2 sc1 / buffer_inv sc1) + agent-scope atomic + sc1 polling
//   variant 1  all XCDs, no fences: data through sc1 stores / sc1 loads, agent-scope atomic + sc1 polling
//   variant 2  one XCD (the workgroups that find themselves on XCC 0 stay, the others leave at once): data through
//              plain stores (write-through L1) and sc0 loads (L1 bypass), atomic executed in the L2, sc0 polling
// build: hipcc --offload-arch=gfx950 -O3 -o probe_barrier probe_barrier.hip ; run: ./probe_barrier <team> <rounds>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr unsigned kSpin = 1u << 22;

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__device__ __forceinline__ unsigned ld_sc0_u32(const unsigned *p)
{
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sc0_f64(double *p, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ double ld_l2_f64(const double *p)     // an atomic executes in the L2: OR 0 returns what the L2 holds
{
    unsigned long long v = __hip_atomic_fetch_or((unsigned long long *)p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __longlong_as_double((long long)v);
}
__device__ __forceinline__ double ld_sc0_f64(const double *p)
{
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

struct Args {
    unsigned *ctr;      // [0] barrier arrivals, [32] error flag, [64] team ticket, [96] started, [128..] started per xcc
    double *slots;      // one 128-byte line per participant
    unsigned *bad;      // mismatches
    int variant, rounds, launched;
};

__global__ __launch_bounds__(256) void k_probe(Args a)
{
    __shared__ int s_rank, s_team, s_ok;
    const int tid = threadIdx.x;
    unsigned *arrive = a.ctr, *err = a.ctr + 32, *ticket = a.ctr + 64, *started = a.ctr + 96;
    if (tid == 0) {
        int rank = -1, team = 0;
        if (a.variant >= 2) {
            const unsigned x = xcc_id();
            if (x == 0) rank = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (x == 0) {       // the team is complete once every launched workgroup has reported
                unsigned spins = 0;
                while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.launched && ++spins < kSpin)
                    __builtin_amdgcn_s_sleep(2);
                if (spins >= kSpin) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                team = (int)__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            rank = (int)blockIdx.x;
            team = (int)gridDim.x;
        }
        s_rank = rank;
        s_team = team;
    }
    __syncthreads();
    const int rank = s_rank, team = s_team;
    if (rank < 0) return;
    unsigned epoch = 0, bad = 0;
    double *mine = a.slots + (size_t)rank * 16;
    const double *next = a.slots + (size_t)((rank + 1) % team) * 16;
    for (int r = 1; r <= a.rounds; r++) {
        const double val = (double)r * 1024.0 + rank;
        if (tid == 0) {
            if (a.variant == 1 || a.variant == 6) __hip_atomic_store(mine, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (a.variant == 3 || a.variant == 5) st_sc0_f64(mine, val);
            else *mine = val;
        }
        __syncthreads();
        if (tid == 0) {
            bool good = true;
            epoch += (unsigned)team;
            if (a.variant == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned sp = 0; __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch; sp++)
                    if (sp > kSpin) { good = false; break; }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            } else if (a.variant == 1 || a.variant == 6) {
                __builtin_amdgcn_s_waitcnt(0);          // the sc1 store has left
                __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned sp = 0; __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch; sp++)
                    if (sp > kSpin) { good = false; break; }
            } else if (a.variant == 7) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                for (unsigned sp = 0; __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch; sp++)
                    if (sp > kSpin) { good = false; break; }
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // stores have reached the L2
                __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // executed in the L2
                for (unsigned sp = 0; ld_sc0_u32(arrive) < epoch; sp++)
                    if (sp > kSpin) { good = false; break; }
            }
            if (!good) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ok = good;
        }
        __syncthreads();
        if (!s_ok) break;
        if (tid == 0) {
            double got;
            if (a.variant == 0) got = *(volatile const double *)next;
            else if (a.variant == 1 || a.variant >= 6) got = __hip_atomic_load(next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (a.variant >= 4) got = ld_l2_f64(next);
            else got = ld_sc0_f64(next);
            if (got != (double)r * 1024.0 + (rank + 1) % team) {
                bad++;
            }
        }
        // second half: nobody may overwrite its slot before the neighbour has read it
        __syncthreads();
        if (tid == 0) {
            bool good = true;
            epoch += (unsigned)team;
            if (a.variant == 7) {
                __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                for (unsigned sp = 0; __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch; sp++)
                    if (sp > kSpin) { good = false; break; }
            } else if (a.variant >= 2 && a.variant != 6) {
                __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                for (unsigned sp = 0; ld_sc0_u32(arrive) < epoch; sp++)
                    if (sp > kSpin) { good = false; break; }
            } else {
                __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned sp = 0; __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch; sp++)
                    if (sp > kSpin) { good = false; break; }
            }
            if (!good) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ok = good;
        }
        __syncthreads();
        if (!s_ok) break;
    }
    if (tid == 0 && bad) atomicAdd(a.bad, bad);
}

int main(int argc, char **argv)
{
    const int team = argc > 1 ? atoi(argv[1]) : 40;
    const int rounds = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned *ctr, *bad;
    double *slots;
    CK(hipMalloc(&ctr, 4096));
    CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&slots, 128 * 2048));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 8; variant++) {
        if (variant >= 2 && variant <= 5) continue;     // sc0 loads hit the reader's L1 (stale): see the log
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(ctr, 0, 4096));
            CK(hipMemset(bad, 0, 4));
            CK(hipMemset(slots, 0, 128 * 2048));
            Args a{ctr, slots, bad, variant, rounds, variant >= 2 ? team * 8 : team};
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_probe, dim3(a.launched), dim3(256), 0, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned h[256], hb = 0;
            CK(hipMemcpy(h, ctr, sizeof(h), hipMemcpyDeviceToHost));
            CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            if (rep == 1)
                printf("variant %d  team %3u  %6.3f us per synchronisation (2 per round, %d rounds)  mismatches %u  timeout %u\n", variant,
                       variant >= 2 ? h[64] : (unsigned)team, ms * 1e3 / (2.0 * rounds), rounds, hb, h[32]);
        }
    }
    return 0;
}
