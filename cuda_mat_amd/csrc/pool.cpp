// pool.cpp -- device memory of libcudamat_hip.so comes from a recycling pool (round 5).
//
// Why: the set-up of a preconditioned solve at the BASELINE sizes allocates ~70 GB in ~340 calls and frees most of it
// again, and the runtime's hipMalloc is not uniformly cheap: `scripts/alloc_churn_probe.py` (ten 8 GB allocations, freed,
// repeated) reads 0.3 ms per call -- and 2.4 SECONDS for one call in every eleven, i.e. once per ~88 GB that a process has
// allocated in total.  Whichever stage of a long-lived process crosses that mark pays it: the resident solver's
// factorisation stage 0.9 s instead of 0.13, a drop-in call 1.2-2.3 s instead of 0.34.  Blocks that are freed
// stay with the library and serve later requests (best fit, split when much larger, merged with free neighbours of the
// same segment when freed), so a process goes to the driver only while its working set still grows.
//
// Semantics kept: pool_free() is a device-wide synchronisation point like hipFree (code that frees a buffer right after
// enqueueing its last use relies on that); contents are whatever the last user left -- exactly as with hipMalloc, which
// promises nothing (every consumer in this library initialises what it reads: the GPU suite runs on recycled blocks).
// Requests below 1 MB go straight to the runtime.  cudamat_pool_trim() and an out-of-memory condition return every
// free segment to the driver; the switch POOL = 0 (read once per process) turns the pool off.
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <map>
#include <mutex>

#include "config.h"

namespace cm {

namespace {

constexpr size_t kGranule = 2u << 20;          // block sizes are multiples of 2 MB (the runtime's own alignment)
constexpr size_t kSmall = 1u << 20;            // smaller requests are not pooled
constexpr size_t kSplitSlack = 64u << 20;      // a free block is split when what is left over is at least this
constexpr size_t kKeepFree = (size_t)96 << 30; // free bytes the pool keeps per device before it trims

struct Block {
    size_t size;
    bool free;
    char *seg;            // base of the hipMalloc'd segment this block was cut from
};

struct DevicePool {
    std::map<char *, Block> blocks;                  // every pooled block by address
    std::multimap<size_t, char *> free_by_size;
    size_t free_bytes = 0;
};

std::mutex g_mu;
std::map<int, DevicePool> g_pools;
std::atomic<unsigned> g_generation{0};
int g_enabled = -1;

bool enabled()
{
    if (g_enabled < 0) g_enabled = config_from_env().pool ? 1 : 0;     // (process-wide, read once: not a per-context switch)
    return g_enabled != 0;
}

void unlist_free(DevicePool &dp, char *base, size_t size)
{
    auto range = dp.free_by_size.equal_range(size);
    for (auto it = range.first; it != range.second; ++it)
        if (it->second == base) { dp.free_by_size.erase(it); break; }
    dp.free_bytes -= size;
}

void list_free(DevicePool &dp, char *base, size_t size)
{
    dp.free_by_size.emplace(size, base);
    dp.free_bytes += size;
}

// return every segment that is one free block to the driver; `keep`: stop once no more than this many free bytes are left
void trim_locked(DevicePool &dp, size_t keep)
{
    for (auto it = dp.blocks.begin(); it != dp.blocks.end() && dp.free_bytes > keep;) {
        Block &b = it->second;
        char *base = it->first;
        auto next = std::next(it);
        const bool whole_segment = b.free && b.seg == base && (next == dp.blocks.end() || next->second.seg != base);
        if (whole_segment) {
            unlist_free(dp, base, b.size);
            (void)hipFree(base);
            g_generation.fetch_add(1);
            it = dp.blocks.erase(it);
        } else {
            it = next;
        }
    }
}

}  // namespace

bool pool_enabled() { return enabled(); }

hipError_t pool_malloc(void **out, size_t bytes)
{
    if (!out) return hipErrorInvalidValue;
    *out = nullptr;
    if (!enabled() || bytes < kSmall) return hipMalloc(out, bytes ? bytes : 1);
    const size_t want = (bytes + kGranule - 1) / kGranule * kGranule;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_mu);
    DevicePool &dp = g_pools[dev];
    auto it = dp.free_by_size.lower_bound(want);
    if (it != dp.free_by_size.end()) {
        char *base = it->second;
        const size_t have = it->first;
        Block &b = dp.blocks[base];
        unlist_free(dp, base, have);
        if (have - want >= kSplitSlack) {                  // cut the request off the front, the rest stays free
            char *rest = base + want;
            dp.blocks[rest] = Block{have - want, true, b.seg};
            list_free(dp, rest, have - want);
            b.size = want;
        }
        b.free = false;
        *out = base;
        return hipSuccess;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipErrorOutOfMemory) {                        // give the driver back what the pool holds and try once more
        (void)hipGetLastError();
        trim_locked(dp, 0);
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) return e;
    dp.blocks[(char *)p] = Block{want, false, (char *)p};
    *out = p;
    return hipSuccess;
}

hipError_t pool_free(void *p)
{
    if (!p) return hipSuccess;
    if (!enabled()) return hipFree(p);
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lock(g_mu);
        DevicePool &dp = g_pools[dev];
        if (dp.blocks.find((char *)p) == dp.blocks.end()) {
            // not ours on this device (a small request, or a block of another device's pool: look there before giving up)
            bool found = false;
            for (auto &kv : g_pools)
                if (kv.second.blocks.count((char *)p)) { dev = kv.first; found = true; break; }
            if (!found) return hipFree(p);
        }
    }
    // hipFree is a device-wide synchronisation point and callers rely on it: nothing that still runs may see the block reused
    int cur = dev;
    (void)hipGetDevice(&cur);
    if (cur != dev) (void)hipSetDevice(dev);
    const hipError_t es = hipDeviceSynchronize();
    if (cur != dev) (void)hipSetDevice(cur);
    std::lock_guard<std::mutex> lock(g_mu);
    DevicePool &dp = g_pools[dev];
    auto it = dp.blocks.find((char *)p);
    if (it == dp.blocks.end() || it->second.free) return hipErrorInvalidValue;
    it->second.free = true;
    // merge with the free neighbours of the same segment
    auto next = std::next(it);
    if (next != dp.blocks.end() && next->second.free && next->second.seg == it->second.seg && it->first + it->second.size == next->first) {
        unlist_free(dp, next->first, next->second.size);
        it->second.size += next->second.size;
        dp.blocks.erase(next);
    }
    if (it != dp.blocks.begin()) {
        auto prev = std::prev(it);
        if (prev->second.free && prev->second.seg == it->second.seg && prev->first + prev->second.size == it->first) {
            unlist_free(dp, prev->first, prev->second.size);
            prev->second.size += it->second.size;
            dp.blocks.erase(it);
            it = prev;
        }
    }
    list_free(dp, it->first, it->second.size);
    if (dp.free_bytes > kKeepFree) trim_locked(dp, kKeepFree);
    return es;
}

// keep the first `bytes` of a pooled block, hand the rest back to the free list (a caller that asked for more than it keeps:
// the placement search of spmv_pb.hip walks device memory in steps larger than the array it places)
void pool_shrink(void *p, size_t bytes)
{
    if (!p || !enabled()) return;
    const size_t want = (bytes + kGranule - 1) / kGranule * kGranule;
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto &kv : g_pools) {
        DevicePool &dp = kv.second;
        auto it = dp.blocks.find((char *)p);
        if (it == dp.blocks.end()) continue;
        Block &b = it->second;
        if (b.free || b.size < want + kSplitSlack) return;
        char *rest = (char *)p + want;
        const size_t rest_size = b.size - want;
        b.size = want;
        // (nothing of this process ever touched the tail except the placement probe, which has completed: no synchronisation)
        auto ins = dp.blocks.emplace(rest, Block{rest_size, true, b.seg}).first;
        auto next = std::next(ins);
        if (next != dp.blocks.end() && next->second.free && next->second.seg == b.seg && rest + rest_size == next->first) {
            unlist_free(dp, next->first, next->second.size);
            ins->second.size += next->second.size;
            dp.blocks.erase(next);
        }
        list_free(dp, rest, ins->second.size);
        return;
    }
}

// the driver allocation (segment) a pooled block was cut from; false for pointers the pool does not know
bool pool_segment_of(const void *p, char **seg_base, size_t *block_bytes)
{
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto &kv : g_pools) {
        auto it = kv.second.blocks.find((char *)p);
        if (it == kv.second.blocks.end()) continue;
        if (seg_base) *seg_base = it->second.seg;
        if (block_bytes) *block_bytes = it->second.size;
        return true;
    }
    return false;
}

// one allocated block becomes two allocated blocks, the second starting `offset` bytes in (a multiple of the pool's 2 MB
// granule); each is then freed on its own.  false: not a pooled block in use, or the offset does not fit
bool pool_split(void *p, size_t offset)
{
    if (!p || offset == 0 || offset % kGranule) return false;
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto &kv : g_pools) {
        DevicePool &dp = kv.second;
        auto it = dp.blocks.find((char *)p);
        if (it == dp.blocks.end()) continue;
        Block &b = it->second;
        if (b.free || offset >= b.size) return false;
        dp.blocks[(char *)p + offset] = Block{b.size - offset, false, b.seg};
        b.size = offset;
        return true;
    }
    return false;
}

// counts the segments handed back to the driver: whoever remembers something about pooled ADDRESSES (the memory classes of
// spmv_pb.hip's placement) forgets it when this moves
unsigned pool_generation() { return g_generation.load(); }

// everything the pool holds free goes back to the driver (cudamat_pool_trim, out-of-memory retries)
void pool_trim()
{
    std::lock_guard<std::mutex> lock(g_mu);
    int before = 0;
    (void)hipGetDevice(&before);
    for (auto &kv : g_pools) {
        (void)hipSetDevice(kv.first);
        trim_locked(kv.second, 0);
    }
    (void)hipSetDevice(before);
}

// free bytes the pool of the current device could hand out without asking the driver
size_t pool_free_bytes()
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_pools.find(dev);
    return it == g_pools.end() ? 0 : it->second.free_bytes;
}

}  // namespace cm
