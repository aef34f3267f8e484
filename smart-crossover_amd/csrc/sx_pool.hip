// Device memory pool of the library: every hipMalloc / hipFree of the kernels' host code goes through sx_dmalloc / sx_dfree
// (sx_internal.h), which keep freed blocks for the next request of the same size class instead of returning them to the
// driver.  Why: a crossover (the reference's run_perturb_algorithm, lp_methods/algorithms.py:42-74, is called once per LP and
// its solver allocates per call) allocates and frees ~1.3 GB of factors, sweep responses and matrices at the headline
// size and ~10 GB at config-5 size, and a pipeline of crossovers asks for the same sizes again and again: the second call
// of a process makes no driver allocation at all (tests/test_gpu_pool.py).  Measured (profiles/r04/in_bench_slowdown.md):
// 3.6 % of a config-5-size crossover, nothing at the headline size.
//
// Semantics kept from hipFree: sx_dfree waits for the device (hipFree synchronises implicitly; callers rely on it when
// they free buffers of kernels still in flight).  Not kept: fresh driver pages read as zeros, pooled blocks do not --
// nothing in the library reads memory it has not written.
//
// Policy: size classes of 8 steps per octave (<= 12.5 % over-allocation), exact class match; blocks above
// SX_POOL_BLOCK_MAX (default 8 GiB) bypass the pool; at most SX_POOL_MAX (default 16 GiB) cached per device -- beyond it
// the largest cached blocks go back to the driver; a failed hipMalloc empties the pool and tries again.  SX_POOL=0: off.
#include "sx_internal.h"

#include <map>
#include <mutex>
#include <unordered_map>

namespace {

struct Pool {
    std::mutex mu;
    // per device: cached blocks by class size, live blocks by address
    std::map<int, std::multimap<size_t, void *>> cached;
    std::map<int, size_t> cached_bytes;
    struct Live {
        int dev;
        size_t bytes;
        bool pooled;
    };
    std::unordered_map<void *, Live> live;
    uint64_t hits = 0, misses = 0;
    size_t live_bytes = 0;
    size_t max_bytes = static_cast<size_t>(16) << 30;
    size_t block_max = static_cast<size_t>(8) << 30;
    int enabled = 1;
    Pool() {
        if (const char *e = getenv("SX_POOL")) enabled = atoi(e) != 0;
        if (const char *e = getenv("SX_POOL_MAX")) max_bytes = static_cast<size_t>(strtoull(e, nullptr, 10));
        if (const char *e = getenv("SX_POOL_BLOCK_MAX")) block_max = static_cast<size_t>(strtoull(e, nullptr, 10));
    }
};

// leaked on purpose: a static destructor would call hipFree after the runtime has shut down
Pool &pool() {
    static Pool *p = new Pool();
    return *p;
}

size_t size_class(size_t bytes) {
    if (bytes <= 4096) return (bytes + 511) & ~static_cast<size_t>(511);
    size_t p2 = static_cast<size_t>(1) << (63 - __builtin_clzll(bytes));
    size_t step = p2 >> 3;
    return (bytes + step - 1) & ~(step - 1);
}

// (lock held) give cached blocks of `dev` back to the driver until at most `keep` bytes stay, largest first
void shrink(Pool &P, int dev, size_t keep) {
    auto &mm = P.cached[dev];
    size_t &have = P.cached_bytes[dev];
    while (have > keep && !mm.empty()) {
        auto it = std::prev(mm.end());
        (void)hipFree(it->second);
        have -= it->first;
        mm.erase(it);
    }
}

} // namespace

hipError_t sx_pool_malloc(void **p, size_t bytes) {
    if (!p) return hipErrorInvalidValue;
    *p = nullptr;
    Pool &P = pool();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (bytes == 0) bytes = 8;
    const bool pooled = P.enabled && bytes <= P.block_max;
    const size_t cls = pooled ? size_class(bytes) : bytes;
    if (pooled) {
        std::lock_guard<std::mutex> lock(P.mu);
        auto &mm = P.cached[dev];
        auto it = mm.find(cls);
        if (it != mm.end()) {
            *p = it->second;
            mm.erase(it);
            P.cached_bytes[dev] -= cls;
            P.live[*p] = {dev, cls, true};
            P.live_bytes += cls;
            ++P.hits;
            return hipSuccess;
        }
    }
    // (outside the lock: tens of GB take a second, and the context's helper thread allocates beside the caller)
    hipError_t e = hipMalloc(p, cls);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        {
            std::lock_guard<std::mutex> lock(P.mu);
            shrink(P, dev, 0);
        }
        e = hipMalloc(p, cls);
        if (e != hipSuccess) {
            *p = nullptr;
            return e;
        }
    }
    std::lock_guard<std::mutex> lock(P.mu);
    P.live[*p] = {dev, cls, pooled};
    P.live_bytes += cls;
    ++P.misses;
    return hipSuccess;
}

hipError_t sx_pool_free(void *p) {
    if (!p) return hipSuccess;
    Pool &P = pool();
    int dev = -1;
    size_t cls = 0;
    bool keep = false;
    {
        std::lock_guard<std::mutex> lock(P.mu);
        auto it = P.live.find(p);
        if (it != P.live.end()) {
            keep = it->second.pooled;
            dev = it->second.dev;
            cls = it->second.bytes;
            P.live_bytes -= std::min(P.live_bytes, cls);
            P.live.erase(it);
        }
    }
    if (!keep) return hipFree(p); // not ours, or a block too large to keep
    // hipFree's implicit synchronisation: kernels that still use the block must have finished before it is handed out again
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != dev) (void)hipSetDevice(dev);
    hipError_t e = hipDeviceSynchronize();
    if (cur != dev && cur >= 0) (void)hipSetDevice(cur);
    std::lock_guard<std::mutex> lock(P.mu);
    if (!P.enabled || cls > P.max_bytes) return hipFree(p);
    P.cached[dev].insert({cls, p});
    P.cached_bytes[dev] += cls;
    if (P.cached_bytes[dev] > P.max_bytes) shrink(P, dev, P.max_bytes);
    return e;
}

// Cached blocks back to the driver (all devices).  For a host that wants the memory for something else between calls.
SX_API int sx_pool_trim(void) {
    Pool &P = pool();
    std::lock_guard<std::mutex> lock(P.mu);
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (auto &kv : P.cached) {
        if (kv.second.empty()) continue;
        (void)hipSetDevice(kv.first);
        (void)hipDeviceSynchronize();
        shrink(P, kv.first, 0);
    }
    if (cur >= 0) (void)hipSetDevice(cur);
    return SX_OK;
}

// cached / live bytes (all devices) and the number of requests served from the pool / by the driver since the library was loaded
SX_API int sx_pool_stats(uint64_t *cached_bytes, uint64_t *live_bytes, uint64_t *hits, uint64_t *misses) {
    Pool &P = pool();
    std::lock_guard<std::mutex> lock(P.mu);
    uint64_t c = 0;
    for (auto &kv : P.cached_bytes) c += kv.second;
    if (cached_bytes) *cached_bytes = c;
    if (live_bytes) *live_bytes = P.live_bytes;
    if (hits) *hits = P.hits;
    if (misses) *misses = P.misses;
    return SX_OK;
}
