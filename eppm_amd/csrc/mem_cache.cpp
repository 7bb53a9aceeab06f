// mem_cache.cpp -- what a destroyed context gives back and the next one of its size takes: slabs, pinned staging buffers, streams.
#include "api_internal.h"

using namespace eppm;

// A program in the reference's demo style makes a fresh object per pair: allocation is inside the window main.cpp times (:63-66).
// hipMalloc of a 95 MB slab costs ~1 ms, hipFree ~3 ms (it drains the device), a pinned staging buffer ~0.3 ms.  Blocks a destroyed
// context gives back are therefore kept -- a few, bounded in bytes -- and handed to the next context that asks for exactly that size.
// Nothing depends on a block's contents: every plane is written before it is read (a fresh hipMalloc block is not zeroed either).
#ifndef EPPM_MEM_CACHE
#define EPPM_MEM_CACHE 1
#endif
namespace {
struct CachedBlock { int device; bool pinned; size_t bytes; void* p; };
std::mutex g_memcache_mu;
std::vector<CachedBlock> g_memcache;
constexpr size_t kMemCacheDeviceBytes = (size_t)3 << 30, kMemCachePinnedBytes = (size_t)512 << 20;
constexpr int kMemCacheBlocks = 12;
}  // namespace

hipError_t cache_alloc(void** p, size_t bytes, bool pinned, int device)
{
    if (EPPM_MEM_CACHE) {
        std::lock_guard<std::mutex> lk(g_memcache_mu);
        for (size_t i = 0; i < g_memcache.size(); i++)
            if (g_memcache[i].pinned == pinned && g_memcache[i].bytes == bytes && g_memcache[i].device == device) {
                *p = g_memcache[i].p;
                g_memcache.erase(g_memcache.begin() + i);
                return hipSuccess;
            }
    }
    hipError_t e = pinned ? hipHostMalloc(p, bytes, hipHostMallocDefault) : hipMalloc(p, bytes);
    if (e != hipSuccess && EPPM_MEM_CACHE) {           // never let kept blocks cause a failure that would not happen without them
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lk(g_memcache_mu);
            for (const CachedBlock& b : g_memcache) { if (b.pinned) (void)hipHostFree(b.p); else (void)hipFree(b.p); }
            g_memcache.clear();
        }
        e = pinned ? hipHostMalloc(p, bytes, hipHostMallocDefault) : hipMalloc(p, bytes);
    }
    return e;
}
// the block must be idle (the caller has synchronised the stream that used it)
void cache_free(void* p, size_t bytes, bool pinned, int device)
{
    if (!p) return;
    if (EPPM_MEM_CACHE) {
        std::lock_guard<std::mutex> lk(g_memcache_mu);
        size_t held = 0;
        int n = 0;
        for (const CachedBlock& b : g_memcache) if (b.pinned == pinned) { held += b.bytes; n++; }
        if (n < kMemCacheBlocks && held + bytes <= (pinned ? kMemCachePinnedBytes : kMemCacheDeviceBytes)) {
            g_memcache.push_back(CachedBlock{device, pinned, bytes, p});
            return;
        }
    }
    if (pinned) (void)hipHostFree(p); else (void)hipFree(p);
}
// Streams likewise: creating one costs about a millisecond (a hardware queue behind it), destroying one as much.  A destroyed context's
// own stream -- idle: eppm_destroy synchronises it first -- goes to a small per-device pool.
namespace {
struct PooledStream { int device; hipStream_t s; };
std::vector<PooledStream> g_streams;          // g_memcache_mu
}
hipError_t pooled_stream_create(hipStream_t* out, int device)
{
    if (EPPM_MEM_CACHE) {
        std::lock_guard<std::mutex> lk(g_memcache_mu);
        for (size_t i = 0; i < g_streams.size(); i++)
            if (g_streams[i].device == device) { *out = g_streams[i].s; g_streams.erase(g_streams.begin() + i); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void pooled_stream_destroy(hipStream_t s, int device)
{
    if (!s) return;
    if (EPPM_MEM_CACHE) {
        std::lock_guard<std::mutex> lk(g_memcache_mu);
        if (g_streams.size() < 16) { g_streams.push_back(PooledStream{device, s}); return; }
    }
    (void)hipStreamDestroy(s);
}
// gives every cached block back to the runtime (memory accounting, tests): slabs, pinned staging buffers, pooled streams, and the
// generator tables no context uses any more (with their drawn-ahead numbers -- up to 512 MB each -- and retired smaller tables).
// What stays resident afterwards: the tables of contexts that still exist, and the three look-up tables per (device, radius) (< 1 KB).
extern "C" int eppm_release_cached_memory(void)
{
    {
        std::lock_guard<std::mutex> lk(g_memcache_mu);
        for (const CachedBlock& b : g_memcache) { if (b.pinned) (void)hipHostFree(b.p); else (void)hipFree(b.p); }
        g_memcache.clear();
        for (const PooledStream& p : g_streams) (void)hipStreamDestroy(p.s);
        g_streams.clear();
    }
    rngtab_release_idle();
    return EPPM_OK;
}
