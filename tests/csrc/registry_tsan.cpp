// registry_tsan.cpp -- the host-memory registry (eppm_amd/csrc/host_registry.h: refcounts, aliases, `closing`, bounded waits) under
// ThreadSanitizer on the CPU, with the two runtime calls (hipHostRegister / hipHostUnregister) replaced by fakes that keep a "pinned"
// flag per block.  T threads x N random operations on B blocks: register the block, register a range inside it, hold it for a transfer
// (acquire .. release), give a registration back, query.  Checked all along:
//   * the fakes: a block is pinned at most once at a time and unpinned only while pinned;
//   * a transfer in flight (between acquire and release) never sees its block unpinned -- the contract the last owner's wait exists for;
//   * a thread that OWNS a registration finds its block pinned and registered (the lost-owner race of round 5: an owner arriving while
//     the last one leaves must end up owning a pinned block) unless it is itself inside the window in which another owner is leaving;
//   * an unregister of something this thread registered never reports "not registered" (the alias bookkeeping of round 4);
// and at the end, when every thread has given back what it holds: no block, no alias, nothing pinned, pins == unpins.
// Built by tests/test_abi_cpu.py with -fsanitize=thread; exit code 0 and no sanitizer report = pass.
// usage: registry_tsan [threads] [ops per thread]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "host_registry.h"

namespace {
constexpr int kBlocks = 6;
constexpr size_t kBlockBytes = 4096;
char* g_mem[kBlocks];
std::atomic<int> g_pinned[kBlocks];
std::atomic<long> g_pins{0}, g_unpins{0}, g_errors{0};

int block_of(const void* p)
{
    for (int b = 0; b < kBlocks; b++)
        if ((const char*)p >= g_mem[b] && (const char*)p < g_mem[b] + kBlockBytes) return b;
    return -1;
}
void fail(const char* what, int b)
{
    fprintf(stderr, "registry_tsan: %s (block %d)\n", what, b);
    g_errors++;
}
int fake_pin(void* p, size_t)
{
    const int b = block_of(p);
    if (b < 0) { fail("pin of an unknown pointer", b); return 1; }
    if ((char*)p != g_mem[b]) return 1;          // a range inside a block nobody has registered: this fake pins whole blocks only (add reports kPinFailed)
    if (g_pinned[b].exchange(1) != 0) fail("block pinned twice", b);
    g_pins++;
    return 0;
}
int fake_unpin(void* p)
{
    const int b = block_of(p);
    if (b < 0) { fail("unpin of an unknown pointer", b); return 1; }
    if (g_pinned[b].exchange(0) != 1) fail("unpin of a block that is not pinned", b);
    g_unpins++;
    return 0;
}
eppm::HostRegistry g_reg(fake_pin, fake_unpin, /*wait_ms*/ 3);
}  // namespace

eppm::HostRegistry* eppm::default_host_registry() { return &g_reg; }

static void worker(int tid, long ops)
{
    unsigned s = 12345u + 7919u * (unsigned)tid;
    auto rnd = [&](unsigned n) { s = s * 1664525u + 1013904223u; return (s >> 8) % n; };
    std::vector<void*> mine;                 // pointers this thread registered and has not given back
    for (long i = 0; i < ops; i++) {
        const int b = (int)rnd(kBlocks);
        unsigned op = rnd(8);
        if (op < 2 && mine.size() >= 2) op = 2;          // at most two registrations per thread: blocks really come and go (their last owner leaves ~10^4 times a run)
        switch (op) {
            case 0: case 1: {                // register the whole block, or a range inside it (an alias once the block is known)
                const size_t off = rnd(2) ? 0 : 64 * (1 + rnd(8));
                void* p = g_mem[b] + off;
                const eppm::HostRegistry::Status st = g_reg.add(p, off ? 256 : kBlockBytes);
                if (st == eppm::HostRegistry::kOk) mine.push_back(p);
                else if (st != eppm::HostRegistry::kPinFailed) fail("register: unexpected status", b);      // (an inner range of an unknown block: the fake refuses it)
                break;
            }
            case 2: case 3: {                // give one registration back
                if (mine.empty()) break;
                const size_t k = rnd((unsigned)mine.size());
                const eppm::HostRegistry::Status st = g_reg.remove(mine[k]);
                if (st == eppm::HostRegistry::kOk) { mine[k] = mine.back(); mine.pop_back(); }
                else if (st == eppm::HostRegistry::kNotRegistered) fail("unregister: a registration this thread holds was not found", block_of(mine[k]));
                else if (st != eppm::HostRegistry::kBusy && st != eppm::HostRegistry::kBeingUnregistered) fail("unregister: unexpected status", block_of(mine[k]));
                break;
            }
            case 4: case 5: case 6: {        // a transfer: acquire, "copy", release; the block stays pinned throughout
                eppm::HostHold hold;
                if (hold.add(g_mem[b] + 128, 512)) {
                    for (int spin = 0; spin < 20; spin++)
                        if (g_pinned[b].load() != 1) { fail("block unpinned under a transfer in flight", b); break; }
                    if (rnd(16) == 0) std::this_thread::yield();
                    if (rnd(2048) == 0) std::this_thread::sleep_for(std::chrono::milliseconds(6));      // longer than the last owner's bounded wait (3 ms here): its unregister must fail with kBusy
                    if (g_pinned[b].load() != 1) fail("block unpinned under a transfer in flight", b);
                }
                break;
            }
            default: {                       // query; an owner of the whole block sees it pinned
                const bool reg = g_reg.registered(g_mem[b], kBlockBytes);
                if (reg && g_pinned[b].load() != 1) {
                    // registered() and the load are two steps: only a block this thread itself keeps registered cannot have gone in between
                    bool own = false;
                    for (void* p : mine) own = own || block_of(p) == b;
                    if (own) fail("a registered block this thread owns is not pinned", b);
                }
                break;
            }
        }
    }
    // give everything back (kBusy: a transfer of another thread is in flight -- retry)
    while (!mine.empty()) {
        const eppm::HostRegistry::Status st = g_reg.remove(mine.back());
        if (st == eppm::HostRegistry::kOk) mine.pop_back();
        else if (st != eppm::HostRegistry::kBusy && st != eppm::HostRegistry::kBeingUnregistered) { fail("final unregister failed", block_of(mine.back())); mine.pop_back(); }
    }
}

int main(int argc, char** argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 8;
    const long ops = argc > 2 ? atol(argv[2]) : 100000;
    for (int b = 0; b < kBlocks; b++) { g_mem[b] = (char*)malloc(kBlockBytes); g_pinned[b] = 0; }
    // memory that arrives pinned (eppm_host_alloc) with a range registered on top, freed while transfers come and go
    char* owned = (char*)malloc(kBlockBytes);
    g_reg.add_owned(owned, kBlockBytes);
    if (g_reg.add(owned + 64, 128) != eppm::HostRegistry::kOk || g_reg.remove(owned + 64) != eppm::HostRegistry::kOk) fail("owner on top of an owned block", -1);
    if (g_reg.remove(owned) != eppm::HostRegistry::kOwnedBlock) fail("an owned block must be freed, not unregistered", -1);
    // the bounded wait, deterministically: the only owner unregisters while a transfer of another thread is in flight for longer than the
    // wait -> kBusy, the block stays pinned and registered, and a retry after the transfer succeeds
    {
        if (g_reg.add(g_mem[0], kBlockBytes) != eppm::HostRegistry::kOk) fail("scenario: register", 0);
        std::atomic<int> holding{0};
        std::thread xfer([&] {
            eppm::HostHold h;
            if (!h.add(g_mem[0], 256)) fail("scenario: acquire", 0);
            holding = 1;
            std::this_thread::sleep_for(std::chrono::milliseconds(40));
            if (g_pinned[0].load() != 1) fail("scenario: block unpinned under a transfer in flight", 0);
        });
        while (!holding.load()) std::this_thread::yield();
        if (g_reg.remove(g_mem[0]) != eppm::HostRegistry::kBusy) fail("scenario: the last owner must not unpin under a transfer in flight", 0);
        if (!g_reg.registered(g_mem[0], kBlockBytes) || g_pinned[0].load() != 1) fail("scenario: the block must stay registered and pinned after kBusy", 0);
        xfer.join();
        if (g_reg.remove(g_mem[0]) != eppm::HostRegistry::kOk || g_pinned[0].load() != 0) fail("scenario: retry after the transfer", 0);
    }
    std::vector<std::thread> ts;
    for (int t = 0; t < threads; t++) ts.emplace_back(worker, t, ops);
    std::thread freer([&] {
        for (int k = 0; k < 200; k++) {
            eppm::HostHold h;
            h.add(owned, 64);
        }
    });
    for (auto& t : ts) t.join();
    freer.join();
    if (g_reg.remove_owned(owned) != eppm::HostRegistry::kOk) fail("eppm_host_free of an idle owned block", -1);
    if (g_reg.blocks() != 0 || g_reg.aliases() != 0) fail("entries left in the registry", -1);
    for (int b = 0; b < kBlocks; b++)
        if (g_pinned[b].load() != 0) fail("block left pinned", b);
    if (g_pins.load() != g_unpins.load()) fail("pins != unpins", -1);
    printf("registry_tsan: %d threads x %ld operations, %ld pins, %ld unpins, %ld errors\n", threads, ops, g_pins.load(), g_unpins.load(), g_errors.load());
    for (int b = 0; b < kBlocks; b++) free(g_mem[b]);
    free(owned);
    return g_errors.load() ? 1 : 0;
}
