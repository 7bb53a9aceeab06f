// host_registry.h -- caller memory pinned for DMA (eppm_host_register / eppm_host_alloc): which blocks are pinned, who owns a
// registration, which transfers are in flight on a block.  set_images reads registered images and compute writes registered flow planes
// directly over PCIe -- no staging copy on either side (set_data / compute_flow's cudaMemcpy legs, driver :159-168, :299-306, read and
// write the caller's memory too; ownership contract: SURVEY section 8(b), driver :101-104, :170-209).
//
// Header-only and free of HIP: the two runtime calls (pin = hipHostRegister, unpin = hipHostUnregister) are injected, so that the
// refcount / alias / closing / wait logic -- plain host C++, and where rounds 4 and 5 each found a race -- runs under ThreadSanitizer on
// the CPU (tests/csrc/registry_tsan.cpp, part of the -m "not gpu" suite).  host_registry.cpp binds it to HIP and to the C ABI.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <vector>

namespace eppm {

class HostRegistry {
public:
    enum Status { kOk = 0, kNotRegistered, kOwnedBlock, kBusy, kBeingFreed, kPinFailed, kBeingUnregistered };
    typedef int (*PinFn)(void* p, size_t bytes);          // 0 = pinned
    typedef int (*UnpinFn)(void* p);                      // 0 = unpinned
    HostRegistry(PinFn pin, UnpinFn unpin, int wait_ms = 5000) : pin_(pin), unpin_(unpin), wait_ms_(wait_ms) {}

    // refs: register calls outstanding on the block (several owners may register the same block -- runeppm --gpus N --pin shares its
    // images between the workers' objects -- and it is unpinned when the LAST of them unregisters); users: DMA transfers of contexts in
    // flight on it (held from the look-up that decides "read / write in place" until the transfer has completed: a look-up and its
    // hipMemcpyAsync are one critical step with respect to unregistration).  closing: the last owner is waiting for the users to drain
    // before it unpins -- no new transfer starts on the block (acquire skips it, the caller stages), and a concurrent register of a
    // registered (not owned) block REVIVES it: the waiter then leaves the pinning to the new owner.
    struct Block { size_t bytes; bool owned; int refs; int users; bool closing; };

    bool registered(const void* p, size_t bytes)
    {
        if (!p) return false;
        std::lock_guard<std::mutex> lk(mu_);
        auto it = covering(p, bytes);
        return it != reg_.end() && !it->second.closing;
    }
    // the block covering [p, p + bytes) marked in use (0 when there is none): the caller DMAs from / into it, then release(base)
    uintptr_t acquire(const void* p, size_t bytes)
    {
        if (!p) return 0;
        std::lock_guard<std::mutex> lk(mu_);
        auto it = covering(p, bytes);
        if (it == reg_.end() || it->second.closing) return 0;
        it->second.users++;
        return it->first;
    }
    void release(uintptr_t base)
    {
        if (!base) return;
        std::lock_guard<std::mutex> lk(mu_);
        auto it = reg_.find(base);
        if (it != reg_.end() && it->second.users > 0 && --it->second.users == 0) cv_.notify_all();
    }
    // eppm_host_register.  pin_status: what the injected pin call returned when the result is kPinFailed.
    Status add(void* p, size_t bytes, int* pin_status = nullptr)
    {
        std::lock_guard<std::mutex> lk(mu_);          // held across the check and the pin call: two threads registering one block
        auto it = covering(p, bytes);
        if (it != reg_.end()) {
            if (it->second.closing) {
                if (it->second.owned) return kBeingFreed;
                it->second.closing = false;           // the last owner was on its way out: this owner keeps the pages pinned
                cv_.notify_all();
            }
            it->second.refs++;
            if (it->first != (uintptr_t)p) alias_.emplace((uintptr_t)p, it->first);
            return kOk;
        }
        const int e = pin_(p, bytes);
        if (e != 0) { if (pin_status) *pin_status = e; return kPinFailed; }
        reg_[(uintptr_t)p] = Block{bytes, false, 1, 0, false};
        return kOk;
    }
    // eppm_host_unregister
    Status remove(void* p, int* unpin_status = nullptr)
    {
        std::unique_lock<std::mutex> lk(mu_);
        const uintptr_t key = (uintptr_t)p;
        auto al = alias_.find(key);
        auto it = (al != alias_.end()) ? reg_.find(al->second) : reg_.find(key);
        if (it == reg_.end()) return kNotRegistered;
        if (it->second.owned) {      // a range inside eppm_host_alloc memory was registered on top: drop that owner; the block itself goes with eppm_host_free
            if (it->second.refs <= 1) return kOwnedBlock;
            it->second.refs--;
            drop_alias(key);
            return kOk;
        }
        if (it->second.refs > 1) { it->second.refs--; drop_alias(key); return kOk; }          // another owner still holds the registration
        // no owner left to give up: the last one is inside its wait below (a second unregister of the same registration)
        if (it->second.refs == 0) return kBeingUnregistered;
        // last owner: wait for the transfers in flight on the block (a context of another thread between its look-up and the end of its
        // copy); bounded, so that unregistering under one's own pending eppm_compute_begin_into is an error and not a deadlock.  The wait
        // drops the lock: the block is marked closing meanwhile (no new transfer starts on it), and a thread that registers it again in
        // that window becomes its owner -- the pages then stay pinned.  The alias entry goes only when this owner is really gone, so a
        // retry after kBusy finds the block through the same pointer.
        const uintptr_t base = it->first;
        it->second.refs = 0;
        it->second.closing = true;
        const bool idle = wait(lk, [&] {
            auto q = reg_.find(base);
            return q == reg_.end() || !q->second.closing || q->second.users == 0;
        });
        it = reg_.find(base);
        if (it == reg_.end()) { drop_alias(key); return kOk; }
        if (!it->second.closing) { drop_alias(key); return kOk; }              // revived by a concurrent register: its owner now
        if (!idle || it->second.users != 0) {
            it->second.refs = 1; it->second.closing = false;
            return kBusy;
        }
        reg_.erase(it);
        drop_alias(key);
        // unpinned under the lock: a thread that registers the same block right now must find either the entry or unpinned pages
        const int e = unpin_((void*)base);
        if (e != 0) { if (unpin_status) *unpin_status = e; return kPinFailed; }
        return kOk;
    }
    // eppm_host_alloc: memory that arrives pinned and is freed, not unpinned
    void add_owned(void* p, size_t bytes)
    {
        std::lock_guard<std::mutex> lk(mu_);
        reg_[(uintptr_t)p] = Block{bytes, true, 1, 0, false};
    }
    // eppm_host_free: kOk = the entry is gone and the caller frees the memory
    Status remove_owned(void* p)
    {
        std::unique_lock<std::mutex> lk(mu_);
        auto it = reg_.find((uintptr_t)p);
        if (it == reg_.end() || !it->second.owned) return kNotRegistered;
        it->second.closing = true;                     // no new transfer starts on memory that is about to go
        const bool idle = wait(lk, [&] { auto q = reg_.find((uintptr_t)p); return q == reg_.end() || q->second.users == 0; });
        it = reg_.find((uintptr_t)p);
        if (it == reg_.end()) return kOk;
        if (!idle) { it->second.closing = false; return kBusy; }
        reg_.erase(it);
        return kOk;
    }
    size_t blocks()
    {
        std::lock_guard<std::mutex> lk(mu_);
        return reg_.size();
    }
    size_t aliases()
    {
        std::lock_guard<std::mutex> lk(mu_);
        return alias_.size();
    }

private:
    // the bounded wait of the last owner.  A system_clock deadline: libstdc++ then waits with pthread_cond_timedwait, which every
    // ThreadSanitizer runtime intercepts (the steady-clock form, pthread_cond_clockwait, is unknown to GCC 11's and makes it report the
    // mutex as locked twice)
    template <class Pred>
    bool wait(std::unique_lock<std::mutex>& lk, Pred pred)
    {
        return cv_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(wait_ms_), pred);
    }
    std::map<uintptr_t, Block>::iterator covering(const void* p, size_t bytes)          // mu_ held
    {
        auto it = reg_.upper_bound((uintptr_t)p);
        if (it == reg_.begin()) return reg_.end();
        --it;
        return ((uintptr_t)p + bytes <= it->first + it->second.bytes) ? it : reg_.end();
    }
    void drop_alias(uintptr_t p)                                                              // mu_ held; one entry (equal keys map to one base)
    {
        auto al = alias_.find(p);
        if (al != alias_.end()) alias_.erase(al);
    }
    PinFn pin_;
    UnpinFn unpin_;
    int wait_ms_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::map<uintptr_t, Block> reg_;
    std::multimap<uintptr_t, uintptr_t> alias_;       // pointer registered INSIDE an existing block -> that block's base
};

// the registry a program's transfers go through: defined once per program (host_registry.cpp for the library, the test driver for itself)
__attribute__((visibility("hidden"))) HostRegistry* default_host_registry();

// releases what a call acquired, on every return path
struct HostHold {
    HostRegistry* reg;
    std::vector<uintptr_t> v;
    explicit HostHold(HostRegistry* r = default_host_registry()) : reg(r) {}
    bool add(const void* p, size_t bytes) { const uintptr_t b = reg->acquire(p, bytes); if (b) v.push_back(b); return b != 0; }
    // both planes or neither: a plane that is held is a plane the copy engine will write
    bool add2(const void* p, const void* q, size_t bytes)
    {
        const uintptr_t a = reg->acquire(p, bytes);
        if (!a) return false;
        const uintptr_t b = reg->acquire(q, bytes);
        if (!b) { reg->release(a); return false; }
        v.push_back(a); v.push_back(b);
        return true;
    }
    void release() { for (uintptr_t b : v) reg->release(b); v.clear(); }
    ~HostHold() { release(); }
};

}  // namespace eppm
