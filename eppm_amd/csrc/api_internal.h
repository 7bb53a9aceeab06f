// api_internal.h -- what the translation units behind the C ABI of libeppm_hip.so (include/eppm.h) share: error plumbing, parameter
// checks, host-side look-up tables, the PatchMatch generator object, the block / stream cache and the PatchMatch host driver.  Private to
// eppm_amd/csrc; everything here has hidden visibility (the library exports the C ABI, the drop-in class and the kernels, nothing else).
//   api_common.cpp         errors, version, default / checked parameters, host look-up tables, pyramid dimensions
//   host_registry.h/.cpp   caller memory registered for DMA (header-only logic, tested under ThreadSanitizer on the CPU)
//   rng_tables.cpp         XORWOW generator objects and their shared read-only tables
//   mem_cache.cpp          slabs, pinned staging buffers and streams of destroyed contexts, kept for the next one
//   pm_driver.cpp          baoCudaPatchMatch's host loop (sweep forms by iteration, search), shared by contexts and stage launchers
//   context.cpp            eppm_ctx: create / destroy / set_images / compute / planes / stage times (the class's init, set_data, compute_flow)
//   device_api.cpp         device-memory plumbing of the ABI (malloc / memcpy / NUMA binding)
//   launchers_ref_abi.cpp  the reference's live extern "C" stage launchers and the sub-stage entry points of the parity tests
//   test_hooks.cpp         libeppm_hip_test.so only: include/eppm_test.h
#pragma once

#include <ctype.h>
#include <math.h>
#include <sched.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "eppm_internal.h"
#include "host_registry.h"

#define EPPM_HIDDEN __attribute__((visibility("hidden")))

// ---- errors (api_common.cpp) ----
EPPM_HIDDEN int set_err(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// a failed HIP call leaves a sticky "last error"; it is consumed here so that it cannot surface in a later, unrelated call
#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            (void)hipGetLastError();                                                                         \
            return set_err(EPPM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                                    \
    } while (0)
#define CHK(expr)                     \
    do {                              \
        int r_ = (expr);              \
        if (r_ != EPPM_OK) return r_; \
    } while (0)

// ---- parameters, host look-up tables, pyramid geometry (api_common.cpp) ----
EPPM_HIDDEN int check_params(const eppm_params& p);
EPPM_HIDDEN void host_pm_lut(int R, std::vector<float>& v);       // gs[0..R], cn[0..8] (+ the tolerance library's td[256], ta[256])
EPPM_HIDDEN void host_wmf_lut(std::vector<float>& v);
EPPM_HIDDEN void host_blf_lut(std::vector<float>& v);
EPPM_HIDDEN int pyr_init_dim(int* arrH, int* arrW, int h, int w, int maxDepth, double ratio);
EPPM_HIDDEN int upload_lut(float** dst, const std::vector<float>& v);
// the two-level index of eppm_device.cuh: DeltaTab over the 598 distinct L-inf distances of unorm8 texels: t1[kd] and, per slot of t2, the
// distance it stands for (gaps between the members of a group repeat a neighbour); built once, checked exhaustively
struct DeltaIndex { int32_t t1[256]; std::vector<float> dval; };
EPPM_HIDDEN const DeltaIndex& delta_index();
// `head` followed by a DeltaTab whose values the device computes with the formula they replace (which: see launch_delta_values)
EPPM_HIDDEN int upload_lut_delta(float** dst, const std::vector<float>& head, int which);
EPPM_HIDDEN int upload_pm_lut(float** dst, int R);        // gs[0..R], cn[0..8], then the library's term tables (exact: DeltaTab; tolerance: td, ta)
EPPM_HIDDEN int upload_wmf_lut(float** dst);              // g[0..WMF_RADIUS], DeltaTab of the range weight
EPPM_HIDDEN int upload_blf_lut(float** dst);              // g[0..2*POSTPROC_BLF_SIG_S], DeltaTab of the range weight

// ---- kernel-variant switches of the parity tests ----
// The product library has none: the functions below are constants.  libeppm_hip_test.so (the same objects, with the translation units
// that read a switch compiled once more with -DEPPM_TEST_HOOKS, plus test_hooks.cpp; include/eppm_test.h) exports eppm_test_set_option,
// which sets the DEFAULTS a context copies when it is created (eppm_ctx::opt_*) and what the context-less stage launchers read; a context
// in use is never affected.
#ifdef EPPM_TEST_HOOKS
EPPM_HIDDEN extern std::atomic<int> g_opt_rand_table;       // "rand_table": 0 = contexts created afterwards draw while they search (the form above 512 MB)
EPPM_HIDDEN extern std::atomic<int> g_opt_sweep_spec;       // "sweep_spec": -1 by iteration, 0 never, 1 always, 2 always and without the work list, 3 always in the merged form
EPPM_HIDDEN extern std::atomic<int> g_opt_no_split;         // "c2f_no_split"
static inline int opt_rand_table() { return g_opt_rand_table.load(); }
static inline int opt_sweep_spec() { return g_opt_sweep_spec.load(); }
static inline int opt_no_split() { return g_opt_no_split.load(); }
#else
static constexpr int opt_rand_table() { return 1; }
static constexpr int opt_sweep_spec() { return -1; }
static constexpr int opt_no_split() { return 0; }
#endif

// ---- caller memory registered for DMA: host_registry.h (header-only logic), bound to HIP by host_registry.cpp ----

// ---- PatchMatch generator object (rng_tables.cpp) ----
namespace eppm { struct RngTables; }
struct eppm_pm_rng {
    eppm::RngTables* tables = nullptr; // shared, read-only (rngtab_acquire): init_tab, iter_tab, skip_mat point into it
    const int16_t* rand_tab = nullptr; // contexts: the search launches' numbers drawn ahead (rngtab_rand_table), or NULL
    size_t rand_stride = 0;            // shorts per launch
    int device = 0, w = 0, h = 0, gx = 0, gy = 0, G = 0, per_lane = 0;
    unsigned long long seed = 0;
    uint32_t* init_tab = nullptr;
    uint32_t* iter_tab = nullptr;
    uint32_t* work[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [problem][ping-pong]
    bool own_work = true;              // false: the states live in a context's slab (one set per pair of the batch)
    int cur[2] = {0, 0};
    uint32_t* skip_mat = nullptr;
    uint32_t skip_weyl = 0;
    eppm::PmRngDev dev() const
    {
        eppm::PmRngDev d;
        d.init_tab = init_tab; d.iter_tab = iter_tab; d.skip_mat = skip_mat;
        d.skip_weyl = skip_weyl; d.per_lane = per_lane; d.gx = gx; d.gy = gy;
        return d;
    }
};
EPPM_HIDDEN int rng_create(eppm_pm_rng** out, int w, int h, const eppm_params& p, bool alloc_work = true);
EPPM_HIDDEN void rng_free(eppm_pm_rng* r);
EPPM_HIDDEN void rngtab_release_idle();        // eppm_release_cached_memory: the tables no generator uses any more

// ---- blocks and streams of destroyed contexts (mem_cache.cpp) ----
EPPM_HIDDEN hipError_t cache_alloc(void** p, size_t bytes, bool pinned, int device);
EPPM_HIDDEN void cache_free(void* p, size_t bytes, bool pinned, int device);        // the block must be idle
EPPM_HIDDEN hipError_t pooled_stream_create(hipStream_t* out, int device);
EPPM_HIDDEN void pooled_stream_destroy(hipStream_t s, int device);

// ---- baoCudaPatchMatch's host loop (pm_driver.cpp) ----
EPPM_HIDDEN eppm::PmProblem mk_problem(const eppm::PlanesH& P, float* cost, int16_t* nnf, int16_t* nnf_alt, eppm_pm_rng* rng, int k, float* spec = nullptr,
                                       int32_t* scand = nullptr, uint32_t* wl = nullptr, int16_t* seed = nullptr);
EPPM_HIDDEN void search(eppm::PmBatch& b, eppm_pm_rng* rng, const float* lut, const eppm_params& prm, hipStream_t s, int launch_no = -1);
EPPM_HIDDEN bool sweep_list_on(int mode);
EPPM_HIDDEN void sweep(eppm::PmBatch& b, const float* lut, const eppm_params& prm, int dir, hipStream_t s, bool speculative = false);
EPPM_HIDDEN void jump(eppm::PmBatch& b, const float* lut, const eppm_params& prm, hipStream_t s);
EPPM_HIDDEN void neighbor(eppm::PmBatch& b, const float* lut, const eppm_params& prm, int launches, hipStream_t s);
EPPM_HIDDEN void run_patchmatch(eppm::PmBatch& b, eppm_pm_rng* rng, const float* lut, const eppm_params& prm, hipStream_t s, int spec_mode);
#ifndef EPPM_SWEEP_CACHE
#define EPPM_SWEEP_CACHE 1
#endif

// ---- state of the context-less launchers that test_hooks.cpp and the colour entry points share (launchers_ref_abi.cpp) ----
EPPM_HIDDEN int launcher_finish();
