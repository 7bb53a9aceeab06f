// rng_tables.cpp -- the PatchMatch generator objects (eppm_pm_rng) and their shared read-only tables: every block's XORWOW start
// states, the GF(2) skip matrix and the numbers of a run's search launches drawn ahead (DESIGN.md sections 3.4, 4).
#include "api_internal.h"

using namespace eppm;

// The read-only tables of a generator -- every block's start states for the init draw and the first search, and the GF(2) skip matrix --
// depend only on (device, w, h, num_guess, seed): building them walks every block's stream on the host (0.4 M draws at 1024x436) and
// raises a 160x160 bit matrix to a power, 1.5 ms of the 3.3 ms a context takes to create.  Contexts of one geometry share one copy;
// an entry nobody uses stays cached (a fresh object per pair is the window the reference's own demo times) until eight such pile up.
namespace eppm {
struct RngTables {
    int device, w, h, G;
    unsigned long long seed;
    uint32_t *init_tab = nullptr, *iter_tab = nullptr, *skip_mat = nullptr;
    uint32_t skip_weyl = 0;
    int per_lane = 0, refs = 0;
    unsigned long long last_use = 0;
    // the numbers of the first rand_iters search launches of a run, drawn ahead (PmRngDev::rand_tab): [launch][block][G][512] int16
    // (these three under build_mu, not under the global lock: building a table allocates, launches and synchronises)
    std::mutex build_mu;
    int16_t* rand_tab = nullptr;
    int rand_iters = 0;
    std::vector<void*> retired;        // smaller tables older contexts may still read; freed with the entry
};
}  // namespace eppm
namespace {
std::mutex g_rngtab_mu;
std::vector<RngTables*> g_rngtab;
unsigned long long g_rngtab_clock = 0;
void rngtab_free(RngTables* t)
{
    (void)hipFree(t->init_tab); (void)hipFree(t->iter_tab); (void)hipFree(t->skip_mat); (void)hipFree(t->rand_tab);
    for (void* p : t->retired) (void)hipFree(p);
    delete t;
}
}  // namespace

static int rngtab_acquire(RngTables** out, int device, int w, int h, const eppm_params& p)
{
    std::lock_guard<std::mutex> lk(g_rngtab_mu);
    for (RngTables* t : g_rngtab)
        if (t->device == device && t->w == w && t->h == h && t->G == p.num_guess && t->seed == p.seed) {
            t->refs++; t->last_use = ++g_rngtab_clock;
            *out = t;
            return EPPM_OK;
        }
    RngTables* t = new RngTables();
    t->device = device; t->w = w; t->h = h; t->G = p.num_guess; t->seed = p.seed;
    const int gx = (w + kBlock - 1) / kBlock, gy = (h + kBlock - 1) / kBlock, nb = gx * gy;
    t->per_lane = 512 * t->G / 64;
    const size_t words = (size_t)nb * 64 * 6;
    std::vector<uint32_t> it(words), st(words);
    // walk every block's stream once: lane l of the init draw starts at draw 8*l, lane l of a search at
    // 512 + per_lane*l (curand_init(seed, block_id, 0): kernel.cu:68)
    for (int b = 0; b < nb; b++) {
        XorwowState s;
        xorwow_init(&s, p.seed, (unsigned long long)b);
        for (int q = 0; q < 512; q++) {
            if ((q & 7) == 0) memcpy(&it[((size_t)b * 64 + (q >> 3)) * 6], &s, 24);
            xorwow_next(&s);
        }
        for (int q = 0; q < 512 * t->G; q++) {
            if (q % t->per_lane == 0) memcpy(&st[((size_t)b * 64 + q / t->per_lane) * 6], &s, 24);
            xorwow_next(&s);
        }
    }
    const unsigned long long skip = (unsigned long long)(512 * t->G - t->per_lane);
    std::vector<uint32_t> mat(160 * 5);
    xorwow_skip_matrix(skip, mat.data());
    t->skip_weyl = 362437u * (uint32_t)skip;
    hipError_t e = hipMalloc(&t->init_tab, words * 4);
    if (e == hipSuccess) e = hipMalloc(&t->iter_tab, words * 4);
    if (e == hipSuccess) e = hipMalloc(&t->skip_mat, mat.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(t->init_tab, it.data(), words * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->iter_tab, st.data(), words * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->skip_mat, mat.data(), mat.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rngtab_free(t);
        return set_err(EPPM_ERR_HIP, "generator tables: %s", hipGetErrorString(e));
    }
    t->refs = 1; t->last_use = ++g_rngtab_clock;
    g_rngtab.push_back(t);
    *out = t;
    return EPPM_OK;
}
// The random numbers of `iters` search launches drawn ahead, once per (device, geometry, num_guess, seed): the block streams are re-seeded
// on every PatchMatch call (kernel.cu:68, :160), so every run of a geometry draws the same numbers.  Drawn on the device by the code the
// search itself uses (k_pm_rand_table = its drawing wave), launch after launch, from the first search's lane states.  The search then
// needs no drawing wave (G instead of G + 1 waves per workgroup), no generator state and none of the GF(2) jumps: 6.6 % of its
// instructions.  6.9 MB at 1024x436 (112 blocks x 10 launches x 6 guesses x 512 shorts), 125 MB at 3840x2160; above 512 MB: not built.
#ifndef EPPM_RAND_TABLE
#define EPPM_RAND_TABLE 1
#endif
static const int16_t* rngtab_rand_table(RngTables* t, int iters, size_t* stride)
{
    const int gx = (t->w + kBlock - 1) / kBlock, gy = (t->h + kBlock - 1) / kBlock, nb = gx * gy;
    *stride = (size_t)nb * 512 * t->G;
    if (!EPPM_RAND_TABLE || iters < 1 || !opt_rand_table()) return nullptr;
    // the caller holds a reference on the entry (rngtab_acquire): it cannot go away.  Only contexts of this very (device, geometry,
    // num_guess, seed) wait for each other here; creating and destroying contexts of any other kind goes on meanwhile.
    std::lock_guard<std::mutex> lk(t->build_mu);
    if (t->rand_iters >= iters) return t->rand_tab;
    const size_t bytes = *stride * 2 * (size_t)iters;
    if (bytes > ((size_t)512 << 20)) return nullptr;
    int16_t* tab = nullptr;
    uint32_t* work = nullptr;
    const size_t state_bytes = (size_t)nb * 64 * 6 * 4;
    if (hipMalloc(&tab, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMalloc(&work, state_bytes) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(tab); return nullptr; }
    PmRngDev d;
    d.init_tab = t->init_tab; d.iter_tab = t->iter_tab; d.skip_mat = t->skip_mat; d.skip_weyl = t->skip_weyl; d.per_lane = t->per_lane; d.gx = gx; d.gy = gy;
    hipError_t e = hipMemcpy(work, t->iter_tab, state_bytes, hipMemcpyDeviceToDevice);
    for (int it = 0; it < iters && e == hipSuccess; it++) launch_pm_rand_table(d, work, tab + (size_t)it * *stride, t->G, nullptr);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    (void)hipFree(work);
    if (e != hipSuccess) { (void)hipGetLastError(); (void)hipFree(tab); return nullptr; }
    if (t->rand_tab) t->retired.push_back(t->rand_tab);
    t->rand_tab = tab;
    t->rand_iters = iters;
    return tab;
}

static void rngtab_release(RngTables* t)
{
    if (!t) return;
    std::lock_guard<std::mutex> lk(g_rngtab_mu);
    t->refs--;
    // keep at most eight unused entries: drop the least recently used ones beyond that
    for (;;) {
        int idle = 0, oldest = -1;
        for (int i = 0; i < (int)g_rngtab.size(); i++)
            if (g_rngtab[i]->refs == 0) { idle++; if (oldest < 0 || g_rngtab[i]->last_use < g_rngtab[oldest]->last_use) oldest = i; }
        if (idle <= 8) break;
        rngtab_free(g_rngtab[oldest]);
        g_rngtab.erase(g_rngtab.begin() + oldest);
    }
}

int rng_create(eppm_pm_rng** out, int w, int h, const eppm_params& p, bool alloc_work)
{
    eppm_pm_rng* r = new eppm_pm_rng();
    HIPCHK(hipGetDevice(&r->device));
    r->w = w; r->h = h; r->G = p.num_guess; r->seed = p.seed;
    r->gx = (w + kBlock - 1) / kBlock; r->gy = (h + kBlock - 1) / kBlock;
    const int tr = rngtab_acquire(&r->tables, r->device, w, h, p);
    if (tr != EPPM_OK) { delete r; return tr; }
    r->per_lane = r->tables->per_lane;
    r->init_tab = r->tables->init_tab; r->iter_tab = r->tables->iter_tab; r->skip_mat = r->tables->skip_mat;
    r->skip_weyl = r->tables->skip_weyl;
    const size_t words = (size_t)r->gx * r->gy * 64 * 6;
    r->own_work = alloc_work;
    if (!alloc_work) r->rand_tab = rngtab_rand_table(r->tables, p.num_iter, &r->rand_stride);      // contexts; the stand-alone generator objects stream
    if (alloc_work) {       // (a context's states live in its slab and are set by k_pm_init_field at the start of every PatchMatch run)
        for (int k = 0; k < 2; k++)
            for (int q = 0; q < 2; q++) HIPCHK(hipMalloc(&r->work[k][q], words * 4));
        HIPCHK(hipMemcpy(r->work[0][0], r->iter_tab, words * 4, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(r->work[1][0], r->iter_tab, words * 4, hipMemcpyDeviceToDevice));
    }
    *out = r;
    return EPPM_OK;
}

void rng_free(eppm_pm_rng* r)
{
    if (!r) return;
    if (r->own_work)
        for (int k = 0; k < 2; k++)
            for (int q = 0; q < 2; q++) (void)hipFree(r->work[k][q]);
    rngtab_release(r->tables);
    delete r;
}

void rngtab_release_idle()
{
    std::lock_guard<std::mutex> lk(g_rngtab_mu);
    for (size_t i = 0; i < g_rngtab.size();) {
        if (g_rngtab[i]->refs == 0) { rngtab_free(g_rngtab[i]); g_rngtab.erase(g_rngtab.begin() + i); }
        else i++;
    }
}
