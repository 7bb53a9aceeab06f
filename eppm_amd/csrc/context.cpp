// context.cpp -- eppm_ctx, the object behind class bao_flow_patchmatch_multiscale_cuda: create / destroy (init, _destroy: driver
// :112-157, :170-209), set_images (set_data :159-168 + _prepare_data :212-215), compute (compute_flow :217-306), planes, stage times.
#include "api_internal.h"

using namespace eppm;

// ---------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------

struct StageEv { const char* name; hipEvent_t a, b; };

// One context = a batch of `npairs` independent pairs of one size (1 for the plain eppm_create).  Every device plane of
// pair k lives at the same offset inside pair k's SLAB and the slabs are `stride` bytes apart in one allocation, so every
// launch covers all active pairs: it gets pair 0's pointers and {n_active, stride} (eppm_internal.h: Batch), and
// blockIdx.z / .y selects the pair.  The pointer members below are pair 0's; ping-pong swaps apply to every pair alike.
struct eppm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int opt_sweep_spec = -1, opt_no_split = 0;     // kernel-variant switches, copied from the process defaults at creation (test support)
    eppm_params prm;
    int h = 0, w = 0, nl = 0;
    int npairs = 1, n_active = 1;
    char* slab = nullptr;
    size_t stride = 0;
    int H[kMaxLevels], W[kMaxLevels];
    size_t ipitch[kMaxLevels], cpitch[kMaxLevels];   // bytes
    uint32_t *raw1 = nullptr, *raw2 = nullptr;
    size_t raw_pitch = 0;
    uint32_t *img1[kMaxLevels] = {}, *img2[kMaxLevels] = {}, *tmpu[kMaxLevels] = {};
    uint8_t *cen1[kMaxLevels] = {}, *cen2[kMaxLevels] = {};
    void *pk1[kMaxLevels] = {}, *pk2[kMaxLevels] = {};       // float4 texel planes {r,g,b,census}, linear (pitch = w)
    uint32_t *pc1[kMaxLevels] = {}, *pc2[kMaxLevels] = {};   // the same texels in 4 bytes, at the levels the LDS-window refine runs on
    uint32_t *pp1 = nullptr, *pp2 = nullptr;                 // tolerance library: column-parity planes of pc at the PatchMatch level (PlanesH::pp1)
    int pp_pitch = 0, pp_pad = 0;
    int16_t *nnf1 = nullptr, *nnf2 = nullptr, *nnf_tmp = nullptr, *nnf_tmp2 = nullptr;
    float *cost1 = nullptr, *cost2 = nullptr;
    float *spec1 = nullptr, *spec2 = nullptr;   // evaluation cache of the sweeps (PmProblem::spec / scand): four direction planes each
    int32_t *scand1 = nullptr, *scand2 = nullptr;
    uint32_t *wl1 = nullptr, *wl2 = nullptr;    // work lists of the speculative sweeps (PmProblem::wl)
    int16_t *seed1 = nullptr, *seed2 = nullptr; // merged form: the field before each direction's sweep (PmProblem::seed), four planes each
    uint32_t* wmf_ws = nullptr;        // work lists + counters of the weighted median
    bool flow_pending = false;         // eppm_compute_begin issued, eppm_compute_end not yet
    float *flow[kMaxLevels] = {}, *flow_tmp[kMaxLevels] = {};
    float* c2f_cost9[kMaxLevels] = {};  // 9 candidates x 4 passes costs per pixel, only for levels whose refine launch is split
    float *lut_pm = nullptr, *lut_wmf = nullptr, *lut_blf = nullptr;
    eppm_pm_rng* rng = nullptr;
    float* d_uv = nullptr;              // planar u | v of the final flow (host-pointer boundary), in the slab
    uint32_t* d_color = nullptr;        // colour-coded flow (optional output), in the slab
    uint32_t* h_color = nullptr;        // pinned, allocated on first use
    uint8_t* d_rgb = nullptr;           // staging for host RGB input (both frames), in the slab
    // pinned staging for images / flows in memory the caller did NOT register (eppm_host_register), allocated on the first such
    // call; the image staging is double-buffered (an event per buffer marks its H2D done), so staging pair i+1 never waits for
    // the stream to drain
    size_t slab_bytes = 0, h_rgb_bytes = 0, h_flow_bytes = 0;      // sizes of the cacheable blocks (cache_alloc / cache_free)
    uint8_t* h_rgb[2] = {nullptr, nullptr};   // each npairs x both frames
    hipEvent_t ev_rgb[2] = {nullptr, nullptr};
    hipEvent_t ev_h2d = nullptr;        // marks the DMA reads of registered caller images
    int rgb_cur = 0;
    float* h_flow = nullptr;            // npairs x (u plane | v plane)
    std::vector<float*> out_u, out_v;   // per active pair: where eppm_compute_begin_into sent the planes directly (NULL: staging)
    HostHold out_hold;                  // the registered blocks those planes lie in, in use until eppm_compute_end
    bool have_images = false, have_flow = false;
    int timing = 0;                     // 0 off, 1 every stage, 2 only the dominant kernel (the candidate refine)
    std::vector<StageEv> ev;
    std::vector<StageEv> ev_prep;
    std::vector<hipEvent_t> ev_pool;    // events are created once and reused: no hipEventCreate in a steady-state step
    Batch bt() const { return Batch{n_active, stride}; }
    template <class T> T* of_pair(T* p, int k) const { return (T*)((char*)p + (size_t)k * stride); }
};

static PlanesH planes(const eppm_ctx* c, int l, bool swap)
{
    PlanesH p;
    p.pk1 = swap ? c->pk2[l] : c->pk1[l];
    p.pk2 = swap ? c->pk1[l] : c->pk2[l];
    p.w = c->W[l]; p.h = c->H[l];
    p.pitch = c->W[l];
    p.pc1 = swap ? c->pc2[l] : c->pc1[l];
    p.pc2 = swap ? c->pc1[l] : c->pc2[l];
    if (l == c->nl - 1 && c->pp1) {
        p.pp1 = swap ? c->pp2 : c->pp1;
        p.pp2 = swap ? c->pp1 : c->pp2;
        p.pp_pitch = c->pp_pitch; p.pp_pad = c->pp_pad;
    }
    return p;
}

static hipEvent_t pool_event(eppm_ctx* c)
{
    hipEvent_t e = nullptr;
    if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
}
static bool stage_on(const eppm_ctx* c, bool dominant) { return c->timing == 1 || (c->timing == 2 && dominant); }
static void stage_begin(eppm_ctx* c, std::vector<StageEv>& v, const char* name, bool dominant = false)
{
    if (!stage_on(c, dominant)) return;
    StageEv e;
    e.name = name;
    e.a = pool_event(c);
    e.b = pool_event(c);
    (void)hipEventRecord(e.a, c->stream);
    v.push_back(e);
}
static void stage_end(eppm_ctx* c, std::vector<StageEv>& v, bool dominant = false)
{
    if (!stage_on(c, dominant)) return;
    (void)hipEventRecord(v.back().b, c->stream);
}
static void clear_events(eppm_ctx* c, std::vector<StageEv>& v)
{
    for (auto& e : v) { c->ev_pool.push_back(e.a); c->ev_pool.push_back(e.b); }
    v.clear();
}

extern "C" int eppm_destroy(eppm_ctx* c)
{
    if (!c) return EPPM_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->out_hold.release();
    clear_events(c, c->ev);
    clear_events(c, c->ev_prep);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    cache_free(c->slab, c->slab_bytes, false, c->device);
    if (c->h_color) (void)hipHostFree(c->h_color);
    for (int q = 0; q < 2; q++) {
        cache_free(c->h_rgb[q], c->h_rgb_bytes, true, c->device);
        if (c->ev_rgb[q]) (void)hipEventDestroy(c->ev_rgb[q]);
    }
    if (c->ev_h2d) (void)hipEventDestroy(c->ev_h2d);
    cache_free(c->h_flow, c->h_flow_bytes, true, c->device);
    rng_free(c->rng);
    if (c->own_stream && c->stream) pooled_stream_destroy(c->stream, c->device);
    delete c;
    return EPPM_OK;
}

// The three look-up tables (a few hundred bytes, functions of the patch radius only) are uploaded once per (device, radius) and shared by
// every context: three hipMalloc + three synchronous copies + three hipFree per context were a millisecond of the create / destroy pair.
static int shared_luts(int device, int R, float** pm, float** wmf, float** blf)
{
    struct Entry { int device, R; float *pm, *wmf, *blf; };
    static std::mutex mu;
    static std::vector<Entry> tab;
    std::lock_guard<std::mutex> lk(mu);
    for (const Entry& e : tab)
        if (e.device == device && e.R == R) { *pm = e.pm; *wmf = e.wmf; *blf = e.blf; return EPPM_OK; }
    Entry e{device, R, nullptr, nullptr, nullptr};
    CHK(upload_pm_lut(&e.pm, R));
    CHK(upload_wmf_lut(&e.wmf));
    CHK(upload_blf_lut(&e.blf));
    tab.push_back(e);
    *pm = e.pm; *wmf = e.wmf; *blf = e.blf;
    return EPPM_OK;
}

// Lays the planes of ONE pair out in a slab (256-byte aligned offsets, pitched rows padded to 256 bytes), allocates
// npairs slabs in one block and points the context's members at pair 0's planes.
static int ctx_alloc(eppm_ctx* c)
{
    const int h = c->h, w = c->w;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    auto pitch_of = [](size_t row_bytes) { return (row_bytes + 255) & ~(size_t)255; };
    struct Fix { void** dst; size_t off; };
    std::vector<Fix> fix;
    auto plane = [&](void** dst, size_t bytes) { fix.push_back(Fix{dst, take(bytes)}); };
    c->raw_pitch = pitch_of((size_t)w * 4);
    plane((void**)&c->raw1, c->raw_pitch * h);
    plane((void**)&c->raw2, c->raw_pitch * h);
    for (int i = 0; i < c->nl; i++) {
        c->ipitch[i] = pitch_of((size_t)c->W[i] * 4);
        c->cpitch[i] = pitch_of((size_t)c->W[i]);
        const size_t n = (size_t)c->W[i] * c->H[i];
        plane((void**)&c->img1[i], c->ipitch[i] * c->H[i]);
        plane((void**)&c->img2[i], c->ipitch[i] * c->H[i]);
        plane((void**)&c->tmpu[i], c->ipitch[i] * c->H[i]);
        plane(&c->pk1[i], n * 16);
        plane(&c->pk2[i], n * 16);
        plane((void**)&c->pc1[i], n * 4);      // (the refine levels' window kernels and the PatchMatch level's random search)
        plane((void**)&c->pc2[i], n * 4);
        plane((void**)&c->cen1[i], c->cpitch[i] * c->H[i]);
        plane((void**)&c->cen2[i], c->cpitch[i] * c->H[i]);
        plane((void**)&c->flow[i], n * 8);
        plane((void**)&c->flow_tmp[i], n * 8);
        if (i < c->nl - 1 && c2f_refine_wants_split(c->W[i], c->H[i], c->prm.patch_r, 1, c->opt_no_split != 0)) plane((void**)&c->c2f_cost9[i], n * 36 * 4);
    }
    const int L = c->nl - 1;
    const size_t n2 = (size_t)c->W[L] * c->H[L];
#ifdef EPPM_TOL
    c->pp_pad = (c->prm.patch_r + 2) & ~1;                  // even and >= R + 1: a target column is in [0, w], a sample within R of it
    c->pp_pitch = parity_pitch(c->W[L], c->pp_pad);
    plane((void**)&c->pp1, (size_t)2 * c->H[L] * c->pp_pitch * 4);
    plane((void**)&c->pp2, (size_t)2 * c->H[L] * c->pp_pitch * 4);
#endif
    plane((void**)&c->nnf1, n2 * 4);
    plane((void**)&c->nnf2, n2 * 4);
    plane((void**)&c->nnf_tmp, n2 * 4);
    plane((void**)&c->nnf_tmp2, n2 * 4);
    plane((void**)&c->cost1, n2 * 4);
    plane((void**)&c->cost2, n2 * 4);
    plane((void**)&c->spec1, n2 * 4 * 4);
    plane((void**)&c->spec2, n2 * 4 * 4);
    plane((void**)&c->scand1, n2 * 4 * 4);
    plane((void**)&c->scand2, n2 * 4 * 4);
    plane((void**)&c->wl1, pm_worklist_words(c->W[L], c->H[L], c->prm.seg_len) * 4);
    plane((void**)&c->wl2, pm_worklist_words(c->W[L], c->H[L], c->prm.seg_len) * 4);
    plane((void**)&c->seed1, n2 * 4 * 4);
    plane((void**)&c->seed2, n2 * 4 * 4);
    plane((void**)&c->wmf_ws, wmf_workspace_words(c->W[L], c->H[L], c->prm.wmf_iters) * 4);
    plane((void**)&c->d_rgb, (size_t)h * w * 3 * 2);
    plane((void**)&c->d_color, (size_t)h * w * 4);
    plane((void**)&c->d_uv, (size_t)h * w * 8);
    CHK(rng_create(&c->rng, c->W[L], c->H[L], c->prm, false));
    const size_t rng_bytes = (size_t)c->rng->gx * c->rng->gy * 64 * 6 * 4;
    for (int k = 0; k < 2; k++)
        for (int q = 0; q < 2; q++) plane((void**)&c->rng->work[k][q], rng_bytes);
    c->stride = (off + 4095) & ~(size_t)4095;
    // every texel plane is addressed with 32-bit byte offsets from ITS OWN base; the slab stride itself is 64-bit
    c->slab_bytes = c->stride * c->npairs;
    {
        const hipError_t e = cache_alloc((void**)&c->slab, c->slab_bytes, false, c->device);
        if (e != hipSuccess) { (void)hipGetLastError(); return set_err(EPPM_ERR_HIP, "hipMalloc of %zu bytes (%d slab(s)) failed: %s", c->slab_bytes, c->npairs, hipGetErrorString(e)); }
    }
    for (const Fix& f : fix) *f.dst = c->slab + f.off;
    CHK(shared_luts(c->device, c->prm.patch_r, &c->lut_pm, &c->lut_wmf, &c->lut_blf));
    c->out_u.assign(c->npairs, nullptr);
    c->out_v.assign(c->npairs, nullptr);
    return EPPM_OK;
}

extern "C" int eppm_create_batch(eppm_ctx** out, int h, int w, int device, const eppm_params* params, int npairs)
{
    if (!out) return set_err(EPPM_ERR_ARG, "eppm_create: NULL out");
    *out = nullptr;
    if (npairs < 1 || npairs > 4096) return set_err(EPPM_ERR_ARG, "eppm_create_batch: npairs %d out of range [1,4096]", npairs);
    if (h < 4 || w < 4 || h > 32767 || w > 32767) return set_err(EPPM_ERR_ARG, "eppm_create: size %dx%d out of range (NNF coordinates are int16)", w, h);
    if ((unsigned long long)h * (unsigned long long)w * 16ULL >= (1ULL << 32))
        return set_err(EPPM_ERR_ARG, "eppm_create: size %dx%d out of range (texel planes are addressed with 32-bit byte offsets)", w, h);
    eppm_params p;
    eppm_default_params(&p);
    if (params) p = *params;
    CHK(check_params(p));
    HIPCHK(hipSetDevice(device));
    eppm_ctx* c = new eppm_ctx();
    c->device = device; c->prm = p; c->h = h; c->w = w; c->npairs = npairs; c->n_active = 1;
    c->opt_sweep_spec = opt_sweep_spec(); c->opt_no_split = opt_no_split();
    c->nl = pyr_init_dim(c->H, c->W, h, w, p.levels, 0.5f);
    const int L = c->nl - 1;
    if (c->H[L] < 1 || c->W[L] < 1 || (c->W[L] + p.seg_len - 1) / p.seg_len > 1024 || (c->H[L] + p.seg_len - 1) / p.seg_len > 1024) {
        delete c;
        return set_err(EPPM_ERR_ARG, "eppm_create: unsupported size %dx%d", w, h);
    }
    const hipError_t e = pooled_stream_create(&c->stream, c->device);
    if (e != hipSuccess) { delete c; return set_err(EPPM_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    c->own_stream = true;
    int r = ctx_alloc(c);
    if (r != EPPM_OK) { eppm_destroy(c); return r; }
    *out = c;
    return EPPM_OK;
}

extern "C" int eppm_create(eppm_ctx** out, int h, int w, int device, const eppm_params* params)
{
    return eppm_create_batch(out, h, w, device, params, 1);
}

extern "C" int eppm_batch_size(const eppm_ctx* c) { return c ? c->npairs : 0; }

extern "C" int eppm_set_stream(eppm_ctx* c, void* s)
{
    if (!c) return set_err(EPPM_ERR_ARG, "NULL ctx");
    if (c->own_stream && c->stream) { (void)hipStreamSynchronize(c->stream); pooled_stream_destroy(c->stream, c->device); }
    c->stream = (hipStream_t)s;
    c->own_stream = false;
    return EPPM_OK;
}

extern "C" int eppm_num_levels(const eppm_ctx* c) { return c ? c->nl : 0; }
extern "C" int eppm_level_dims(const eppm_ctx* c, int level, int* h, int* w)
{
    if (!c || level < 0 || level >= c->nl) return set_err(EPPM_ERR_ARG, "bad level");
    if (h) *h = c->H[level];
    if (w) *w = c->W[level];
    return EPPM_OK;
}
extern "C" int eppm_enable_stage_timing(eppm_ctx* c, int on)
{
    if (!c) return set_err(EPPM_ERR_ARG, "NULL ctx");
    c->timing = (on == 2) ? 2 : (on != 0);
    return EPPM_OK;
}

// ---- prepare: refine :1060-1071 + .cuh:642-664.  The two frames of every active pair share every launch; the raw
// RGBA planes of the active pairs are in the slabs already. ----
static int prepare(eppm_ctx* c)
{
    stage_begin(c, c->ev_prep, "prepare");
    hipStream_t s = c->stream;
    const Batch bt = c->bt();
    uint32_t **p1 = c->img1, **p2 = c->img2, **tmp = c->tmpu;
    const int p0 = (int)(c->ipitch[0] / 4);
    launch_gauss_rgba2(p1[0], c->raw1, p2[0], c->raw2, p0, c->H[0], c->W[0], .5f, 2, s, bt);    // refine :1063-1064
    const float ratio = 0.5f;                                                             // PYR_RATIO
    const float baseSigma = (1 / ratio - 1);
    const int n = (int)(log(0.25) / (double)logf(ratio));   // C++ float overload in the reference: n = 1
    const float nSigma = baseSigma * n;
    for (int i = 1; i < c->nl; i++) {
        // source level j, blur (sigma, radius), resize ratio r: .cuh:647-663
        const int j = (i <= n) ? 0 : i - n;
        const float sigma = (i <= n) ? baseSigma * i : nSigma;
        const float r = (i <= n) ? (float)pow(ratio, i) : (float)pow(ratio, i) * c->W[0] / c->W[j];
        const int radius = (int)(sigma * 3);
        const int pj = (int)(c->ipitch[j] / 4), pi = (int)(c->ipitch[i] / 4);
        if (gauss_decimate2_ok(c->H[i], c->W[i], c->H[j], c->W[j], r, radius)) {
            // exact 2:1 step: blur only the pixels the decimation keeps (a quarter of the level)
            launch_gauss_decimate2(p1[i], p1[j], p2[i], p2[j], 2, pi, c->H[i], c->W[i], pj, c->H[j], c->W[j], sigma, radius, s, bt);
        } else {
            for (int k = 0; k < 2; k++) {
                uint32_t** pyr = k ? p2 : p1;
                launch_gauss_rgba(tmp[j], pyr[j], pj, c->H[j], c->W[j], sigma, radius, s, bt);
                launch_resize_rgba(pyr[i], pi, c->H[i], c->W[i], tmp[j], pj, c->H[j], c->W[j], r, s, bt);
            }
        }
    }
    CensusBatch cb;
    cb.n = 0;
    for (int k = 0; k < 2; k++)
        for (int i = 0; i < c->nl; i++) {
            CensusJob& J = cb.job[cb.n++];
            J.census = k ? c->cen2[i] : c->cen1[i]; J.cpitch = (int)c->cpitch[i];
            J.texels = k ? c->pk2[i] : c->pk1[i];   J.tpitch = c->W[i];
            J.img = k ? c->img2[i] : c->img1[i];    J.ipitch = (int)(c->ipitch[i] / 4);
            J.w = c->W[i]; J.h = c->H[i]; J.first_block = 0;
            J.packed = k ? c->pc2[i] : c->pc1[i];
        }
    launch_census_batch(cb, s, bt);
    if (c->pp1) {
        const int L = c->nl - 1;
        launch_parity_planes(c->pp1, c->pp_pitch, c->pp_pad, c->pc1[L], c->W[L], c->W[L], c->H[L], s, bt);
        launch_parity_planes(c->pp2, c->pp_pitch, c->pp_pad, c->pc2[L], c->W[L], c->W[L], c->H[L], s, bt);
    }
    stage_end(c, c->ev_prep);
    HIPCHK(hipGetLastError());
    c->have_images = true;
    c->have_flow = false;
    return EPPM_OK;
}

// host RGB of pairs 0..n-1 -> H2D -> RGBA planes (bao_rgb2rgba, alpha = 0) -> prepare.  An image inside memory registered with
// eppm_host_register / eppm_host_alloc is read by the copy engine where it lies; any other image goes through the context's pinned
// staging (one host copy), which is double-buffered.
static int set_images_host_impl(eppm_ctx* c, int n, const uint8_t* const* rgb1, const uint8_t* const* rgb2, size_t row_stride, HostHold& hold)
{
    if (row_stride < (size_t)c->w * 3) return set_err(EPPM_ERR_ARG, "eppm_set_images: row_stride %zu < 3*w", row_stride);
    HIPCHK(hipSetDevice(c->device));
    const size_t row = (size_t)c->w * 3, img = row * c->h, span = row_stride * (c->h - 1) + row;
    for (int k = 0; k < n; k++)
        if (!rgb1[k] || !rgb2[k]) return set_err(EPPM_ERR_ARG, "eppm_set_images: NULL image");
    uint8_t* stage = nullptr;
    bool staged = false, direct = false;
    // (`hold`: registered blocks read in place stay in use until their DMA has completed -- the end of the call)
    for (int k = 0; k < n; k++)
        for (int f = 0; f < 2; f++) {
            const uint8_t* src = f ? rgb2[k] : rgb1[k];
            uint8_t* dst = c->of_pair(c->d_rgb, k) + (size_t)f * img;
            if (hold.add(src, span)) {
                if (row_stride == row) HIPCHK(hipMemcpyAsync(dst, src, img, hipMemcpyHostToDevice, c->stream));
                else HIPCHK(hipMemcpy2DAsync(dst, row, src, row_stride, row, c->h, hipMemcpyHostToDevice, c->stream));
                direct = true;
                continue;
            }
            if (!stage) {
                const int q = c->rgb_cur;
                if (!c->h_rgb[q]) {
                    c->h_rgb_bytes = img * 2 * c->npairs;
                    HIPCHK(cache_alloc((void**)&c->h_rgb[q], c->h_rgb_bytes, true, c->device));
                    HIPCHK(hipEventCreateWithFlags(&c->ev_rgb[q], hipEventDisableTiming));
                } else {
                    HIPCHK(hipEventSynchronize(c->ev_rgb[q]));      // the H2D that last read this buffer (two set_images ago)
                }
                stage = c->h_rgb[q];
            }
            uint8_t* h = stage + ((size_t)k * 2 + f) * img;
            if (row_stride == row) memcpy(h, src, img);
            else
                for (int y = 0; y < c->h; y++) memcpy(h + (size_t)y * row, src + (size_t)y * row_stride, row);
            HIPCHK(hipMemcpyAsync(dst, h, img, hipMemcpyHostToDevice, c->stream));
            staged = true;
        }
    if (staged) {
        HIPCHK(hipEventRecord(c->ev_rgb[c->rgb_cur], c->stream));
        c->rgb_cur ^= 1;
    }
    if (direct) {
        if (!c->ev_h2d) HIPCHK(hipEventCreateWithFlags(&c->ev_h2d, hipEventDisableTiming));
        HIPCHK(hipEventRecord(c->ev_h2d, c->stream));
    }
    c->n_active = n;
    const int p0 = (int)(c->raw_pitch / 4);
    launch_rgb_to_rgba(c->raw1, p0, c->d_rgb, c->h, c->w, c->stream, c->bt());
    launch_rgb_to_rgba(c->raw2, p0, c->d_rgb + img, c->h, c->w, c->stream, c->bt());
    const int r = prepare(c);
    // set_data's contract (a synchronous cudaMemcpy in the reference, driver :165-166): when the call returns the caller may reuse
    // its images.  Staged images were copied above; for images read in place, wait for their DMA (the kernels are queued already).
    if (direct) HIPCHK(hipEventSynchronize(c->ev_h2d));
    return r;
}
static int set_images_host(eppm_ctx* c, int n, const uint8_t* const* rgb1, const uint8_t* const* rgb2, size_t row_stride)
{
    HostHold hold;
    const int r = set_images_host_impl(c, n, rgb1, rgb2, row_stride, hold);
    if (r != EPPM_OK && !hold.v.empty()) (void)hipStreamSynchronize(c->stream);     // nothing may still read the blocks when `hold` lets them go
    return r;
}

extern "C" int eppm_set_images(eppm_ctx* c, const uint8_t* rgb1, const uint8_t* rgb2, size_t row_stride)
{
    if (!c || !rgb1 || !rgb2) return set_err(EPPM_ERR_ARG, "eppm_set_images: NULL argument");
    return set_images_host(c, 1, &rgb1, &rgb2, row_stride);
}

extern "C" int eppm_batch_set_images(eppm_ctx* c, int n, const uint8_t* const* rgb1, const uint8_t* const* rgb2, size_t row_stride)
{
    if (!c || !rgb1 || !rgb2) return set_err(EPPM_ERR_ARG, "eppm_batch_set_images: NULL argument");
    if (n < 1 || n > c->npairs) return set_err(EPPM_ERR_ARG, "eppm_batch_set_images: %d pairs, context holds %d", n, c->npairs);
    return set_images_host(c, n, rgb1, rgb2, row_stride);
}

// device-resident RGBA of pairs 0..n-1: copied into the slabs' raw planes in stream order (the caller's planes are not
// read after the copies complete, and never in place), then prepare
static int set_images_device(eppm_ctx* c, int n, const void* const* d1, const void* const* d2, size_t pitch)
{
    if (pitch < (size_t)c->w * 4 || (pitch & 3)) return set_err(EPPM_ERR_ARG, "eppm_set_images_device: bad pitch %zu", pitch);
    HIPCHK(hipSetDevice(c->device));
    for (int k = 0; k < n; k++) {
        if (!d1[k] || !d2[k]) return set_err(EPPM_ERR_ARG, "eppm_set_images_device: NULL image");
        HIPCHK(hipMemcpy2DAsync(c->of_pair(c->raw1, k), c->raw_pitch, d1[k], pitch, (size_t)c->w * 4, c->h, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpy2DAsync(c->of_pair(c->raw2, k), c->raw_pitch, d2[k], pitch, (size_t)c->w * 4, c->h, hipMemcpyDeviceToDevice, c->stream));
    }
    c->n_active = n;
    return prepare(c);
}

extern "C" int eppm_set_images_device(eppm_ctx* c, const void* d1, const void* d2, size_t pitch)
{
    if (!c || !d1 || !d2) return set_err(EPPM_ERR_ARG, "eppm_set_images_device: NULL argument");
    return set_images_device(c, 1, &d1, &d2, pitch);
}

extern "C" int eppm_batch_set_images_device(eppm_ctx* c, int n, const void* const* d_rgba1, const void* const* d_rgba2, size_t pitch)
{
    if (!c || !d_rgba1 || !d_rgba2) return set_err(EPPM_ERR_ARG, "eppm_batch_set_images_device: NULL argument");
    if (n < 1 || n > c->npairs) return set_err(EPPM_ERR_ARG, "eppm_batch_set_images_device: %d pairs, context holds %d", n, c->npairs);
    return set_images_device(c, n, d_rgba1, d_rgba2, pitch);
}

static int compute_all(eppm_ctx* c)
{
    if (!c->have_images) return set_err(EPPM_ERR_STATE, "eppm_compute: no images set");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const Batch bt = c->bt();
    const int L = c->nl - 1;                                            // pm_layer, driver :219
    const int lw = c->W[L], lh = c->H[L];

    stage_begin(c, c->ev, "patchmatch");
    {
        PmBatch b;
        b.n = 2; b.cpitch = lw; b.npitch = lw; b.npairs = bt.n; b.stride = bt.stride;
        b.cache_plane = (size_t)lw * lh;
        b.seed_plane = (size_t)lw * lh * 2;
        b.wl_units = pm_worklist_units(lw, lh, c->prm.seg_len);
        b.p[0] = mk_problem(planes(c, L, false), c->cost1, c->nnf1, c->nnf_tmp, c->rng, 0, c->spec1, EPPM_SWEEP_CACHE ? c->scand1 : nullptr, sweep_list_on(c->opt_sweep_spec) ? c->wl1 : nullptr, c->seed1);     // driver :223
        b.p[1] = mk_problem(planes(c, L, true), c->cost2, c->nnf2, c->nnf_tmp2, c->rng, 1, c->spec2, EPPM_SWEEP_CACHE ? c->scand2 : nullptr, sweep_list_on(c->opt_sweep_spec) ? c->wl2 : nullptr, c->seed2);     // driver :224
        run_patchmatch(b, c->rng, c->lut_pm, c->prm, s, c->opt_sweep_spec);
    }
    stage_end(c, c->ev);

    stage_begin(c, c->ev, "l2_post");
    launch_lr_check(c->nnf1, c->cost1, c->nnf2, lw, lh, lw, lw, s, bt);                                      // driver :233
    launch_lr_check(c->nnf2, c->cost2, c->nnf1, lw, lh, lw, lw, s, bt);
    launch_outlier(c->nnf_tmp, c->cost1, c->nnf1, lw, lh, lw, lw, s, bt);                                    // driver :237
    std::swap(c->nnf1, c->nnf_tmp);
    if (launch_wmf(c->nnf1, c->nnf_tmp, c->img1[L], (int)(c->ipitch[L] / 4), lw, lh, lw, c->lut_wmf, c->prm.wmf_iters, 1,      // driver :239
                   c->wmf_ws, s, bt) != c->nnf1)
        std::swap(c->nnf1, c->nnf_tmp);
    launch_fill_holes(c->nnf_tmp, c->nnf1, c->img1[L], (int)(c->ipitch[L] / 4), lw, lh, lw, s, bt);          // driver :240
    std::swap(c->nnf1, c->nnf_tmp);
    launch_nnf2flow(c->flow[L], lw, c->nnf1, lw, lw, lh, s, bt);                                             // driver :258
    stage_end(c, c->ev);

    static const char* up_names[] = {"upsample_L0", "upsample_L1", "upsample_L2", "upsample_L3", "upsample_L4", "upsample_L5", "upsample_L6"};
    static const char* rf_names[] = {"c2f_refine_L0", "c2f_refine_L1", "c2f_refine_L2", "c2f_refine_L3", "c2f_refine_L4", "c2f_refine_L5", "c2f_refine_L6"};
    static const char* bl_names[] = {"flow_blf_L0", "flow_blf_L1", "flow_blf_L2", "flow_blf_L3", "flow_blf_L4", "flow_blf_L5", "flow_blf_L6"};
    for (int l = L - 1; l >= 0; l--) {                                                                       // driver :275-282
        stage_begin(c, c->ev, up_names[l]);
        launch_resize_flow(c->flow[l], c->H[l], c->W[l], c->flow[l + 1], c->H[l + 1], c->W[l + 1], 2.0f, 2.0f, s, bt);   // refine :1082-1083
        stage_end(c, c->ev);
        stage_begin(c, c->ev, rf_names[l], true);
        launch_c2f_refine(planes(c, l, false), c->flow[l], c->lut_pm, c->prm.patch_r, c->c2f_cost9[l], s, bt, c->opt_no_split != 0);   // refine :1086
        stage_end(c, c->ev, true);
        stage_begin(c, c->ev, bl_names[l]);
        launch_flow_blf(c->flow_tmp[l], c->flow[l], c->img1[l], (int)(c->ipitch[l] / 4), c->W[l], c->H[l], c->W[l], c->lut_blf, s, bt);  // driver :280
        std::swap(c->flow[l], c->flow_tmp[l]);
        stage_end(c, c->ev);
    }
    stage_begin(c, c->ev, "flow_blf_final");
    launch_flow_blf(c->flow_tmp[0], c->flow[0], c->img1[0], (int)(c->ipitch[0] / 4), c->W[0], c->H[0], c->W[0], c->lut_blf, s, bt);      // driver :289
    std::swap(c->flow[0], c->flow_tmp[0]);
    stage_end(c, c->ev);
    HIPCHK(hipGetLastError());
    c->have_flow = true;
    return EPPM_OK;
}

extern "C" int eppm_compute_device(eppm_ctx* c, void* d_flow)
{
    if (!c) return set_err(EPPM_ERR_ARG, "NULL ctx");
    CHK(compute_all(c));
    if (d_flow) HIPCHK(hipMemcpyAsync(d_flow, c->flow[0], (size_t)c->h * c->w * 8, hipMemcpyDeviceToDevice, c->stream));
    return EPPM_OK;
}

extern "C" int eppm_batch_compute_device(eppm_ctx* c, void* const* d_flows)
{
    if (!c) return set_err(EPPM_ERR_ARG, "NULL ctx");
    CHK(compute_all(c));
    if (d_flows)
        for (int k = 0; k < c->n_active; k++)
            if (d_flows[k]) HIPCHK(hipMemcpyAsync(d_flows[k], c->of_pair(c->flow[0], k), (size_t)c->h * c->w * 8, hipMemcpyDeviceToDevice, c->stream));
    return EPPM_OK;
}

// compute_flow split in two so that a host thread can keep several contexts in flight: begin enqueues the whole path, the
// de-interleave (on the device) and the device-to-host copies and returns; end waits.  When begin knows the destination planes
// and they lie in registered memory, the copy engine writes them directly; otherwise the planes land in the context's pinned
// staging and end copies them out.
static int compute_begin_impl(eppm_ctx* c, int n_out, float* const* u, float* const* v)
{
    CHK(compute_all(c));
    const size_t n = (size_t)c->h * c->w;
    launch_split_flow(c->d_uv, c->flow[0], (int)n, c->stream, c->bt());                                                           // driver :302-306, on the device
    for (int k = 0; k < c->n_active; k++) {                                                                                       // driver :299
        float* du = (u && k < n_out) ? u[k] : nullptr;
        float* dv = (v && k < n_out) ? v[k] : nullptr;
        const float* src = c->of_pair(c->d_uv, k);
        if (du && dv && c->out_hold.add2(du, dv, n * 4)) {          // both planes in registered memory, held until eppm_compute_end
            HIPCHK(hipMemcpyAsync(du, src, n * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(dv, src + n, n * 4, hipMemcpyDeviceToHost, c->stream));
            c->out_u[k] = du; c->out_v[k] = dv;
            continue;
        }
        if (!c->h_flow) {
            c->h_flow_bytes = n * 8 * c->npairs;
            HIPCHK(cache_alloc((void**)&c->h_flow, c->h_flow_bytes, true, c->device));
        }
        HIPCHK(hipMemcpyAsync(c->h_flow + (size_t)k * n * 2, src, n * 8, hipMemcpyDeviceToHost, c->stream));
        c->out_u[k] = c->out_v[k] = nullptr;
    }
    c->flow_pending = true;
    return EPPM_OK;
}
static int compute_begin(eppm_ctx* c, int n_out, float* const* u, float* const* v)
{
    const int r = compute_begin_impl(c, n_out, u, v);
    if (r != EPPM_OK && !c->out_hold.v.empty()) {
        // eppm_compute_end will refuse to run (nothing is pending): the planes held so far must not stay in use until the context dies.
        // Copies already queued into them drain first.
        (void)hipStreamSynchronize(c->stream);
        c->out_hold.release();
    }
    return r;
}

extern "C" int eppm_compute_begin(eppm_ctx* c)
{
    if (!c) return set_err(EPPM_ERR_ARG, "eppm_compute_begin: NULL ctx");
    return compute_begin(c, 0, nullptr, nullptr);
}

extern "C" int eppm_compute_begin_into(eppm_ctx* c, float* u, float* v)
{
    if (!c || !u || !v) return set_err(EPPM_ERR_ARG, "eppm_compute_begin_into: NULL argument");
    return compute_begin(c, 1, &u, &v);
}

extern "C" int eppm_batch_compute_begin_into(eppm_ctx* c, float* const* u, float* const* v)
{
    if (!c || !u || !v) return set_err(EPPM_ERR_ARG, "eppm_batch_compute_begin_into: NULL argument");
    return compute_begin(c, c->n_active, u, v);
}

static int compute_end(eppm_ctx* c, int n_out, float* const* u, float* const* v)
{
    if (!c->flow_pending) return set_err(EPPM_ERR_STATE, "eppm_compute_end without eppm_compute_begin");
    HIPCHK(hipSetDevice(c->device));
    const hipError_t es = hipStreamSynchronize(c->stream);
    c->out_hold.release();              // the copy engine has left the caller's planes (or the stream is broken)
    HIPCHK(es);
    c->flow_pending = false;
    const size_t n = (size_t)c->h * c->w;
    for (int k = 0; k < n_out && k < c->n_active; k++) {
        if (!u[k] || !v[k]) continue;
        // the planes are in the caller's memory already (begin_into, registered), or in the staging buffer: u plane, then v plane
        const float* fu = c->out_u[k] ? c->out_u[k] : c->h_flow + (size_t)k * n * 2;
        const float* fv = c->out_v[k] ? c->out_v[k] : c->h_flow + (size_t)k * n * 2 + n;
        if (u[k] != fu) memcpy(u[k], fu, n * sizeof(float));
        if (v[k] != fv) memcpy(v[k], fv, n * sizeof(float));
    }
    return EPPM_OK;
}

extern "C" int eppm_compute_end(eppm_ctx* c, float* u, float* v)
{
    if (!c || !u || !v) return set_err(EPPM_ERR_ARG, "eppm_compute_end: NULL argument");
    return compute_end(c, 1, &u, &v);
}

extern "C" int eppm_batch_compute_end(eppm_ctx* c, float* const* u, float* const* v)
{
    if (!c || !u || !v) return set_err(EPPM_ERR_ARG, "eppm_batch_compute_end: NULL argument");
    return compute_end(c, c->n_active, u, v);
}

extern "C" int eppm_compute(eppm_ctx* c, float* u, float* v)
{
    if (!c || !u || !v) return set_err(EPPM_ERR_ARG, "eppm_compute: NULL argument");
    CHK(compute_begin(c, 1, &u, &v));
    return compute_end(c, 1, &u, &v);
}

extern "C" int eppm_batch_compute(eppm_ctx* c, float* const* u, float* const* v)
{
    if (!c || !u || !v) return set_err(EPPM_ERR_ARG, "eppm_batch_compute: NULL argument");
    CHK(compute_begin(c, c->n_active, u, v));
    return compute_end(c, c->n_active, u, v);
}

extern "C" int eppm_synchronize(eppm_ctx* c)
{
    if (!c) return set_err(EPPM_ERR_ARG, "NULL ctx");
    HIPCHK(hipStreamSynchronize(c->stream));
    return EPPM_OK;
}

extern "C" int eppm_stage_times(eppm_ctx* c, const char** names, float* ms, int max)
{
    if (!c) return 0;
    (void)hipStreamSynchronize(c->stream);
    int n = 0;
    for (auto* v : {&c->ev_prep, &c->ev})
        for (auto& e : *v) {
            if (n >= max) return n;
            float t = 0;
            if (hipEventElapsedTime(&t, e.a, e.b) != hipSuccess) t = -1;
            names[n] = e.name; ms[n] = t; n++;
        }
    return n;
}

extern "C" int eppm_clear_stage_times(eppm_ctx* c)
{
    if (!c) return set_err(EPPM_ERR_ARG, "NULL ctx");
    (void)hipStreamSynchronize(c->stream);
    clear_events(c, c->ev);
    clear_events(c, c->ev_prep);
    return EPPM_OK;
}

extern "C" int eppm_batch_get_plane(eppm_ctx* c, int pair, const char* name, int level, void* dst, size_t dst_bytes)
{
    if (!c || !name || !dst) return set_err(EPPM_ERR_ARG, "eppm_get_plane: NULL argument");
    if (level < 0 || level >= c->nl) return set_err(EPPM_ERR_ARG, "eppm_get_plane: bad level %d", level);
    if (pair < 0 || pair >= c->npairs) return set_err(EPPM_ERR_ARG, "eppm_get_plane: bad pair %d", pair);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    const int w = c->W[level], h = c->H[level], L = c->nl - 1;
    const void* src = nullptr;
    size_t esz = 0, pitch = 0;
    std::string n(name);
    if (n == "img1" || n == "img2") { src = (n == "img1") ? c->img1[level] : c->img2[level]; esz = 4; pitch = c->ipitch[level]; }
    else if (n == "census1" || n == "census2") { src = (n == "census1") ? c->cen1[level] : c->cen2[level]; esz = 1; pitch = c->cpitch[level]; }
    else if (n == "flow") { src = c->flow[level]; esz = 8; pitch = (size_t)w * 8; }
    else if (level == L && (n == "nnf1" || n == "nnf2")) { src = (n == "nnf1") ? c->nnf1 : c->nnf2; esz = 4; pitch = (size_t)w * 4; }
    else if (level == L && (n == "cost1" || n == "cost2")) { src = (n == "cost1") ? c->cost1 : c->cost2; esz = 4; pitch = (size_t)w * 4; }
    else return set_err(EPPM_ERR_ARG, "eppm_get_plane: unknown plane '%s' at level %d", name, level);
    if (dst_bytes < (size_t)w * h * esz) return set_err(EPPM_ERR_ARG, "eppm_get_plane: dst too small");
    HIPCHK(hipMemcpy2D(dst, (size_t)w * esz, c->of_pair((const char*)src, pair), pitch, (size_t)w * esz, h, hipMemcpyDeviceToHost));
    return EPPM_OK;
}

extern "C" int eppm_get_plane(eppm_ctx* c, const char* name, int level, void* dst, size_t dst_bytes)
{
    return eppm_batch_get_plane(c, 0, name, level, dst, dst_bytes);
}

extern "C" int eppm_compute_color(eppm_ctx* c, uint8_t* rgb, size_t row_stride, float max_disp_x, float max_disp_y)
{
    if (!c || !rgb) return set_err(EPPM_ERR_ARG, "eppm_compute_color: NULL argument");
    if (!c->have_flow) return set_err(EPPM_ERR_STATE, "eppm_compute_color: no flow computed yet");
    if (row_stride < (size_t)c->w * 3) return set_err(EPPM_ERR_ARG, "eppm_compute_color: row_stride %zu < 3*w", row_stride);
    HIPCHK(hipSetDevice(c->device));
    const size_t n = (size_t)c->h * c->w;
    if (!c->h_color) HIPCHK(hipHostMalloc((void**)&c->h_color, n * 4, hipHostMallocDefault));
    launch_flow_to_color(c->d_color, c->flow[0], c->h, c->w, max_disp_x, max_disp_y, c->stream);       // driver :311
    HIPCHK(hipMemcpyAsync(c->h_color, c->d_color, n * 4, hipMemcpyDeviceToHost, c->stream));           // driver :312
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int y = 0; y < c->h; y++)                                                                         // bao_rgba2rgb, driver :313
        for (int x = 0; x < c->w; x++) {
            const uint32_t p = c->h_color[(size_t)y * c->w + x];
            uint8_t* o = rgb + (size_t)y * row_stride + (size_t)x * 3;
            o[0] = (uint8_t)(p & 0xff); o[1] = (uint8_t)((p >> 8) & 0xff); o[2] = (uint8_t)((p >> 16) & 0xff);
        }
    return EPPM_OK;
}
