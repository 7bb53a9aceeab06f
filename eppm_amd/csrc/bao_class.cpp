// bao_class.cpp -- class bao_flow_patchmatch_multiscale_cuda on top of the C ABI
// (reference: bao_flow_patchmatch_multiscale_cuda.cpp:66-168, :217-315).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/bao_flow_patchmatch_multiscale_cuda.h"
#include "../../include/eppm.h"

// Caller blocks the object has seen (bao_alloc lays an image / a flow plane out as ONE contiguous block reachable through
// row-pointer tables, bao_basic.h:124-162).  The reference indexes through the tables on every call (bao_rgb2rgba,
// basic/bao_basic_cuda.h:258-267), so every pixel pointer of a table is checked on every call before the block is read as one piece
// (a vectorised pointer walk, ~0.1 ms per 1024x436 image); set_option("trust_verified_tables", 1) re-checks a table that was verified
// once through its row ends only.  With set_option("pin_caller_buffers", 1) a block is also registered for DMA
// (eppm_host_register), so that the copy engine reads / writes the caller's memory and no host copy remains.
namespace {
struct SeenBlock { const void* table; const void* base; size_t bytes; bool pinned; };
struct ClassPriv {
    SeenBlock blk[8];
    int n = 0, next = 0, evictions = 0;
    bool pin = false;
    bool verify_always = true;      // default: no table is trusted from an earlier call; set_option("trust_verified_tables", 1) turns the trust cache on
};
void release_block(SeenBlock& b)
{
    if (b.pinned) eppm_host_unregister((void*)b.base);
    b = SeenBlock{NULL, NULL, 0, false};
}
// remembers (table, base); registers the block when pinning is on.  Returns true when the block was verified before.
bool seen(ClassPriv* pr, const void* table, const void* base, size_t bytes, bool add)
{
    for (int i = 0; i < pr->n; i++)
        if (pr->blk[i].table == table && pr->blk[i].base == base && pr->blk[i].bytes == bytes) return true;
    if (!add) return false;
    int slot = pr->n < 8 ? pr->n++ : (pr->next++ & 7);
    if (pr->blk[slot].base) { release_block(pr->blk[slot]); pr->evictions++; }
    pr->blk[slot] = SeenBlock{table, base, bytes, false};
    // a caller that cycles through more blocks than the ring holds would pin and unpin one per call (each costs as much as several
    // staged copies): after 16 evictions new blocks are no longer pinned and go through the context's staging buffers
    if (pr->pin && pr->evictions < 16 && eppm_host_register((void*)base, bytes) == EPPM_OK) pr->blk[slot].pinned = true;
    return false;
}
}  // namespace

bao_flow_patchmatch_multiscale_cuda::bao_flow_patchmatch_multiscale_cuda()
    : m_h(0), m_w(0), m_device(0), m_ctx(NULL), m_params(NULL), m_stage(NULL), m_u(NULL), m_v(NULL), m_priv(new ClassPriv())
{
    eppm_params* p = (eppm_params*)malloc(sizeof(eppm_params));
    if (p) eppm_default_params(p);
    m_params = p;
}

bao_flow_patchmatch_multiscale_cuda::~bao_flow_patchmatch_multiscale_cuda()
{
    _destroy();
    free(m_params);
    delete (ClassPriv*)m_priv;
}

bool bao_flow_patchmatch_multiscale_cuda::set_option(const char* name, long long value)
{
    eppm_params* p = (eppm_params*)m_params;
    if (!p || !name) return false;
    if (!strcmp(name, "patch_r")) p->patch_r = (int)value;
    else if (!strcmp(name, "num_iter")) p->num_iter = (int)value;
    else if (!strcmp(name, "search_range")) p->search_range = (int)value;
    else if (!strcmp(name, "num_guess")) p->num_guess = (int)value;
    else if (!strcmp(name, "seg_len")) p->seg_len = (int)value;
    else if (!strcmp(name, "wmf_iters")) p->wmf_iters = (int)value;
    else if (!strcmp(name, "seed")) p->seed = (unsigned long long)value;
    else if (!strcmp(name, "propagation")) p->propagation = (int)value;
    else if (!strcmp(name, "levels")) p->levels = (int)value;
    else if (!strcmp(name, "pin_caller_buffers")) ((ClassPriv*)m_priv)->pin = (value != 0);
    else if (!strcmp(name, "verify_tables_every_call")) ((ClassPriv*)m_priv)->verify_always = (value != 0);
    else if (!strcmp(name, "trust_verified_tables")) ((ClassPriv*)m_priv)->verify_always = (value == 0);
    else return false;
    return true;
}

void bao_flow_patchmatch_multiscale_cuda::_destroy()
{
    if (m_ctx) eppm_destroy(m_ctx);
    m_ctx = NULL;
    ClassPriv* pr = (ClassPriv*)m_priv;
    for (int i = 0; i < pr->n; i++) release_block(pr->blk[i]);
    pr->n = pr->next = pr->evictions = 0;
    free(m_stage); free(m_u); free(m_v);
    m_stage = NULL; m_u = NULL; m_v = NULL;
}

// driver .cpp:106-110
void bao_flow_patchmatch_multiscale_cuda::init(unsigned char*** img1, unsigned char*** img2, int h, int w)
{
    init(h, w);
    set_data(img1, img2);
}

// driver .cpp:112-157
void bao_flow_patchmatch_multiscale_cuda::init(int h, int w)
{
    _destroy();
    m_h = h; m_w = w;
    if (eppm_create(&m_ctx, h, w, m_device, (const eppm_params*)m_params) != EPPM_OK) {
        fprintf(stderr, "bao_flow_patchmatch_multiscale_cuda::init: %s\n", eppm_last_error());
        m_ctx = NULL;
        return;
    }
}

// every pixel pointer of the table against the one contiguous block: h*w pointer reads, branch-free per row so that the compiler
// vectorises the comparison (0.15-0.25 ms per 1024x436 image)
static bool table_is_block(unsigned char*** img, const unsigned char* base, int h, int w)
{
    const size_t row = (size_t)w * 3;
    for (int i = 0; i < h; i++) {
        unsigned char* const* r = img[i];
        const uintptr_t b = (uintptr_t)(base + (size_t)i * row);
        uintptr_t diff = 0;
        for (int j = 0; j < w; j++) diff |= (uintptr_t)r[j] ^ (b + (uintptr_t)3 * (uintptr_t)j);
        if (diff) return false;
    }
    return true;
}
static void forget(ClassPriv* pr, const void* table)
{
    for (int i = 0; i < pr->n; i++)
        if (pr->blk[i].table == table) release_block(pr->blk[i]);
}

// The image as one contiguous h*w*3 block, or NULL when the row-pointer tables describe any other layout.  img[i][j] points at
// pixel (i,j)'s three bytes (bao_alloc<unsigned char>(h,w,3), bao_basic.h:146-162).  Every pointer is checked on EVERY call (the
// reference indexes through the table on every call).  For a table this object has verified before, whose row ends still
// describe the same block, the full walk is DEFERRED (*deferred = true): set_data enqueues the transfer from the block first and
// walks the table while the copy engine and the first kernels run, and redoes the call through the pointers if the walk fails --
// the same result as checking first, without the walk's time in front of the GPU work.  "trust_verified_tables" skips that walk.
static const unsigned char* contiguous_rgb(void* priv, unsigned char*** img, int h, int w, bool* deferred)
{
    ClassPriv* pr = (ClassPriv*)priv;
    const unsigned char* base = img[0][0];
    const size_t row = (size_t)w * 3;
    for (int i = 0; i < h; i++)
        if (img[i][0] != base + (size_t)i * row || img[i][w - 1] != base + (size_t)i * row + (size_t)3 * (w - 1)) return NULL;
    if (seen(pr, img, base, row * h, false)) {
        *deferred = pr->verify_always;
        return base;
    }
    if (!table_is_block(img, base, h, w)) return NULL;
    seen(pr, img, base, row * h, true);
    return base;
}

// driver .cpp:159-168 (bao_rgb2rgba indexes through the row tables, bao_basic_cuda.h:258-267)
bool bao_flow_patchmatch_multiscale_cuda::set_data(unsigned char*** img1, unsigned char*** img2)
{
    if (!m_ctx || !img1 || !img2) return false;
    bool defer_a = false, defer_b = false;
    const unsigned char* a = contiguous_rgb(m_priv, img1, m_h, m_w, &defer_a);
    const unsigned char* b = contiguous_rgb(m_priv, img2, m_h, m_w, &defer_b);
    auto gather = [&](unsigned char* dst, unsigned char*** img) {      // any other layout goes through the pointers, as bao_rgb2rgba does
        for (int i = 0; i < m_h; i++)
            for (int j = 0; j < m_w; j++)
                for (int c = 0; c < 3; c++) dst[((size_t)i * m_w + j) * 3 + c] = img[i][j][c];
    };
    auto staged = [&](bool first) -> unsigned char* {
        if (!m_stage) m_stage = (unsigned char*)malloc((size_t)m_h * m_w * 3 * 2);
        return m_stage ? m_stage + (first ? 0 : (size_t)m_h * m_w * 3) : NULL;
    };
    if (!a) { unsigned char* d = staged(true); if (!d) return false; gather(d, img1); a = d; }
    if (!b) { unsigned char* d = staged(false); if (!d) return false; gather(d, img2); b = d; }
    if (eppm_set_images(m_ctx, a, b, (size_t)m_w * 3) != EPPM_OK) {
        fprintf(stderr, "bao_flow_patchmatch_multiscale_cuda::set_data: %s\n", eppm_last_error());
        return false;
    }
    // the deferred walks, while the device works on what was just enqueued
    const bool bad_a = defer_a && !table_is_block(img1, a, m_h, m_w), bad_b = defer_b && !table_is_block(img2, b, m_h, m_w);
    if (bad_a || bad_b) {
        // the caller rewrote pixel pointers of a table it had passed before: the images are what the pointers say, not the block
        ClassPriv* pr = (ClassPriv*)m_priv;
        if (bad_a) { forget(pr, img1); unsigned char* d = staged(true); if (!d) return false; gather(d, img1); a = d; }
        if (bad_b) { forget(pr, img2); unsigned char* d = staged(false); if (!d) return false; gather(d, img2); b = d; }
        if (eppm_set_images(m_ctx, a, b, (size_t)m_w * 3) != EPPM_OK) {
            fprintf(stderr, "bao_flow_patchmatch_multiscale_cuda::set_data: %s\n", eppm_last_error());
            return false;
        }
    }
    return true;
}

// disp[i] is a row of w floats; bao_alloc<float>(h,w) makes the rows one contiguous block (bao_basic.h:124-133)
static float* contiguous_plane(void* priv, float** disp, int h, int w)
{
    for (int i = 1; i < h; i++)
        if (disp[i] != disp[0] + (size_t)i * w) return NULL;
    seen((ClassPriv*)priv, disp, disp[0], sizeof(float) * h * w, true);
    return disp[0];
}

// driver .cpp:217-315
void bao_flow_patchmatch_multiscale_cuda::compute_flow(float** disp1_x, float** disp1_y, unsigned char*** color_flow)
{
    if (!m_ctx || !disp1_x || !disp1_y) return;
    float* u = contiguous_plane(m_priv, disp1_x, m_h, m_w);
    float* v = contiguous_plane(m_priv, disp1_y, m_h, m_w);
    const bool direct = (u && v);
    if (!direct) {
        if (!m_u) m_u = (float*)malloc(sizeof(float) * m_h * m_w);
        if (!m_v) m_v = (float*)malloc(sizeof(float) * m_h * m_w);
        if (!m_u || !m_v) return;
        u = m_u; v = m_v;
    }
    if (eppm_compute(m_ctx, u, v) != EPPM_OK) {
        fprintf(stderr, "bao_flow_patchmatch_multiscale_cuda::compute_flow: %s\n", eppm_last_error());
        return;
    }
    if (!direct)
        for (int i = 0; i < m_h; i++) {
            memcpy(disp1_x[i], m_u + (size_t)i * m_w, sizeof(float) * m_w);
            memcpy(disp1_y[i], m_v + (size_t)i * m_w, sizeof(float) * m_w);
        }
    if (color_flow != NULL) {
        // bao_cuda_convert_flow_to_colorshow(d_colorflow, flow, h, w, 20, 20) on the device flow, D2H, bao_rgba2rgb: driver .cpp:308-314
        if (!m_stage) m_stage = (unsigned char*)malloc((size_t)m_h * m_w * 3 * 2);
        if (!m_stage) return;
        unsigned char* rgb = m_stage;          // h*w*3 of the RGB staging buffer
        if (eppm_compute_color(m_ctx, rgb, (size_t)m_w * 3, 20, 20) != EPPM_OK) {
            fprintf(stderr, "bao_flow_patchmatch_multiscale_cuda::compute_flow (color): %s\n", eppm_last_error());
            return;
        }
        for (int i = 0; i < m_h; i++)
            for (int j = 0; j < m_w; j++)
                for (int c = 0; c < 3; c++) color_flow[i][j][c] = rgb[((size_t)i * m_w + j) * 3 + c];
    }
}
