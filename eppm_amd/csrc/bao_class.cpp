// bao_class.cpp -- class bao_flow_patchmatch_multiscale_cuda on top of the C ABI
// (reference: bao_flow_patchmatch_multiscale_cuda.cpp:66-168, :217-315).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/bao_flow_patchmatch_multiscale_cuda.h"
#include "../../include/eppm.h"

bao_flow_patchmatch_multiscale_cuda::bao_flow_patchmatch_multiscale_cuda()
    : m_h(0), m_w(0), m_device(0), m_ctx(NULL), m_params(NULL), m_stage(NULL), m_u(NULL), m_v(NULL)
{
    eppm_params* p = (eppm_params*)malloc(sizeof(eppm_params));
    if (p) eppm_default_params(p);
    m_params = p;
}

bao_flow_patchmatch_multiscale_cuda::~bao_flow_patchmatch_multiscale_cuda()
{
    _destroy();
    free(m_params);
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
    else return false;
    return true;
}

void bao_flow_patchmatch_multiscale_cuda::_destroy()
{
    if (m_ctx) eppm_destroy(m_ctx);
    m_ctx = NULL;
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
    m_stage = (unsigned char*)malloc((size_t)h * w * 3 * 2);
    m_u = (float*)malloc(sizeof(float) * h * w);
    m_v = (float*)malloc(sizeof(float) * h * w);
}

// driver .cpp:159-168 (bao_rgb2rgba indexes through the row tables, bao_basic_cuda.h:258-267)
bool bao_flow_patchmatch_multiscale_cuda::set_data(unsigned char*** img1, unsigned char*** img2)
{
    if (!m_ctx || !img1 || !img2) return false;
    unsigned char* a = m_stage;
    unsigned char* b = m_stage + (size_t)m_h * m_w * 3;
    // img[i][j] points at pixel (i,j)'s three bytes; bao_alloc lays a row out contiguously (bao_basic.h:146-162), which is
    // checked per row -- then the row is one memcpy; any other layout goes through the pointers as bao_rgb2rgba does
    auto gather = [&](unsigned char* dst, unsigned char*** img) {
        for (int i = 0; i < m_h; i++) {
            unsigned char** row = img[i];
            const unsigned char* base = row[0];
            bool contiguous = true;
            for (int j = 1; j < m_w; j++)
                if (row[j] != base + (size_t)3 * j) { contiguous = false; break; }
            unsigned char* d = dst + (size_t)i * m_w * 3;
            if (contiguous) memcpy(d, base, (size_t)m_w * 3);
            else
                for (int j = 0; j < m_w; j++)
                    for (int c = 0; c < 3; c++) d[(size_t)j * 3 + c] = row[j][c];
        }
    };
    gather(a, img1);
    gather(b, img2);
    if (eppm_set_images(m_ctx, a, b, (size_t)m_w * 3) != EPPM_OK) {
        fprintf(stderr, "bao_flow_patchmatch_multiscale_cuda::set_data: %s\n", eppm_last_error());
        return false;
    }
    return true;
}

// driver .cpp:217-315
void bao_flow_patchmatch_multiscale_cuda::compute_flow(float** disp1_x, float** disp1_y, unsigned char*** color_flow)
{
    if (!m_ctx || !disp1_x || !disp1_y) return;
    if (eppm_compute(m_ctx, m_u, m_v) != EPPM_OK) {
        fprintf(stderr, "bao_flow_patchmatch_multiscale_cuda::compute_flow: %s\n", eppm_last_error());
        return;
    }
    for (int i = 0; i < m_h; i++) {                          // disp[i] is a row of w floats (bao_alloc<float>(h,w), bao_basic.h:124-133)
        memcpy(disp1_x[i], m_u + (size_t)i * m_w, sizeof(float) * m_w);
        memcpy(disp1_y[i], m_v + (size_t)i * m_w, sizeof(float) * m_w);
    }
    if (color_flow != NULL) {
        // bao_cuda_convert_flow_to_colorshow(d_colorflow, flow, h, w, 20, 20) on the device flow, D2H, bao_rgba2rgb: driver .cpp:308-314
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
