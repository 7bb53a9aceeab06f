// ref_io_shim.cpp -- test infrastructure.  extern "C" doorways into the REFERENCE's own host-side
// I/O code (compiled where it lies under /root/reference by oracle/Makefile, output only in
// oracle/_ref/).  Used by tests to check eppm_amd's PPM reader, .flo writer/reader and EPE
// against the reference implementation itself (kind "reference", not a restatement).
#include "bao_basic.h"
#include "bao_flow_tools.h"
typedef unsigned char uchar;                       // imageLib/Image.h:… defines it for colorcode.h; only the prototype is needed here
#include "../3rdparty/middlebury/colorcode.h"

extern "C" {

// bao_basic.cpp:137-218
int refio_load_ppm(const char* filename, unsigned char* image, int h, int w)
{
    int nc = 0;
    bao_loadimage_ppm((char*)filename, image, h, w, &nc);
    return nc;
}

// bao_flow_tools.cpp:49-62 -> flowIO.cpp:122-163
void refio_save_flo(const char* filename, const float* u, const float* v, int h, int w)
{
    float** dx = bao_alloc<float>(h, w);
    float** dy = bao_alloc<float>(h, w);
    memcpy(dx[0], u, sizeof(float) * h * w);
    memcpy(dy[0], v, sizeof(float) * h * w);
    bao_save_flo_file(filename, dx, dy, h, w);
    bao_free(dx);
    bao_free(dy);
}

// bao_flow_tools.cpp:33-47
int refio_load_flo(const char* filename, float* u, float* v, int h, int w)
{
    int fh = 0, fw = 0;
    bao_read_flo_file_size(filename, fh, fw);
    if (fh != h || fw != w) return -1;
    float** dx = bao_alloc<float>(h, w);
    float** dy = bao_alloc<float>(h, w);
    bao_load_flo_file(filename, dx, dy, h, w);
    memcpy(u, dx[0], sizeof(float) * h * w);
    memcpy(v, dy[0], sizeof(float) * h * w);
    bao_free(dx);
    bao_free(dy);
    return 0;
}

// bao_flow_tools.cpp:64-111
void refio_flow_error(const float* u, const float* v, const float* gu, const float* gv, int h, int w, float* epe, float* aae)
{
    float** a = bao_alloc<float>(h, w); float** b = bao_alloc<float>(h, w);
    float** c = bao_alloc<float>(h, w); float** d = bao_alloc<float>(h, w);
    memcpy(a[0], u, sizeof(float) * h * w); memcpy(b[0], v, sizeof(float) * h * w);
    memcpy(c[0], gu, sizeof(float) * h * w); memcpy(d[0], gv, sizeof(float) * h * w);
    float e = 0, g = 0;
    bao_calc_flow_error(a, b, c, d, h, w, e, g, 0, false);
    *epe = e; *aae = g;
    bao_free(a); bao_free(b); bao_free(c); bao_free(d);
}

// bao_flow_tools.cpp:64-111 with a border
void refio_flow_error_border(const float* u, const float* v, const float* gu, const float* gv, int h, int w, int border, float* epe, float* aae)
{
    float** a = bao_alloc<float>(h, w); float** b = bao_alloc<float>(h, w);
    float** c = bao_alloc<float>(h, w); float** d = bao_alloc<float>(h, w);
    memcpy(a[0], u, sizeof(float) * h * w); memcpy(b[0], v, sizeof(float) * h * w);
    memcpy(c[0], gu, sizeof(float) * h * w); memcpy(d[0], gv, sizeof(float) * h * w);
    float e = 0, g = 0;
    bao_calc_flow_error(a, b, c, d, h, w, e, g, border, false);
    *epe = e; *aae = g;
    bao_free(a); bao_free(b); bao_free(c); bao_free(d);
}

// bao_flow_tools.cpp:114-141
float refio_flow_error_percentage(const float* u, const float* v, const float* gu, const float* gv, int h, int w, int thresh, unsigned char* emap)
{
    float** a = bao_alloc<float>(h, w); float** b = bao_alloc<float>(h, w);
    float** c = bao_alloc<float>(h, w); float** d = bao_alloc<float>(h, w);
    unsigned char** m = emap ? bao_alloc<unsigned char>(h, w) : 0;
    memcpy(a[0], u, sizeof(float) * h * w); memcpy(b[0], v, sizeof(float) * h * w);
    memcpy(c[0], gu, sizeof(float) * h * w); memcpy(d[0], gv, sizeof(float) * h * w);
    const float r = bao_calc_flow_error_percentage(a, b, c, d, h, w, thresh, m);
    if (m) { memcpy(emap, m[0], (size_t)h * w); bao_free(m); }
    bao_free(a); bao_free(b); bao_free(c); bao_free(d);
    return r;
}

// bao_flow_tools.cpp:166-197
void refio_flow_cutoff(float* uo, float* vo, const float* u, const float* v, int h, int w, int cutoff, int cut_invalid)
{
    float** a = bao_alloc<float>(h, w); float** b = bao_alloc<float>(h, w);
    float** c = bao_alloc<float>(h, w); float** d = bao_alloc<float>(h, w);
    memcpy(a[0], u, sizeof(float) * h * w); memcpy(b[0], v, sizeof(float) * h * w);
    bao_flow_cutoff(c, d, a, b, h, w, cutoff, cut_invalid != 0);
    memcpy(uo, c[0], sizeof(float) * h * w); memcpy(vo, d[0], sizeof(float) * h * w);
    bao_free(a); bao_free(b); bao_free(c); bao_free(d);
}

// bao_flow_tools.cpp:200-231
void refio_flow_to_color(unsigned char* rgb, const float* u, const float* v, int h, int w)
{
    float** a = bao_alloc<float>(h, w); float** b = bao_alloc<float>(h, w);
    unsigned char*** c = bao_alloc<unsigned char>(h, w, 3);
    memcpy(a[0], u, sizeof(float) * h * w); memcpy(b[0], v, sizeof(float) * h * w);
    bao_convert_flow_to_colorshow(c, a, b, h, w);
    memcpy(rgb, c[0][0], (size_t)h * w * 3);
    bao_free(a); bao_free(b); bao_free(c);
}

// 3rdparty/middlebury/colorcode.cpp:61-85 -- the CPU routine the reference's device colour coding (basic/bao_basic_cuda.cuh:776-807)
// is a port of; pix is B,G,R.  fx, fy already divided by the maximum radius.
void refio_compute_color(float fx, float fy, unsigned char* pix) { computeColor(fx, fy, pix); }

// bao_basic.h:196-211
int refio_pyr_init_dim(int* arrH, int* arrW, int h, int w, int maxDepth, float ratio)
{
    int *ah = 0, *aw = 0;
    int n = bao_pyr_init_dim(ah, aw, h, w, maxDepth, (BAO_FLOAT)ratio);
    for (int i = 0; i < n; i++) { arrH[i] = ah[i]; arrW[i] = aw[i]; }
    bao_pyr_destroy_dim(ah, aw);
    return n;
}
}
