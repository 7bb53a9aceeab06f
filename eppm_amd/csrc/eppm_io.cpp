// eppm_io.cpp -- file formats and error metrics of the reference's CLI path (main.cpp:56-69):
// PPM reader (basic/bao_basic.cpp:137-218), Middlebury .flo (3rdparty/middlebury/flowIO.cpp:5-20,
// :48-163), EPE/AAE (basic/bao_flow_tools.cpp:64-111).  No GPU code here.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/eppm.h"

static int read_ppm_header(FILE* f, int* type, int* w, int* h)
{
    // "P<n>\n", then comment lines starting with '#', then "w h", then one line with maxval
    if (fgetc(f) != 'P') return -1;
    if (fscanf(f, "%d\n", type) != 1) return -1;
    char line[2048];
    *w = *h = 0;
    while (fgets(line, sizeof(line), f)) {
        if (line[0] == '#') continue;
        if (sscanf(line, "%d %d", w, h) != 2) return -1;
        break;
    }
    if (*w <= 0 || *h <= 0) return -1;
    if (!fgets(line, 100, f)) return -1;   // maxval line
    return 0;
}

extern "C" int eppm_ppm_size(const char* filename, int* h, int* w)
{
    if (!filename || !h || !w) return EPPM_ERR_ARG;
    FILE* f = fopen(filename, "rb");
    if (!f) return EPPM_ERR_ARG;
    int type = 0;
    const int r = read_ppm_header(f, &type, w, h);
    fclose(f);
    return r == 0 ? EPPM_OK : EPPM_ERR_ARG;
}

extern "C" int eppm_load_ppm(const char* filename, uint8_t* image, int h, int w, int* channels)
{
    if (!filename || !image || h <= 0 || w <= 0) return EPPM_ERR_ARG;
    FILE* f = fopen(filename, "rb");
    if (!f) return EPPM_ERR_ARG;
    int type = 0, fw = 0, fh = 0;
    if (read_ppm_header(f, &type, &fw, &fh) != 0) { fclose(f); return EPPM_ERR_ARG; }
    int rc = EPPM_OK;
    if (type == 6 || type == 5) {
        const int nc = (type == 6) ? 3 : 1;
        if (channels) *channels = nc;
        memset(image, 0, (size_t)h * w * nc);                       // as the reference does before fread
        size_t got = fread(image, 1, (size_t)h * w * nc, f);
        (void)got;                                                  // short files leave zeros (reference behaviour)
    } else {
        rc = EPPM_ERR_ARG;                                          // ASCII variants are not used by the flow path
    }
    fclose(f);
    return rc;
}

extern "C" int eppm_save_flo(const char* filename, const float* u, const float* v, int h, int w)
{
    if (!filename || !u || !v || h <= 0 || w <= 0) return EPPM_ERR_ARG;
    const char* dot = strrchr(filename, '.');
    if (!dot || strcmp(dot, ".flo") != 0) return EPPM_ERR_ARG;     // flowIO.cpp:127-133
    FILE* f = fopen(filename, "wb");
    if (!f) return EPPM_ERR_ARG;
    fwrite("PIEH", 1, 4, f);
    const int32_t ww = w, hh = h;
    fwrite(&ww, 4, 1, f);
    fwrite(&hh, 4, 1, f);
    std::vector<float> row((size_t)w * 2);
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) { row[2 * x] = u[(size_t)y * w + x]; row[2 * x + 1] = v[(size_t)y * w + x]; }
        if (fwrite(row.data(), 4, row.size(), f) != row.size()) { fclose(f); return EPPM_ERR_ARG; }
    }
    fclose(f);
    return EPPM_OK;
}

extern "C" int eppm_flo_size(const char* filename, int* h, int* w)
{
    if (!filename || !h || !w) return EPPM_ERR_ARG;
    FILE* f = fopen(filename, "rb");
    if (!f) return EPPM_ERR_ARG;
    float tag = 0;
    int32_t ww = 0, hh = 0;
    const bool ok = fread(&tag, 4, 1, f) == 1 && fread(&ww, 4, 1, f) == 1 && fread(&hh, 4, 1, f) == 1 && tag == 202021.25f &&
                    ww >= 1 && ww <= 99999 && hh >= 1 && hh <= 99999;       // flowIO.cpp:70-84
    fclose(f);
    if (!ok) return EPPM_ERR_ARG;
    *h = hh; *w = ww;
    return EPPM_OK;
}

extern "C" int eppm_load_flo(const char* filename, float* u, float* v, int h, int w)
{
    int fh = 0, fw = 0;
    if (!u || !v || eppm_flo_size(filename, &fh, &fw) != EPPM_OK || fh != h || fw != w) return EPPM_ERR_ARG;
    FILE* f = fopen(filename, "rb");
    if (!f) return EPPM_ERR_ARG;
    fseek(f, 12, SEEK_SET);
    std::vector<float> row((size_t)w * 2);
    for (int y = 0; y < h; y++) {
        if (fread(row.data(), 4, row.size(), f) != row.size()) { fclose(f); return EPPM_ERR_ARG; }
        for (int x = 0; x < w; x++) { u[(size_t)y * w + x] = row[2 * x]; v[(size_t)y * w + x] = row[2 * x + 1]; }
    }
    fclose(f);
    return EPPM_OK;
}

// basic/bao_flow_tools.cpp:64-111 (border = 0): a pixel counts when the ground truth is non-zero and known
extern "C" int eppm_flow_error(const float* u, const float* v, const float* gu, const float* gv, int h, int w, float* epe, float* aae)
{
    if (!u || !v || !gu || !gv || h <= 0 || w <= 0) return EPPM_ERR_ARG;
    int num_valid = 0;
    float total_angle = 0, total_epe = 0;
    for (size_t i = 0; i < (size_t)h * w; i++) {
        const float gtuu = gu[i], gtvv = gv[i];
        if ((fabs(gtuu) > 0 && fabs(gtuu) <= 1e9) || (fabs(gtvv) > 0 && fabs(gtvv) <= 1e9)) {
            num_valid++;
            const float uu = u[i], vv = v[i];
            const float cos_val = (uu * gtuu + vv * gtvv + 1.0f) / (sqrt(uu * uu + vv * vv + 1.0f) * sqrt(gtuu * gtuu + gtvv * gtvv + 1.0f));
            const float angle_val = acos(cos_val);
            total_angle += angle_val;
            const float epe_val = sqrt((uu - gtuu) * (uu - gtuu) + (vv - gtvv) * (vv - gtvv));
            total_epe += epe_val;
        }
    }
    if (num_valid > 0) {
        if (aae) *aae = (total_angle / num_valid) * 180.0f / 3.14159f;
        if (epe) *epe = total_epe / num_valid;
    } else {
        if (aae) *aae = 0;
        if (epe) *epe = 0;
    }
    return EPPM_OK;
}
