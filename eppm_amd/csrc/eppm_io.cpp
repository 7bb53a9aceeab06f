// eppm_io.cpp -- file formats and error metrics of the reference's CLI path (main.cpp:56-69):
// PPM reader (basic/bao_basic.cpp:137-218), Middlebury .flo (3rdparty/middlebury/flowIO.cpp:5-20,
// :48-163), EPE/AAE (basic/bao_flow_tools.cpp:64-111).  No GPU code here.
#include <limits.h>
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

// basic/bao_flow_tools.cpp:64-111: a pixel counts when the ground truth is non-zero and known; `border` pixels on every side are left out
extern "C" int eppm_flow_error_border(const float* u, const float* v, const float* gu, const float* gv, int h, int w, int border, float* epe, float* aae)
{
    if (!u || !v || !gu || !gv || h <= 0 || w <= 0 || border < 0) return EPPM_ERR_ARG;
    int num_valid = 0;
    float total_angle = 0, total_epe = 0;
    for (int y = border; y < h - border; y++)
      for (int x = border; x < w - border; x++) {
        const size_t i = (size_t)y * w + x;
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
extern "C" int eppm_flow_error(const float* u, const float* v, const float* gu, const float* gv, int h, int w, float* epe, float* aae)
{
    return eppm_flow_error_border(u, v, gu, gv, h, w, 0, epe, aae);
}

// basic/bao_flow_tools.cpp:114-141: fraction of the pixels with a known ground truth whose end-point error exceeds error_thresh;
// error_map (h*w bytes, or NULL): 255 where it does, 0 elsewhere
extern "C" int eppm_flow_error_percentage(const float* u, const float* v, const float* gu, const float* gv, int h, int w, int error_thresh,
                                          uint8_t* error_map, float* fraction)
{
    if (!u || !v || !gu || !gv || !fraction || h <= 0 || w <= 0) return EPPM_ERR_ARG;
    if (error_map) memset(error_map, 0, (size_t)h * w);
    int num_valid = 0, num_correct = 0;
    for (size_t i = 0; i < (size_t)h * w; i++) {
        const float gtuu = gu[i], gtvv = gv[i];
        if (fabs(gtuu) <= 1e9 || fabs(gtvv) <= 1e9) {
            num_valid++;
            const float uu = u[i], vv = v[i];
            const float epe_val = sqrt((uu - gtuu) * (uu - gtuu) + (vv - gtvv) * (vv - gtvv));
            if (epe_val <= error_thresh) num_correct++;
            else if (error_map) error_map[i] = 255;
        }
    }
    *fraction = num_valid > 0 ? 1.0f - float(num_correct) / float(num_valid) : 0.0f;
    return EPPM_OK;
}

// basic/bao_flow_tools.cpp:166-197: both components clamped to [-|cutoff|, |cutoff|]; unknown vectors (a component above 1e9 in
// magnitude, flowIO.cpp:37-41) pass through unless cut_invalid is set
extern "C" int eppm_flow_cutoff(float* u_out, float* v_out, const float* u, const float* v, int h, int w, int cutoff, int cut_invalid)
{
    if (!u_out || !v_out || !u || !v || h <= 0 || w <= 0) return EPPM_ERR_ARG;
    const int c = (cutoff == INT_MIN) ? INT_MAX : abs(cutoff);      // fabs(INT_MIN) does not fit an int: undefined in the reference, the largest cutoff here
    for (size_t i = 0; i < (size_t)h * w; i++) {
        const float x = u[i], y = v[i];
        if (!cut_invalid && (fabs(x) > 1e9 || fabs(y) > 1e9)) { u_out[i] = x; v_out[i] = y; continue; }
        const float lx = (x < c) ? x : (float)c, ly = (y < c) ? y : (float)c;           // __min(val, cutoff), then __max(., -cutoff)
        u_out[i] = (lx > -c) ? lx : (float)-c;
        v_out[i] = (ly > -c) ? ly : (float)-c;
    }
    return EPPM_OK;
}

// Host colour coding of a flow field (basic/bao_flow_tools.cpp:200-231 on Middlebury's computeColor, colorcode.cpp:30-85): vectors are
// scaled by the largest known radius of the field, unknown vectors are black; rgb: h*w*3 bytes, R,G,B.  (The device routine of the
// optional color_flow output, k_color.hip, is the reference's CUDA port of the same wheel with a fixed scale.)
namespace {
struct Wheel {
    int n = 0, c[60][3];
    Wheel()
    {
        const int RY = 15, YG = 6, GC = 4, CB = 11, BM = 13, MR = 6;
        auto put = [&](int r, int g, int b) { c[n][0] = r; c[n][1] = g; c[n][2] = b; n++; };
        for (int i = 0; i < RY; i++) put(255, 255 * i / RY, 0);
        for (int i = 0; i < YG; i++) put(255 - 255 * i / YG, 255, 0);
        for (int i = 0; i < GC; i++) put(0, 255, 255 * i / GC);
        for (int i = 0; i < CB; i++) put(0, 255 - 255 * i / CB, 255);
        for (int i = 0; i < BM; i++) put(255 * i / BM, 0, 255);
        for (int i = 0; i < MR; i++) put(255, 0, 255 - 255 * i / MR);
    }
};
}  // namespace
extern "C" int eppm_flow_to_color_host(uint8_t* rgb, const float* u, const float* v, int h, int w)
{
    if (!rgb || !u || !v || h <= 0 || w <= 0) return EPPM_ERR_ARG;
    static const Wheel W;
    float maxrad = -1;
    for (size_t i = 0; i < (size_t)h * w; i++) {
        const float fx = u[i], fy = v[i];
        if (!(fabs(fx) <= 1e9) || !(fabs(fy) <= 1e9)) continue;
        const float rad = sqrt(fx * fx + fy * fy);
        maxrad = (maxrad > rad) ? maxrad : rad;
    }
    // A field with no motion (maxrad 0) or no known vector (-1) divides 0 by 0 in the reference and indexes the wheel with (int)NaN --
    // undefined there (a crash on x86); here the scale is 1 (Middlebury's own color_flow tool: "if (maxrad == 0) maxrad = 1"), and a
    // vector with a NaN component is drawn black like an unknown one.
    if (!(maxrad > 0)) maxrad = 1;
    for (size_t i = 0; i < (size_t)h * w; i++) {
        uint8_t* o = rgb + i * 3;
        if (!(fabs(u[i]) <= 1e9) || !(fabs(v[i]) <= 1e9)) { o[0] = o[1] = o[2] = 0; continue; }
        const float fx = u[i] / maxrad, fy = v[i] / maxrad;
        const float rad = sqrt(fx * fx + fy * fy);
        const float a = atan2(-fy, -fx) / M_PI;
        const float fk = (a + 1.0f) / 2.0f * (W.n - 1);
        const int k0 = (int)fk, k1 = (k0 + 1) % W.n;
        const float f = fk - k0;
        for (int b = 0; b < 3; b++) {
            const float col0 = W.c[k0][b] / 255.0f, col1 = W.c[k1][b] / 255.0f;
            float col = (1 - f) * col0 + f * col1;
            if (rad <= 1) col = 1 - rad * (1 - col);
            else col *= .75;
            o[b] = (uint8_t)(int)(255.0 * col);            // computeColor writes B,G,R; bao_convert_flow_to_colorshow swaps to R,G,B
        }
    }
    return EPPM_OK;
}
