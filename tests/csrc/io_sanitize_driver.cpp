// Host I/O and flow tools of the library (eppm_amd/csrc/eppm_io.cpp, no GPU code) under ASan + UBSan: malformed PPM / .flo files and
// degenerate flow fields.  Built and run by tests/test_abi_cpu.py::test_host_io_under_sanitizers; prints "ok" when every call returned
// what it should and the sanitizers stayed silent.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/eppm.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL line %d: %s\n", __LINE__, #c); fails++; } } while (0)

static std::string put(const std::string& dir, const char* name, const void* data, size_t n)
{
    const std::string p = dir + "/" + name;
    FILE* f = fopen(p.c_str(), "wb");
    if (n) fwrite(data, 1, n, f);
    fclose(f);
    return p;
}
static std::string puts_(const std::string& dir, const char* name, const std::string& s) { return put(dir, name, s.data(), s.size()); }

int main(int argc, char** argv)
{
    const std::string d = argc > 1 ? argv[1] : "/tmp";
    int h = 0, w = 0, nc = 0;
    // ---- PPM headers
    CHECK(eppm_ppm_size((d + "/missing.ppm").c_str(), &h, &w) != EPPM_OK);
    CHECK(eppm_ppm_size(NULL, &h, &w) != EPPM_OK);
    CHECK(eppm_ppm_size(puts_(d, "empty.ppm", "").c_str(), &h, &w) != EPPM_OK);
    CHECK(eppm_ppm_size(puts_(d, "p_only.ppm", "P").c_str(), &h, &w) != EPPM_OK);
    CHECK(eppm_ppm_size(puts_(d, "notppm.ppm", "GIF89a").c_str(), &h, &w) != EPPM_OK);
    CHECK(eppm_ppm_size(puts_(d, "neg.ppm", "P6\n-5 10\n255\n").c_str(), &h, &w) != EPPM_OK);
    CHECK(eppm_ppm_size(puts_(d, "zero.ppm", "P6\n0 0\n255\n").c_str(), &h, &w) != EPPM_OK);
    CHECK(eppm_ppm_size(puts_(d, "nodims.ppm", "P6\n# only a comment\n").c_str(), &h, &w) != EPPM_OK);
    CHECK(eppm_ppm_size(puts_(d, "nomax.ppm", "P6\n4 3\n").c_str(), &h, &w) != EPPM_OK);
    CHECK(eppm_ppm_size(puts_(d, "garbage.ppm", "P6\nfoo bar\n255\n").c_str(), &h, &w) != EPPM_OK);
    CHECK(eppm_ppm_size(puts_(d, "bigtype.ppm", "P99999999999999999999\n4 3\n255\n").c_str(), &h, &w) == EPPM_OK || true);   // any answer, no UB
    {
        std::string longc = "P6\n#" + std::string(5000, 'x') + "\n4 3\n255\n";           // a comment longer than the line buffer
        (void)eppm_ppm_size(puts_(d, "longcomment.ppm", longc).c_str(), &h, &w);          // the reference's reader has the same 2 KB line: any answer, no overrun
    }
    {
        std::string ok = "P6\n# c1\n# c2\n4 3\n255\n";
        for (int i = 0; i < 36; i++) ok.push_back((char)(i * 7));
        const std::string p = puts_(d, "ok.ppm", ok);
        CHECK(eppm_ppm_size(p.c_str(), &h, &w) == EPPM_OK && h == 3 && w == 4);
        std::vector<uint8_t> img(36, 0xee);
        CHECK(eppm_load_ppm(p.c_str(), img.data(), 3, 4, &nc) == EPPM_OK && nc == 3 && img[35] == (uint8_t)(35 * 7));
        CHECK(eppm_load_ppm(p.c_str(), NULL, 3, 4, &nc) != EPPM_OK);
        CHECK(eppm_load_ppm(p.c_str(), img.data(), 0, 4, &nc) != EPPM_OK);
        CHECK(eppm_load_ppm(p.c_str(), img.data(), 3, 4, NULL) == EPPM_OK);
        // short file: the missing tail stays zero (reference: memset, then fread)
        const std::string sh = puts_(d, "short.ppm", ok.substr(0, ok.size() - 10));
        std::fill(img.begin(), img.end(), 0xee);
        CHECK(eppm_load_ppm(sh.c_str(), img.data(), 3, 4, &nc) == EPPM_OK && img[35] == 0 && img[25] == (uint8_t)(25 * 7));
        // grey (P5) and ASCII (P3: refused)
        std::string p5 = "P5\n4 3\n255\n" + std::string(12, 'a');
        CHECK(eppm_load_ppm(puts_(d, "g.pgm", p5).c_str(), img.data(), 3, 4, &nc) == EPPM_OK && nc == 1 && img[11] == 'a');
        CHECK(eppm_load_ppm(puts_(d, "a.ppm", "P3\n1 1\n255\n1 2 3\n").c_str(), img.data(), 1, 1, &nc) != EPPM_OK);
    }
    // ---- .flo
    {
        const float tag = 202021.25f;
        std::vector<float> u(12), v(12), u2(12), v2(12);
        for (int i = 0; i < 12; i++) { u[i] = i * 0.5f; v[i] = -i; }
        const std::string p = d + "/a.flo";
        CHECK(eppm_save_flo(p.c_str(), u.data(), v.data(), 3, 4) == EPPM_OK);
        CHECK(eppm_save_flo((d + "/a.txt").c_str(), u.data(), v.data(), 3, 4) != EPPM_OK);
        CHECK(eppm_save_flo((d + "/noext").c_str(), u.data(), v.data(), 3, 4) != EPPM_OK);
        CHECK(eppm_save_flo((d + "/no/such/dir/a.flo").c_str(), u.data(), v.data(), 3, 4) != EPPM_OK);
        CHECK(eppm_save_flo(p.c_str(), NULL, v.data(), 3, 4) != EPPM_OK);
        CHECK(eppm_flo_size(p.c_str(), &h, &w) == EPPM_OK && h == 3 && w == 4);
        CHECK(eppm_load_flo(p.c_str(), u2.data(), v2.data(), 3, 4) == EPPM_OK && u2 == u && v2 == v);
        CHECK(eppm_load_flo(p.c_str(), u2.data(), v2.data(), 4, 3) != EPPM_OK);           // size mismatch
        CHECK(eppm_load_flo(p.c_str(), NULL, v2.data(), 3, 4) != EPPM_OK);
        // truncated body, wrong tag, absurd sizes, header only
        std::vector<char> raw(12 + 96);
        FILE* f = fopen(p.c_str(), "rb"); CHECK(fread(raw.data(), 1, raw.size(), f) == raw.size()); fclose(f);
        CHECK(eppm_load_flo(put(d, "trunc.flo", raw.data(), raw.size() - 5).c_str(), u2.data(), v2.data(), 3, 4) != EPPM_OK);
        CHECK(eppm_flo_size(put(d, "hdr.flo", raw.data(), 7).c_str(), &h, &w) != EPPM_OK);
        CHECK(eppm_flo_size(put(d, "e.flo", raw.data(), 0).c_str(), &h, &w) != EPPM_OK);
        std::vector<char> bad = raw; bad[0] ^= 1;
        CHECK(eppm_flo_size(put(d, "tag.flo", bad.data(), bad.size()).c_str(), &h, &w) != EPPM_OK);
        for (int32_t dim : {0, -1, 100000, 0x7fffffff}) {
            std::vector<char> b2 = raw; memcpy(&b2[4], &dim, 4);
            CHECK(eppm_flo_size(put(d, "dim.flo", b2.data(), b2.size()).c_str(), &h, &w) != EPPM_OK);
            b2 = raw; memcpy(&b2[8], &dim, 4);
            CHECK(eppm_flo_size(put(d, "dim.flo", b2.data(), b2.size()).c_str(), &h, &w) != EPPM_OK);
        }
        (void)tag;
    }
    // ---- flow tools on degenerate fields
    {
        const int H = 5, W = 7, N = H * W;
        const float big = 1e10f, nan = nanf(""), inf = INFINITY;
        std::vector<float> z(N, 0.0f), unk(N, big), mix(N), gu(N), gv(N), o1(N), o2(N);
        for (int i = 0; i < N; i++) { mix[i] = (i % 5 == 0) ? nan : (i % 7 == 0) ? inf : (i % 3 == 0) ? big : (float)(i - 17) * 3.5f; gu[i] = (float)(i % 4) - 1.5f; gv[i] = (i % 6 == 0) ? big : 0.25f * i; }
        std::vector<uint8_t> rgb(N * 3), emap(N);
        float epe = -1, aae = -1, frac = -1;
        // colour coding: a static scene, a field with no known vector, NaN / inf / unknown components
        CHECK(eppm_flow_to_color_host(rgb.data(), z.data(), z.data(), H, W) == EPPM_OK && rgb[0] == 255 && rgb[1] == 255 && rgb[2] == 255);
        CHECK(eppm_flow_to_color_host(rgb.data(), unk.data(), unk.data(), H, W) == EPPM_OK && rgb[0] == 0 && rgb[N * 3 - 1] == 0);
        CHECK(eppm_flow_to_color_host(rgb.data(), mix.data(), z.data(), H, W) == EPPM_OK && rgb[0] == 0);
        CHECK(eppm_flow_to_color_host(rgb.data(), z.data(), mix.data(), H, W) == EPPM_OK);
        CHECK(eppm_flow_to_color_host(NULL, z.data(), z.data(), H, W) != EPPM_OK);
        CHECK(eppm_flow_to_color_host(rgb.data(), z.data(), z.data(), 0, W) != EPPM_OK);
        // errors
        CHECK(eppm_flow_error(z.data(), z.data(), z.data(), z.data(), H, W, &epe, &aae) == EPPM_OK && epe == 0 && aae == 0);      // no valid pixel
        CHECK(eppm_flow_error(mix.data(), mix.data(), gu.data(), gv.data(), H, W, &epe, &aae) == EPPM_OK);
        CHECK(eppm_flow_error(z.data(), z.data(), gu.data(), gv.data(), H, W, NULL, NULL) == EPPM_OK);
        CHECK(eppm_flow_error_border(z.data(), z.data(), gu.data(), gv.data(), H, W, 100, &epe, &aae) == EPPM_OK && epe == 0);     // border swallows the image
        CHECK(eppm_flow_error_border(z.data(), z.data(), gu.data(), gv.data(), H, W, -1, &epe, &aae) != EPPM_OK);
        CHECK(eppm_flow_error_border(z.data(), z.data(), gu.data(), gv.data(), H, W, 0x7fffffff, &epe, &aae) == EPPM_OK);
        CHECK(eppm_flow_error_percentage(mix.data(), z.data(), gu.data(), gv.data(), H, W, 3, emap.data(), &frac) == EPPM_OK && frac >= 0 && frac <= 1);
        CHECK(eppm_flow_error_percentage(z.data(), z.data(), unk.data(), unk.data(), H, W, 3, NULL, &frac) == EPPM_OK && frac == 0);
        CHECK(eppm_flow_error_percentage(z.data(), z.data(), gu.data(), gv.data(), H, W, 3, NULL, NULL) != EPPM_OK);
        // cutoff
        for (int c : {0, 5, -5, 0x7fffffff, (int)0x80000000}) {
            CHECK(eppm_flow_cutoff(o1.data(), o2.data(), mix.data(), gv.data(), H, W, c, 0) == EPPM_OK);
            CHECK(eppm_flow_cutoff(o1.data(), o2.data(), mix.data(), gv.data(), H, W, c, 1) == EPPM_OK);
        }
        CHECK(eppm_flow_cutoff(o1.data(), o2.data(), mix.data(), gv.data(), H, W, 5, 1) == EPPM_OK && o2[6] == 5.0f);            // unknown vector cut when asked
        CHECK(eppm_flow_cutoff(o1.data(), o2.data(), mix.data(), gv.data(), H, W, 5, 0) == EPPM_OK && o2[6] == big);
        CHECK(eppm_flow_cutoff(o1.data(), NULL, mix.data(), gv.data(), H, W, 5, 0) != EPPM_OK);
    }
    if (fails) return 1;
    printf("ok\n");
    return 0;
}
