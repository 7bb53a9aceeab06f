// runeppm -- CLI with the I/O contract of the reference's demo (main.cpp:36-79): read two P6 PPMs, run
// init + compute_flow through the drop-in class, print the time of that window, write flow.flo.
// Usage: runeppm [img1.ppm img2.ppm [out.flo]]   (defaults: frame10.ppm frame11.ppm flow.flo)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bao_flow_patchmatch_multiscale_cuda.h"
#include "eppm.h"

template <typename T>
static T*** alloc3(int n, int r, int c, std::vector<T>& store, std::vector<T*>& rows, std::vector<T**>& planes)
{
    store.assign((size_t)n * r * c, T());
    rows.resize((size_t)n * r);
    planes.resize(n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < r; j++) rows[(size_t)i * r + j] = &store[((size_t)i * r + j) * c];
    for (int i = 0; i < n; i++) planes[i] = &rows[(size_t)i * r];
    return planes.data();
}

int main(int argc, char** argv)
{
    const char* f1 = argc > 2 ? argv[1] : "frame10.ppm";
    const char* f2 = argc > 2 ? argv[2] : "frame11.ppm";
    const char* fo = argc > 3 ? argv[3] : "flow.flo";
    int h = 0, w = 0, h2 = 0, w2 = 0;
    if (eppm_ppm_size(f1, &h, &w) != EPPM_OK || eppm_ppm_size(f2, &h2, &w2) != EPPM_OK || h != h2 || w != w2) {
        fprintf(stderr, "cannot read %s / %s (or sizes differ)\n", f1, f2);
        return 1;
    }
    std::vector<unsigned char> s1, s2;
    std::vector<unsigned char*> r1, r2;
    std::vector<unsigned char**> p1, p2;
    unsigned char*** img1 = alloc3<unsigned char>(h, w, 3, s1, r1, p1);   // bao_alloc<unsigned char>(h,w,3), main.cpp:42-43
    unsigned char*** img2 = alloc3<unsigned char>(h, w, 3, s2, r2, p2);
    int nch = 0;
    printf("loading image ... \n");
    eppm_load_ppm(f1, img1[0][0], h, w, &nch);
    eppm_load_ppm(f2, img2[0][0], h, w, &nch);
    std::vector<float> u((size_t)h * w, 0.f), v((size_t)h * w, 0.f);
    std::vector<float*> ur(h), vr(h);
    for (int i = 0; i < h; i++) { ur[i] = &u[(size_t)i * w]; vr[i] = &v[(size_t)i * w]; }

    printf("Processing (image size %d * %d * %d)...\n", w, h, nch);
    bao_flow_patchmatch_multiscale_cuda eppm;
    auto t0 = std::chrono::steady_clock::now();
    eppm.init(img1, img2, h, w);                         // main.cpp:63-64: the reference's timed window
    eppm.compute_flow(ur.data(), vr.data());
    auto t1 = std::chrono::steady_clock::now();
    printf("GPU: %.3f s (init + compute_flow)\n", std::chrono::duration<double>(t1 - t0).count());
    t0 = std::chrono::steady_clock::now();
    eppm.set_data(img1, img2);
    eppm.compute_flow(ur.data(), vr.data());
    t1 = std::chrono::steady_clock::now();
    printf("GPU: %.3f s (set_data + compute_flow, steady state)\n", std::chrono::duration<double>(t1 - t0).count());
    printf("Saving flo file...%d*%d\n", h, w);
    if (eppm_save_flo(fo, u.data(), v.data(), h, w) != EPPM_OK) { fprintf(stderr, "cannot write %s\n", fo); return 1; }
    return 0;
}
