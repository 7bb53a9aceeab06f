// runeppm -- CLI with the I/O contract of the reference's demo (main.cpp:36-79): read two P6 PPMs, run
// init + compute_flow through the drop-in class, print the time of that window, write flow.flo.
//
//   runeppm [options] [img1.ppm img2.ppm [out.flo]]          (defaults: frame10.ppm frame11.ppm flow.flo)
//
//   --size WxH        synthetic pair instead of files: smooth value noise, image 2 = image 1 translated by
//                     (+5,-3) px; the end-point error against that translation is printed
//   --seed N          PatchMatch RNG seed (1234, bao_pmflow_kernel.cu:68); also seeds the synthetic images
//   --levels N        pyramid depth (PYR_MAX_DEPTH 3, defs.h:31)
//   --patch-r N       patch radius (PATCH_R 9, defs.h:44)
//   --iters N         PatchMatch iterations (NUM_ITER 10, defs.h:45)
//   --propagation M   0 segmented sweeps (live in the reference), 1 jump flood, 2 4-neighbour
//   --pairs P         process the pair P times in steady state (set_data + compute_flow); throughput is printed
//   --gpus G          G worker threads, one context per GPU; the P pairs are dealt round-robin (pair i -> GPU i mod G)
//   --batch B         each worker runs its pairs B at a time through a batch context (eppm_create_batch: every kernel launch
//                     covers the B pairs); default 1 = the drop-in class, one pair per launch
//   --pin             set_option("pin_caller_buffers", 1): the class registers the image and flow blocks for DMA, no host copies
//   --out file.flo    output name (same as the third positional argument)
//   --gt file.flo     print EPE / AAE of the result against a ground-truth .flo (bao_flow_tools.cpp:64-111)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "bao_flow_patchmatch_multiscale_cuda.h"
#include "eppm.h"

template <typename T>
struct Array3 {       // bao_alloc<T>(n,r,c): one contiguous block reachable through row-pointer tables (bao_basic.h:146-162)
    std::vector<T> store;
    std::vector<T*> rows;
    std::vector<T**> planes;
    Array3(int n, int r, int c) : store((size_t)n * r * c), rows((size_t)n * r), planes(n)
    {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < r; j++) rows[(size_t)i * r + j] = &store[((size_t)i * r + j) * c];
        for (int i = 0; i < n; i++) planes[i] = &rows[(size_t)i * r];
    }
    T*** p() { return planes.data(); }
};

struct Options {
    const char *f1 = "frame10.ppm", *f2 = "frame11.ppm", *fo = "flow.flo", *gt = nullptr;
    int sw = 0, sh = 0, pairs = 1, gpus = 1, batch = 1;
    std::vector<std::pair<std::string, long long>> opts;
};

static unsigned hash32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// smooth value noise, three octaves, per channel
static void synth_image(unsigned char* rgb, int h, int w, int ox, int oy, unsigned seed)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) {
                float acc = 0.f, amp = 1.f, tot = 0.f;
                for (int o = 0; o < 3; o++) {
                    const int cell = 32 >> (2 * o);
                    const float fx = (float)(x + ox + 4096) / cell, fy = (float)(y + oy + 4096) / cell;
                    const int ix = (int)fx, iy = (int)fy;
                    const float tx = fx - ix, ty = fy - iy;
                    float v[2][2];
                    for (int b = 0; b < 2; b++)
                        for (int a = 0; a < 2; a++)
                            v[b][a] = (hash32((unsigned)(ix + a) * 73856093u ^ (unsigned)(iy + b) * 19349663u ^ (unsigned)(c + 3 * o) * 83492791u ^ seed) & 0xffff) / 65535.f;
                    const float sx = tx * tx * (3 - 2 * tx), sy = ty * ty * (3 - 2 * ty);
                    acc += amp * ((v[0][0] * (1 - sx) + v[0][1] * sx) * (1 - sy) + (v[1][0] * (1 - sx) + v[1][1] * sx) * sy);
                    tot += amp;
                    amp *= 0.5f;
                }
                rgb[((size_t)y * w + x) * 3 + c] = (unsigned char)(255.f * acc / tot);
            }
}

static bool apply_opt(eppm_params& p, const std::string& name, long long v)
{
    if (name == "patch_r") p.patch_r = (int)v;
    else if (name == "num_iter") p.num_iter = (int)v;
    else if (name == "seed") p.seed = (unsigned long long)v;
    else if (name == "propagation") p.propagation = (int)v;
    else if (name == "levels") p.levels = (int)v;
    else if (name == "pin_caller_buffers" || name == "verify_tables_every_call" || name == "trust_verified_tables") return true;      // options of the class, not of eppm_params
    else return false;
    return true;
}

static int usage()
{
    fprintf(stderr, "usage: runeppm [--size WxH] [--seed N] [--levels N] [--patch-r N] [--iters N] [--propagation M]\n"
                    "               [--pin] [--pairs P] [--gpus G] [--batch B] [--gt file.flo] [--out file.flo] [img1.ppm img2.ppm [out.flo]]\n");
    return 2;
}

int main(int argc, char** argv)
{
    Options o;
    std::vector<const char*> pos;
    for (int i = 1; i < argc; i++) {
        const char* a = argv[i];
        auto val = [&](long long* v) { if (i + 1 >= argc) return false; *v = atoll(argv[++i]); return true; };
        long long v = 0;
        if (!strcmp(a, "--size")) {
            if (i + 1 >= argc || sscanf(argv[++i], "%dx%d", &o.sw, &o.sh) != 2) return usage();
        } else if (!strcmp(a, "--seed")) { if (!val(&v)) return usage(); o.opts.push_back({"seed", v}); }
        else if (!strcmp(a, "--levels")) { if (!val(&v)) return usage(); o.opts.push_back({"levels", v}); }
        else if (!strcmp(a, "--patch-r")) { if (!val(&v)) return usage(); o.opts.push_back({"patch_r", v}); }
        else if (!strcmp(a, "--iters")) { if (!val(&v)) return usage(); o.opts.push_back({"num_iter", v}); }
        else if (!strcmp(a, "--propagation")) { if (!val(&v)) return usage(); o.opts.push_back({"propagation", v}); }
        else if (!strcmp(a, "--pairs")) { if (!val(&v) || v < 1) return usage(); o.pairs = (int)v; }
        else if (!strcmp(a, "--gpus")) { if (!val(&v) || v < 1) return usage(); o.gpus = (int)v; }
        else if (!strcmp(a, "--batch")) { if (!val(&v) || v < 1) return usage(); o.batch = (int)v; }
        else if (!strcmp(a, "--pin")) o.opts.push_back({"pin_caller_buffers", 1});
        else if (!strcmp(a, "--gt")) { if (i + 1 >= argc) return usage(); o.gt = argv[++i]; }
        else if (!strcmp(a, "--out")) { if (i + 1 >= argc) return usage(); o.fo = argv[++i]; }
        else if (a[0] == '-' && a[1] == '-') return usage();
        else pos.push_back(a);
    }
    if (pos.size() == 1 || pos.size() > 3) return usage();
    if (pos.size() >= 2) { o.f1 = pos[0]; o.f2 = pos[1]; }
    if (pos.size() == 3) o.fo = pos[2];

    int h = 0, w = 0, nch = 3;
    if (o.sw > 0) { w = o.sw; h = o.sh; }
    else {
        int h2 = 0, w2 = 0;
        if (eppm_ppm_size(o.f1, &h, &w) != EPPM_OK || eppm_ppm_size(o.f2, &h2, &w2) != EPPM_OK || h != h2 || w != w2) {
            fprintf(stderr, "cannot read %s / %s (or sizes differ)\n", o.f1, o.f2);
            return 1;
        }
    }
    Array3<unsigned char> img1(h, w, 3), img2(h, w, 3);                  // bao_alloc<unsigned char>(h,w,3), main.cpp:42-43
    printf("loading image ... \n");
    if (o.sw > 0) {
        unsigned seed = 1234;
        for (auto& kv : o.opts) if (kv.first == "seed") seed = (unsigned)kv.second;
        synth_image(img1.store.data(), h, w, 0, 0, seed);
        synth_image(img2.store.data(), h, w, -5, 3, seed);              // I2(x,y) = I1(x-5, y+3): flow (+5,-3)
    } else {
        if (eppm_load_ppm(o.f1, img1.store.data(), h, w, &nch) != EPPM_OK || eppm_load_ppm(o.f2, img2.store.data(), h, w, &nch) != EPPM_OK) {
            fprintf(stderr, "cannot read %s / %s\n", o.f1, o.f2);
            return 1;
        }
    }
    std::vector<float> u((size_t)h * w, 0.f), v((size_t)h * w, 0.f);
    std::vector<float*> ur(h), vr(h);
    for (int i = 0; i < h; i++) { ur[i] = &u[(size_t)i * w]; vr[i] = &v[(size_t)i * w]; }

    printf("Processing (image size %d * %d * %d)...\n", w, h, nch);
    {
        bao_flow_patchmatch_multiscale_cuda eppm;
        for (auto& kv : o.opts)
            if (!eppm.set_option(kv.first.c_str(), kv.second)) return usage();
        auto t0 = std::chrono::steady_clock::now();
        eppm.init(img1.p(), img2.p(), h, w);                             // main.cpp:63-64: the reference's timed window
        if (!eppm.handle()) return 1;
        eppm.compute_flow(ur.data(), vr.data());
        auto t1 = std::chrono::steady_clock::now();
        printf("GPU: %.3f s (init + compute_flow)\n", std::chrono::duration<double>(t1 - t0).count());
    }

    // steady state: contexts created once, pairs streamed through set_data + compute_flow
    {
        std::vector<std::thread> workers;
        std::vector<int> failed(o.gpus, 0);
        std::atomic<int> ready(0);
        std::vector<double> t_begin(o.gpus, 0.0), t_end(o.gpus, 0.0);
        const auto epoch = std::chrono::steady_clock::now();
        auto now = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - epoch).count(); };
        int ndev = 1;
        if (eppm_device_count(&ndev) != EPPM_OK || ndev < 1) ndev = 1;
        for (int g = 0; g < o.gpus; g++)
            workers.emplace_back([&, g]() {
                const int dev = g % ndev;          // more workers than devices: they share (one context each; all read the same pinned images)
                if (o.gpus > 1) eppm_bind_thread_to_device(dev, nullptr, nullptr);     // this worker's host side next to its GPU (NUMA node of the PCIe slot)
                if (o.batch > 1) {       // batch context through the C ABI: B pairs per launch sequence
                    eppm_params prm;
                    eppm_default_params(&prm);
                    for (auto& kv : o.opts) apply_opt(prm, kv.first, kv.second);
                    eppm_ctx* c = nullptr;
                    if (eppm_create_batch(&c, h, w, dev, &prm, o.batch) != EPPM_OK) { failed[g] = 1; ready++; return; }
                    std::vector<std::vector<float>> bu(o.batch, std::vector<float>((size_t)h * w)), bv(o.batch, std::vector<float>((size_t)h * w));
                    std::vector<const uint8_t*> a1(o.batch, img1.store.data()), a2(o.batch, img2.store.data());
                    std::vector<float*> pu(o.batch), pv(o.batch);
                    for (int k = 0; k < o.batch; k++) { pu[k] = bu[k].data(); pv[k] = bv[k].data(); }
                    ready++;
                    while (ready.load() < o.gpus) std::this_thread::yield();
                    t_begin[g] = now();
                    int mine = 0;
                    for (int p = g; p < o.pairs; p += o.gpus) mine++;
                    for (int done = 0; done < mine; done += o.batch) {
                        const int n = (mine - done < o.batch) ? mine - done : o.batch;
                        if (eppm_batch_set_images(c, n, a1.data(), a2.data(), (size_t)w * 3) != EPPM_OK || eppm_batch_compute(c, pu.data(), pv.data()) != EPPM_OK) {
                            failed[g] = 1;
                            eppm_destroy(c);
                            return;
                        }
                        for (int k = 0; k < n; k++)
                            if (bu[k] != u || bv[k] != v) failed[g] = 2;      // every pair of a batch == the single-pair flow
                    }
                    t_end[g] = now();
                    eppm_destroy(c);
                    return;
                }
                bao_flow_patchmatch_multiscale_cuda e;
                e.set_device(dev);
                for (auto& kv : o.opts) e.set_option(kv.first.c_str(), kv.second);
                e.init(h, w);
                if (!e.handle()) { failed[g] = 1; ready++; return; }
                std::vector<float> lu((size_t)h * w), lv((size_t)h * w);
                std::vector<float*> lur(h), lvr(h);
                for (int i = 0; i < h; i++) { lur[i] = &lu[(size_t)i * w]; lvr[i] = &lv[(size_t)i * w]; }
                ready++;
                while (ready.load() < o.gpus) std::this_thread::yield();     // all contexts exist before the window opens
                t_begin[g] = now();
                for (int p = g; p < o.pairs; p += o.gpus) {
                    if (!e.set_data(img1.p(), img2.p())) { failed[g] = 1; return; }
                    e.compute_flow(lur.data(), lvr.data());
                }
                t_end[g] = now();
                if (g == 0 && (lu != u || lv != v)) failed[g] = 2;       // every run of a pair gives the same flow
            });
        for (auto& t : workers) t.join();
        double tb = 1e30, te = 0;
        for (int g = 0; g < o.gpus; g++) { if (t_begin[g] < tb) tb = t_begin[g]; if (t_end[g] > te) te = t_end[g]; }
        const double dt = te - tb;
        for (int g = 0; g < o.gpus; g++)
            if (failed[g]) { fprintf(stderr, "worker %d failed (%s)\n", g, failed[g] == 2 ? "flow differs between runs" : eppm_last_error()); return 1; }
        printf("GPU: %.3f s (%d x (set_data + compute_flow) on %d GPU(s), %d pair(s) per launch, init hoisted): %.2f Mflow-vectors/s\n", dt,
               o.pairs, o.gpus, o.batch, (double)o.pairs * h * w / dt / 1e6);
    }

    if (o.sw > 0) {
        std::vector<float> gu((size_t)h * w, 5.f), gv((size_t)h * w, -3.f);
        float epe = 0, aae = 0;
        eppm_flow_error(u.data(), v.data(), gu.data(), gv.data(), h, w, &epe, &aae);
        printf("EPE %.4f px, AAE %.4f deg against the synthetic translation (+5,-3)\n", epe, aae);
    }
    if (o.gt) {
        int gh = 0, gw = 0;
        std::vector<float> gu((size_t)h * w), gv((size_t)h * w);
        if (eppm_flo_size(o.gt, &gh, &gw) != EPPM_OK || gh != h || gw != w || eppm_load_flo(o.gt, gu.data(), gv.data(), h, w) != EPPM_OK) {
            fprintf(stderr, "cannot read ground truth %s\n", o.gt);
            return 1;
        }
        float epe = 0, aae = 0;
        eppm_flow_error(u.data(), v.data(), gu.data(), gv.data(), h, w, &epe, &aae);
        printf("EPE %.4f px, AAE %.4f deg against %s\n", epe, aae, o.gt);
    }
    printf("Saving flo file...%d*%d\n", h, w);
    if (eppm_save_flo(o.fo, u.data(), v.data(), h, w) != EPPM_OK) { fprintf(stderr, "cannot write %s\n", o.fo); return 1; }
    return 0;
}
