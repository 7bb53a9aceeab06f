// class_tables.cpp -- GPU test driver (tests/test_configs_gpu.py::test_class_reads_through_rewritten_pointer_tables).
// The reference reads its images through the row-pointer tables on every call (bao_rgb2rgba, basic/bao_basic_cuda.h:258-267), so a caller
// may rewrite the pixel pointers of a table between two set_data calls.  The drop-in class reads a bao_alloc-shaped table as one block;
// this driver checks that it notices when a table it has seen before stops describing that block -- with both ends of every row left
// in place, the case only the full pointer walk catches -- in the default mode and with pinned caller buffers, and that the opt-in
// trust cache is what it says.  Prints "OK" or the first failure.
#include <cstdio>
#include <cstring>
#include <vector>

#include "bao_flow_patchmatch_multiscale_cuda.h"

struct Img {
    int h, w;
    std::vector<unsigned char> store;
    std::vector<unsigned char*> cols;
    std::vector<unsigned char**> rows;
    Img(int h_, int w_) : h(h_), w(w_), store((size_t)h_ * w_ * 3), cols((size_t)h_ * w_), rows(h_)
    {
        for (int i = 0; i < h; i++) { rows[i] = &cols[(size_t)i * w]; for (int j = 0; j < w; j++) rows[i][j] = &store[((size_t)i * w + j) * 3]; }
    }
    unsigned char*** p() { return rows.data(); }
};
struct Plane {
    std::vector<float> store;
    std::vector<float*> rows;
    Plane(int h, int w) : store((size_t)h * w), rows(h) { for (int i = 0; i < h; i++) rows[i] = &store[(size_t)i * w]; }
};
static void fill(Img& m, int dx, int dy, unsigned seed)
{
    for (int y = 0; y < m.h; y++)
        for (int x = 0; x < m.w; x++)
            for (int c = 0; c < 3; c++) {
                const int sx = x - dx, sy = y - dy;
                unsigned v = (unsigned)(sx * 7 + sy * 13 + c * 29) * 2654435761u + seed;
                v ^= v >> 15;
                m.store[((size_t)y * m.w + x) * 3 + c] = (unsigned char)(96 + 40 * ((sx / 6 + sy / 5 + c) & 3) + (v & 15));
            }
}

int main(int argc, char** argv)
{
    const int h = 120, w = 160;
    const bool pin = argc > 1 && !strcmp(argv[1], "--pin"), trust = argc > 1 && !strcmp(argv[1], "--trust");
    Img a(h, w), b(h, w), c(h, w);
    fill(a, 0, 0, 1); fill(b, 3, -2, 1); fill(c, -4, 1, 7);
    Plane u0(h, w), v0(h, w), u1(h, w), v1(h, w), ur(h, w), vr(h, w);

    // what the pair (c, b) gives through a fresh object and a pristine table
    {
        bao_flow_patchmatch_multiscale_cuda ref;
        ref.init(c.p(), b.p(), h, w);
        if (!ref.handle()) { printf("no context\n"); return 1; }
        ref.compute_flow(ur.rows.data(), vr.rows.data());
    }
    bao_flow_patchmatch_multiscale_cuda e;
    if (pin) e.set_option("pin_caller_buffers", 1);
    if (trust) e.set_option("trust_verified_tables", 1);
    e.init(h, w);
    if (!e.handle()) { printf("no context\n"); return 1; }
    for (int rep = 0; rep < 2; rep++) {          // twice: the second call sees a table it has verified before
        if (!e.set_data(a.p(), b.p())) { printf("set_data failed\n"); return 1; }
        e.compute_flow(u0.rows.data(), v0.rows.data());
    }
    if (u0.store == ur.store && v0.store == vr.store) { printf("the two pairs give the same flow: the test proves nothing\n"); return 1; }
    // the caller now points the INTERIOR pixels of table a at image c; both ends of every row still point into a's block
    for (int i = 0; i < h; i++)
        for (int j = 1; j < w - 1; j++) a.rows[i][j] = &c.store[((size_t)i * w + j) * 3];
    for (int i = 0; i < h; i++)                  // and image c's border columns into a's block, so that "through the pointers" is image c everywhere
        for (int k = 0; k < 3; k++) { a.store[((size_t)i * w) * 3 + k] = c.store[((size_t)i * w) * 3 + k]; a.store[((size_t)i * w + w - 1) * 3 + k] = c.store[((size_t)i * w + w - 1) * 3 + k]; }
    if (!e.set_data(a.p(), b.p())) { printf("set_data failed\n"); return 1; }
    e.compute_flow(u1.rows.data(), v1.rows.data());
    const bool through_pointers = (u1.store == ur.store && v1.store == vr.store);
    if (trust) {
        // documented: the opt-in trust cache re-checks row ends only, so it reads a's block (whose interior is still image a)
        if (through_pointers) { printf("trust cache did a full walk?\n"); return 1; }
        printf("OK trust\n");
        return 0;
    }
    if (!through_pointers) { printf("the class read the block although the table no longer describes it\n"); return 1; }
    // and the object keeps working with pristine tables afterwards
    if (!e.set_data(c.p(), b.p())) { printf("set_data failed\n"); return 1; }
    e.compute_flow(u1.rows.data(), v1.rows.data());
    if (!(u1.store == ur.store && v1.store == vr.store)) { printf("flow differs after the fallback\n"); return 1; }
    printf("OK\n");
    return 0;
}
