/* Runs the whole oracle path on a small random pair under AddressSanitizer + UBSan (CPU only; GPU sanitizers are
 * not available on this pool).  Build: gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined
 *   -ffp-contract=off -mavx2 -mfma -I oracle tests/csrc/oracle_sanitize_driver.c oracle/eppm_oracle.c -lm */
#include <stdio.h>
#include <stdlib.h>
#include "eppm_oracle.h"

int main(void)
{
    const int h = 52, w = 70;
    uint8_t* a = malloc((size_t)h * w * 3);
    uint8_t* b = malloc((size_t)h * w * 3);
    unsigned s = 12345;
    for (int i = 0; i < h * w * 3; i++) { s = s * 1664525u + 1013904223u; a[i] = s >> 24; }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) b[(y * w + x) * 3 + c] = a[(y * w + (x + 2 < w ? x + 2 : w - 1)) * 3 + c];
    float* u = malloc(sizeof(float) * h * w);
    float* v = malloc(sizeof(float) * h * w);
    orc_params p;
    for (int mode = 0; mode < 3; mode++) {
        orc_default_params(&p);
        p.propagation = mode;
        p.num_iter = 3;
        orc_dump d;
        if (orc_compute_flow(a, b, h, w, &p, u, v, &d) != 0) return 2;
        orc_free_dump(&d);
    }
    for (int levels = 1; levels <= 4; levels++) {          /* PYR_MAX_DEPTH as a run-time parameter */
        orc_default_params(&p);
        p.levels = levels;
        p.num_iter = 2;
        orc_dump d;
        if (orc_compute_flow(a, b, h, w, &p, u, v, &d) != 0) return 3;
        if (d.n_levels != levels) return 4;
        orc_free_dump(&d);
    }
    double su = 0;
    for (int i = 0; i < h * w; i++) su += u[i];
    printf("ok mean u %.4f\n", su / (h * w));
    free(a); free(b); free(u); free(v);
    return 0;
}
