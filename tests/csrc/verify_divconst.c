/* Exhaustive check (CPU, hardware fmaf) that the 2-operation constant division used by the HIP
 * kernels (eppm_device.cuh: div_const) equals the IEEE quotient the oracle computes, for every float the path can produce:
 *   zh = fl(1/c), zl = fl(1/c - zh) (1/c in double);  q = fmaf(x, zh, x*zl)   ==   x / c
 * Domains: x = 0 and every float in [2^-30, 4] for the range/AD sigmas (the path's smallest non-zero argument is a squared difference
 * of two unorm8 values, > 2^-17); integers 0..255 for unorm8.  Also checks that the other operand order is NOT exact (a guard against
 * a silent swap).
 * Build: gcc -O2 -mfma -ffp-contract=off -fopenmp verify_divconst.c -o verify_divconst */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float divc(float x, float c, float rc_unused)
{
    (void)rc_unused;
    const double rc = 1.0 / (double)c;
    const float zh = (float)rc, zl = (float)(rc - (double)zh);
    return fmaf(x, zh, x * zl);
}
static inline float divc_swapped(float x, float c)
{
    const double rc = 1.0 / (double)c;
    const float zh = (float)rc, zl = (float)(rc - (double)zh);
    return fmaf(x, zl, x * zh);
}

static long check(float c, const char* name)
{
    const float rc = 1.0f / c;
    long bad = 0;
    uint32_t lo, hi;
    float flo = 0x1p-30f, fhi = 4.0f;
    memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (uint32_t u = lo; u <= hi; u++) {
        float x; memcpy(&x, &u, 4);
        float a = divc(x, c, rc), b = x / c;
        if (memcmp(&a, &b, 4) != 0) bad++;
    }
    if (divc(0.0f, c, rc) != 0.0f) bad++;
    {
        float a = divc(-0.3f, c, rc), b = -0.3f / c;          /* negative arguments: the kernels pass -(d*d) */
        if (memcmp(&a, &b, 4) != 0) bad++;
        long swapped_bad = 0;
        for (uint32_t u = lo; u <= lo + (1u << 24); u += 97) { float x; memcpy(&x, &u, 4); float p = divc_swapped(x, c), q = x / c; if (memcmp(&p, &q, 4) != 0) swapped_bad++; }
        if (swapped_bad == 0) { printf("%s: the swapped operand order is unexpectedly exact\n", name); }
    }
    printf("%s c=%a rc=%a mismatches=%ld\n", name, c, rc, bad);
    return bad;
}

int main(void)
{
    long bad = 0;
    bad += check(0.1f * 0.1f, "LAMBDA_AD^2 / PM_SIG_R^2");
    bad += check(0.02f * 0.02f, "WMF_SIG_R^2 / BLF_SIG_R^2");
    const float r255 = 1.0f / 255.0f;
    long bad8 = 0;
    for (int i = 0; i < 256; i++) {
        float a = divc((float)i, 255.0f, r255), b = (float)i / 255.0f;
        if (memcmp(&a, &b, 4) != 0) bad8++;
    }
    printf("unorm8 r255=%a mismatches=%ld\n", r255, bad8);
    return (bad + bad8) ? 1 : 0;
}
