/* Exhaustive check (CPU, hardware fmaf) that the 3-operation constant division used by the HIP
 * kernels equals the IEEE quotient the oracle computes, for every float the path can produce:
 *   q0 = x*rc; r = fmaf(-c, q0, x); q = fmaf(r, rc, q0)   ==   x / c
 * Domains: x = 0 and every float in [2^-24, 4] for the range/AD sigmas; integers 0..255 for unorm8.
 * Build: gcc -O2 -mfma -ffp-contract=off -fopenmp verify_divconst.c -o verify_divconst */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float divc(float x, float c, float rc)
{
    float q0 = x * rc;
    float r = fmaf(-c, q0, x);
    return fmaf(r, rc, q0);
}

static long check(float c, const char* name)
{
    const float rc = 1.0f / c;
    long bad = 0;
    uint32_t lo, hi;
    float flo = 0x1p-24f, fhi = 4.0f;
    memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (uint32_t u = lo; u <= hi; u++) {
        float x; memcpy(&x, &u, 4);
        float a = divc(x, c, rc), b = x / c;
        if (memcmp(&a, &b, 4) != 0) bad++;
    }
    if (divc(0.0f, c, rc) != 0.0f) bad++;
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
