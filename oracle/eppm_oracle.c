/*
 * eppm_oracle.c -- CPU restatement of the EPPM optical-flow hot path (TEST INFRASTRUCTURE ONLY).
 * See eppm_oracle.h for the parity status ("parity unpinned" against the CUDA original)
 * and the determinism rules.  Citations are file:line under /root/reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off is part of the specification: the only fused operations are the
 * explicit fmaf() calls in orc_fast_exp.
 */
#include "eppm_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * constants: defs.h:31-76 and the file-local #defines
 * ---------------------------------------------------------------------------------------- */
#define PYR_RATIO 0.5f                    /* defs.h:33 */
#define PM_SIG_R 0.1f                     /* defs.h:48 */
#define LAMBDA_AD 0.1f                    /* defs.h:51 */
#define LAMBDA_CENSUS 0.3f                /* defs.h:52 */
#define CENSUS_MAX_DIFF 8                 /* bao_pmflow_kernel.cu:32 */
#define WMF_RADIUS 4                      /* defs.h:58 */
#define WMF_SIG_S (WMF_RADIUS * 1.0f)     /* defs.h:59 */
#define WMF_SIG_R 0.02f                   /* defs.h:60 */
#define POSTPROC_BLF_SIG_S 5              /* defs.h:64 */
#define POSTPROC_BLF_SIG_R 0.02f          /* refine :752 */
#define POSTPROC_BLF_RADIUS (2 * POSTPROC_BLF_SIG_S) /* refine :753 */
#define STAT_RADIUS 6                     /* defs.h:68 */
#define STAT_COUNT_THRESH ((2 * STAT_RADIUS + 1) * (2 * STAT_RADIUS + 1) / 2) /* refine :146 */
#define STAT_SIM_THRESH 2                 /* refine :147 */
#define INVALID_LOCATION (-10000)         /* refine :46 */
#define DIFF_THRESH 0                     /* refine :51 */
#define UNKNOWN_FLOW_THRESH 1e9           /* defs.h:85 */
#define UNKNOWN_FLOW 1e10                 /* defs.h:90 */
#define BLOCK_DIM 16                      /* bao_pmflow_kernel.cu:42-43 */

/* plane-fitting coefficients, bao_pmflow_kernel.cu:319-332 */
static const float kPlaneCoef[4][4] = {
    /* u_x,    u_y,     v_x,     v_y */
    {0.0f,    0.0f,    0.0f,    0.0f},
    {0.177f, -0.011f, -0.003f,  0.301f},   /* COEF_FL_*    */
    {0.125f, -0.357f,  0.009f,  0.308f},   /* COEF_LEFT_*  */
    {0.205f,  0.370f,  0.011f,  0.296f},   /* COEF_RIGHT_* */
};

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float fmax2(float a, float b) { return a > b ? a : b; }

void orc_default_params(orc_params* p)
{
    p->patch_r = 9; p->num_iter = 10; p->search_range = 30; p->num_guess = 6;
    p->seg_len = 10; p->wmf_iters = 20; p->seed = 1234ULL; p->dump_stages = 0; p->propagation = 0; p->levels = 3;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n < 1 ? 1 : n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Variants: OTHER legal readings of the racy / unspecified parts of the reference (tools/parity_envelope.py measures how far
 * each moves the flow away from the default reading; they never serve as the parity oracle).  All zero = the lockstep oracle.
 *   sweep_order   0 lockstep (seeds read before any walk, pixel L visited by segment 1 first);
 *                 1 serial: the segments of a line run one after the other in sweep direction and read their seed when they
 *                   start, i.e. propagation along the whole line (a grid whose earlier blocks finish first,
 *                   bao_pmflow_kernel.cu:1059-1076 read nnf[start] live);
 *                 2 lockstep seeds, but pixel L is visited by segment 0 first (the other order of the two racing writes).
 *   post_inplace  bit mask: 1 outlier removal, 2 weighted median, 4 hole filling, 8 flow smoothing run in place in raster order
 *                   (a thread sees the results of every thread before it: refine :149-193, :206-259, :297-371, :764-799 read and
 *                   write one buffer) instead of Jacobi.  Raster order is the far end of what a grid can do; for the outlier
 *                   vote it is degenerate (an invalidated pixel stops supporting its neighbours and the whole field cascades).
 *   exp_mode      1: libm expf (correctly rounded-ish) wherever the reference calls __expf, instead of the shared 2-ulp formula.
 *   seed_variant  1: another seed scrambling (seed ^ golden ratio before cuRAND's constants): a different but equally
 *                   plausible random stream.
 * ---------------------------------------------------------------------------------------- */
static struct { int sweep_order, post_inplace, exp_mode, seed_variant; } g_var = {0, 0, 0, 0};
/* Tolerance-arithmetic variants (tools/tolerance_envelope.py: how far would an integer-domain table form of the patch term move
 * the flow? -- the design study behind libeppm_hip_tol.so; never the parity oracle).
 *   mode  bit 1: 1 - exp(-d^2/s) = TD[kd], exp(-(a^2+b^2)/s) = TA[ka]*TA[kb] with k the integer L-inf distance of the u8 texels
 *         bit 2: cost_sum advances by fmaf(cost, weight, cost_sum)
 *         bit 4: (PatchMatch scope) the canonical order of the tolerance kernels: chunks of half a row (a third at radius 17) summed
 *                from zero, the chunk sums added one after the other (eppm_device.cuh: PatchSum); the refine keeps one chain
 *         bit 8: hardware-exp class: expf() wherever the table form does not apply (smoothing, weighted median)
 *         bit 16: (refine scope only) the weight as ONE hardware exp2 of a summed argument, exp2(log2(gs_j gs_i) - c (ka^2 + kb^2)),
 *                 c = log2(e) / (255^2 s) in float: the form the tolerance library's refine kernel uses (LDS bound otherwise)
 *   scope bit 1: PatchMatch (cost field, sweeps, search)   bit 2: candidate refine   bit 4: smoothing / weighted median weights */
static struct { int mode, scope; } g_tol = {0, 0};
static float g_tol_td[256], g_tol_ta[256], g_tol_tw[256];
void orc_set_tol_variant(int mode, int scope)
{
    g_tol.mode = mode; g_tol.scope = scope;
    const double s = (double)(LAMBDA_AD * LAMBDA_AD);
    for (int k = 0; k < 256; k++) {
        const double d = (double)k / 255.0;
        g_tol_td[k] = (float)(1.0 - exp(-(d * d) / s));
        g_tol_ta[k] = (float)(exp(-(d * d) / s) * 16777216.0);     /* 2^24 per factor: the common scale cancels in cost_sum / weight_sum */
        g_tol_tw[k] = (float)exp(-(d * d) / (double)(WMF_SIG_R * WMF_SIG_R));      /* == POSTPROC_BLF_SIG_R^2 */
    }
}
void orc_set_variant(int sweep_order, int post_inplace, int exp_mode, int seed_variant)
{
    g_var.sweep_order = sweep_order; g_var.post_inplace = post_inplace; g_var.exp_mode = exp_mode; g_var.seed_variant = seed_variant;
}

/* ------------------------------------------------------------------------------------------
 * __expf restated.  CUDA's __expf(x) is ex2.approx(x * log2(e)) (libdevice __nv_fast_expf; the
 * reference is built without -use_fast_math / -ftz, CMakeLists.txt:12-27, so subnormal results
 * are NOT flushed).  The SFU's ex2.approx is not specified bit for bit, so both the oracle and
 * the HIP kernels use this one float32 formula (max rel. error 1.7e-7 for normal results, i.e.
 * the same 2-ulp class as ex2.approx):
 *     y = x * log2e;  n = rint(y);  f = y - n in [-.5,.5];
 *     2^f by a degree-5 polynomial evaluated with fmaf (Horner);  result = ldexpf(p, n),
 * one correctly rounded scaling: gradual underflow below 2^-126, exactly 0 below 2^-150.
 * Used at: bao_pmflow_kernel.cu:285,291; refine :201,759; bao_basic_cuda.cuh:453.
 * ---------------------------------------------------------------------------------------- */
float orc_fast_exp(float x)
{
    if (g_var.exp_mode == 1) return expf(x);     /* variant only, see orc_set_variant */
    float y = x * 0x1.715476p+0f;
    if (y > 127.0f) y = 127.0f;      /* never reached on this path (all arguments are <= 0) */
    if (y < -1000.0f) y = -1000.0f;  /* keeps (int)n defined; the result is 0 from y < -150.5 on */
    const float n = rintf(y);
    const float f = y - n;
    float p = fmaf(0x1.5bba14p-10f, f, 0x1.3cea88p-7f);
    p = fmaf(p, f, 0x1.c6b752p-5f);
    p = fmaf(p, f, 0x1.ebf9bcp-3f);
    p = fmaf(p, f, 0x1.62e42ap-1f);
    p = fmaf(p, f, 1.0f);
    const int ni = (int)n;
    if (ni >= -126) {                /* 2^n is a normal float: one IEEE multiplication (may round into the subnormals) */
        union { uint32_t u; float f; } two_n;
        two_n.u = (uint32_t)(ni + 127) << 23;
        return p * two_n.f;
    }
    return ldexpf(p, ni);
}

/* unorm8 -> float of cudaReadModeNormalizedFloat: c/255 rounded to nearest */
static float g_unorm[256];
static int g_unorm_ready = 0;
static void init_unorm(void)
{
    if (g_unorm_ready) return;
    for (int i = 0; i < 256; i++) g_unorm[i] = (float)i / 255.0f;
    g_unorm_ready = 1;
}

/* LUTs: bao_pmflow_kernel.cu:670-687 */
void orc_pm_luts(int patch_r, float* gs, float* cn)
{
    const float sig_s = 0.5f * patch_r;                 /* PM_SIG_S, defs.h:47 */
    for (int i = 0; i <= patch_r; i++) gs[i] = expf(-(i * i) / (sig_s * sig_s));
    for (int i = 0; i <= CENSUS_MAX_DIFF; i++)
        cn[i] = 1 - expf(-(float)(i * i) / (LAMBDA_CENSUS * CENSUS_MAX_DIFF * LAMBDA_CENSUS * CENSUS_MAX_DIFF));
}
/* refine :270-275 */
void orc_wmf_lut(float* g)
{
    for (int i = 0; i <= WMF_RADIUS; i++) g[i] = expf(-(float)(i * i) / (WMF_SIG_S * WMF_SIG_S));
}
/* refine :811-816 */
void orc_blf_lut(float* g)
{
    for (int i = 0; i <= POSTPROC_BLF_RADIUS; i++)
        g[i] = expf(-(float)(i * i) / (float)(POSTPROC_BLF_SIG_S * POSTPROC_BLF_SIG_S));
}

/* ------------------------------------------------------------------------------------------
 * XORWOW.  cuRAND's default generator (curandState = curandStateXORWOW), used by the
 * reference at bao_pmflow_kernel.cu:68,94-95,1546-1547.  cuRAND is not part of
 * /root/reference (CUDA toolkit "> 5.0", README.md:19, no pinned version); this restates
 * the published algorithm: Marsaglia's xorwow (G. Marsaglia, "Xorshift RNGs", JSS 8(14),
 * 2003, p.5) with cuRAND's documented seeding (seed scrambling, then a skip of
 * subsequence * 2^67 draws, offset 0), cf. curand_kernel.h _curand_init_scratch/curand().
 * The 2^67 skip is done with the GF(2) transition matrix of the xorshift part raised to the
 * 2^67-th power (the Weyl counter d advances by 362437 * 2^67 = 0 mod 2^32).
 * No reference test pins any value of this stream: "RNG stream: parity unpinned".
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t r[160][5]; } mat160;

static inline void xorwow_step_v(uint32_t v[5])
{
    uint32_t t = v[0] ^ (v[0] >> 2);
    v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
    v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

uint32_t orc_xorwow_next(orc_xorwow* s)
{
    xorwow_step_v(s->v);
    s->d += 362437u;
    return s->v[4] + s->d;
}

static void vecmat(const uint32_t v[5], const mat160* m, uint32_t out[5])
{
    uint32_t a[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 32; j++)
            if (v[i] & (1u << j)) {
                const uint32_t* row = m->r[i * 32 + j];
                for (int k = 0; k < 5; k++) a[k] ^= row[k];
            }
    memcpy(out, a, sizeof(a));
}
static void matmul(const mat160* a, const mat160* b, mat160* out)
{
    mat160* t = (mat160*)malloc(sizeof(mat160));
    for (int i = 0; i < 160; i++) vecmat(a->r[i], b, t->r[i]);
    memcpy(out, t, sizeof(mat160));
    free(t);
}
static void mat_step(mat160* m) /* the one-draw transition matrix */
{
    for (int i = 0; i < 160; i++) {
        uint32_t v[5] = {0, 0, 0, 0, 0};
        v[i / 32] = 1u << (i % 32);
        xorwow_step_v(v);
        memcpy(m->r[i], v, sizeof(v));
    }
}

#define SEQ_POW_MAX 40
static mat160* g_seq_pow = NULL;   /* g_seq_pow[k] = M^(2^(67+k)) */
static mat160* g_one_pow = NULL;   /* g_one_pow[k] = M^(2^k), k < 64 */
static void init_jump(void)
{
#pragma omp critical(orc_jump)
    {
        if (!g_seq_pow) {
            mat160* one = (mat160*)malloc(sizeof(mat160) * 64);
            mat160* seq = (mat160*)malloc(sizeof(mat160) * SEQ_POW_MAX);
            mat_step(&one[0]);
            for (int k = 1; k < 64; k++) matmul(&one[k - 1], &one[k - 1], &one[k]);
            mat160 cur;
            matmul(&one[63], &one[63], &cur);                                  /* 2^64 */
            for (int k = 64; k < 67; k++) matmul(&cur, &cur, &cur);            /* 2^67 */
            seq[0] = cur;
            for (int k = 1; k < SEQ_POW_MAX; k++) matmul(&seq[k - 1], &seq[k - 1], &seq[k]);
            g_one_pow = one;
            g_seq_pow = seq;
        }
    }
}

void orc_xorwow_skip(orc_xorwow* s, unsigned long long n)
{
    init_jump();
    s->d += 362437u * (uint32_t)n;
    for (int k = 0; k < 64; k++)
        if (n & (1ULL << k)) vecmat(s->v, &g_one_pow[k], s->v);
}

void orc_xorwow_init(orc_xorwow* s, unsigned long long seed, unsigned long long subsequence)
{
    init_jump();
    if (g_var.seed_variant == 1) seed ^= 0x9E3779B97F4A7C15ULL;     /* variant only, see orc_set_variant */
    /* seed scrambling (curand_kernel.h, _curand_init_scratch) */
    uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    s->d = 6615241u + t1 + t0;
    s->v[0] = 123456789u + t0;
    s->v[1] = 362436069u ^ t0;
    s->v[2] = 521288629u + t1;
    s->v[3] = 88675123u ^ t1;
    s->v[4] = 5783321u + t0;
    /* skipahead_sequence(subsequence): subsequence * 2^67 draws */
    for (int k = 0; k < SEQ_POW_MAX; k++)
        if (subsequence & (1ULL << k)) vecmat(s->v, &g_seq_pow[k], s->v);
}

/* ------------------------------------------------------------------------------------------
 * pyramid geometry: basic/bao_basic.h:196-211 (maxDepth overload)
 * ---------------------------------------------------------------------------------------- */
int orc_pyr_init_dim(int* arrH, int* arrW, int h, int w, int max_depth, float ratio)
{
    if (max_depth == 0) max_depth = 1;
    int n = max_depth;
    if (n <= 0) n = 1;
    arrH[0] = h; arrW[0] = w;
    for (int i = 1; i < n; i++) {
        arrH[i] = (int)((double)h * pow(ratio, i));   /* BAO_FLOAT is double, bao_basic.h:56 */
        arrW[i] = (int)((double)w * pow(ratio, i));
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * prepare
 * ---------------------------------------------------------------------------------------- */
/* bao_basic_cuda.h:258-267 */
void orc_rgb2rgba(orc_uchar4* out, const uint8_t* rgb, int h, int w)
{
    for (int i = 0; i < h * w; i++) {
        out[i].x = rgb[3 * i]; out[i].y = rgb[3 * i + 1]; out[i].z = rgb[3 * i + 2]; out[i].w = 0;
    }
}

/* basic/bao_basic_cuda.cuh:437-467.  Dense (2r+1)^2 Gaussian, clamp-to-edge, weight
 * recomputed per tap with __expf, float accumulation in tap order (dy outer, dx inner),
 * one division per channel, float->u8 by truncation. */
void orc_gauss_filter_rgba(orc_uchar4* out, const orc_uchar4* in, int h, int w, float sigma, int radius_i)
{
    const float radius = (float)radius_i;
    sigma = sigma * sigma * 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float vx = 0, vy = 0, vz = 0, vw = 0, sum = 0;
            for (int dy = (int)-radius; dy <= radius; dy++)
                for (int dx = (int)-radius; dx <= radius; dx++) {
                    int cy = imax(0, imin(h - 1, y + dy));
                    int cx = imax(0, imin(w - 1, x + dx));
                    float weight = orc_fast_exp(-(float)(dy * dy + dx * dx) / sigma);
                    orc_uchar4 t = in[(size_t)cy * w + cx];
                    vx += t.x * weight; vy += t.y * weight; vz += t.z * weight; vw += t.w * weight;
                    sum += weight;
                }
            vx /= sum; vy /= sum; vz /= sum; vw /= sum;
            orc_uchar4 r;
            r.x = (uint8_t)vx; r.y = (uint8_t)vy; r.z = (uint8_t)vz; r.w = (uint8_t)vw;
            out[(size_t)y * w + x] = r;
        }
}

/* basic/bao_basic_cuda.cuh:565-601 (uchar4 specialisation) */
void orc_resize_rgba(orc_uchar4* out, int outH, int outW, const orc_uchar4* in, int h, int w, float ratio)
{
    const float div_scale = 1.f / ratio;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < outH; y++)
        for (int x = 0; x < outW; x++) {
            float fx = (float)(x + 1) * div_scale - 1;
            float fy = (float)(y + 1) * div_scale - 1;
            int xx = (int)fx, yy = (int)fy;
            float dx = fmax2(fminf(fx - xx, 1), 0);
            float dy = fmax2(fminf(fy - yy, 1), 0);
            float rx = 0, ry = 0, rz = 0, rw = 0;
            for (int m = 0; m <= 1; m++)
                for (int n = 0; n <= 1; n++) {
                    int u = imax(0, imin(w - 1, xx + m));
                    int v = imax(0, imin(h - 1, yy + n));
                    float s = fabsf(1 - m - dx) * fabsf(1 - n - dy);
                    orc_uchar4 t = in[(size_t)v * w + u];
                    rx += (t.x * s); ry += (t.y * s); rz += (t.z * s); rw += (t.w * s);
                }
            orc_uchar4 r;
            r.x = (uint8_t)rx; r.y = (uint8_t)ry; r.z = (uint8_t)rz; r.w = (uint8_t)rw;
            out[(size_t)y * outW + x] = r;
        }
}

/* bao_pmflow_census_kernel.cu:39-90: bit k = lum(neigh_k) > lum(centre), clamp addressing */
static inline float lum_of(orc_uchar4 p)
{
    return 0.3f * g_unorm[p.x] + 0.6f * g_unorm[p.y] + 0.1f * g_unorm[p.z];
}
void orc_census(uint8_t* census, const orc_uchar4* img, int h, int w)
{
    init_unorm();
    static const int ox[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
    static const int oy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float c = lum_of(img[(size_t)y * w + x]);
            unsigned r = 0;
            for (int k = 0; k < 8; k++) {
                int cx = iclamp(x + ox[k], 0, w - 1), cy = iclamp(y + oy[k], 0, h - 1);
                if (lum_of(img[(size_t)cy * w + cx]) > c) r += (1u << k);
            }
            census[(size_t)y * w + x] = (uint8_t)r;
        }
}

/* refine :1060-1071 + basic/bao_basic_cuda.cuh:642-664 */
void orc_prepare(orc_uchar4** img_pyr, uint8_t** census_pyr, const orc_uchar4* raw, const int* arrH, const int* arrW, int n_levels)
{
    const float ratio = PYR_RATIO;
    orc_gauss_filter_rgba(img_pyr[0], raw, arrH[0], arrW[0], .5f, 2);        /* refine :1063 */
    /* construct_gauss_pyramid_pitched: pPyr[0] == d_img so no copy (.cuh:646) */
    float baseSigma = (1 / ratio - 1);
    /* .cuh:649 "int n=log(0.25)/log(ratio);" is C++: log(float) resolves to the float overload, and
     * logf(0.5f) = -0.69314718246 is larger in magnitude than ln 2, so the quotient is 1.9999999945 and
     * n = 1 (not 2): level 2 is built from level 1 through the else-branch below. */
    int n = (int)(log(0.25) / (double)logf(ratio));
    float nSigma = baseSigma * n;
    orc_uchar4** tmp = (orc_uchar4**)calloc(n_levels, sizeof(orc_uchar4*));
    for (int i = 0; i < n_levels; i++) tmp[i] = (orc_uchar4*)malloc(sizeof(orc_uchar4) * arrH[i] * arrW[i]);
    for (int i = 1; i < n_levels; i++) {
        if (i <= n) {
            float sigma = baseSigma * i;
            orc_gauss_filter_rgba(tmp[0], img_pyr[0], arrH[0], arrW[0], sigma, (int)(sigma * 3));
            orc_resize_rgba(img_pyr[i], arrH[i], arrW[i], tmp[0], arrH[0], arrW[0], (float)pow(ratio, i));
        } else {
            orc_gauss_filter_rgba(tmp[i - n], img_pyr[i - n], arrH[i - n], arrW[i - n], nSigma, (int)(nSigma * 3));
            orc_resize_rgba(img_pyr[i], arrH[i], arrW[i], tmp[i - n], arrH[i - n], arrW[i - n],
                            (float)pow(ratio, i) * arrW[0] / arrW[i - n]);
        }
    }
    for (int i = 0; i < n_levels; i++) free(tmp[i]);
    free(tmp);
    for (int i = 0; i < n_levels; i++) orc_census(census_pyr[i], img_pyr[i], arrH[i], arrW[i]);  /* refine :1067-1070 */
}

/* ------------------------------------------------------------------------------------------
 * patch cost: bao_pmflow_kernel.cu:255-301.  tex2D = point sample, clamp, u8/255 (A.2).
 * ---------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } rgbf;
static inline rgbf tex_rgb(const orc_uchar4* img, int w, int h, int x, int y)
{
    x = iclamp(x, 0, w - 1); y = iclamp(y, 0, h - 1);
    orc_uchar4 p = img[(size_t)y * w + x];
    rgbf r = {g_unorm[p.x], g_unorm[p.y], g_unorm[p.z]};
    return r;
}
static inline unsigned tex_u8(const uint8_t* c, int w, int h, int x, int y)
{
    x = iclamp(x, 0, w - 1); y = iclamp(y, 0, h - 1);
    return c[(size_t)y * w + x];
}
static inline float max_abs_diff(rgbf a, rgbf b)
{
    return fmax2(fmax2(fabsf(a.x - b.x), fabsf(a.y - b.y)), fabsf(a.z - b.z));
}
static inline int popcount8(unsigned v) { int n = 0; while (v) { n++; v &= v - 1; } return n; }

/* one sample of the inner loop, :275-295; (sx1,sy1) / (sx2,sy2) are the texel coordinates */
static inline void patch_sample(const orc_uchar4* img1, const orc_uchar4* img2, const uint8_t* c1, const uint8_t* c2,
                                int w, int h, rgbf center1, rgbf center2, int sx1, int sy1, int sx2, int sy2,
                                float gs_j, float gs_i, const float* cn, float* cost_sum, float* weight_sum)
{
    rgbf p1 = tex_rgb(img1, w, h, sx1, sy1);
    rgbf p2 = tex_rgb(img2, w, h, sx2, sy2);
    unsigned k1 = tex_u8(c1, w, h, sx1, sy1);
    unsigned k2 = tex_u8(c2, w, h, sx2, sy2);
    int hamming = popcount8(k1 ^ k2);
    float cost = max_abs_diff(p1, p2);
    cost = 1 - orc_fast_exp(-(cost * cost) / (LAMBDA_AD * LAMBDA_AD));
    cost += cn[hamming];
    float weight = max_abs_diff(center1, p1);
    weight *= weight;
    float temp = max_abs_diff(center2, p2);
    temp *= temp;
    weight = orc_fast_exp(-(weight + temp) / (PM_SIG_R * PM_SIG_R));
    weight *= gs_j * gs_i;
    cost *= weight;
    *cost_sum += cost;
    *weight_sum += weight;
}

/* tolerance-arithmetic variant of one sample (orc_set_tol_variant): integer L-inf distances of the u8 texels index two tables */
static inline int iabs_(int v) { return v < 0 ? -v : v; }
static inline int max_abs_diff_u8(orc_uchar4 a, orc_uchar4 b)
{
    return imax(imax(iabs_((int)a.x - (int)b.x), iabs_((int)a.y - (int)b.y)), iabs_((int)a.z - (int)b.z));
}
static inline orc_uchar4 tex_u8x4(const orc_uchar4* img, int w, int h, int x, int y)
{
    x = iclamp(x, 0, w - 1); y = iclamp(y, 0, h - 1);
    return img[(size_t)y * w + x];
}
static int g_tol_exp2_now = 0;
#pragma omp threadprivate(g_tol_exp2_now)
static inline void patch_sample_tol(const orc_uchar4* img1, const orc_uchar4* img2, const uint8_t* c1, const uint8_t* c2,
                                    int w, int h, orc_uchar4 center1, orc_uchar4 center2, int sx1, int sy1, int sx2, int sy2,
                                    float gs_j, float gs_i, const float* cn, float* cost_sum, float* weight_sum)
{
    const orc_uchar4 p1 = tex_u8x4(img1, w, h, sx1, sy1), p2 = tex_u8x4(img2, w, h, sx2, sy2);
    const int hamming = popcount8(tex_u8(c1, w, h, sx1, sy1) ^ tex_u8(c2, w, h, sx2, sy2));
    const float cost = g_tol_td[max_abs_diff_u8(p1, p2)] + cn[hamming];
    float weight;
    if (g_tol_exp2_now) {
        const float c = (float)(1.4426950408889634 / (255.0 * 255.0 * (double)(LAMBDA_AD * LAMBDA_AD)));
        const float ka = (float)max_abs_diff_u8(center1, p1), kb = (float)max_abs_diff_u8(center2, p2);
        const float lsrc = fmaf(-c, ka * ka, log2f(gs_j * gs_i) + 24.0f);         /* once per source sample on the GPU; +24: common scale */
        weight = exp2f(fmaf(-c, kb * kb, lsrc));
        if (weight < 0x1p-126f) weight = 0.0f;                                    /* v_exp_f32 flushes: what vanishes is what the exact formula rounds to 0 */
    } else {
        const float wa = g_tol_ta[max_abs_diff_u8(center1, p1)] * (gs_j * gs_i);      /* the source half, hoisted on the GPU */
        weight = wa * g_tol_ta[max_abs_diff_u8(center2, p2)];
    }
    if (g_tol.mode & 2) *cost_sum = fmaf(cost, weight, *cost_sum); else *cost_sum += cost * weight;
    *weight_sum += weight;
}

float orc_patch_dist(const orc_uchar4* img1, const orc_uchar4* img2, const uint8_t* c1, const uint8_t* c2,
                     int w, int h, int R, const float* gs, const float* cn, int x1, int y1, int x2, int y2)
{
    init_unorm();
    if ((g_tol.mode & 1) && (g_tol.scope & 1)) {
        const orc_uchar4 k1 = tex_u8x4(img1, w, h, x1, y1), k2 = tex_u8x4(img2, w, h, x2, y2);
        float cost_sum = 0.0f, weight_sum = 0.0f;
        /* mode bit 4: the canonical order of the tolerance library's PatchMatch costs (eppm_device.cuh: PatchSum): row-major samples in
         * chunks of half a row (a third at radius 17), each chunk summed from zero, the chunk sums added one after the other */
        const int S = R + 1, CS = (R == 17) ? 6 : (R + 2) / 2;
        for (int i = -R; i <= R; i += 2) {
            float cr = 0.0f, wr = 0.0f;
            int left = CS, jj = 0;
            for (int j = -R; j <= R; j += 2, jj++) {
                float* pc = (g_tol.mode & 4) ? &cr : &cost_sum;
                float* pw = (g_tol.mode & 4) ? &wr : &weight_sum;
                patch_sample_tol(img1, img2, c1, c2, w, h, k1, k2, x1 + j, y1 + i, x2 + j, y2 + i, gs[abs(j)], gs[abs(i)], cn, pc, pw);
                if ((g_tol.mode & 4) && (--left == 0 || jj == S - 1)) { cost_sum += cr; weight_sum += wr; cr = wr = 0.0f; left = CS; }
            }
        }
        return cost_sum / weight_sum;
    }
    rgbf center1 = tex_rgb(img1, w, h, x1, y1);
    rgbf center2 = tex_rgb(img2, w, h, x2, y2);
    float cost_sum = 0.0f, weight_sum = 0.0f;
    for (int i = -R; i <= R; i += 2)        /* "skip pixels": stride-2 grid, :269-272 */
        for (int j = -R; j <= R; j += 2)
            patch_sample(img1, img2, c1, c2, w, h, center1, center2, x1 + j, y1 + i, x2 + j, y2 + i,
                         gs[abs(j)], gs[abs(i)], cn, &cost_sum, &weight_sum);
    return cost_sum / weight_sum;
}

/* bao_pmflow_kernel.cu:334-513: min over 4 affine-warped passes, float target coordinates
 * then floor (point sampling).  Coordinates are formed left to right without contraction. */
float orc_patch_dist_planefit(const orc_uchar4* img1, const orc_uchar4* img2, const uint8_t* c1, const uint8_t* c2,
                              int w, int h, int R, const float* gs, const float* cn, int x1, int y1, int x2, int y2)
{
    init_unorm();
    rgbf center1 = tex_rgb(img1, w, h, x1, y1);
    rgbf center2 = tex_rgb(img2, w, h, x2, y2);
    const float uu = (float)(x2 - x1);
    const float vv = (float)(y2 - y1);
    const int tol = (g_tol.mode & 1) && (g_tol.scope & 2);
    g_tol_exp2_now = tol && (g_tol.mode & 16);
    const orc_uchar4 k1 = tex_u8x4(img1, w, h, x1, y1), k2 = tex_u8x4(img2, w, h, x2, y2);
    float c4[4];
    for (int pass = 0; pass < 4; pass++) {
        const float* cf = kPlaneCoef[pass];
        float cost_sum = 0.0f, weight_sum = 0.0f;
        for (int i = -R; i <= R; i += 2)
            for (int j = -R; j <= R; j += 2) {
                float cx1 = (float)(x1 + j);
                float cy1 = (float)(y1 + i);
                float cx2, cy2;
                if (pass == 0) { cx2 = cx1 + uu; cy2 = cy1 + vv; }
                else {
                    cx2 = cx1 + uu + (j)*cf[0] + (i)*cf[1];
                    cy2 = cy1 + vv + (j)*cf[2] + (i)*cf[3];
                }
                if (tol)          /* the refine's sums advance in sample order, one chain per (pass, candidate) */
                    patch_sample_tol(img1, img2, c1, c2, w, h, k1, k2, (int)floorf(cx1), (int)floorf(cy1),
                                     (int)floorf(cx2), (int)floorf(cy2), gs[abs(j)], gs[abs(i)], cn, &cost_sum, &weight_sum);
                else
                patch_sample(img1, img2, c1, c2, w, h, center1, center2, (int)floorf(cx1), (int)floorf(cy1),
                             (int)floorf(cx2), (int)floorf(cy2), gs[abs(j)], gs[abs(i)], cn, &cost_sum, &weight_sum);
            }
        c4[pass] = cost_sum / weight_sum;
    }
    /* __min(cost1,__min(cost2,__min(cost3,cost4))) :512 with __min(a,b) = (a<b)?a:b
     * (basic/bao_basic_cuda.h:45); the nesting is kept because it decides what a NaN does */
    g_tol_exp2_now = 0;
    float m34 = (c4[2] < c4[3]) ? c4[2] : c4[3];
    float m234 = (c4[1] < m34) ? c4[1] : m34;
    return (c4[0] < m234) ? c4[0] : m234;
}

/* ------------------------------------------------------------------------------------------
 * PatchMatch
 * ---------------------------------------------------------------------------------------- */
/* bao_pmflow_kernel.cu:50-109 */
void orc_gen_rand_field(orc_xorwow* states, orc_short2* nnf, int w, int h, unsigned long long seed)
{
    const int gx = (w + BLOCK_DIM - 1) / BLOCK_DIM, gy = (h + BLOCK_DIM - 1) / BLOCK_DIM;
    for (int by = 0; by < gy; by++)
        for (int bx = 0; bx < gx; bx++) {
            int block_id = by * gx + bx;
            orc_xorwow st;
            orc_xorwow_init(&st, seed, (unsigned long long)block_id);       /* :68 */
            for (int i = 0; i < BLOCK_DIM; i++)
                for (int j = 0; j < BLOCK_DIM; j++) {
                    uint32_t r1 = orc_xorwow_next(&st);
                    uint32_t r2 = orc_xorwow_next(&st);
                    int x = bx * BLOCK_DIM + j, y = by * BLOCK_DIM + i;
                    if (x < w && y < h) {
                        nnf[(size_t)y * w + x].x = (int16_t)(r1 % (uint32_t)(w + 1));   /* :98 */
                        nnf[(size_t)y * w + x].y = (int16_t)(r2 % (uint32_t)(h + 1));   /* :99 */
                    }
                }
            states[block_id] = st;
        }
}

/* :636-645 */
void orc_cost_field(float* cost, const orc_short2* nnf, const orc_uchar4* img1, const orc_uchar4* img2,
                    const uint8_t* c1, const uint8_t* c2, int w, int h, const orc_params* p)
{
    float gs[64], cn[9];
    orc_pm_luts(p->patch_r, gs, cn);
    init_unorm();
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            orc_short2 d = nnf[(size_t)y * w + x];
            cost[(size_t)y * w + x] = orc_patch_dist(img1, img2, c1, c2, w, h, p->patch_r, gs, cn, x, y, d.x, d.y);
        }
}

/* one segment walk.  along = coordinate along the sweep, line = the other coordinate.
 * Forward: :1049-1076 (rows), :1104-1131 (columns).  Reverse: :1078-1102, :1133-1160. */
typedef struct {
    float* cost; orc_short2* nnf; const orc_uchar4 *img1, *img2; const uint8_t *c1, *c2;
    int w, h, R; const float *gs, *cn;
} pm_ctx;

static inline size_t pix_index(const pm_ctx* c, int is_row, int line, int along)
{
    return is_row ? (size_t)line * c->w + along : (size_t)along * c->w + line;
}

static void seg_walk(const pm_ctx* c, int is_row, int reverse, int line, int seg, int L, orc_short2 prev)
{
    const int len = is_row ? c->w : c->h;     /* extent along the sweep */
    int start, end;
    if (!reverse) {
        start = (seg == 0) ? 0 : seg * L - 1;
        end = imin(len - 1, start + L);
        for (int i = start + 1; i <= end; i++) {
            size_t idx = pix_index(c, is_row, line, i);
            float cur_best = c->cost[idx];
            if (is_row) prev.x = (int16_t)imin(prev.x + 1, c->w - 1);
            else        prev.y = (int16_t)imin(prev.y + 1, c->h - 1);
            int px = is_row ? i : line, py = is_row ? line : i;
            float cv = orc_patch_dist(c->img1, c->img2, c->c1, c->c2, c->w, c->h, c->R, c->gs, c->cn, px, py, prev.x, prev.y);
            if (cv < cur_best) { c->nnf[idx] = prev; c->cost[idx] = cv; }
            else prev = c->nnf[idx];
        }
    } else {
        start = (seg + 1) * L;
        if (start >= len) start = len - 1;
        end = seg * L;
        for (int i = start - 1; i >= end; i--) {
            size_t idx = pix_index(c, is_row, line, i);
            float cur_best = c->cost[idx];
            if (is_row) prev.x = (int16_t)imax(prev.x - 1, 0);
            else        prev.y = (int16_t)imax(prev.y - 1, 0);
            int px = is_row ? i : line, py = is_row ? line : i;
            float cv = orc_patch_dist(c->img1, c->img2, c->c1, c->c2, c->w, c->h, c->R, c->gs, c->cn, px, py, prev.x, prev.y);
            if (cv < cur_best) { c->nnf[idx] = prev; c->cost[idx] = cv; }
            else prev = c->nnf[idx];
        }
    }
}

/* One of the four directional kernels of baoSegPropagate (:1167-1181), lockstep order:
 * seeds are read before any walk writes; forward segments are replayed high to low so that
 * segment 1's visit of pixel L precedes segment 0's; reverse segments low to high (their
 * ranges are disjoint and each seed lies in the NEXT segment's range). */
void orc_seg_propagate_dir(float* cost, orc_short2* nnf, const orc_uchar4* img1, const orc_uchar4* img2,
                           const uint8_t* c1, const uint8_t* c2, int w, int h, const orc_params* p, int dir)
{
    float gs[64], cn[9];
    orc_pm_luts(p->patch_r, gs, cn);
    init_unorm();
    pm_ctx c = {cost, nnf, img1, img2, c1, c2, w, h, p->patch_r, gs, cn};
    const int is_row = (dir == 0 || dir == 2), reverse = (dir >= 2), L = p->seg_len;
    const int len = is_row ? w : h, lines = is_row ? h : w;
    const int nseg = (len + L - 1) / L;
#pragma omp parallel for schedule(dynamic, 1)
    for (int line = 0; line < lines; line++) {
        orc_short2* seeds = (orc_short2*)malloc(sizeof(orc_short2) * nseg);
        for (int s = 0; s < nseg; s++) {
            int start;
            if (!reverse) start = (s == 0) ? 0 : s * L - 1;
            else { start = (s + 1) * L; if (start >= len) start = len - 1; }
            seeds[s] = nnf[pix_index(&c, is_row, line, start)];
        }
        if (g_var.sweep_order == 1) {          /* variant: serial along the line, live seeds */
            for (int k = 0; k < nseg; k++) {
                const int s = reverse ? nseg - 1 - k : k;
                int start;
                if (!reverse) start = (s == 0) ? 0 : s * L - 1;
                else { start = (s + 1) * L; if (start >= len) start = len - 1; }
                seg_walk(&c, is_row, reverse, line, s, L, nnf[pix_index(&c, is_row, line, start)]);
            }
        } else if (g_var.sweep_order == 2 && !reverse) {   /* variant: segment 0 reaches pixel L before segment 1 */
            for (int s = 0; s < nseg; s++) seg_walk(&c, is_row, 0, line, s, L, seeds[s]);
        } else
        if (!reverse) for (int s = nseg - 1; s >= 0; s--) seg_walk(&c, is_row, 0, line, s, L, seeds[s]);
        else          for (int s = 0; s < nseg; s++)      seg_walk(&c, is_row, 1, line, s, L, seeds[s]);
        free(seeds);
    }
}

/* d_jump_propagate + baoJumpPropagate, kernel.cu:800-857 (every call site is commented out in the reference,
 * :1813; offered as an option because it has no serial chains).  One launch per step size 32..1: each pixel tries
 * the matches of its four neighbours at distance step, shifted by that distance, in the order left, right, up,
 * down with strict <; candidates outside the image are skipped.  Jacobi per launch. */
void orc_jump_propagate(float* cost, orc_short2* nnf, const orc_uchar4* img1, const orc_uchar4* img2,
                        const uint8_t* c1, const uint8_t* c2, int w, int h, const orc_params* p)
{
    float gs[64], cn[9];
    orc_pm_luts(p->patch_r, gs, cn);
    init_unorm();
    orc_short2* in = (orc_short2*)malloc(sizeof(orc_short2) * w * h);
    for (int step = 32; step >= 1; step /= 2) {
        memcpy(in, nnf, sizeof(orc_short2) * w * h);
#pragma omp parallel for schedule(dynamic, 4)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                orc_short2 best = in[(size_t)y * w + x];
                float best_cost = cost[(size_t)y * w + x];
                orc_short2 nb[4];
                int nx, ny;
                if ((nx = x - step) >= 0) { nb[0] = in[(size_t)y * w + nx]; nb[0].x = (int16_t)(nb[0].x - step); } else { nb[0].x = -999; nb[0].y = -999; }
                if ((nx = x + step) < w)  { nb[1] = in[(size_t)y * w + nx]; nb[1].x = (int16_t)(nb[1].x + step); } else { nb[1].x = -999; nb[1].y = -999; }
                if ((ny = y - step) >= 0) { nb[2] = in[(size_t)ny * w + x]; nb[2].y = (int16_t)(nb[2].y - step); } else { nb[2].x = -999; nb[2].y = -999; }
                if ((ny = y + step) < h)  { nb[3] = in[(size_t)ny * w + x]; nb[3].y = (int16_t)(nb[3].y + step); } else { nb[3].x = -999; nb[3].y = -999; }
                for (int k = 0; k < 4; k++) {
                    orc_short2 d = nb[k];
                    if (d.x < 0 || d.y < 0 || d.x >= w || d.y >= h) continue;
                    float cv = orc_patch_dist(img1, img2, c1, c2, w, h, p->patch_r, gs, cn, x, y, d.x, d.y);
                    if (cv < best_cost) { best = d; best_cost = cv; }
                }
                nnf[(size_t)y * w + x] = best;
                cost[(size_t)y * w + x] = best_cost;
            }
    }
    free(in);
}

/* d_neighbor_propagate + baoParallelPropagate, kernel.cu:720-795 (disabled there: ten launches per iteration,
 * :1804-1809).  Each pixel tries the matches of its upper, lower, left, right neighbours UNSHIFTED (the
 * absolute target is copied, :777-782), in that order with strict <.  Neighbours outside the image are the
 * clamped border pixel (:735-765).  The original loads the rim only from threads on the edge of a full
 * 16x16 block, so in a partial block the right/lower neighbour of the last image column/row is unloaded
 * shared memory; the defined behaviour here is the clamped image neighbour everywhere.  Jacobi per launch. */
void orc_parallel_propagate(float* cost, orc_short2* nnf, const orc_uchar4* img1, const orc_uchar4* img2,
                            const uint8_t* c1, const uint8_t* c2, int w, int h, const orc_params* p)
{
    float gs[64], cn[9];
    orc_pm_luts(p->patch_r, gs, cn);
    init_unorm();
    orc_short2* in = (orc_short2*)malloc(sizeof(orc_short2) * w * h);
    memcpy(in, nnf, sizeof(orc_short2) * w * h);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            orc_short2 best = in[(size_t)y * w + x];
            float best_cost = cost[(size_t)y * w + x];
            const int uy = y > 0 ? y - 1 : 0, ly = y < h - 1 ? y + 1 : h - 1;
            const int lx = x > 0 ? x - 1 : 0, rx = x < w - 1 ? x + 1 : w - 1;
            const orc_short2 nb[4] = { in[(size_t)uy * w + x], in[(size_t)ly * w + x], in[(size_t)y * w + lx], in[(size_t)y * w + rx] };
            for (int k = 0; k < 4; k++) {
                float cv = orc_patch_dist(img1, img2, c1, c2, w, h, p->patch_r, gs, cn, x, y, nb[k].x, nb[k].y);
                if (cv < best_cost) { best = nb[k]; best_cost = cv; }
            }
            nnf[(size_t)y * w + x] = best;
            cost[(size_t)y * w + x] = best_cost;
        }
    free(in);
}

/* :1519-1586 */
void orc_random_search(orc_xorwow* states, float* cost, orc_short2* nnf, const orc_uchar4* img1, const orc_uchar4* img2,
                       const uint8_t* c1, const uint8_t* c2, int w, int h, const orc_params* p)
{
    float gs[64], cn[9];
    orc_pm_luts(p->patch_r, gs, cn);
    init_unorm();
    const int gx = (w + BLOCK_DIM - 1) / BLOCK_DIM, gy = (h + BLOCK_DIM - 1) / BLOCK_DIM;
    const int G = p->num_guess;
#pragma omp parallel for schedule(dynamic, 1)
    for (int block_id = 0; block_id < gx * gy; block_id++) {
        const int by = block_id / gx, bx = block_id % gx;
        orc_xorwow st = states[block_id];
        /* thread (0,0) fills the 16x16 table once per guess, :1540-1554 */
        int16_t (*sr)[BLOCK_DIM * BLOCK_DIM][2] = malloc(sizeof(int16_t) * G * BLOCK_DIM * BLOCK_DIM * 2);
        for (int k = 0; k < G; k++)
            for (int t = 0; t < BLOCK_DIM * BLOCK_DIM; t++) {
                uint32_t r1 = orc_xorwow_next(&st);
                uint32_t r2 = orc_xorwow_next(&st);
                sr[k][t][0] = (int16_t)r1;
                sr[k][t][1] = (int16_t)r2;
            }
        states[block_id] = st;                                                  /* :1567 */
        for (int i = 0; i < BLOCK_DIM; i++)
            for (int j = 0; j < BLOCK_DIM; j++) {
                const int x = bx * BLOCK_DIM + j, y = by * BLOCK_DIM + i;
                if (x >= w || y >= h) continue;
                const size_t idx = (size_t)y * w + x;
                orc_short2 best = nnf[idx];
                float best_cost = cost[idx];
                orc_short2 guess[16];
                int mag = p->search_range;
                for (int k = 0; k < G; k++) {
                    uint32_t rdn1 = (uint32_t)(int32_t)sr[k][i * BLOCK_DIM + j][0];   /* short -> unsigned int, :1558-1559 */
                    uint32_t rdn2 = (uint32_t)(int32_t)sr[k][i * BLOCK_DIM + j][1];
                    int16_t xmin = (int16_t)imax(best.x - mag, 0), xmax = (int16_t)imin(best.x + mag + 1, w + 1);
                    int16_t ymin = (int16_t)imax(best.y - mag, 0), ymax = (int16_t)imin(best.y + mag + 1, h + 1);
                    guess[k].x = (int16_t)((uint32_t)(int32_t)xmin + rdn1 % (uint32_t)(xmax - xmin));
                    guess[k].y = (int16_t)((uint32_t)(int32_t)ymin + rdn2 % (uint32_t)(ymax - ymin));
                    if (mag / 2 >= 1 /* SEARCH_RADIUS_MIN */) mag /= 2;
                }
                for (int k = 0; k < G; k++) {
                    float cv = orc_patch_dist(img1, img2, c1, c2, w, h, p->patch_r, gs, cn, x, y, guess[k].x, guess[k].y);
                    if (cv < best_cost) { best = guess[k]; best_cost = cv; }
                }
                nnf[idx] = best;
                cost[idx] = best_cost;
            }
        free(sr);
    }
}

/* :1760-1826 */
void orc_patchmatch(orc_short2* nnf, float* cost, const orc_uchar4* img1, const orc_uchar4* img2,
                    const uint8_t* c1, const uint8_t* c2, int w, int h, const orc_params* p, int iters_done)
{
    const int gx = (w + BLOCK_DIM - 1) / BLOCK_DIM, gy = (h + BLOCK_DIM - 1) / BLOCK_DIM;
    orc_xorwow* states = (orc_xorwow*)malloc(sizeof(orc_xorwow) * gx * gy);
    orc_gen_rand_field(states, nnf, w, h, p->seed);
    orc_cost_field(cost, nnf, img1, img2, c1, c2, w, h, p);
    const int iters = (iters_done < 0) ? p->num_iter : iters_done;
    for (int it = 0; it < iters; it++) {
        if (p->propagation == 1) orc_jump_propagate(cost, nnf, img1, img2, c1, c2, w, h, p);
        else if (p->propagation == 2) for (int q = 0; q < 10; q++) orc_parallel_propagate(cost, nnf, img1, img2, c1, c2, w, h, p);
        else for (int dir = 0; dir < 4; dir++) orc_seg_propagate_dir(cost, nnf, img1, img2, c1, c2, w, h, p, dir);
        orc_random_search(states, cost, nnf, img1, img2, c1, c2, w, h, p);
    }
    free(states);
}

/* ------------------------------------------------------------------------------------------
 * level-2 post-processing
 * ---------------------------------------------------------------------------------------- */
static void lr_pass(orc_short2* nnf, float* cost, const orc_short2* nnf2, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t idx = (size_t)y * w + x;
            orc_short2 d = nnf[idx];
            int bad;
            if (d.y < 0 || d.y >= h || d.x < 0 || d.x >= w) bad = 1;
            else {
                orc_short2 d2 = nnf2[(size_t)d.y * w + d.x];
                bad = (abs(d2.x - x) > DIFF_THRESH || abs(d2.y - y) > DIFF_THRESH);
            }
            if (bad) { nnf[idx].x = INVALID_LOCATION; nnf[idx].y = INVALID_LOCATION; cost[idx] = FLT_MAX; }
        }
}
/* refine :53-92: two launches; the second sees the first one's invalid marks */
void orc_left_right_check(orc_short2* nnf1, float* cost1, orc_short2* nnf2, float* cost2, int w, int h)
{
    lr_pass(nnf1, cost1, nnf2, w, h);
    lr_pass(nnf2, cost2, nnf1, w, h);
}

/* refine :149-193 (Jacobi) */
void orc_outlier_removal(orc_short2* nnf, float* cost, int w, int h)
{
    orc_short2* in_copy = (orc_short2*)malloc(sizeof(orc_short2) * w * h);
    memcpy(in_copy, nnf, sizeof(orc_short2) * w * h);
    const int inplace = (g_var.post_inplace & 1) != 0;                    /* variant: read the buffer being written, raster order */
    const orc_short2* in = inplace ? nnf : in_copy;
#pragma omp parallel for schedule(static) if (!inplace)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            orc_short2 cur = in[(size_t)y * w + x];
            if (cur.x < 0 && cur.y < 0) continue;
            cur.x = (int16_t)(cur.x - x); cur.y = (int16_t)(cur.y - y);
            int count = 0;
            for (int dy = -STAT_RADIUS; dy <= STAT_RADIUS; dy++)
                for (int dx = -STAT_RADIUS; dx <= STAT_RADIUS; dx++) {
                    int cy = y + dy, cx = x + dx;
                    if (cx < 0 || cy < 0 || cx >= w || cy >= h) continue;
                    orc_short2 nb = in[(size_t)cy * w + cx];
                    nb.x = (int16_t)(nb.x - cx); nb.y = (int16_t)(nb.y - cy);
                    if (abs(nb.x - cur.x) <= STAT_SIM_THRESH && abs(nb.y - cur.y) <= STAT_SIM_THRESH) count++;
                }
            if (count < STAT_COUNT_THRESH) {
                nnf[(size_t)y * w + x].x = INVALID_LOCATION; nnf[(size_t)y * w + x].y = INVALID_LOCATION;
                cost[(size_t)y * w + x] = FLT_MAX;
            }
        }
    free(in_copy);
}

/* refine :198-204 */
static inline float wmf_weight(rgbf a, rgbf b, int dx, int dy, const float* g)
{
    float delta_r = max_abs_diff(a, b);
    float coef_r = orc_fast_exp(-(delta_r * delta_r) / (WMF_SIG_R * WMF_SIG_R));
    float coef_s = g[dx] * g[dy];
    return coef_r * coef_s;
}
/* refine :206-286 (Jacobi per launch) */
void orc_weighted_median(orc_short2* nnf, const orc_uchar4* img, int w, int h, int num_iter, int only_occ)
{
    float g[WMF_RADIUS + 1];
    orc_wmf_lut(g);
    init_unorm();
    orc_short2* in_copy = (orc_short2*)malloc(sizeof(orc_short2) * w * h);
    const int inplace = (g_var.post_inplace & 2) != 0;                    /* variant: read the buffer being written, raster order */
    const orc_short2* in = inplace ? nnf : in_copy;
    for (int it = 0; it < num_iter; it++) {
        memcpy(in_copy, nnf, sizeof(orc_short2) * w * h);
#pragma omp parallel for schedule(dynamic, 2) if (!inplace)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                orc_short2 out = in[(size_t)y * w + x];
                if (only_occ && out.x >= 0 && out.y >= 0) continue;
                rgbf center = tex_rgb(img, w, h, x, y);
                float minCostSum = FLT_MAX;
                for (int dy = -WMF_RADIUS; dy <= WMF_RADIUS; dy++)
                    for (int dx = -WMF_RADIUS; dx <= WMF_RADIUS; dx++) {
                        int cy = y + dy, cx = x + dx;
                        if (cx < 0 || cy < 0 || cx >= w || cy >= h) continue;
                        orc_short2 cand = in[(size_t)cy * w + cx];
                        if (cand.x < 0 || cand.y < 0) continue;
                        cand.x = (int16_t)(cand.x - cx); cand.y = (int16_t)(cand.y - cy);
                        float costSum = 0.0f, weightSum = 0.0f;
                        for (int dy2 = -WMF_RADIUS; dy2 <= WMF_RADIUS; dy2++)
                            for (int dx2 = -WMF_RADIUS; dx2 <= WMF_RADIUS; dx2++) {
                                int cy2 = y + dy2, cx2 = x + dx2;
                                if (cx2 < 0 || cy2 < 0 || cx2 >= w || cy2 >= h) continue;
                                orc_short2 cur = in[(size_t)cy2 * w + cx2];
                                if (cur.x < 0 || cur.y < 0) continue;
                                cur.x = (int16_t)(cur.x - cx2); cur.y = (int16_t)(cur.y - cy2);
                                rgbf pix = tex_rgb(img, w, h, cx2, cy2);
                                float wgt = wmf_weight(center, pix, abs(dx2), abs(dy2), g);
                                costSum += wgt * imax(abs(cand.x - cur.x), abs(cand.y - cur.y));
                                weightSum += wgt;
                            }
                        if (weightSum > 0.0f && costSum < minCostSum) {
                            minCostSum = costSum;
                            out.x = (int16_t)(cand.x + x);
                            out.y = (int16_t)(cand.y + y);
                        }
                    }
                if (out.x < 0 || out.y < 0) continue;
                nnf[(size_t)y * w + x] = out;
            }
    }
    free(in_copy);
}

/* refine :297-371 (Jacobi) */
void orc_fill_holes(orc_short2* nnf, const orc_uchar4* img, int w, int h)
{
    init_unorm();
    orc_short2* in_copy = (orc_short2*)malloc(sizeof(orc_short2) * w * h);
    memcpy(in_copy, nnf, sizeof(orc_short2) * w * h);
    const int inplace = (g_var.post_inplace & 4) != 0;                    /* variant: read the buffer being written, raster order */
    const orc_short2* in = inplace ? nnf : in_copy;
#pragma omp parallel for schedule(static) if (!inplace)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            orc_short2 cur = in[(size_t)y * w + x];
            if (cur.x >= 0 && cur.y >= 0) continue;
            orc_short2 nd[4] = {cur, cur, cur, cur};
            int nx[4] = {x, x, x, x}, ny[4] = {y, y, y, y};
            for (int cx = x - 1; cx >= 0; cx--) { nd[0] = in[(size_t)y * w + cx]; if (nd[0].x >= 0 && nd[0].y >= 0) { nx[0] = cx; ny[0] = y; break; } }
            for (int cx = x + 1; cx < w; cx++)  { nd[1] = in[(size_t)y * w + cx]; if (nd[1].x >= 0 && nd[1].y >= 0) { nx[1] = cx; ny[1] = y; break; } }
            for (int cy = y - 1; cy >= 0; cy--) { nd[2] = in[(size_t)cy * w + x]; if (nd[2].x >= 0 && nd[2].y >= 0) { nx[2] = x; ny[2] = cy; break; } }
            for (int cy = y + 1; cy < h; cy++)  { nd[3] = in[(size_t)cy * w + x]; if (nd[3].x >= 0 && nd[3].y >= 0) { nx[3] = x; ny[3] = cy; break; } }
            rgbf curPix = tex_rgb(img, w, h, x, y);
            float minPixDiff = FLT_MAX;
            for (int i = 0; i < 4; i++) {
                rgbf np = tex_rgb(img, w, h, nx[i], ny[i]);
                float pd = max_abs_diff(curPix, np);
                if (pd < minPixDiff && nd[i].x >= 0 && nd[i].y >= 0) {
                    minPixDiff = pd;
                    cur.x = (int16_t)(nd[i].x - nx[i]);
                    cur.y = (int16_t)(nd[i].y - ny[i]);
                }
            }
            cur.x = (int16_t)(cur.x + x);
            cur.y = (int16_t)(cur.y + y);
            nnf[(size_t)y * w + x] = cur;
        }
    free(in_copy);
}

/* refine :636-655 */
void orc_nnf2flow(orc_float2* flow, const orc_short2* nnf, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            orc_short2 d = nnf[(size_t)y * w + x];
            orc_float2 f;
            if (d.x <= INVALID_LOCATION || d.y <= INVALID_LOCATION) { f.x = (float)UNKNOWN_FLOW; f.y = (float)UNKNOWN_FLOW; }
            else { f.x = (float)(d.x - x); f.y = (float)(d.y - y); }
            flow[(size_t)y * w + x] = f;
        }
}

/* ------------------------------------------------------------------------------------------
 * coarse to fine
 * ---------------------------------------------------------------------------------------- */
/* basic/bao_basic_cuda.cuh:511-537 */
void orc_resize_flow(orc_float2* out, int outH, int outW, const orc_float2* in, int h, int w, float ratio)
{
    const float div_scale = 1.f / ratio;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < outH; y++)
        for (int x = 0; x < outW; x++) {
            float fx = (float)(x + 1) * div_scale - 1;
            float fy = (float)(y + 1) * div_scale - 1;
            int xx = (int)fx, yy = (int)fy;
            float dx = fmax2(fminf(fx - xx, 1), 0);
            float dy = fmax2(fminf(fy - yy, 1), 0);
            float rx = 0, ry = 0;
            for (int m = 0; m <= 1; m++)
                for (int n = 0; n <= 1; n++) {
                    int u = imax(0, imin(w - 1, xx + m));
                    int v = imax(0, imin(h - 1, yy + n));
                    float s = fabsf(1 - m - dx) * fabsf(1 - n - dy);
                    orc_float2 t = in[(size_t)v * w + u];
                    rx += t.x * s; ry += t.y * s;
                }
            out[(size_t)y * outW + x].x = rx;
            out[(size_t)y * outW + x].y = ry;
        }
}

/* basic/bao_basic_cuda.cuh:135-142 */
void orc_mul_scalar(orc_float2* f, float s, int h, int w)
{
    for (size_t i = 0; i < (size_t)h * w; i++) { f[i].x = f[i].x * s; f[i].y = f[i].y * s; }
}

/* float -> short as cvt.rzi.s16.f32 (truncate, saturate) */
static inline int16_t f2short(float f)
{
    if (!(f > -32768.0f)) return (int16_t)-32768;   /* also NaN -> 0 in PTX; NaN never occurs here */
    if (f > 32767.0f) return (int16_t)32767;
    return (int16_t)(int)f;
}

/* bao_pmflow_kernel.cu:2005-2041 */
void orc_c2f_refine(orc_float2* flow, const orc_uchar4* img1, const orc_uchar4* img2, const uint8_t* c1, const uint8_t* c2,
                    int w, int h, const orc_params* p)
{
    float gs[64], cn[9];
    orc_pm_luts(p->patch_r, gs, cn);   /* LUTs left in constant memory by baoComputeCostField */
    init_unorm();
#pragma omp parallel for schedule(dynamic, 2)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            orc_float2 fv = flow[(size_t)y * w + x];
            if (fv.x > UNKNOWN_FLOW_THRESH || fv.y > UNKNOWN_FLOW_THRESH) {
                flow[(size_t)y * w + x].x = 0; flow[(size_t)y * w + x].y = 0;
                continue;
            }
            int16_t cxs[3], cys[3];
            cxs[1] = (int16_t)(f2short(fv.x) + x);
            cys[1] = (int16_t)(f2short(fv.y) + y);
            cxs[0] = (int16_t)(cxs[1] - 1); cys[0] = (int16_t)(cys[1] - 1);
            cxs[2] = (int16_t)(cxs[1] + 1); cys[2] = (int16_t)(cys[1] + 1);
            int bx = cxs[1], by = cys[1];
            float min_cost = 999999;
            for (int m = 0; m < 3; m++)
                for (int n = 0; n < 3; n++) {
                    if (cxs[m] < 0 || cys[n] < 0 || cxs[m] >= w || cys[n] >= h) continue;
                    float cv = orc_patch_dist_planefit(img1, img2, c1, c2, w, h, p->patch_r, gs, cn, x, y, cxs[m], cys[n]);
                    if (cv < min_cost) { min_cost = cv; bx = cxs[m]; by = cys[n]; }
                }
            flow[(size_t)y * w + x].x = (float)(bx - x);
            flow[(size_t)y * w + x].y = (float)(by - y);
        }
}

/* refine :756-799 (Jacobi) */
void orc_flow_smoothing(orc_float2* flow, const orc_uchar4* img, int w, int h)
{
    float g[POSTPROC_BLF_RADIUS + 1];
    orc_blf_lut(g);
    init_unorm();
    orc_float2* in_copy = (orc_float2*)malloc(sizeof(orc_float2) * w * h);
    memcpy(in_copy, flow, sizeof(orc_float2) * w * h);
    const int inplace = (g_var.post_inplace & 8) != 0;                    /* variant: read the buffer being written, raster order */
    const orc_float2* in = inplace ? flow : in_copy;
#pragma omp parallel for schedule(static) if (!inplace)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            rgbf center = tex_rgb(img, w, h, x, y);
            float nx = 0.f, ny = 0.f, wsum = 0.f;
            for (int dy = -POSTPROC_BLF_RADIUS; dy <= POSTPROC_BLF_RADIUS; dy++)
                for (int dx = -POSTPROC_BLF_RADIUS; dx <= POSTPROC_BLF_RADIUS; dx++) {
                    int cy = y + dy, cx = x + dx;
                    if (cx < 0 || cy < 0 || cx >= w || cy >= h) continue;
                    orc_float2 cf = in[(size_t)cy * w + cx];
                    if (cf.x > UNKNOWN_FLOW_THRESH || cf.y > UNKNOWN_FLOW_THRESH) continue;
                    rgbf pix = tex_rgb(img, w, h, cx, cy);
                    float delta_r = max_abs_diff(center, pix);
                    float coef_r = (g_tol.scope & 4) ? g_tol_tw[max_abs_diff_u8(tex_u8x4(img, w, h, x, y), tex_u8x4(img, w, h, cx, cy))]
                                 : orc_fast_exp(-(delta_r * delta_r) / (POSTPROC_BLF_SIG_R * POSTPROC_BLF_SIG_R));
                    float coef_s = g[abs(dx)] * g[abs(dy)];
                    float wgt = coef_r * coef_s;
                    nx += wgt * cf.x; ny += wgt * cf.y; wsum += wgt;
                }
            if (wsum != 0) {
                flow[(size_t)y * w + x].x = nx / wsum;
                flow[(size_t)y * w + x].y = ny / wsum;
            }
        }
    free(in_copy);
}

/* ------------------------------------------------------------------------------------------
 * the whole path: bao_flow_patchmatch_multiscale_cuda.cpp:159-168 (set_data) + :217-306
 * ---------------------------------------------------------------------------------------- */
void orc_free_dump(orc_dump* d)
{
    for (int i = 0; i < 8; i++) {
        free(d->img1[i]); free(d->img2[i]); free(d->cen1[i]); free(d->cen2[i]); free(d->flow[i]); free(d->flow_c2f[i]);
    }
    free(d->nnf1_pm); free(d->nnf2_pm); free(d->cost1_pm); free(d->cost2_pm);
    free(d->nnf1_lr); free(d->nnf1_out); free(d->nnf1_wmf); free(d->nnf1_fill);
    memset(d, 0, sizeof(*d));
}

static void* dup_mem(const void* p, size_t n) { void* q = malloc(n); memcpy(q, p, n); return q; }

int orc_compute_flow(const uint8_t* rgb1, const uint8_t* rgb2, int h, int w, const orc_params* p,
                     float* u, float* v, orc_dump* dump)
{
    const int NL = (p->levels >= 1 && p->levels <= 8) ? p->levels : 3;   /* PYR_MAX_DEPTH, defs.h:31 */
    int arrH[8], arrW[8];
    orc_pyr_init_dim(arrH, arrW, h, w, NL, PYR_RATIO); /* driver :116 */
    orc_uchar4 *raw1 = malloc(sizeof(orc_uchar4) * h * w), *raw2 = malloc(sizeof(orc_uchar4) * h * w);
    orc_rgb2rgba(raw1, rgb1, h, w);                    /* driver :161-162 */
    orc_rgb2rgba(raw2, rgb2, h, w);
    orc_uchar4 *img1[8] = {0}, *img2[8] = {0};
    uint8_t *cen1[8] = {0}, *cen2[8] = {0};
    orc_float2* flow[8] = {0};
    for (int i = 0; i < NL; i++) {
        size_t n = (size_t)arrH[i] * arrW[i];
        img1[i] = malloc(sizeof(orc_uchar4) * n); img2[i] = malloc(sizeof(orc_uchar4) * n);
        cen1[i] = malloc(n); cen2[i] = malloc(n);
        flow[i] = malloc(sizeof(orc_float2) * n);
    }
    orc_prepare(img1, cen1, raw1, arrH, arrW, NL);     /* driver :212-215 */
    orc_prepare(img2, cen2, raw2, arrH, arrW, NL);
    free(raw1); free(raw2);
    if (dump) {
        memset(dump, 0, sizeof(*dump));
        dump->n_levels = NL;
        for (int i = 0; i < NL; i++) {
            size_t n = (size_t)arrH[i] * arrW[i];
            dump->arrH[i] = arrH[i]; dump->arrW[i] = arrW[i];
            dump->img1[i] = dup_mem(img1[i], sizeof(orc_uchar4) * n); dump->img2[i] = dup_mem(img2[i], sizeof(orc_uchar4) * n);
            dump->cen1[i] = dup_mem(cen1[i], n); dump->cen2[i] = dup_mem(cen2[i], n);
        }
    }

    const int L = NL - 1;                              /* pm_layer, driver :219 */
    const int lw = arrW[L], lh = arrH[L];
    const size_t ln = (size_t)lw * lh;
    orc_short2 *nnf1 = malloc(sizeof(orc_short2) * ln), *nnf2 = malloc(sizeof(orc_short2) * ln);
    float *cost1 = malloc(sizeof(float) * ln), *cost2 = malloc(sizeof(float) * ln);
    orc_patchmatch(nnf1, cost1, img1[L], img2[L], cen1[L], cen2[L], lw, lh, p, -1);   /* driver :223 */
    orc_patchmatch(nnf2, cost2, img2[L], img1[L], cen2[L], cen1[L], lw, lh, p, -1);   /* driver :224 */
    if (dump) {
        dump->nnf1_pm = dup_mem(nnf1, sizeof(orc_short2) * ln); dump->nnf2_pm = dup_mem(nnf2, sizeof(orc_short2) * ln);
        dump->cost1_pm = dup_mem(cost1, sizeof(float) * ln); dump->cost2_pm = dup_mem(cost2, sizeof(float) * ln);
    }
    orc_left_right_check(nnf1, cost1, nnf2, cost2, lw, lh);                            /* driver :233 */
    if (dump) dump->nnf1_lr = dup_mem(nnf1, sizeof(orc_short2) * ln);
    orc_outlier_removal(nnf1, cost1, lw, lh);                                          /* driver :237 */
    if (dump) dump->nnf1_out = dup_mem(nnf1, sizeof(orc_short2) * ln);
    orc_weighted_median(nnf1, img1[L], lw, lh, p->wmf_iters, 1);                       /* driver :239 */
    if (dump) dump->nnf1_wmf = dup_mem(nnf1, sizeof(orc_short2) * ln);
    orc_fill_holes(nnf1, img1[L], lw, lh);                                             /* driver :240 */
    if (dump) dump->nnf1_fill = dup_mem(nnf1, sizeof(orc_short2) * ln);
    orc_nnf2flow(flow[L], nnf1, lw, lh);                                               /* driver :258 */
    if (dump) dump->flow[L] = dup_mem(flow[L], sizeof(orc_float2) * ln);

    for (int l = L - 1; l >= 0; l--) {                                                 /* driver :275-282 */
        /* baoCudaBLF_C2F, refine :1076-1087 */
        orc_resize_flow(flow[l], arrH[l], arrW[l], flow[l + 1], arrH[l + 1], arrW[l + 1], 1.f / PYR_RATIO);
        orc_mul_scalar(flow[l], 2.0f, arrH[l], arrW[l]);
        orc_c2f_refine(flow[l], img1[l], img2[l], cen1[l], cen2[l], arrW[l], arrH[l], p);
        if (dump) dump->flow_c2f[l] = dup_mem(flow[l], sizeof(orc_float2) * arrH[l] * arrW[l]);
        orc_flow_smoothing(flow[l], img1[l], arrW[l], arrH[l]);                        /* driver :280 */
        if (dump && l > 0) dump->flow[l] = dup_mem(flow[l], sizeof(orc_float2) * arrH[l] * arrW[l]);
        /* driver :281: WMF on m_disp_vec1_pyramid[l] is dead work, no effect on the flow */
    }
    orc_flow_smoothing(flow[0], img1[0], arrW[0], arrH[0]);                            /* driver :289 */
    if (dump) { free(dump->flow[0]); dump->flow[0] = dup_mem(flow[0], sizeof(orc_float2) * h * w); }   /* levels == 1: replaces the pre-smoothing copy */
    for (size_t i = 0; i < (size_t)h * w; i++) { u[i] = flow[0][i].x; v[i] = flow[0][i].y; }  /* driver :302-306 */

    free(nnf1); free(nnf2); free(cost1); free(cost2);
    for (int i = 0; i < NL; i++) { free(img1[i]); free(img2[i]); free(cen1[i]); free(cen2[i]); free(flow[i]); }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Flow colour coding (next: n4).  basic/bao_basic_cuda.cuh:743-845: colour wheel :751-775 (the table of
 * 3rdparty/middlebury/colorcode.cpp:30-59), _d_bao_compute_flow_color :776-807, the float2 kernel :816-829,
 * launcher :839-845 (max_rad = sqrt(mx*mx + my*my), the float overload); called by the driver at :311 with (20,20).
 * CUDA's atan2f is a third-party routine that is not specified bit for bit: orc_color_atan2 is the restatement the
 * HIP kernel shares (Cephes-style reduction to [0, tan(pi/8)], degree-4 polynomial in t^2, IEEE operations in this
 * order, no contraction).  Against a CUDA build an 8-bit channel may differ by one level where atan2f rounds differently.
 * ---------------------------------------------------------------------------------------- */
static float orc_color_atan2(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    float t = (mx == 0.0f) ? 0.0f : mn / mx;
    float base = 0.0f;
    if (t > 0.4142135679721832275390625f) {
        base = 0.785398185253143310546875f;
        t = (t - 1.0f) / (t + 1.0f);
    }
    const float z = t * t;
    float p = 8.05374449538e-2f * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    float r = base + (p * z * t + t);
    if (ay > ax) r = 1.57079637050628662109375f - r;
    if (signbit(x)) r = 3.1415927410125732421875f - r;
    return copysignf(r, y);
}

static void orc_color_wheel(int wheel[60][3], int* ncols)
{
    const int RY = 15, YG = 6, GC = 4, CB = 11, BM = 13, MR = 6;
    int k = 0, i;
#define SETCOLS(r, g, b) do { wheel[k][0] = (r); wheel[k][1] = (g); wheel[k][2] = (b); k++; } while (0)
    for (i = 0; i < RY; i++) SETCOLS(255, 255 * i / RY, 0);
    for (i = 0; i < YG; i++) SETCOLS(255 - 255 * i / YG, 255, 0);
    for (i = 0; i < GC; i++) SETCOLS(0, 255, 255 * i / GC);
    for (i = 0; i < CB; i++) SETCOLS(0, 255 - 255 * i / CB, 255);
    for (i = 0; i < BM; i++) SETCOLS(255 * i / BM, 0, 255);
    for (i = 0; i < MR; i++) SETCOLS(255, 0, 255 - 255 * i / MR);
#undef SETCOLS
    *ncols = k;
}

/* rgba: h*w {R,G,B,0}; flow: h*w float2 */
void orc_flow_to_color(orc_uchar4* rgba, const orc_float2* flow, int h, int w, float max_disp_x, float max_disp_y)
{
    int wheel[60][3], ncols;
    orc_color_wheel(wheel, &ncols);
    const float max_rad = sqrtf(max_disp_x * max_disp_x + max_disp_y * max_disp_y);         /* .cuh:844 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float vx = flow[y * w + x].x, vy = flow[y * w + x].y;
            orc_uchar4 out = {0, 0, 0, 0};
            if (fabsf(vx) < 999999 && fabsf(vy) < 999999) {                                 /* .cuh:825 */
                const float fx = vx / max_rad, fy = vy / max_rad;
                const float rad = sqrtf(fx * fx + fy * fy);                                  /* __fsqrt_rn, :778 */
                const float a = orc_color_atan2(-fy, -fx) / 3.14159f;
                const float fk = (a + 1.0f) / 2.0f * (float)(ncols - 1);
                const int k0 = (int)fk;
                const int k1 = (k0 + 1) % ncols;
                const float f = fk - (float)k0;
                unsigned char pix[3];
                for (int b = 0; b < 3; b++) {
                    const float col0 = (float)wheel[k0][b] / 255.0f;
                    const float col1 = (float)wheel[k1][b] / 255.0f;
                    float col = (1 - f) * col0 + f * col1;
                    if (rad <= 1) col = 1 - rad * (1 - col);
                    else col = (float)((double)col * .75);                                   /* col *= .75 in double, :795 */
                    pix[b] = (unsigned char)(int)(255.0 * (double)col);                      /* :797 */
                }
                out.x = pix[0]; out.y = pix[1]; out.z = pix[2];                              /* pix.x = pixval[2] = wheel channel 0 (R), :799-803 */
            }
            rgba[y * w + x] = out;
        }
}
