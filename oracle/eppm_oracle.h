/*
 * eppm_oracle.h -- CPU restatement of the EPPM optical-flow hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product (eppm_amd/) never does.
 *
 * It restates, function by function, the live CUDA path of linchaobao/EPPM
 * (bao_pmflow_kernel.cu, bao_pmflow_refine_kernel.cu, bao_pmflow_census_kernel.cu,
 * basic/bao_basic_cuda.cuh, bao_flow_patchmatch_multiscale_cuda.cpp).  Each function
 * cites the reference file:line it follows.
 *
 * PARITY STATUS: "parity unpinned" against the original CUDA binary -- the reference
 * ships no tests, no golden .flo, no CPU path, and it is racy and depends on cuRAND
 * and on the SFU __expf.  What IS pinned: the host-side pieces
 * the reference does define bit-exactly (pyramid dimensions, LUT formulas, .flo/PPM
 * formats, census bit order, KATs derivable by hand), see tests/.
 *
 * Determinism rules (the racy original does not define an order; we define the
 * "lockstep" order a single resident grid would produce):
 *   - every kernel: all threads read their inputs before any thread writes (Jacobi);
 *   - SegPropagate: all segments advance step by step together; segment seeds are
 *     read at step 0; the doubly visited forward pixel 10 is visited by segment 1
 *     (its step 1) before segment 0 (its step 10);
 *   - RandomSearch: all six guesses are generated from the pre-search best;
 *   - __expf / tex2D normalisation: one defined float32 formula (orc_fast_exp, c/255.0f);
 *   - no FMA contraction except the explicit fmaf() inside orc_fast_exp.
 */
#ifndef EPPM_ORACLE_H_
#define EPPM_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint8_t x, y, z, w; } orc_uchar4;
typedef struct { int16_t x, y; } orc_short2;
typedef struct { float x, y; } orc_float2;

/* runtime parameters; defaults = defs.h:31-76 + file-local #defines */
typedef struct {
    int   patch_r;        /* PATCH_R 9            defs.h:44 */
    int   num_iter;       /* NUM_ITER 10          defs.h:45 */
    int   search_range;   /* SEARCH_RANGE 30      defs.h:36 */
    int   num_guess;      /* NUM_RAND_GUESS 6     defs.h:38 */
    int   seg_len;        /* PROP_SEG_LENGTH 10   bao_pmflow_kernel.cu:979 */
    int   wmf_iters;      /* 20, driver :239 */
    unsigned long long seed; /* 1234, bao_pmflow_kernel.cu:68 */
    int   dump_stages;    /* oracle-only: keep intermediate planes */
    int   propagation;    /* 0: baoSegPropagate (live, kernel.cu:1812); 1: baoJumpPropagate (:843-857, disabled there); 2: 10x baoParallelPropagate (:790-795, disabled at :1804-1809) */
    int   levels;         /* PYR_MAX_DEPTH 3        defs.h:31; 1..8, PatchMatch runs at level levels-1 */
} orc_params;

void  orc_default_params(orc_params* p);
/* OTHER legal readings of the racy / unspecified parts of the reference, for measuring how far they move the flow
 * (tools/parity_envelope.py); all zero = the lockstep oracle, the only reading used for parity.  See eppm_oracle.c. */
void  orc_set_variant(int sweep_order, int post_inplace, int exp_mode, int seed_variant);
/* Tolerance-arithmetic variants: the integer-domain table form of the patch term and freer summation orders that
 * libeppm_hip_tol.so uses (tools/tolerance_envelope.py measures their end-point error against the default oracle; (0, 0) = off).
 * mode / scope bits: see eppm_oracle.c.  Never the parity oracle. */
void  orc_set_tol_variant(int mode, int scope);

/* ---- arithmetic building blocks ---- */
float orc_fast_exp(float x);                       /* restates __expf (see .c) */
void  orc_pm_luts(int patch_r, float* gs /*[patch_r+1]*/, float* cn /*[9]*/);   /* kernel.cu:670-687 */
void  orc_wmf_lut(float* g /*[5]*/);               /* refine :270-275 */
void  orc_blf_lut(float* g /*[11]*/);              /* refine :811-816 */

/* ---- XORWOW (cuRAND default generator, restated from the published algorithm) ---- */
typedef struct { uint32_t v[5]; uint32_t d; } orc_xorwow;
void     orc_xorwow_init(orc_xorwow* s, unsigned long long seed, unsigned long long subsequence);
uint32_t orc_xorwow_next(orc_xorwow* s);
/* advance a state by n draws through the GF(2) jump matrix (used to test the product's skip-ahead) */
void     orc_xorwow_skip(orc_xorwow* s, unsigned long long n);

/* ---- pyramid geometry: basic/bao_basic.h:196-211 ---- */
int   orc_pyr_init_dim(int* arrH, int* arrW, int h, int w, int max_depth, float ratio);

/* ---- prepare: refine :1060-1071 ---- */
void  orc_rgb2rgba(orc_uchar4* out, const uint8_t* rgb, int h, int w);  /* bao_basic_cuda.h:258-267 */
void  orc_gauss_filter_rgba(orc_uchar4* out, const orc_uchar4* in, int h, int w, float sigma, int radius); /* .cuh:437-467 */
void  orc_resize_rgba(orc_uchar4* out, int outH, int outW, const orc_uchar4* in, int h, int w, float ratio); /* .cuh:565-601 */
void  orc_census(uint8_t* census, const orc_uchar4* img, int h, int w);   /* census :45-90 */
/* builds img pyramid (levels 0..n-1, tightly packed w*h) and census pyramid from a raw RGBA image */
void  orc_prepare(orc_uchar4** img_pyr, uint8_t** census_pyr, const orc_uchar4* raw, const int* arrH, const int* arrW, int n_levels);

/* ---- PatchMatch: kernel.cu:1760-1826 ---- */
float orc_patch_dist(const orc_uchar4* img1, const orc_uchar4* img2, const uint8_t* c1, const uint8_t* c2,
                     int w, int h, int patch_r, const float* gs, const float* cn,
                     int x1, int y1, int x2, int y2);                     /* :255-301 */
float orc_patch_dist_planefit(const orc_uchar4* img1, const orc_uchar4* img2, const uint8_t* c1, const uint8_t* c2,
                     int w, int h, int patch_r, const float* gs, const float* cn,
                     int x1, int y1, int x2, int y2);                     /* :334-513 */
void  orc_gen_rand_field(orc_xorwow* states, orc_short2* nnf, int w, int h, unsigned long long seed); /* :50-109 */
void  orc_cost_field(float* cost, const orc_short2* nnf, const orc_uchar4* img1, const orc_uchar4* img2,
                     const uint8_t* c1, const uint8_t* c2, int w, int h, const orc_params* p);        /* :636-645 */
/* dir: 0 row fwd, 1 col fwd, 2 row rev, 3 col rev (launch order of :1167-1181) */
void  orc_seg_propagate_dir(float* cost, orc_short2* nnf, const orc_uchar4* img1, const orc_uchar4* img2,
                     const uint8_t* c1, const uint8_t* c2, int w, int h, const orc_params* p, int dir);
/* baoJumpPropagate, kernel.cu:800-857: steps 32,16,8,4,2,1; Jacobi per launch */
void  orc_jump_propagate(float* cost, orc_short2* nnf, const orc_uchar4* img1, const orc_uchar4* img2,
                     const uint8_t* c1, const uint8_t* c2, int w, int h, const orc_params* p);
/* baoParallelPropagate, kernel.cu:720-795: one launch (the disabled call site runs ten per iteration, :1804-1809) */
void  orc_parallel_propagate(float* cost, orc_short2* nnf, const orc_uchar4* img1, const orc_uchar4* img2,
                     const uint8_t* c1, const uint8_t* c2, int w, int h, const orc_params* p);
void  orc_random_search(orc_xorwow* states, float* cost, orc_short2* nnf, const orc_uchar4* img1, const orc_uchar4* img2,
                     const uint8_t* c1, const uint8_t* c2, int w, int h, const orc_params* p);        /* :1519-1586 */
/* iters_done: stop after that many iterations (<0 = p->num_iter); for per-iteration parity */
void  orc_patchmatch(orc_short2* nnf, float* cost, const orc_uchar4* img1, const orc_uchar4* img2,
                     const uint8_t* c1, const uint8_t* c2, int w, int h, const orc_params* p, int iters_done);

/* ---- level-2 post-processing: refine ---- */
void  orc_left_right_check(orc_short2* nnf1, float* cost1, orc_short2* nnf2, float* cost2, int w, int h);   /* :53-92 */
void  orc_outlier_removal(orc_short2* nnf, float* cost, int w, int h);                                       /* :149-193 */
void  orc_weighted_median(orc_short2* nnf, const orc_uchar4* img, int w, int h, int num_iter, int only_occlusion); /* :198-286 */
void  orc_fill_holes(orc_short2* nnf, const orc_uchar4* img, int w, int h);                                   /* :297-390 */
void  orc_nnf2flow(orc_float2* flow, const orc_short2* nnf, int w, int h);                                    /* :636-655 */

/* ---- coarse to fine: refine :1076-1087 ---- */
void  orc_resize_flow(orc_float2* out, int outH, int outW, const orc_float2* in, int h, int w, float ratio);  /* .cuh:511-537 */
void  orc_mul_scalar(orc_float2* f, float s, int h, int w);                                                   /* .cuh:135-142 */
void  orc_c2f_refine(orc_float2* flow, const orc_uchar4* img1, const orc_uchar4* img2, const uint8_t* c1, const uint8_t* c2,
                     int w, int h, const orc_params* p);                                                      /* kernel.cu:2005-2041 */
void  orc_flow_smoothing(orc_float2* flow, const orc_uchar4* img, int w, int h);                              /* refine :764-799 */

/* ---- whole path: driver :159-168 + :217-306.  rgb1/rgb2: h*w*3 bytes; u,v: h*w floats ---- */
typedef struct {
    int n_levels; int arrH[8]; int arrW[8];
    orc_uchar4* img1[8]; orc_uchar4* img2[8]; uint8_t* cen1[8]; uint8_t* cen2[8];
    orc_short2 *nnf1_pm, *nnf2_pm;  float *cost1_pm, *cost2_pm;    /* after PatchMatch */
    orc_short2 *nnf1_lr, *nnf1_out, *nnf1_wmf, *nnf1_fill;          /* after LR / outlier / WMF / fill */
    orc_float2* flow[8];           /* flow pyramid: [2] after nnf2flow, [1],[0] after C2F+smoothing (+final) */
    orc_float2* flow_c2f[8];       /* after the C2F candidate refine, before smoothing */
} orc_dump;
void  orc_free_dump(orc_dump* d);
int   orc_compute_flow(const uint8_t* rgb1, const uint8_t* rgb2, int h, int w, const orc_params* p,
                       float* u, float* v, orc_dump* dump /* may be NULL */);

/* colour coding of a flow field (basic/bao_basic_cuda.cuh:776-845; driver :311 passes 20,20) */
void  orc_flow_to_color(orc_uchar4* rgba, const orc_float2* flow, int h, int w, float max_disp_x, float max_disp_y);
int   orc_num_threads(void);
void  orc_set_num_threads(int n);    /* OpenMP threads of the following calls (bench.py's single-thread CPU baseline) */

#ifdef __cplusplus
}
#endif
#endif
