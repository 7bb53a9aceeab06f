// eppm_internal.h -- host-side declarations of the kernel launch wrappers (one per stage).
// All pitches here are in ELEMENTS of the plane's type unless the name says bytes.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/eppm.h"

namespace eppm {

// geometry constants shared by host and device code
constexpr int kMaxLevels = 8;
constexpr int kNumLevels = 3;        // PYR_MAX_DEPTH, defs.h:31
constexpr int kBlock = 16;       // BLOCK_DIM_X/Y, bao_pmflow_kernel.cu:42-43
constexpr int kMaxS = 32;        // samples per patch row: patch_r + 1 <= 32
constexpr int kWmfRadius = 4;    // defs.h:58
constexpr int kBlfRadius = 10;   // 2*POSTPROC_BLF_SIG_S, refine :753
constexpr int kDeltaSlots = 800; // entries of eppm_device.cuh: DeltaTab::t2 (799 are used: api_common.cpp delta_index)

// Batch of independent pairs processed by ONE launch: every device plane of pair k lives at the same offset inside
// pair k's slab, and the slabs are `stride` bytes apart, so a kernel finds pair k's planes by adding k*stride to the
// pointers it was given for pair 0 (blockIdx.z, or .y for 1-D grids, selects the pair).  {1, 0} = a single pair.
struct Batch {
    int n;
    size_t stride;
};
constexpr Batch kOnePair = {1, 0};

struct PlanesH {            // host-side mirror of eppm::Planes: float4 texel planes {r,g,b,census bits}, pitch in pixels
    const void* pk1;
    const void* pk2;
    int w, h, pitch;
    // optional (NULL: absent): the same texels in 4 bytes, {R, G, B, census} as one word per pixel, same pitch.  Kernels that stage
    // a whole tile / window in LDS read these and convert while storing (a quarter of the HBM and L2 bytes for the same values)
    const uint32_t* pc1 = nullptr;
    const uint32_t* pc2 = nullptr;
    // optional, tolerance library only (NULL: absent): COLUMN-PARITY planes of the 4-byte texels at the PatchMatch level.  The patch samples
    // every other column, so the S samples of a patch row are S CONSECUTIVE words of the plane of their column parity: 3 gathers per row of 10
    // samples instead of 10 (the kernels that give a lane a whole evaluation run at the L1's lane rate).  Layout: word [p][y][i] = texel
    // (clamp(2i + p - pp_pad, 0, w - 1), y) -- replicate padding = the clamp addressing of the texture model; pp_pitch words per row.
    const uint32_t* pp1 = nullptr;
    const uint32_t* pp2 = nullptr;
    int pp_pitch = 0, pp_pad = 0;
};
inline int parity_pitch(int w, int pad) { return (w + 2 * pad + 1) / 2 + 1; }      // words per row of one parity plane
// the two parity planes of one image (k_prepare.hip); pc = its 4-byte texel plane, pitch pc_pitch words
void launch_parity_planes(uint32_t* pp, int pp_pitch, int pad, const uint32_t* pc, int pc_pitch, int w, int h, hipStream_t s, Batch bt = kOnePair);

// ---- prepare (k_prepare.hip) ----
void launch_gauss_rgba(uint32_t* out, const uint32_t* in, int pitch_px, int h, int w, float sigma, int radius, hipStream_t s, Batch bt = kOnePair);
// blur + exact 2:1 decimation in one kernel (only the kept pixels are blurred); use when gauss_decimate2_ok()
bool gauss_decimate2_ok(int outH, int outW, int h, int w, float ratio, int radius);
void launch_gauss_decimate2(uint32_t* out0, const uint32_t* in0, uint32_t* out1, const uint32_t* in1, int nimg, int out_pitch_px, int outH,
                            int outW, int pitch_px, int h, int w, float sigma, int radius, hipStream_t s, Batch bt = kOnePair);
// the same blur on two images of equal geometry in one launch
void launch_gauss_rgba2(uint32_t* out0, const uint32_t* in0, uint32_t* out1, const uint32_t* in1, int pitch_px, int h, int w, float sigma,
                        int radius, hipStream_t s, Batch bt = kOnePair);
// census (+ texel plane) of several planes in one launch
struct CensusJob { uint8_t* census; int cpitch; void* texels; int tpitch; const uint32_t* img; int ipitch; int w, h; int first_block;
                   uint32_t* packed = nullptr; };   // optional 4-byte texel plane {R,G,B,census}, pitch tpitch
struct CensusBatch { int n; CensusJob job[2 * kMaxLevels]; };
void launch_census_batch(CensusBatch& B, hipStream_t s, Batch bt = kOnePair);
void launch_resize_rgba(uint32_t* out, int out_pitch_px, int outH, int outW, const uint32_t* in, int in_pitch_px, int h, int w,
                        float ratio, hipStream_t s, Batch bt = kOnePair);
// census plane and (optionally, texels != NULL) the float4 texel plane the patch kernels read
void launch_census(uint8_t* census, int cpitch, void* texels, int tpitch, const uint32_t* img, int ipitch, int w, int h, hipStream_t s);
void launch_pack(void* texels, int tpitch, const uint32_t* img, int ipitch, const uint8_t* census, int cpitch, int w, int h, hipStream_t s);
// DeltaTab values (eppm_device.cuh): t2[i] = f(t2[i]) on the device, which = 0 patch data term, 1 smoothing / weighted-median weight
void launch_delta_values(float* t2, int n, int which, hipStream_t s);
void launch_rgb_to_rgba(uint32_t* out, int pitch_px, const uint8_t* rgb, int h, int w, hipStream_t s, Batch bt = kOnePair);

// ---- PatchMatch (k_patchmatch.hip) ----
// One PatchMatch problem = (source planes, target planes, NNF, cost).  The forward (1->2) and backward (2->1)
// problems of a pair have the same size and run in the same launches (blockIdx.z / blockIdx.y selects one).
struct PmProblem {
    PlanesH P;
    float* cost;
    int16_t* nnf;        // short2, current
    int16_t* nnf_alt;    // short2, ping-pong partner for the sweeps
    // Evaluation cache of the sweeps, one (candidate, cost) entry per pixel and sweep direction: four planes each (cost pitch),
    // direction d at element offset d * PmBatch::cache_plane.  The patch cost is a pure function of (pixel, candidate) while the
    // images stand, so a sweep that meets the candidate it evaluated for this pixel last time -- the neighbour's match did not
    // change between two iterations, the normal case once the field has converged -- takes the cost from here.  spec doubles as
    // phase A's hand-over plane to phase B (k_patchmatch.hip).  scand == NULL: no cache (every candidate is evaluated).
    float* spec = nullptr;
    int32_t* scand = nullptr;   // x | y << 16 of the cached candidate; -1 = empty (no candidate has both coordinates -1)
    // Work list of the speculative sweeps (k_patchmatch.hip): phase A names the chains on which some candidate of the rejection path
    // would be ACCEPTED -- every other chain leaves its pixels as they are, and phase B walks the listed chains only.  Layout:
    // word 0, 1: list lengths of the even / odd sweeps of a run (ping-pong: a sweep's phase A clears the other one);
    // [16, 16 + units): stamp per unit (two adjacent segments of a line) = 1 + number of the last sweep that listed it;
    // [16 + units, 16 + 2 * units): the list.  NULL: phase B walks every chain.
    // Merged form (one phase A for the four sweeps of an iteration): words 8 + 4 * (iteration & 1) + d: list length of direction d;
    // [16 + (2 + d) * units, ..): stamps of direction d = 1 + the iteration that listed the unit last; [16 + (6 + d) * units, ..): its list.
    uint32_t* wl = nullptr;
    // Merged form of the speculative sweeps (k_patchmatch.hip, k_pm_spec_all): the field as it stood before each direction's sweep of the
    // current iteration, four short2 planes (direction d at int16 offset d * PmBatch::seed_plane) -- where the in-place sweeps read their seeds.
    int16_t* seed = nullptr;
    uint32_t* rng_work;       // [nblocks][64][6] XORWOW lane states read by the random search
    uint32_t* rng_work_next;  // ... written by it (ping-pong: four workgroups read each block's state, one advances it)
};
struct PmBatch {
    PmProblem p[2];      // the problems of pair 0; pair k's are `stride` bytes further (every pointer of PmProblem)
    int n;               // problems per pair: 1 or 2
    int cpitch, npitch;  // elements
    int npairs = 1;      // a launch covers n * npairs problems
    size_t stride = 0;
    size_t cache_plane = 0;   // elements per direction plane of PmProblem::spec / scand
    int wl_units = 0;         // capacity of PmProblem::wl (pm_worklist_units)
    int sweep_seq = 0;        // number of the next sweep of this PatchMatch run (0, 1, ..): launch_pm_sweep counts
    size_t seed_plane = 0;    // int16 elements per direction plane of PmProblem::seed
    int merged_it = 0;        // merged form: number of the iteration (list parity, stamps)
    int seg_len = 0, nseg_row = 0, nseg_col = 0;   // merged form: the sweeps' geometry (every kernel lists for every direction)
};
// units (pairs of adjacent segments of a line) of the larger of the row and column sweeps; words of PmProblem::wl
int pm_worklist_units(int w, int h, int seg_len);
// words of PmProblem::wl: 16 counters | stamps, list of the two-launch form (units each) | stamps, lists of the merged form (4 x units each)
inline size_t pm_worklist_words(int w, int h, int seg_len) { return 16 + 10 * (size_t)pm_worklist_units(w, h, seg_len); }
// RNG tables shared by both problems (same seed, same block ids: the reference re-initialises the states on
// every baoCudaPatchMatch call, kernel.cu:160); see xorwow_host.cpp
struct PmRngDev {
    const uint32_t* init_tab;    // [nblocks][64][6] lane l at draw 8*l              (init field)
    const uint32_t* iter_tab;    // [nblocks][64][6] lane l at draw 512 + per_lane*l (first search)
    const uint32_t* skip_mat;    // [160][5] GF(2) matrix: advance by (512*G - per_lane) draws
    uint32_t skip_weyl;          // 362437 * (512*G - per_lane)
    int per_lane;                // draws per lane per search = 512*G/64
    int gx, gy;
    // The numbers of search launch `it` of a run, drawn ahead: [block][G][512] int16 (the shorts the search stores in LDS); NULL: the
    // search draws them itself from rng_work.  The block streams depend on (seed, geometry, num_guess) only -- the reference re-seeds on
    // every call (kernel.cu:68, :160) -- so the numbers every PatchMatch run of a geometry draws are the same constants.
    const int16_t* rand_tab = nullptr;
};
// fills tab[block][512*G] with the draws of ONE search launch from the lane states in `work` and advances `work` (in place) to the
// next launch's position: exactly what wave G of k_pm_random_search does
void launch_pm_rand_table(const PmRngDev& rng, uint32_t* work, int16_t* tab, int G, hipStream_t s);
void launch_pm_init_field(const PmBatch& b, const PmRngDev& rng, hipStream_t s);
void launch_pm_cost_field(const PmBatch& b, const float* lut, int R, hipStream_t s);
// one directional sweep; returns true when the result is in nnf_alt (caller swaps nnf/nnf_alt)
// speculative: the two-launch form for iterations in which few candidates are accepted (k_patchmatch.hip, k_pm_sweep_spec); same results
bool launch_pm_sweep(PmBatch& b, const float* lut, int R, int seg_len, int dir, hipStream_t s, bool speculative = false);
// The four sweeps of one iteration in the merged speculative form (k_pm_spec_all + four in-place launches over the listed chains): same
// results, in place in nnf.  Returns false (nothing launched) when the problems lack the planes or the radius has no instantiation.
bool launch_pm_sweeps_merged(PmBatch& b, const float* lut, int R, int seg_len, int iteration, hipStream_t s);
// one jump-flood launch (step = neighbour distance); reads nnf, writes nnf_alt (caller swaps)
void launch_pm_jump(const PmBatch& b, const float* lut, int R, int step, hipStream_t s);
// one 4-neighbour propagation launch (d_neighbor_propagate); reads nnf, writes nnf_alt (caller swaps)
void launch_pm_neighbor(const PmBatch& b, const float* lut, int R, hipStream_t s);
void launch_pm_random_search(const PmBatch& b, const PmRngDev& rng, const float* lut, int R, int search_range, int num_guess,
                             hipStream_t s);

// ---- level-2 post-processing (k_post.hip) ----
void launch_lr_check(int16_t* nnf1, float* cost1, const int16_t* nnf2, int w, int h, int cost_pitch, int nnf_pitch, hipStream_t s, Batch bt = kOnePair);
void launch_outlier(int16_t* nnf_out, float* cost, const int16_t* nnf_in, int w, int h, int cost_pitch, int nnf_pitch, hipStream_t s, Batch bt = kOnePair);
// all num_iter Jacobi launches; ping-pongs buf_a (input) / buf_b, ws = 2*w*h + num_iter + 2 uint32 words; returns the result buffer
size_t wmf_workspace_words(int w, int h, int num_iter);
int16_t* launch_wmf(int16_t* buf_a, int16_t* buf_b, const uint32_t* img, int ipitch, int w, int h, int nnf_pitch,
                    const float* wmf_lut, int num_iter, int only_occlusion, uint32_t* ws, hipStream_t s, Batch bt = kOnePair);
void launch_fill_holes(int16_t* nnf_out, const int16_t* nnf_in, const uint32_t* img, int ipitch, int w, int h, int nnf_pitch,
                       hipStream_t s, Batch bt = kOnePair);
void launch_nnf2flow(float* flow, int flow_pitch, const int16_t* nnf, int nnf_pitch, int w, int h, hipStream_t s, Batch bt = kOnePair);

// ---- coarse to fine (k_c2f.hip) ----
void launch_resize_flow(float* out, int outH, int outW, const float* in, int h, int w, float ratio, float post_scale, hipStream_t s, Batch bt = kOnePair);
void launch_mul_scalar(float* flow, float scale, int h, int w, hipStream_t s);
bool c2f_refine_wants_split(int w, int h, int R, int npairs = 1, bool no_split = false);
bool c2f_window_span(int R, int* span_x, int* span_y);          // test support: admissible centre spread of the LDS-window kernels
// cost9: scratch of 36 floats per pixel for launches that c2f_refine_wants_split(), or NULL
void launch_c2f_refine(const PlanesH& P, float* flow, const float* lut, int R, float* cost9, hipStream_t s, Batch bt = kOnePair, bool no_split = false);
void launch_flow_blf(float* out, const float* in, const uint32_t* img, int ipitch, int w, int h, int flow_pitch,
                     const float* blf_lut, hipStream_t s, Batch bt = kOnePair);

// interleaved float2 flow -> planar u | v (2*n floats) on the device: compute_flow's de-interleave, driver :302-306
void launch_split_flow(float* uv, const float* flow, int n, hipStream_t s, Batch bt = kOnePair);

// ---- flow colour coding (k_color.hip) ----
// rgba: h*w packed R | G<<8 | B<<16 (alpha 0); flow: h*w float2
void launch_flow_to_color(uint32_t* rgba, const float* flow, int h, int w, float max_disp_x, float max_disp_y, hipStream_t s, Batch bt = kOnePair);

// ---- probes (k_prepare.hip) ----
void launch_probe_delta(const float* x, float* y, int n, const float* delta_tab, hipStream_t s);
void launch_probe(const float* x, float* y, int n, int which, hipStream_t s);

// ---- host XORWOW (xorwow_host.cpp) ----
struct XorwowState { uint32_t v[5]; uint32_t d; };
void xorwow_init(XorwowState* s, unsigned long long seed, unsigned long long subsequence);
uint32_t xorwow_next(XorwowState* s);
// out[160*5]: row-vector GF(2) matrix advancing the xorshift words by n draws
void xorwow_skip_matrix(unsigned long long n, uint32_t* out);

}  // namespace eppm
