// pm_driver.cpp -- baoCudaPatchMatch's host loop (bao_pmflow_kernel.cu:1760-1826): random field, cost field, then num_iter x [four
// segmented sweeps in the form the iteration calls for + random search]; shared by the contexts (context.cpp) and the
// reference-signature launcher (launchers_ref_abi.cpp).
#include "api_internal.h"

using namespace eppm;

// ---- baoCudaPatchMatch (kernel.cu:1760-1826) for one problem or for the forward+backward pair at once ----
PmProblem mk_problem(const PlanesH& P, float* cost, int16_t* nnf, int16_t* nnf_alt, eppm_pm_rng* rng, int k, float* spec, int32_t* scand, uint32_t* wl, int16_t* seed)
{
    PmProblem p;
    p.P = P; p.cost = cost; p.nnf = nnf; p.nnf_alt = nnf_alt; p.spec = spec; p.scand = scand; p.wl = wl; p.seed = seed;
    p.rng_work = rng ? rng->work[k][rng->cur[k]] : nullptr;
    p.rng_work_next = rng ? rng->work[k][rng->cur[k] ^ 1] : nullptr;
    return p;
}
// one random search on the batch; afterwards the advanced RNG states are the current ones
void search(PmBatch& b, eppm_pm_rng* rng, const float* lut, const eppm_params& prm, hipStream_t s, int launch_no)
{
    PmRngDev d = rng->dev();
    if (rng->rand_tab && launch_no >= 0) d.rand_tab = rng->rand_tab + (size_t)launch_no * rng->rand_stride;
    launch_pm_random_search(b, d, lut, prm.patch_r, prm.search_range, prm.num_guess, s);
    for (int k = 0; k < b.n; k++) {
        std::swap(b.p[k].rng_work, b.p[k].rng_work_next);
        rng->cur[k] ^= 1;
    }
}
// The sweeps of an iteration run in the speculative two-launch form (k_patchmatch.hip: k_pm_sweep_spec + phase B) once most
// candidates are rejected: in the third iteration (index 2) one step in eight to one in four still follows an accepted
// candidate, from the fourth on fewer than one in ten (tools/sweep_stats.py), and a step that follows a rejection needs no
// dependent evaluation.  Same results either way; from iteration 2 / 3 measured equal within 0.5 %, from 0 or 1 slower.
#ifndef EPPM_SPEC_FROM_ITER
#define EPPM_SPEC_FROM_ITER 2
#endif
#ifndef EPPM_SWEEP_LIST
#define EPPM_SWEEP_LIST 1          // work list of the speculative sweeps: phase B walks only the chains phase A found an accepted candidate on
#endif
bool sweep_list_on(int mode) { return EPPM_SWEEP_LIST && mode != 2; }
// A launch over one 1024x436 pair (two problems of 28 k pixels) is too small for the two-launch form to pay: phase A's evaluations
// are one wave per SIMD, and the classic kernel at 32 lanes per chain finishes in 27 us where phase A + phase B take 19 + 16.  From
// about a hundred thousand pixels per launch on (two such pairs; one 1920x1080 or 3840x2160 pair) the speculative form wins.
#ifndef EPPM_SPEC_MIN_PIXELS
#define EPPM_SPEC_MIN_PIXELS 100000
#endif
static bool sweep_speculative(int iteration, long long pixels, int m)
{
    return m < 0 ? (iteration >= EPPM_SPEC_FROM_ITER && pixels >= EPPM_SPEC_MIN_PIXELS) : m != 0;
}
// From this iteration on the four speculative sweeps share ONE phase A (k_patchmatch.hip, k_pm_spec_all: the merged form): the field has
// converged far enough that a phase-A launch costs its launch, and four of them per iteration are three too many.  The threshold depends
// on the size of a PROBLEM, not of the launch (round 5, A/B within one lease, profiles/r05x_c_merged_threshold_by_size.txt): the merged
// phase A touches every pixel for four directions at once, and on a 480x270 or 960x540 problem that pays two iterations later than on
// a 256x109 one -- 1920x1080: 187.4-188.0 Mflow-vectors/s from the eighth iteration against 186.4-187.0 from the sixth, 3840x2160 R = 17:
// 139.9-140.8 ms against 141.0-141.9; eight 1024x436 pairs per launch: from the fifth to the eighth equal within the noise.  Mode 3 of
// the test switch forces the merged form from the first iteration.
#ifndef EPPM_MERGED_FROM_ITER
#define EPPM_MERGED_FROM_ITER 5
#endif
#ifndef EPPM_MERGED_FROM_ITER_LARGE
#define EPPM_MERGED_FROM_ITER_LARGE 7        // problems of more than EPPM_MERGED_LARGE_PIXELS pixels
#endif
#ifndef EPPM_MERGED_LARGE_PIXELS
#define EPPM_MERGED_LARGE_PIXELS 65536
#endif
static bool sweeps_merged(int iteration, long long pixels, long long problem_pixels, int m)
{
    const int from = problem_pixels > EPPM_MERGED_LARGE_PIXELS ? EPPM_MERGED_FROM_ITER_LARGE : EPPM_MERGED_FROM_ITER;
    return m == 3 || (m < 0 && from >= 0 && iteration >= from && sweep_speculative(iteration, pixels, m));
}
// one directional sweep on the batch; keeps the result in p[k].nnf (swaps the ping-pong pair when needed)
void sweep(PmBatch& b, const float* lut, const eppm_params& prm, int dir, hipStream_t s, bool speculative)
{
    if (launch_pm_sweep(b, lut, prm.patch_r, prm.seg_len, dir, s, speculative))
        for (int k = 0; k < b.n; k++) std::swap(b.p[k].nnf, b.p[k].nnf_alt);
}
// baoJumpPropagate: six Jacobi launches (kernel.cu:849-854); an even number of swaps
void jump(PmBatch& b, const float* lut, const eppm_params& prm, hipStream_t s)
{
    for (int step = 32; step >= 1; step /= 2) {
        launch_pm_jump(b, lut, prm.patch_r, step, s);
        for (int k = 0; k < b.n; k++) std::swap(b.p[k].nnf, b.p[k].nnf_alt);
    }
}
// baoParallelPropagate (kernel.cu:790-795): `launches` Jacobi launches; the disabled call site runs ten per
// iteration (:1804-1809)
void neighbor(PmBatch& b, const float* lut, const eppm_params& prm, int launches, hipStream_t s)
{
    for (int q = 0; q < launches; q++) {
        launch_pm_neighbor(b, lut, prm.patch_r, s);
        for (int k = 0; k < b.n; k++) std::swap(b.p[k].nnf, b.p[k].nnf_alt);
    }
}
// returns with the NNF of problem k in b.p[k].nnf (an even number of sweeps: the caller's buffer)
void run_patchmatch(PmBatch& b, eppm_pm_rng* rng, const float* lut, const eppm_params& prm, hipStream_t s, int spec_mode)
{
    b.sweep_seq = 0;
    launch_pm_init_field(b, rng->dev(), s);
    launch_pm_cost_field(b, lut, prm.patch_r, s);
    for (int it = 0; it < prm.num_iter; it++) {
        if (prm.propagation == 1) jump(b, lut, prm, s);
        else if (prm.propagation == 2) neighbor(b, lut, prm, 10, s);
        else {
            const long long problem_pixels = (long long)b.p[0].P.w * b.p[0].P.h, pixels = problem_pixels * b.n * b.npairs;
            if (sweeps_merged(it, pixels, problem_pixels, spec_mode) && launch_pm_sweeps_merged(b, lut, prm.patch_r, prm.seg_len, it, s)) { /* in place */ }
            else for (int dir = 0; dir < 4; dir++) sweep(b, lut, prm, dir, s, sweep_speculative(it, pixels, spec_mode));
        }
        search(b, rng, lut, prm, s, it);
    }
}

