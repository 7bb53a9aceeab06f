/*
 * eppm_test.h -- test support of the EPPM engine: exported by libeppm_hip_test.so ONLY.
 *
 * libeppm_hip_test.so is libeppm_hip.so's own objects (every kernel, the launchers, the C++ class: the same .o files) with
 * eppm_api.cpp compiled once more with -DEPPM_TEST_HOOKS and k_probe.hip added (eppm_amd/csrc/Makefile).  It exports everything
 * include/eppm.h declares plus the four entry points below; the product library exports none of them and has no switch a host
 * program could flip: `nm -D libeppm_hip.so | grep -c "eppm_test\|eppm_probe"` is 0 (tests/test_abi_cpu.py).
 * The parity tests load the test library; bench.py, smoke(), the CLI and the C++ class link the product library.
 */
#ifndef EPPM_TEST_H_
#define EPPM_TEST_H_

#include "eppm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* switches with which the parity tests steer launches onto a specific kernel variant (a host program never needs them;
 * every variant computes the same bits).  A call sets the DEFAULT that contexts created afterwards copy, and what the context-less stage
 * launchers below read; a context that exists already is not affected.  "c2f_no_split" = 1: the candidate refine is never split over
 * several workgroups per tile, so that small images run the LDS-window kernels too.  "sweep_spec": -1 (default) the sweeps of PatchMatch
 * iterations >= 2 (the third on) run in the speculative two-launch form when a launch covers at least 100 000 pixels (two 1024x436 pairs,
 * one 1920x1080 pair), 0 never, 1 always (also in eppm_pm_seg_propagate, which otherwise runs the classic form), 2 always and without
 * the work list (phase B walks every chain), 3 always and in the merged form (one phase A for the four sweeps of an iteration, the form the
 * library takes by itself from the sixth iteration on -- from the eighth on problems of more than 65 536 pixels, i.e. the quarter-resolution
 * level of 1920x1080 and 3840x2160 pairs).  "rand_table": 1 (default) a context's random searches read numbers drawn ahead per geometry,
 * 0 they draw while they search -- the form a context takes by itself when the table would exceed 512 MB. */
int  eppm_test_set_option(const char* name, int value);
/* admissible spread (max - min, pixels) of a 16x16 tile's candidate centres for which the LDS-window refine kernels stage the
 * target window; wider tiles take the per-access path inside the same launch (patch_r 9 or 17) */
int  eppm_probe_c2f_window(int patch_r, int* span_x, int* span_y);
/* device-side arithmetic probes (parity of the shared float formulas): y[i] = f(x[i]) for n host floats */
int  eppm_probe_fast_exp(const float* x, float* y, int n);
int  eppm_probe_div_const(const float* x, float* y, int n, int which); /* 0: /(.1f*.1f) 1: /(.02f*.02f) 2: unorm8 (x = 0..255) */
/* the range terms that are read from a table of the 598 possible L-inf distances of unorm8 texels instead of being evaluated
 * (eppm_device.cuh: DeltaTab): y[i] = table(x[i]); which = 0: 1 - exp(-d^2/(.1f*.1f)) of the patch data term, 1: exp(-d^2/(.02f*.02f)) of
 * the smoothing and weighted-median weights.  x must be such distances (|a/255 - b/255| of two bytes, as floats). */
int  eppm_probe_delta_table(const float* x, float* y, int n, int which);

#ifdef __cplusplus
}
#endif
#endif /* EPPM_TEST_H_ */
