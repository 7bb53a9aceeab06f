// test_hooks.cpp -- libeppm_hip_test.so only (include/eppm_test.h): the kernel-variant switches and arithmetic probes of the parity
// tests.  The product library is linked without this file and exports none of it.
#define EPPM_TEST_HOOKS 1
#include "api_internal.h"

using namespace eppm;

std::atomic<int> g_opt_rand_table{1};
std::atomic<int> g_opt_sweep_spec{-1};
std::atomic<int> g_opt_no_split{0};

static int probe(const float* x, float* y, int n, int which)
{
    float *dx = nullptr, *dy = nullptr;
    HIPCHK(hipMalloc(&dx, (size_t)n * 4));
    HIPCHK(hipMalloc(&dy, (size_t)n * 4));
    HIPCHK(hipMemcpy(dx, x, (size_t)n * 4, hipMemcpyHostToDevice));
    launch_probe(dx, dy, n, which, nullptr);
    HIPCHK(hipMemcpy(y, dy, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy);
    return launcher_finish();
}
extern "C" int eppm_test_set_option(const char* name, int value)
{
    if (!name) return set_err(EPPM_ERR_ARG, "eppm_test_set_option: NULL name");
    if (!strcmp(name, "c2f_no_split")) { g_opt_no_split.store(value); return EPPM_OK; }
    if (!strcmp(name, "sweep_spec")) { g_opt_sweep_spec.store(value); return EPPM_OK; }
    if (!strcmp(name, "rand_table")) { g_opt_rand_table.store(value); return EPPM_OK; }
    return set_err(EPPM_ERR_ARG, "eppm_test_set_option: unknown option '%s'", name);
}
extern "C" int eppm_probe_c2f_window(int patch_r, int* span_x, int* span_y)
{
    if (!span_x || !span_y) return set_err(EPPM_ERR_ARG, "eppm_probe_c2f_window: NULL argument");
    if (!c2f_window_span(patch_r, span_x, span_y)) return set_err(EPPM_ERR_ARG, "no LDS-window refine kernel for patch_r %d", patch_r);
    return EPPM_OK;
}
extern "C" int eppm_probe_fast_exp(const float* x, float* y, int n) { return probe(x, y, n, 0); }
extern "C" int eppm_probe_div_const(const float* x, float* y, int n, int which) { return probe(x, y, n, 1 + which); }
// y[i] = the table form of a range term at the distance x[i] (which = 0: 1 - exp(-d^2 / LAMBDA_AD^2) of the patch data term -- exact
// library only, the tolerance library has no such table --, 1: exp(-d^2 / SIG_R^2) of the smoothing / weighted-median weights)
extern "C" int eppm_probe_delta_table(const float* x, float* y, int n, int which)
{
    if (!x || !y || n < 1 || which < 0 || which > 1) return set_err(EPPM_ERR_ARG, "eppm_probe_delta_table: bad argument");
    float *lut = nullptr, *dx = nullptr, *dy = nullptr;
    size_t head = 0;
    if (which == 0) {
#ifdef EPPM_TOL
        return set_err(EPPM_ERR_ARG, "eppm_probe_delta_table: the tolerance library's patch term has no delta table");
#else
        CHK(upload_pm_lut(&lut, 9));
        head = 9 + 1 + 9;
#endif
    } else {
        CHK(upload_blf_lut(&lut));
        head = kBlfRadius + 1;
    }
    HIPCHK(hipMalloc(&dx, (size_t)n * 4));
    HIPCHK(hipMalloc(&dy, (size_t)n * 4));
    HIPCHK(hipMemcpy(dx, x, (size_t)n * 4, hipMemcpyHostToDevice));
    launch_probe_delta(dx, dy, n, lut + head, nullptr);
    HIPCHK(hipMemcpy(y, dy, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(lut);
    return launcher_finish();
}
