// k_probe.hip -- test support, linked into libeppm_hip_test.so only (include/eppm_test.h): the shared float formulas of
// eppm_device.cuh evaluated on the device for arrays of host floats, so that the parity tests can compare them bit for bit with the
// oracle's restatement (tests/test_parity_gpu.py: test_fast_exp_bits, test_div_const_bits).
#include "eppm_device.cuh"
#include "eppm_internal.h"

namespace eppm {

// ---------------------------------------------------------------------------------------------------
// arithmetic probes for the parity tests of the shared float formulas
// ---------------------------------------------------------------------------------------------------
__global__ void k_probe(const float* __restrict__ x, float* __restrict__ y, int n, int which)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float r;
    if (which == 0) r = fast_exp(v);
    else if (which == 1) r = div_ad2(v);
    else if (which == 2) r = div_wmf2(v);
    else r = unorm8(v);
    y[i] = r;
}
// y[i] = delta_lookup(the DeltaTab at `tab`, x[i]): the table form of a range term against the formula it stands for
__global__ __launch_bounds__(256) void k_probe_delta(const float* __restrict__ x, float* __restrict__ y, int n, const float* __restrict__ tab)
{
    __shared__ DeltaTab s_D, s_E;               // both forms of the look-up (LDS addresses / offsets): they must agree, or the answer is NaN
    load_delta_tab(s_D, tab, threadIdx.x, 256);
    load_delta_tab<false>(s_E, tab, threadIdx.x, 256);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float a = delta_lookup(s_D, x[i]), b = delta_lookup_off(s_E, x[i]);
        y[i] = (__float_as_uint(a) == __float_as_uint(b)) ? a : __uint_as_float(0x7fc00000u);
    }
}
void launch_probe_delta(const float* x, float* y, int n, const float* tab, hipStream_t s)
{
    hipLaunchKernelGGL(k_probe_delta, dim3((n + 255) / 256), dim3(256), 0, s, x, y, n, tab);
}
void launch_probe(const float* x, float* y, int n, int which, hipStream_t s)
{
    hipLaunchKernelGGL(k_probe, dim3((n + 255) / 256), dim3(256), 0, s, x, y, n, which);
}

}  // namespace eppm
