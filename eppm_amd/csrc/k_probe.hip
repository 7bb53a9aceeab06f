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
void launch_probe(const float* x, float* y, int n, int which, hipStream_t s)
{
    hipLaunchKernelGGL(k_probe, dim3((n + 255) / 256), dim3(256), 0, s, x, y, n, which);
}

}  // namespace eppm
