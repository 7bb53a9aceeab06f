// terms_rate -- attainable VALU issue rate for the instruction mix of one patch-cost term (patch_terms, the body
// of every PatchMatch / refine inner loop) with NO memory traffic: inputs live in registers and are perturbed
// with full-rate integer ops each iteration.  Reported: time per term per wave on one SIMD, at WPS waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
//        -I eppm_amd/csrc -I include -o terms_rate terms_rate.hip
#include "eppm_device.cuh"
#include <cstdio>
using namespace eppm;

__global__ __launch_bounds__(256) void k_terms(float* out, const float* lut, int iters, unsigned seed)
{
    __shared__ PatchLutT<10> L;
    load_patch_lut(L, lut, 9, threadIdx.x, 256);
    __syncthreads();
    const rgbf c1 = {0.3f + threadIdx.x * 1e-3f, 0.4f, 0.5f};
    rgbf c2[3] = {{0.31f, 0.41f, 0.52f}, {0.29f, 0.43f, 0.5f}, {0.33f, 0.39f, 0.48f}};
    float cs[3] = {0, 0, 0}, ws[3] = {0, 0, 0};
    unsigned u = seed + threadIdx.x * 2654435761u;
    float4 q1 = make_float4(0.35f, 0.45f, 0.55f, __uint_as_float(4u * (threadIdx.x & 255)));
    float4 q2[3];
    for (int n = 0; n < 3; n++) q2[n] = make_float4(0.3f + 0.01f * n, 0.4f, 0.5f + 0.02f * n, __uint_as_float(4u * ((threadIdx.x + n) & 255)));
#pragma unroll 2
    for (int it = 0; it < iters; it++) {
        // cheap perturbation: keeps the compiler from hoisting anything, costs 4 full-rate integer ops per 3 terms
        u += 0x9E3779B9u;
        q1.x = __uint_as_float((__float_as_uint(q1.x) & 0xfffffff0u) | (u & 15u));
        float a2 = max_abs_diff(c1, texel_rgb(q1));
        a2 *= a2;
        const float gsp = L.gsp[it & 63];
#pragma unroll
        for (int n = 0; n < 3; n++) {
            q2[n].y = __uint_as_float(__float_as_uint(q2[n].y) ^ (u & 7u));
            const rgbf p1 = texel_rgb(q1), p2 = texel_rgb(q2[n]);
            float cost = max_abs_diff(p1, p2);
            cost = one_minus_fast_exp(div_ad2(-(cost * cost)));
            cost += census_cost(L.cnx, __float_as_uint(q1.w), __float_as_uint(q2[n].w));
            float temp = max_abs_diff(c2[n], p2);
            temp *= temp;
            float weight = fast_exp(div_ad2(-(a2 + temp)));
            weight *= gsp;
            cost *= weight;
            cs[n] += cost;
            ws[n] += weight;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = cs[0] / ws[0] + cs[1] / ws[1] + cs[2] / ws[2];
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float h_lut[19];
    for (int i = 0; i < 10; i++) h_lut[i] = expf(-(float)(i * i) / 20.25f);
    for (int k = 0; k < 9; k++) h_lut[10 + k] = 1 - expf(-(float)(k * k) / 5.76f);
    float *lut, *out;
    hipMalloc(&lut, sizeof(h_lut));
    hipMemcpy(lut, h_lut, sizeof(h_lut), hipMemcpyHostToDevice);
    hipMalloc(&out, (size_t)cus * 16 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int w = 0; w < 30; w++) hipLaunchKernelGGL(k_terms, dim3(cus * 4), dim3(256), 0, 0, out, lut, iters, 1u);    // clock ramp
    hipDeviceSynchronize();
    for (int wps = 1; wps <= 8; wps++) {            // workgroups per CU = waves per SIMD (256 threads = 4 waves, one per SIMD)
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_terms, dim3(cus * wps), dim3(256), 0, 0, out, lut, iters, 7u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double terms_per_simd = (double)iters * 3 * wps;
        printf("waves/SIMD %d: %.3f ms, %.1f ns per term per SIMD, %.1f cycles at 2.3 GHz\n", wps, ms, ms * 1e6 / terms_per_simd,
               ms * 1e-3 * 2.3e9 / terms_per_simd);
    }
    return 0;
}
