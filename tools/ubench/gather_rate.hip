// L1 (TCP) throughput of 16-byte gathers by lane-address pattern, footprint inside the 32 KiB L1: how many clocks does one
// global_load_dwordx4 of a 64-lane wave cost the CU when the lanes' addresses are (a) unrelated, (b) grouped?
// build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip ; run: ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int BYTES> struct Vec;
template <> struct Vec<16> { using T = float4; static __device__ float f(T v) { return v.x + v.w; } };
template <> struct Vec<8> { using T = float2; static __device__ float f(T v) { return v.x + v.y; } };
template <> struct Vec<4> { using T = float; static __device__ float f(T v) { return v; } };

template <int GROUP, int STRIDE, int ALIGN, int BYTES = 16>
__global__ __launch_bounds__(256) void k_gather(const char* __restrict__ buf, unsigned mask, int iters, float* out)
{
    const unsigned lane = threadIdx.x & 63, grp = lane / GROUP, j = lane % GROUP;
    unsigned st = (blockIdx.x * 977u + (threadIdx.x >> 6) * 131u + grp * 2654435761u) | 1u;
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        typename Vec<BYTES>::T v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            st = st * 1664525u + 1013904223u;
            unsigned base = ((st >> 8) & mask) & ~(unsigned)(ALIGN - 1);
            unsigned off = (base + j * STRIDE) & mask & ~(unsigned)(BYTES - 1);
            v[k] = *reinterpret_cast<const typename Vec<BYTES>::T*>(buf + off);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) acc += Vec<BYTES>::f(v[k]);
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int GROUP, int STRIDE, int ALIGN, int BYTES = 16>
static void run(const char* name, const char* buf, unsigned bytes, float* out)
{
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    (void)0; hipLaunchKernelGGL((k_gather<GROUP, STRIDE, ALIGN, BYTES>), dim3(blocks), dim3(256), 0, 0, buf, bytes - 1, 50, out);
    hipEventRecord(a);
    (void)0; hipLaunchKernelGGL((k_gather<GROUP, STRIDE, ALIGN, BYTES>), dim3(blocks), dim3(256), 0, 0, buf, bytes - 1, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = (double)blocks * 4 /*waves*/ * iters * 4 / 256.0;
    printf("%-44s footprint %7u B: %8.3f ms, %6.1f ns per wave-load per CU (%5.1f clk @2.4GHz)\n", name, bytes, ms,
           ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
}

int main()
{
    char* buf; float* out;
    hipMalloc(&buf, 64 << 20); hipMemset(buf, 0, 64 << 20); hipMalloc(&out, 16);
    for (unsigned bytes : {16u << 10, 1u << 20}) {
        run<1, 0, 16>("64 unrelated lanes", buf, bytes, out);
        run<2, 32, 16>("pairs, 32 B apart", buf, bytes, out);
        run<4, 32, 16>("quads, 32 B apart (128 B span)", buf, bytes, out);
        run<4, 16, 16>("quads, contiguous 64 B, 16 B aligned", buf, bytes, out);
        run<4, 16, 64>("quads, contiguous 64 B, 64 B aligned", buf, bytes, out);
        run<8, 16, 16>("octets, contiguous 128 B, 16 B aligned", buf, bytes, out);
        run<8, 16, 128>("octets, contiguous 128 B, 128 B aligned", buf, bytes, out);
        run<16, 16, 16>("16 lanes contiguous 256 B, 16 B aligned", buf, bytes, out);
        run<16, 32, 16>("16 lanes, 32 B apart (512 B span)", buf, bytes, out);
        run<16, 224, 16>("16 lanes, 224 B apart", buf, bytes, out);
        run<64, 16, 1024>("64 lanes contiguous 1 KiB", buf, bytes, out);
        run<1, 0, 4, 4>("4 B loads: 64 unrelated lanes", buf, bytes, out);
        run<4, 8, 4, 4>("4 B loads: quads, 8 B apart", buf, bytes, out);
        run<16, 8, 4, 4>("4 B loads: 16 lanes, 8 B apart (128 B span)", buf, bytes, out);
        run<16, 56, 4, 4>("4 B loads: 16 lanes, 56 B apart", buf, bytes, out);
        run<64, 4, 256, 4>("4 B loads: 64 lanes contiguous 256 B", buf, bytes, out);
        run<1, 0, 8, 8>("8 B loads: 64 unrelated lanes", buf, bytes, out);
        run<4, 16, 8, 8>("8 B loads: quads, 16 B apart", buf, bytes, out);
        run<64, 8, 512, 8>("8 B loads: 64 lanes contiguous 512 B", buf, bytes, out);
    }
    return 0;
}
