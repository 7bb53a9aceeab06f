#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
// L-inf distance of two texels whose channels are INTEGER bit patterns (denormal floats): sub + max3|abs| -> integer bits
__global__ void k_check(const uint32_t* a, const uint32_t* b, uint32_t* o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float ax = __uint_as_float(a[3*i]), ay = __uint_as_float(a[3*i+1]), az = __uint_as_float(a[3*i+2]);
    float bx = __uint_as_float(b[3*i]), by = __uint_as_float(b[3*i+1]), bz = __uint_as_float(b[3*i+2]);
    float d = fmaxf(fmaxf(fabsf(ax - bx), fabsf(ay - by)), fabsf(az - bz));
    o[i] = __float_as_uint(d);
}
#define R8(x) x x x x x x x x
#define R32(x) R8(x) R8(x) R8(x) R8(x)
__global__ __launch_bounds__(256) void k_rate_sub(float* out, uint32_t ua, uint32_t ub, int n)
{
    float v0 = __uint_as_float(ua + threadIdx.x), v1 = __uint_as_float(ub), v2 = __uint_as_float(ub*3), v3 = v1;
    for (int it = 0; it < n; it++) asm volatile(R32("v_sub_f32 %0, %1, %0\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3;
}
__global__ __launch_bounds__(256) void k_rate_max3(float* out, uint32_t ua, uint32_t ub, int n)
{
    float v0 = __uint_as_float(ua + threadIdx.x), v1 = __uint_as_float(ub), v2 = __uint_as_float(ub*3), v3 = v1;
    for (int it = 0; it < n; it++) asm volatile(R32("v_max3_f32 %0, |%1|, |%2|, |%0|\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3;
}
int main()
{
    const int n = 1 << 20;
    std::vector<uint32_t> a(3*n), b(3*n), o(n);
    uint32_t s = 12345;
    for (int i = 0; i < 3*n; i++) { s = s * 1664525u + 1013904223u; a[i] = ((s >> 8) & 255) * 40; s = s * 1664525u + 1013904223u; b[i] = ((s >> 8) & 255) * 40; }
    uint32_t *da, *db, *dout;
    hipMalloc(&da, 12*n); hipMalloc(&db, 12*n); hipMalloc(&dout, 4*n);
    hipMemcpy(da, a.data(), 12*n, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 12*n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(n/256), dim3(256), 0, 0, da, db, dout, n);
    hipMemcpy(o.data(), dout, 4*n, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; i++) {
        uint32_t m = 0;
        for (int c = 0; c < 3; c++) { int d = (int)a[3*i+c] - (int)b[3*i+c]; if (d < 0) d = -d; if ((uint32_t)d > m) m = d; }
        if (o[i] != m) { if (bad < 5) printf("mismatch %d: got %u want %u\n", i, o[i], m); bad++; }
    }
    printf("denormal-int L-inf distance: %ld mismatches of %d\n", bad, n);
    float* out; hipMalloc(&out, 2048*256*4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 20; w++) hipLaunchKernelGGL(k_rate_sub, dim3(2048), dim3(256), 0, 0, out, 0x3f800000u, 0x3f000000u, 4096);
    struct { const char* name; void (*k)(float*, uint32_t, uint32_t, int); uint32_t ua, ub; } tab[] = {
        {"v_sub_f32 normal", k_rate_sub, 0x3f800000u, 0x3f000000u}, {"v_sub_f32 denormal", k_rate_sub, 4000u, 40u},
        {"v_max3_f32|abs| normal", k_rate_max3, 0x3f800000u, 0x3f000000u}, {"v_max3_f32|abs| denormal", k_rate_max3, 4000u, 40u}};
    for (auto& t : tab) {
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(t.k, dim3(2048), dim3(256), 0, 0, out, t.ua, t.ub, 4096);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.3f ms  %6.2f cycles per wave64 instruction per SIMD\n", t.name, ms, ms * 1e-3 * 2.3e9 / (4096.0 * 32 * 8));
    }
    return 0;
}
