// k_color.hip -- Middlebury colour coding of a flow field on the device (reference:
// basic/bao_basic_cuda.cuh:743-845, called by the driver at :311 with (20,20); the colour wheel is
// 3rdparty/middlebury/colorcode.cpp:30-78).  Optional output of compute_flow (the color_flow argument).
//
// Arithmetic as the reference writes it: float throughout except `(int)(255.0*col)` and `col *= .75`, which
// are double (.cuh:795-797); the angle is atan2(-fy,-fx)/3.14159f (:779); a vector is drawn when
// |fx| < 999999 and |fy| < 999999 (:825), black otherwise.  CUDA's atan2f is not specified bit for bit, so --
// like __expf -- oracle and kernel share one restatement (color_atan2: Cephes-style reduction to
// [0, tan(pi/8)] and a degree-4 polynomial in t^2, plain IEEE operations in this order, no contraction);
// against a CUDA build the 8-bit result may differ by one level where atan2f rounds differently.
#include "eppm_device.cuh"
#include "eppm_internal.h"

namespace eppm {

// colour wheel, colorcode.cpp:30-59 / .cuh:751-775: 55 entries (RY 15, YG 6, GC 4, CB 11, BM 13, MR 6)
struct ColorWheel { int n; unsigned char c[60][3]; };
constexpr ColorWheel make_wheel()
{
    ColorWheel W{};
    const int RY = 15, YG = 6, GC = 4, CB = 11, BM = 13, MR = 6;
    int k = 0;
    for (int i = 0; i < RY; i++, k++) { W.c[k][0] = 255; W.c[k][1] = (unsigned char)(255 * i / RY); W.c[k][2] = 0; }
    for (int i = 0; i < YG; i++, k++) { W.c[k][0] = (unsigned char)(255 - 255 * i / YG); W.c[k][1] = 255; W.c[k][2] = 0; }
    for (int i = 0; i < GC; i++, k++) { W.c[k][0] = 0; W.c[k][1] = 255; W.c[k][2] = (unsigned char)(255 * i / GC); }
    for (int i = 0; i < CB; i++, k++) { W.c[k][0] = 0; W.c[k][1] = (unsigned char)(255 - 255 * i / CB); W.c[k][2] = 255; }
    for (int i = 0; i < BM; i++, k++) { W.c[k][0] = (unsigned char)(255 * i / BM); W.c[k][1] = 0; W.c[k][2] = 255; }
    for (int i = 0; i < MR; i++, k++) { W.c[k][0] = 255; W.c[k][1] = 0; W.c[k][2] = (unsigned char)(255 - 255 * i / MR); }
    W.n = k;
    return W;
}
__constant__ ColorWheel c_wheel = make_wheel();

// atan2f restated (see the header comment).  IEEE sign conventions: the sign of a zero x counts (atan2(+-0,-0) = +-pi).
__device__ __forceinline__ float color_atan2(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float t = (mx == 0.0f) ? 0.0f : mn / mx;                 // in [0,1]
    float base = 0.0f;
    if (t > 0.4142135679721832275390625f) {                  // tan(pi/8): atan t = pi/4 + atan((t-1)/(t+1))
        base = 0.785398185253143310546875f;
        t = (t - 1.0f) / (t + 1.0f);
    }
    const float z = t * t;
    float p = 8.05374449538e-2f * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    float r = base + (p * z * t + t);
    if (ay > ax) r = 1.57079637050628662109375f - r;
    if (__builtin_signbitf(x)) r = 3.1415927410125732421875f - r;
    return __builtin_copysignf(r, y);
}

// _d_bao_compute_flow_color, .cuh:776-807: returns R | G<<8 | B<<16 (pix.x = wheel channel 0 ... ; pix.w is left unset there, 0 here)
__device__ __forceinline__ uint32_t flow_color(float fx, float fy)
{
    const float rad = __builtin_sqrtf(fx * fx + fy * fy);
    const float a = color_atan2(-fy, -fx) / 3.14159f;
    const float fk = (a + 1.0f) / 2.0f * (float)(c_wheel.n - 1);
    const int k0 = (int)fk;
    const int k1 = (k0 + 1) % c_wheel.n;
    const float f = fk - (float)k0;
    uint32_t out = 0;
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const float col0 = (float)c_wheel.c[k0][b] / 255.0f;
        const float col1 = (float)c_wheel.c[k1][b] / 255.0f;
        float col = (1 - f) * col0 + f * col1;
        if (rad <= 1) col = 1 - rad * (1 - col);
        else col = (float)((double)col * .75);
        out |= ((uint32_t)(unsigned char)(int)(255.0 * (double)col)) << (8 * b);
    }
    return out;
}

// _d_bao_convert_flow_to_colorshow (float2 form), .cuh:816-829
__global__ __launch_bounds__(256) void k_flow_to_color(uint32_t* __restrict__ rgba_, const float2* __restrict__ flow_, int h, int w, float max_rad, size_t pstride)
{
    uint32_t* __restrict__ rgba = pair_ptr(rgba_, pstride, blockIdx.z);
    const float2* __restrict__ flow = pair_ptr(flow_, pstride, blockIdx.z);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const float2 v = flow[y * w + x];
    uint32_t c = 0;
    if (fabsf(v.x) < 999999 && fabsf(v.y) < 999999) c = flow_color(v.x / max_rad, v.y / max_rad);
    rgba[y * w + x] = c;
}

void launch_flow_to_color(uint32_t* rgba, const float* flow, int h, int w, float max_disp_x, float max_disp_y, hipStream_t s, Batch bt)
{
    const float max_rad = sqrtf(max_disp_x * max_disp_x + max_disp_y * max_disp_y);     // sqrt(float) overload, .cuh:835,844
    dim3 block(64, 4), grid((w + 63) / 64, (h + 3) / 4, bt.n);
    hipLaunchKernelGGL(k_flow_to_color, grid, block, 0, s, rgba, (const float2*)flow, h, w, max_rad, bt.stride);
}

}  // namespace eppm
