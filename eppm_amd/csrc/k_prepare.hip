// k_prepare.hip -- prefilter, pyramid decimation, census transform (reference: basic/bao_basic_cuda.cuh
// :437-467, :565-601; bao_pmflow_census_kernel.cu:39-112).  HBM-bound byte work: one read of each input
// pixel neighbourhood through LDS tiles, one coalesced write.
#include "eppm_device.cuh"
#include "eppm_internal.h"

namespace eppm {

// ---------------------------------------------------------------------------------------------------
// Dense (2r+1)^2 Gaussian on RGBA u8 (.cuh:437-467): clamp-to-edge, weight = fast_exp(-(dy^2+dx^2)/s2)
// recomputed per tap, float accumulation in tap order (dy outer, dx inner), per-channel division,
// float->u8 truncation.  Tile 32x8 outputs per 256-thread block (4 waves), halo staged in LDS so
// every input pixel is fetched once per block, rows coalesced.
// ---------------------------------------------------------------------------------------------------
constexpr int GT_W = 32, GT_H = 8, G_MAXR = 6;

// blockIdx.z = pair * nimg + image: the (one or two) frames of a pair and all pairs of a batch share every launch
__global__ __launch_bounds__(256) void k_gauss_rgba(uint32_t* __restrict__ out0, const uint32_t* __restrict__ in0,
                                                    uint32_t* __restrict__ out1, const uint32_t* __restrict__ in1, int pitch,
                                                    int h, int w, float sigma2, int radius, int nimg, size_t pstride)
{
    const unsigned pair = blockIdx.z / nimg, im = blockIdx.z % nimg;
    uint32_t* __restrict__ out = pair_ptr(im ? out1 : out0, pstride, pair);
    const uint32_t* __restrict__ in = pair_ptr(im ? in1 : in0, pstride, pair);
    __shared__ uint32_t tile[(GT_H + 2 * G_MAXR) * (GT_W + 2 * G_MAXR)];
    __shared__ float wtab[(2 * G_MAXR + 1) * (2 * G_MAXR + 1)];
    const int tw = GT_W + 2 * radius, th = GT_H + 2 * radius;
    const int x0 = blockIdx.x * GT_W, y0 = blockIdx.y * GT_H;
    const int tid = threadIdx.y * GT_W + threadIdx.x;
    for (int t = tid; t < tw * th; t += 256) {
        const int ty = t / tw, tx = t % tw;
        const int cy = max(0, min(h - 1, y0 + ty - radius));
        const int cx = max(0, min(w - 1, x0 + tx - radius));
        tile[t] = in[cy * pitch + cx];
    }
    const int d = 2 * radius + 1;
    for (int t = tid; t < d * d; t += 256) {
        const int dy = t / d - radius, dx = t % d - radius;
        wtab[t] = fast_exp(-(float)(dy * dy + dx * dx) / sigma2);
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= w || y >= h) return;
    float vx = 0, vy = 0, vz = 0, vw = 0, sum = 0;
    for (int dy = 0; dy < d; dy++)
        for (int dx = 0; dx < d; dx++) {
            const float weight = wtab[dy * d + dx];
            const uint32_t p = tile[(threadIdx.y + dy) * tw + threadIdx.x + dx];
            vx += (float)(p & 0xffu) * weight;
            vy += (float)((p >> 8) & 0xffu) * weight;
            vz += (float)((p >> 16) & 0xffu) * weight;
            vw += (float)(p >> 24) * weight;
            sum += weight;
        }
    vx /= sum; vy /= sum; vz /= sum; vw /= sum;
    const uint32_t r = (uint32_t)vx | ((uint32_t)vy << 8) | ((uint32_t)vz << 16) | ((uint32_t)vw << 24);
    out[y * pitch + x] = r;
}

void launch_gauss_rgba(uint32_t* out, const uint32_t* in, int pitch_px, int h, int w, float sigma, int radius, hipStream_t s, Batch bt)
{
    dim3 grid((w + GT_W - 1) / GT_W, (h + GT_H - 1) / GT_H, bt.n), block(GT_W, GT_H);
    hipLaunchKernelGGL(k_gauss_rgba, grid, block, 0, s, out, in, out, in, pitch_px, h, w, sigma * sigma * 2, radius, 1, bt.stride);
}
void launch_gauss_rgba2(uint32_t* out0, const uint32_t* in0, uint32_t* out1, const uint32_t* in1, int pitch_px, int h, int w, float sigma,
                        int radius, hipStream_t s, Batch bt)
{
    dim3 grid((w + GT_W - 1) / GT_W, (h + GT_H - 1) / GT_H, 2 * bt.n), block(GT_W, GT_H);
    hipLaunchKernelGGL(k_gauss_rgba, grid, block, 0, s, out0, in0, out1, in1, pitch_px, h, w, sigma * sigma * 2, radius, 2, bt.stride);
}

// ---------------------------------------------------------------------------------------------------
// Pyramid step blur + decimate (.cuh:647-663) fused for the exact 2:1 case.  At ratio 1/2 the bilinear resize
// reads one tap with weight 1: output (x,y) = blurred pixel (min(2x+1,w-1), min(2y+1,h-1)).  The reference blurs
// every pixel of the finer level into a temp plane and keeps a quarter of them; here only the kept pixels are
// blurred -- the same formula per pixel, so the same bytes.  32x8 outputs per block, source halo in LDS.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gauss_decimate2(uint32_t* __restrict__ out0, const uint32_t* __restrict__ in0,
                                                         uint32_t* __restrict__ out1, const uint32_t* __restrict__ in1, int out_pitch,
                                                         int outH, int outW, int pitch, int h, int w, float sigma2, int radius, int nimg,
                                                         size_t pstride)
{
    const unsigned pair = blockIdx.z / nimg, im = blockIdx.z % nimg;
    uint32_t* __restrict__ out = pair_ptr(im ? out1 : out0, pstride, pair);
    const uint32_t* __restrict__ in = pair_ptr(im ? in1 : in0, pstride, pair);
    __shared__ uint32_t tile[(2 * GT_H + 2 * G_MAXR) * (2 * GT_W + 2 * G_MAXR)];
    __shared__ float wtab[(2 * G_MAXR + 1) * (2 * G_MAXR + 1)];
    const int tw = 2 * GT_W + 2 * radius, th = 2 * GT_H + 2 * radius;
    const int x0 = blockIdx.x * GT_W, y0 = blockIdx.y * GT_H;          // output coordinates
    const int sx0 = 2 * x0 + 1 - radius, sy0 = 2 * y0 + 1 - radius;     // source coordinates of the tile origin
    const int tid = threadIdx.y * GT_W + threadIdx.x;
    for (int t = tid; t < tw * th; t += 256) {
        const int ty = t / tw, tx = t % tw;
        const int cy = max(0, min(h - 1, sy0 + ty));
        const int cx = max(0, min(w - 1, sx0 + tx));
        tile[t] = in[cy * pitch + cx];
    }
    const int d = 2 * radius + 1;
    for (int t = tid; t < d * d; t += 256) {
        const int dy = t / d - radius, dx = t % d - radius;
        wtab[t] = fast_exp(-(float)(dy * dy + dx * dx) / sigma2);
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= outW || y >= outH) return;
    // The kept pixel is min(2x+1, w-1): when the clamp bites (2x+1 = w, possible for the last column of an odd-width
    // level only if outW were rounded up -- it is floor(w/2), so it never does) the tile position would shift; assert by construction.
    const int lx = 2 * threadIdx.x, ly = 2 * threadIdx.y;              // tile position of the window origin of (2x+1, 2y+1)
    float vx = 0, vy = 0, vz = 0, vw = 0, sum = 0;
    for (int dy = 0; dy < d; dy++)
        for (int dx = 0; dx < d; dx++) {
            const float weight = wtab[dy * d + dx];
            const uint32_t p = tile[(ly + dy) * tw + lx + dx];
            vx += (float)(p & 0xffu) * weight;
            vy += (float)((p >> 8) & 0xffu) * weight;
            vz += (float)((p >> 16) & 0xffu) * weight;
            vw += (float)(p >> 24) * weight;
            sum += weight;
        }
    vx /= sum; vy /= sum; vz /= sum; vw /= sum;
    // float -> u8 truncation of the blur, then the resize's own float -> u8 of 1*value + 0*others: the identity
    out[y * out_pitch + x] = (uint32_t)vx | ((uint32_t)vy << 8) | ((uint32_t)vz << 16) | ((uint32_t)vw << 24);
}

// valid when the resize ratio is exactly 1/2 and every kept pixel 2x+1, 2y+1 lies inside the finer level
bool gauss_decimate2_ok(int outH, int outW, int h, int w, float ratio, int radius)
{
    return ratio == 0.5f && radius <= G_MAXR && 2 * (outW - 1) + 1 <= w - 1 && 2 * (outH - 1) + 1 <= h - 1;
}
// two images per launch (out1/in1 may repeat out0/in0 with nimg = 1)
void launch_gauss_decimate2(uint32_t* out0, const uint32_t* in0, uint32_t* out1, const uint32_t* in1, int nimg, int out_pitch_px, int outH,
                            int outW, int pitch_px, int h, int w, float sigma, int radius, hipStream_t s, Batch bt)
{
    dim3 grid((outW + GT_W - 1) / GT_W, (outH + GT_H - 1) / GT_H, nimg * bt.n), block(GT_W, GT_H);
    hipLaunchKernelGGL(k_gauss_decimate2, grid, block, 0, s, out0, in0, out1, in1, out_pitch_px, outH, outW, pitch_px, h, w,
                       sigma * sigma * 2, radius, nimg, bt.stride);
}

// ---------------------------------------------------------------------------------------------------
// Bilinear resize, uchar4 (.cuh:565-601), as written: fx=(x+1)/ratio-1, trunc, 4 taps, trunc to u8.
// For ratio 1/2 and 1/4 the weights are exactly 1,0,0,0 (pixel (2x+1,2y+1) / (4x+3,4y+3)).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize_rgba(uint32_t* __restrict__ out_, int out_pitch, int outH, int outW,
                                                     const uint32_t* __restrict__ in_, int in_pitch, int h, int w, float ratio, size_t pstride)
{
    uint32_t* __restrict__ out = pair_ptr(out_, pstride, blockIdx.z);
    const uint32_t* __restrict__ in = pair_ptr(in_, pstride, blockIdx.z);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= outW || y >= outH) return;
    const float div_scale = 1.f / ratio;
    const float fx = (float)(x + 1) * div_scale - 1;
    const float fy = (float)(y + 1) * div_scale - 1;
    const int xx = (int)fx, yy = (int)fy;
    const float dx = fmaxf(fminf(fx - xx, 1), 0);
    const float dy = fmaxf(fminf(fy - yy, 1), 0);
    float rx = 0, ry = 0, rz = 0, rw = 0;
    for (int m = 0; m <= 1; m++)
        for (int n = 0; n <= 1; n++) {
            const int u = max(0, min(w - 1, xx + m));
            const int v = max(0, min(h - 1, yy + n));
            const float sc = fabsf(1 - m - dx) * fabsf(1 - n - dy);
            const uint32_t p = in[v * in_pitch + u];
            rx += ((float)(p & 0xffu) * sc);
            ry += ((float)((p >> 8) & 0xffu) * sc);
            rz += ((float)((p >> 16) & 0xffu) * sc);
            rw += ((float)(p >> 24) * sc);
        }
    out[y * out_pitch + x] = (uint32_t)rx | ((uint32_t)ry << 8) | ((uint32_t)rz << 16) | ((uint32_t)rw << 24);
}

void launch_resize_rgba(uint32_t* out, int out_pitch_px, int outH, int outW, const uint32_t* in, int in_pitch_px, int h, int w,
                        float ratio, hipStream_t s, Batch bt)
{
    dim3 block(64, 4), grid((outW + 63) / 64, (outH + 3) / 4, bt.n);
    hipLaunchKernelGGL(k_resize_rgba, grid, block, 0, s, out, out_pitch_px, outH, outW, in, in_pitch_px, h, w, ratio, bt.stride);
}

// ---------------------------------------------------------------------------------------------------
// 3x3 census on luminance (census :45-90): bit k = lum(neigh_k) > lum(centre), clamp addressing,
// lum = .3R + .6G + .1B on unorm floats, evaluated left to right.  64x4 tile per block, luminance of
// the (64+2)x(4+2) halo computed once into LDS.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lum_of(uint32_t p)
{
    const rgbf c = unpack_rgb(p);
    return 0.3f * c.x + 0.6f * c.y + 0.1f * c.z;
}

__global__ __launch_bounds__(256) void k_census(uint8_t* __restrict__ census, int cpitch, float4* __restrict__ texels, int tpitch,
                                                const uint32_t* __restrict__ img, int ipitch, int w, int h)
{
    __shared__ float lum[6][66];
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 4;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    for (int t = tid; t < 6 * 66; t += 256) {
        const int ty = t / 66, tx = t % 66;
        const int cy = iclamp(y0 + ty - 1, 0, h - 1), cx = iclamp(x0 + tx - 1, 0, w - 1);
        lum[ty][tx] = lum_of(img[cy * ipitch + cx]);
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= w || y >= h) return;
    const int lx = threadIdx.x + 1, ly = threadIdx.y + 1;
    const float c = lum[ly][lx];
    uint32_t r = 0;
    r += (lum[ly - 1][lx - 1] > c) ? 1u : 0u;
    r += (lum[ly - 1][lx] > c) ? 2u : 0u;
    r += (lum[ly - 1][lx + 1] > c) ? 4u : 0u;
    r += (lum[ly][lx - 1] > c) ? 8u : 0u;
    r += (lum[ly][lx + 1] > c) ? 16u : 0u;
    r += (lum[ly + 1][lx - 1] > c) ? 32u : 0u;
    r += (lum[ly + 1][lx] > c) ? 64u : 0u;
    r += (lum[ly + 1][lx + 1] > c) ? 128u : 0u;
    census[y * cpitch + x] = (uint8_t)r;
    if (texels) texels[y * tpitch + x] = make_texel(img[y * ipitch + x], r);
}

// every level of both frames in ONE launch: a block finds its job by scanning the (at most 16) block offsets
// blockIdx.y = pair of the batch
__global__ __launch_bounds__(256) void k_census_batch(CensusBatch B, size_t pstride)
{
    int j = 0;
    while (j + 1 < B.n && (int)blockIdx.x >= B.job[j + 1].first_block) j++;
    CensusJob J = B.job[j];
    J.census = pair_ptr(J.census, pstride, blockIdx.y);
    J.texels = pair_ptr_opt(J.texels, pstride, blockIdx.y);
    J.packed = pair_ptr_opt(J.packed, pstride, blockIdx.y);
    J.img = pair_ptr(J.img, pstride, blockIdx.y);
    const int b = blockIdx.x - J.first_block, bw = (J.w + 63) / 64;
    __shared__ float lum[6][66];
    const int x0 = (b % bw) * 64, y0 = (b / bw) * 4;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    for (int t = tid; t < 6 * 66; t += 256) {
        const int ty = t / 66, tx = t % 66;
        const int cy = iclamp(y0 + ty - 1, 0, J.h - 1), cx = iclamp(x0 + tx - 1, 0, J.w - 1);
        lum[ty][tx] = lum_of(J.img[cy * J.ipitch + cx]);
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= J.w || y >= J.h) return;
    const int lx = threadIdx.x + 1, ly = threadIdx.y + 1;
    const float c = lum[ly][lx];
    uint32_t r = 0;
    r += (lum[ly - 1][lx - 1] > c) ? 1u : 0u;
    r += (lum[ly - 1][lx] > c) ? 2u : 0u;
    r += (lum[ly - 1][lx + 1] > c) ? 4u : 0u;
    r += (lum[ly][lx - 1] > c) ? 8u : 0u;
    r += (lum[ly][lx + 1] > c) ? 16u : 0u;
    r += (lum[ly + 1][lx - 1] > c) ? 32u : 0u;
    r += (lum[ly + 1][lx] > c) ? 64u : 0u;
    r += (lum[ly + 1][lx + 1] > c) ? 128u : 0u;
    J.census[y * J.cpitch + x] = (uint8_t)r;
    if (J.texels) ((float4*)J.texels)[y * J.tpitch + x] = make_texel(J.img[y * J.ipitch + x], r);
    if (J.packed) J.packed[y * J.tpitch + x] = (J.img[y * J.ipitch + x] & 0xffffffu) | (r << 24);
}
void launch_census_batch(CensusBatch& B, hipStream_t s, Batch bt)
{
    int blocks = 0;
    for (int j = 0; j < B.n; j++) {
        B.job[j].first_block = blocks;
        blocks += ((B.job[j].w + 63) / 64) * ((B.job[j].h + 3) / 4);
    }
    hipLaunchKernelGGL(k_census_batch, dim3(blocks, bt.n), dim3(64, 4), 0, s, B, bt.stride);
}

void launch_census(uint8_t* census, int cpitch, void* texels, int tpitch, const uint32_t* img, int ipitch, int w, int h, hipStream_t s)
{
    dim3 block(64, 4), grid((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(k_census, grid, block, 0, s, census, cpitch, (float4*)texels, tpitch, img, ipitch, w, h);
}

// texel plane from separately supplied image + census planes (the reference-signature launchers receive them apart)
__global__ __launch_bounds__(256) void k_pack(float4* __restrict__ texels, int tpitch, const uint32_t* __restrict__ img, int ipitch,
                                              const uint8_t* __restrict__ census, int cpitch, int w, int h)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    texels[y * tpitch + x] = make_texel(img[y * ipitch + x], census[y * cpitch + x]);
}
void launch_pack(void* texels, int tpitch, const uint32_t* img, int ipitch, const uint8_t* census, int cpitch, int w, int h, hipStream_t s)
{
    dim3 block(64, 4), grid((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(k_pack, grid, block, 0, s, (float4*)texels, tpitch, img, ipitch, census, cpitch, w, h);
}

// The values of a DeltaTab (eppm_device.cuh), computed BY the formula they replace, in place: t2[i] holds a distance d on entry and f(d)
// on return.  which = 0: the patch term's 1 - exp(-d^2 / LAMBDA_AD^2); 1: the smoothing / weighted-median weight exp(-d^2 / SIG_R^2).
__global__ __launch_bounds__(256) void k_delta_values(float* __restrict__ t2, int n, int which)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float d = t2[i];
    t2[i] = (which == 0) ? one_minus_fast_exp(div_ad2(-(d * d))) : fast_exp(div_wmf2(-(d * d)));
}
void launch_delta_values(float* t2, int n, int which, hipStream_t s)
{
    hipLaunchKernelGGL(k_delta_values, dim3((n + 255) / 256), dim3(256), 0, s, t2, n, which);
}

// column-parity planes of a 4-byte texel plane (eppm_internal.h: PlanesH::pp1): word [p][y][i] = pc[y][clamp(2i + p - pad, 0, w - 1)]
__global__ __launch_bounds__(256) void k_parity_planes(uint32_t* __restrict__ pp_, int pp_pitch, int pad, const uint32_t* __restrict__ pc_,
                                                       int pc_pitch, int w, int h, size_t pstride)
{
    uint32_t* __restrict__ pp = pair_ptr(pp_, pstride, blockIdx.z);
    const uint32_t* __restrict__ pc = pair_ptr(pc_, pstride, blockIdx.z);
    const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y >> 1, par = blockIdx.y & 1;
    if (i >= pp_pitch) return;
    const int x = iclamp(2 * i + par - pad, 0, w - 1);
    pp[((size_t)par * h + y) * pp_pitch + i] = pc[y * pc_pitch + x];
}
void launch_parity_planes(uint32_t* pp, int pp_pitch, int pad, const uint32_t* pc, int pc_pitch, int w, int h, hipStream_t s, Batch bt)
{
    dim3 block(256), grid((pp_pitch + 255) / 256, 2 * h, bt.n);
    hipLaunchKernelGGL(k_parity_planes, grid, block, 0, s, pp, pp_pitch, pad, pc, pc_pitch, w, h, bt.stride);
}

// RGB (3 B/px, tightly packed rows) -> RGBA with alpha 0 (bao_rgb2rgba, basic/bao_basic_cuda.h:258-267)
__global__ __launch_bounds__(256) void k_rgb_to_rgba(uint32_t* __restrict__ out_, int pitch, const uint8_t* __restrict__ rgb_, int h, int w, size_t pstride)
{
    uint32_t* __restrict__ out = pair_ptr(out_, pstride, blockIdx.z);
    const uint8_t* __restrict__ rgb = pair_ptr(rgb_, pstride, blockIdx.z);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* p = rgb + ((size_t)y * w + x) * 3;
    out[y * pitch + x] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}
void launch_rgb_to_rgba(uint32_t* out, int pitch_px, const uint8_t* rgb, int h, int w, hipStream_t s, Batch bt)
{
    dim3 block(256), grid((w + 255) / 256, h, bt.n);
    hipLaunchKernelGGL(k_rgb_to_rgba, grid, block, 0, s, out, pitch_px, rgb, h, w, bt.stride);
}

}  // namespace eppm
