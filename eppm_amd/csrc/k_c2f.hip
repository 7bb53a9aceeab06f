// k_c2f.hip -- coarse-to-fine step (reference: basic/bao_basic_cuda.cuh:511-537 float2 bilinear resize,
// :135-142 scalar multiply; bao_pmflow_kernel.cu:2005-2041 plane-fitting candidate refine;
// bao_pmflow_refine_kernel.cu:756-799 joint-bilateral flow smoothing).
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "eppm_device.cuh"
#include "eppm_internal.h"

namespace eppm {

#ifndef EPPM_C2F_UNROLL
#define EPPM_C2F_UNROLL 2
#endif
#ifndef EPPM_BLF_UNROLL
#define EPPM_BLF_UNROLL 21
#endif
#ifdef EPPM_TOL
#define EPPM_C2F_LOG2 true      // the tolerance library's tile kernels keep log2(gs_j gs_i): their weight is one exp2 (eppm_device.cuh)
#else
#define EPPM_C2F_LOG2 false
#endif
#define EPPM_PRAGMA_(x) _Pragma(#x)
#define EPPM_UNROLL(n) EPPM_PRAGMA_(unroll n)

__device__ __forceinline__ Planes to_dev(const PlanesH& h, size_t pstride = 0, unsigned pair = 0)
{
    Planes p;
    p.pk1 = pair_ptr((const float4*)h.pk1, pstride, pair); p.pk2 = pair_ptr((const float4*)h.pk2, pstride, pair);
    p.w = h.w; p.h = h.h; p.pitch = h.pitch;
    return p;
}

// texel (sx, sy) for an LDS tile: from the 4-byte plane when the launch has one (converted here, bit for bit what the float4
// plane holds: both come from make_texel), else from the float4 plane
__device__ __forceinline__ float4 stage_texel(const uint32_t* __restrict__ pc, const float4* __restrict__ pk, unsigned idx)
{
    if (pc) { const uint32_t w = pc[idx]; return make_texel(w, w >> 24); }
    return pk[idx];
}

// .cuh:511-537 as written (m outer over x, n inner over y), then the x post_scale of .cuh:135-142
__global__ __launch_bounds__(256) void k_resize_flow(float* __restrict__ out_, int outH, int outW, const float* __restrict__ in_,
                                                     int h, int w, float ratio, float post_scale, size_t pstride)
{
    float* __restrict__ out = pair_ptr(out_, pstride, blockIdx.z);
    const float* __restrict__ in = pair_ptr(in_, pstride, blockIdx.z);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= outW || y >= outH) return;
    const float div_scale = 1.f / ratio;
    const float fx = (float)(x + 1) * div_scale - 1;
    const float fy = (float)(y + 1) * div_scale - 1;
    const int xx = (int)fx, yy = (int)fy;
    const float dx = fmaxf(fminf(fx - xx, 1), 0);
    const float dy = fmaxf(fminf(fy - yy, 1), 0);
    float rx = 0, ry = 0;
    for (int m = 0; m <= 1; m++)
        for (int n = 0; n <= 1; n++) {
            const int u = max(0, min(w - 1, xx + m));
            const int v = max(0, min(h - 1, yy + n));
            const float sc = fabsf(1 - m - dx) * fabsf(1 - n - dy);
            rx += in[(v * w + u) * 2] * sc;
            ry += in[(v * w + u) * 2 + 1] * sc;
        }
    out[(y * outW + x) * 2] = rx * post_scale;
    out[(y * outW + x) * 2 + 1] = ry * post_scale;
}
void launch_resize_flow(float* out, int outH, int outW, const float* in, int h, int w, float ratio, float post_scale, hipStream_t s, Batch bt)
{
    dim3 block(64, 4), grid((outW + 63) / 64, (outH + 3) / 4, bt.n);
    hipLaunchKernelGGL(k_resize_flow, grid, block, 0, s, out, outH, outW, in, h, w, ratio, post_scale, bt.stride);
}

__global__ __launch_bounds__(256) void k_mul_scalar(float* __restrict__ f, float scale, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f[i] = f[i] * scale;
}
void launch_mul_scalar(float* flow, float scale, int h, int w, hipStream_t s)
{
    const int n = h * w * 2;
    hipLaunchKernelGGL(k_mul_scalar, dim3((n + 255) / 256), dim3(256), 0, s, flow, scale, n);
}

// ---------------------------------------------------------------------------------------------------
// kernel.cu:2005-2041: 3x3 integer candidates (x offset outer, y offset inner) around the truncated
// up-sampled flow; cost = min of 4 affine passes; strict < keeps the first minimum; the centre candidate
// is the initial best with cost 999999.  In place: a thread reads and writes only its own pixel.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_c2f_refine(PlanesH Ph, float* __restrict__ flow_, const float* __restrict__ lut, int R, size_t pstride)
{
    __shared__ EPPM_LUT_ALIGN PatchLut L;
    load_patch_lut(L, lut, R, threadIdx.y * kBlock + threadIdx.x, 256);
    __syncthreads();
    const Planes P = to_dev(Ph, pstride, blockIdx.z);
    float* __restrict__ flow = pair_ptr(flow_, pstride, blockIdx.z);
    const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y * kBlock + threadIdx.y;
    if (x >= P.w || y >= P.h) return;
    const float fvx = flow[(y * P.w + x) * 2], fvy = flow[(y * P.w + x) * 2 + 1];
    if (fvx > kUnknownFlowThresh || fvy > kUnknownFlowThresh) {
        flow[(y * P.w + x) * 2] = 0.0f;
        flow[(y * P.w + x) * 2 + 1] = 0.0f;
        return;
    }
    const int ccx = (int)(int16_t)(f2short(fvx) + x);
    const int ccy = (int)(int16_t)(f2short(fvy) + y);
    int bx = ccx, by = ccy;
    float min_cost = 999999;
#pragma unroll 1
    for (int m = 0; m < 3; m++) {
        const int cx = (int)(int16_t)(ccx + m - 1);
#pragma unroll 1
        for (int n = 0; n < 3; n++) {
            const int cy = (int)(int16_t)(ccy + n - 1);
            if (cx < 0 || cy < 0 || cx >= P.w || cy >= P.h) continue;
            const float cv = patch_dist_planefit(P, L, R, x, y, cx, cy);
            if (cv < min_cost) { min_cost = cv; bx = cx; by = cy; }
        }
    }
    flow[(y * P.w + x) * 2] = (float)(bx - x);
    flow[(y * P.w + x) * 2 + 1] = (float)(by - y);
}
// ---------------------------------------------------------------------------------------------------
// The same stage restructured for CDNA4 (bit-identical results).  The reference evaluates the 36
// (candidate, pass) patch costs of a pixel one after the other, re-fetching and re-converting the 100
// source samples 36 times.  Here the sample loop is outermost within a pass: the source texel comes from
// an LDS tile (16x16 + R halo, clamped at load), its conversion and its range term a^2 are computed once
// per sample and shared by the 9 candidates, whose 9 pairs of running sums advance together -- each sum
// still adds its terms in the reference's i-outer/j-inner order.  Target texels are one 16-byte gather each
// (unorm rgb + census).  The passes run 4th to 1st so the reference's nested
// __min(c1,__min(c2,__min(c3,c4))) becomes a running select with the same NaN behaviour.
// ---------------------------------------------------------------------------------------------------
// Affine passes: the target of sample (i,j) is floor(((x+j + uu) + j*A) + i*B) in float (kernel.cu:334-513).
// x+j+uu = cx+j is an integer M, and for every M an image can produce the float sum rounds so that
//   floor(...) = M + floor(fl(fl(j*A) + fl(i*B)))
// (checked exhaustively for -R <= M < 32764 by tests/test_oracle_cpu.py::test_planefit_offsets; the launcher
// falls back to the generic kernel beyond that).  So the warp of a pass is a table of integer offsets
// (dx, dy) per sample and the per-sample float coordinate arithmetic of the reference becomes one integer add.  Along a sample row dy takes at most two consecutive values: the four
// clamped row offsets a row can need are formed once per row and a per-sample flag picks three of them.
struct C2fOff { int dx16, up; };     // dx16 = (j + x-offset) * 16 (bytes); up = 1 if this sample's dy is the row minimum + 1

template <int R>
struct C2fTables {
    static constexpr int S = R + 1;
    C2fOff off[3][S * S];             // passes 1..3
    int rowdy[3][S];                  // i + min over the row of the y-offset
};

// The tables depend only on the radius and on the reference's coefficients: they are compile-time constants in
// __constant__ memory.  Their index is wave-uniform, so they arrive through the scalar cache (s_load) and cost
// no vector instruction; `up` steers a scalar branch.
constexpr int cfloor(float v) { const int t = (int)v; return ((float)t > v) ? t - 1 : t; }
template <int R>
constexpr C2fTables<R> make_c2f_tables()
{
    constexpr int S = R + 1;
    constexpr float kc[3][4] = {
        {0.177f, -0.011f, -0.003f, 0.301f},
        {0.125f, -0.357f, 0.009f, 0.308f},
        {0.205f, 0.370f, 0.011f, 0.296f},
    };
    C2fTables<R> T{};
    for (int p = 0; p < 3; p++)
        for (int ii = 0; ii < S; ii++) {
            const int i = 2 * ii - R;
            int dy[S] = {};
            int lo = 1 << 30;
            for (int jj = 0; jj < S; jj++) {
                const int j = 2 * jj - R;
                const float fx = (float)(j)*kc[p][0] + (float)(i)*kc[p][1];
                const float fy = (float)(j)*kc[p][2] + (float)(i)*kc[p][3];
                T.off[p][ii * S + jj].dx16 = (j + cfloor(fx)) * 16;
                dy[jj] = i + cfloor(fy);
                lo = dy[jj] < lo ? dy[jj] : lo;
            }
            T.rowdy[p][ii] = lo;
            for (int jj = 0; jj < S; jj++) T.off[p][ii * S + jj].up = dy[jj] - lo;      // 0 or 1
        }
    return T;
}
__constant__ C2fTables<9> c2f_tab9 = make_c2f_tables<9>();
__constant__ C2fTables<17> c2f_tab17 = make_c2f_tables<17>();
template <int R> __device__ __forceinline__ const C2fTables<R>& c2f_tables();
template <> __device__ __forceinline__ const C2fTables<9>& c2f_tables<9>() { return c2f_tab9; }
template <> __device__ __forceinline__ const C2fTables<17>& c2f_tables<17>() { return c2f_tab17; }

// WIN: the target texels come from an LDS window of the target image instead of per-lane gathers (k_c2f_refine_win):
// `s_win` is the window (row stride WW texels, cells pre-clamped to the image at load), `wbase` this lane's byte offset of
// candidate (cx, ccy-1) with zero sample offset.  The sample's offset is a scalar, the three row candidates are WW texels apart.
template <int R, int PASS, bool RAW = false, bool WIN = false, int WW = 0>
__device__ __forceinline__ void c2f_pass(const Planes& P, const PatchLutT<R + 1>& L, const float4* __restrict__ s_src,
                                         int TW, int tx, int ty, int cx16, int wmax16, int ccy, const rgbf c1, const rgbf (&c2)[3], float (&run)[3],
                                         const float4* __restrict__ s_win = nullptr, int wbase = 0)
{
    // candidates (cx, ccy-1), (cx, ccy), (cx, ccy+1): one x offset m, the three y offsets n
    constexpr int S = R + 1;
    constexpr int TP = (PASS == 0) ? 0 : PASS - 1;
    const C2fTables<R>& T = c2f_tables<R>();
#ifdef EPPM_TOL
    const float* __restrict__ lg2 = L.gsp;          // log2(gs_j gs_i), load_patch_lut<true>
    const rgbf c1s = tol_scale_centre(c1), c2s[3] = {tol_scale_centre(c2[0]), tol_scale_centre(c2[1]), tol_scale_centre(c2[2])};
#endif
    float cs[3] = {0.0f, 0.0f, 0.0f}, ws[3] = {0.0f, 0.0f, 0.0f};
    const unsigned pitch16 = (unsigned)P.pitch << 4;
#pragma unroll 1
    for (int ii = 0; ii < S; ii++) {
        // clamped row offsets of the targets: rows ccy-1+dy .. ccy+1+dy (+1 more where dy steps inside the row)
        unsigned Rr[4];
        const int rdy = (PASS == 0) ? 2 * ii - R : T.rowdy[TP][ii];
        const int rb = ccy - 1 + rdy;
        if (!WIN) {
#pragma unroll
            for (int k = 0; k < 4; k++) Rr[k] = __umul24((unsigned)iclamp(rb + k, 0, P.h - 1), pitch16);
        }
EPPM_UNROLL(EPPM_C2F_UNROLL)
        for (int jj = 0; jj < S; jj++) {
            const float4 q1 = s_src[(ty + 2 * ii) * TW + tx + 2 * jj];
            const rgbf p1 = texel_rgb(q1);
            const uint32_t k1 = __float_as_uint(q1.w);
#ifdef EPPM_TOL
            const float lsrc = tol_exp_arg_scaled(c1s, p1, lg2[ii * S + jj]);                 // log2 of the source half of the weight: once per sample
#else
            float a2 = max_abs_diff(c1, p1);
            a2 *= a2;
            const float gsp = L.gsp[ii * S + jj];
#endif
            // byte offsets of the three targets: clamped column (in bytes throughout: no shift) + clamped rows
            const int dx16 = (PASS == 0) ? (2 * jj - R) * 16 : T.off[TP][ii * S + jj].dx16;
            float4 q2[3];                                 // the three gathers are issued back to back, then consumed
            if (WIN) {
                // one vector add (lane base + scalar sample offset), three LDS reads with immediate row offsets
                const int soff = (rdy + ((PASS != 0) ? T.off[TP][ii * S + jj].up : 0)) * (WW * 16) + dx16;
                const char* wp = reinterpret_cast<const char*>(s_win) + (wbase + soff);
#pragma unroll
                for (int n = 0; n < 3; n++) q2[n] = *reinterpret_cast<const float4*>(wp + n * (WW * 16));
            } else {
            const unsigned Xb = (unsigned)med3i(cx16 + dx16, 0, wmax16);
            if (PASS != 0 && T.off[TP][ii * S + jj].up) {         // wave-uniform; compiles to three selects on a scalar condition
#pragma unroll
                for (int n = 0; n < 3; n++) q2[n] = texel_at(P.pk2, Rr[n + 1] + Xb);
            } else {
#pragma unroll
                for (int n = 0; n < 3; n++) q2[n] = texel_at(P.pk2, Rr[n] + Xb);
            }
            }
#pragma unroll
            for (int n = 0; n < 3; n++) {
                const rgbf p2 = texel_rgb(q2[n]);
#ifdef EPPM_TOL
                const float cost = tol_cost(L.tab(), p1, p2, k1, __float_as_uint(q2[n].w));
                patch_accum(cs[n], ws[n], cost, __builtin_amdgcn_exp2f(tol_exp_arg_scaled(c2s[n], p2, lsrc)));
#else
                float cost = max_abs_diff(p1, p2);
                cost = EPPM_DELTA_PATCH ? delta_lookup(L.D, cost) : one_minus_fast_exp(div_ad2(-(cost * cost)));      // the same bits either way (eppm_device.cuh: DeltaTab)
                cost += census_cost(L.cnx, k1, __float_as_uint(q2[n].w));
                float temp = max_abs_diff(c2[n], p2);
                temp *= temp;
                float weight = fast_exp(div_ad2(-(a2 + temp)));
                weight *= gsp;
                cost *= weight;
                cs[n] += cost;
                ws[n] += weight;
#endif
            }
        }
    }
#pragma unroll
    for (int n = 0; n < 3; n++) {
        const float c = cs[n] / ws[n];
        run[n] = (RAW || PASS == 3) ? c : ((c < run[n]) ? c : run[n]);
    }
}

// Waves per SIMD the register allocator plans for.  LDS would allow 5 (29 KB per workgroup), but at 4 the 116-VGPR
// schedule keeps more gathers in flight per wave and is 1.5 % faster than the 92-VGPR one (A/B on one box,
// tools/gpu.sh ab); unroll 1 / 5 of the sample loop and 3 waves are slower, and so is a software-pipelined loop
// that issues the gathers of the next sample pair before computing the current one (161 VGPRs, +6 % instructions,
// +3.5 % time: the kernel waits on VALU issue, not on memory).
#ifndef EPPM_C2F_WAVES
#define EPPM_C2F_WAVES 4
#endif
#ifndef EPPM_C2F_WAVES_MIN
#define EPPM_C2F_WAVES_MIN 2
#endif
#define EPPM_C2F_OCC __attribute__((amdgpu_waves_per_eu(EPPM_C2F_WAVES_MIN, EPPM_C2F_WAVES)))     // (min, max): radius 17 only fits 2
// SPLIT: a launch with few tiles (fewer than 256: under one wave per SIMD) does not fill the chip and runs latency bound.
// Then a tile is given to 3 workgroups (one candidate column m each) or to 4 (one affine pass
// each), whichever divides more evenly over the 256 CUs; the costs of a pixel (9 x 4 passes) go to a scratch plane and
// k_c2f_select replays the reference's nested minimum and candidate loop.  Same costs, same selection order.
template <int R, int SPLIT>
__global__ __launch_bounds__(256) EPPM_C2F_OCC void k_c2f_refine_tiled(PlanesH Ph, float* __restrict__ flow_, const float* __restrict__ lut,
                                                                       float* __restrict__ cost9_, size_t pstride)
{
    float* __restrict__ flow = pair_ptr(flow_, pstride, blockIdx.y);             // blockIdx.y = pair of the batch
    float* __restrict__ cost9 = pair_ptr_opt(cost9_, pstride, blockIdx.y);
    constexpr int TWU = kBlock + 2 * R;                 // used tile width
    // row stride padded to a multiple of 16 texels (256 B): a ds_read_b128 wave access is served in groups made
    // of 8 lanes of one tile row and 8 of the next (MI355X LDS lane groups); with the stride = 0 mod 256 B the two
    // halves fall on disjoint banks (34-texel rows cost a 2-way conflict on about every read)
    constexpr int TW = (TWU + 15) / 16 * 16;
    __shared__ EPPM_LUT_ALIGN PatchLutT<R + 1> L;
    __shared__ float4 s_src[TWU * TW];
    const int tid = threadIdx.y * kBlock + threadIdx.x;
    load_patch_lut<EPPM_C2F_LOG2>(L, lut, R, tid, 256);
    const Planes P = to_dev(Ph, pstride, blockIdx.y);
    // XCD-aware tile order: workgroups are dealt round robin over the 8 XCDs (b % 8), each with its own L2.
    // Give XCD k the k-th contiguous eighth of the row-major tile list so that neighbouring tiles -- which
    // share their R-pixel halos and their target windows -- hit the same L2 (speed only, never correctness).
    const int tiles_x = (P.w + kBlock - 1) / kBlock, tiles = tiles_x * ((P.h + kBlock - 1) / kBlock);
    const int per_xcd = (tiles + 7) / 8;
    const int slot = SPLIT ? (blockIdx.x >> 3) / SPLIT : (blockIdx.x >> 3);  // the workgroups of a tile share an XCD
    const int part = SPLIT ? (blockIdx.x >> 3) % SPLIT : -1;
    const int m_only = (SPLIT == 3) ? part : -1;
    const int tile = (blockIdx.x & 7) * per_xcd + slot;
    if (slot >= per_xcd || tile >= tiles) return;
    const int x0 = (tile % tiles_x) * kBlock, y0 = (tile / tiles_x) * kBlock;
    for (int t = tid; t < TWU * TWU; t += 256) {
        const int ry = t / TWU, rx = t % TWU;
        const int sy = iclamp(y0 + ry - R, 0, P.h - 1), sx = iclamp(x0 + rx - R, 0, P.w - 1);
        s_src[ry * TW + rx] = P.pk1[(unsigned)(sy * P.pitch + sx)];
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= P.w || y >= P.h) return;
    const float fvx = flow[(y * P.w + x) * 2], fvy = flow[(y * P.w + x) * 2 + 1];
    if (fvx > kUnknownFlowThresh || fvy > kUnknownFlowThresh) {
        if (SPLIT == 0) {
            flow[(y * P.w + x) * 2] = 0.0f;
            flow[(y * P.w + x) * 2 + 1] = 0.0f;
        }
        return;
    }
    const int ccx = (int)(int16_t)(f2short(fvx) + x);
    const int ccy = (int)(int16_t)(f2short(fvy) + y);
    const rgbf c1 = texel_rgb(s_src[(threadIdx.y + R) * TW + threadIdx.x + R]);
    int bx = ccx, by = ccy;
    float min_cost = 999999;
#pragma unroll 1
    for (int m = 0; m < 3; m++) {                    // x offset outer, as the reference's candidate loop (kernel.cu:2028)
        if (SPLIT == 3 && m != m_only) continue;
        const int cx = (int)(int16_t)(ccx + m - 1);
        if (cx < 0 || cx >= P.w) continue;           // every candidate of this column is skipped (:2030)
        rgbf c2[3];
#pragma unroll
        for (int n = 0; n < 3; n++) c2[n] = texel_rgb(tex_px(P.pk2, P.pitch, P.w, P.h, cx, ccy + n - 1));
        float run[3];
        const int cx16 = cx << 4, wmax16 = (P.w - 1) << 4;
        if (SPLIT == 4) {                            // this workgroup's pass only, raw cost
            if (part == 3) c2f_pass<R, 3, true>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
            else if (part == 2) c2f_pass<R, 2, true>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
            else if (part == 1) c2f_pass<R, 1, true>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
            else c2f_pass<R, 0, true>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
#pragma unroll
            for (int n = 0; n < 3; n++) cost9[(size_t)(y * P.w + x) * 36 + part * 9 + m * 3 + n] = run[n];
            continue;
        }
        c2f_pass<R, 3>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
        c2f_pass<R, 2>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
        c2f_pass<R, 1>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
        c2f_pass<R, 0>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
        if (SPLIT == 3) {
#pragma unroll
            for (int n = 0; n < 3; n++) cost9[(size_t)(y * P.w + x) * 36 + m * 3 + n] = run[n];      // nested minimum already formed
            continue;
        }
#pragma unroll
        for (int n = 0; n < 3; n++) {
            const int cy = (int)(int16_t)(ccy + n - 1);
            if (cy < 0 || cy >= P.h) continue;
            const float cv = run[n];
            if (cv < min_cost) { min_cost = cv; bx = cx; by = cy; }
        }
    }
    if (SPLIT != 0) return;
    flow[(y * P.w + x) * 2] = (float)(bx - x);
    flow[(y * P.w + x) * 2 + 1] = (float)(by - y);
}


// ---------------------------------------------------------------------------------------------------
// The same tile kernel with the TARGET texels staged in LDS as well.  After up-sampling, the flow of a 16x16 tile is
// nearly constant, so the 3 600 target texels a pixel reads (9 candidates x 4 passes x 100 samples) and those of its
// 255 neighbours fall into one small window of the target image: (tile + flow spread) + candidate +-1 + the sample and
// warp offsets.  A workgroup whose candidate centres span at most WIN_SPAN_X x WIN_SPAN_Y pixels loads that window once
// (rows clamped to the image at load, like the source tile) and every target fetch becomes ONE vector add and an LDS read
// with an immediate offset -- no clamps, no per-candidate address arithmetic, no gather round trips through the texture
// path (the 16-byte gathers kept its addresser ~58 % busy).  A tile whose flow is not coherent enough (motion boundaries)
// takes the per-access path of k_c2f_refine_tiled inside the same workgroup.  Same arithmetic, same order: bit-identical.
// ---------------------------------------------------------------------------------------------------
template <int R>
struct C2fWinGeom {
    // extreme sample offsets over all four passes (pass 0: the plain +-R grid)
    static constexpr int xlo() { int v = -R; const auto T = make_c2f_tables<R>(); for (int p = 0; p < 3; p++) for (int t = 0; t < (R + 1) * (R + 1); t++) v = T.off[p][t].dx16 / 16 < v ? T.off[p][t].dx16 / 16 : v; return v; }
    static constexpr int xhi() { int v = R; const auto T = make_c2f_tables<R>(); for (int p = 0; p < 3; p++) for (int t = 0; t < (R + 1) * (R + 1); t++) v = T.off[p][t].dx16 / 16 > v ? T.off[p][t].dx16 / 16 : v; return v; }
    static constexpr int ylo() { int v = -R; const auto T = make_c2f_tables<R>(); for (int p = 0; p < 3; p++) for (int i = 0; i <= R; i++) v = T.rowdy[p][i] < v ? T.rowdy[p][i] : v; return v; }
    static constexpr int yhi() { int v = R; const auto T = make_c2f_tables<R>(); for (int p = 0; p < 3; p++) for (int i = 0; i <= R; i++) v = T.rowdy[p][i] + 1 > v ? T.rowdy[p][i] + 1 : v; return v; }
};
#ifndef EPPM_C2F_WIN_H
#ifdef EPPM_TOL
#define EPPM_C2F_WIN_H 50
#else
#define EPPM_C2F_WIN_H 48      // the data term's table (4.2 KB, eppm_device.cuh: DeltaTab) and two workgroups per CU: 48 window rows (admissible spread 21)
#endif
#endif
#ifndef EPPM_C2F_PASS2
#define EPPM_C2F_PASS2 1
#endif
#ifndef EPPM_C2F_WIN_WAVES
#define EPPM_C2F_WIN_WAVES 4
#endif

// Two affine passes of one candidate column evaluated together from the LDS window: the source sample, its range term a^2 and
// the spatial weight are formed once per sample for the 6 (pass, row candidate) terms; each of the 6 pairs of running sums
// still adds its terms in the reference's sample order.  outA / outB: raw costs of pass PA / PB for the three row candidates.
template <int R, int PA, int PB, int WW>
__device__ __forceinline__ void c2f_pass2_win(const PatchLutT<R + 1>& L, const float4* __restrict__ s_src, int TW, int tx, int ty,
                                              const rgbf c1, const rgbf (&c2)[3], const float4* __restrict__ s_win, int wbase,
                                              float (&outA)[3], float (&outB)[3])
{
    constexpr int S = R + 1;
    constexpr int TA = (PA == 0) ? 0 : PA - 1, TB = (PB == 0) ? 0 : PB - 1;
    const C2fTables<R>& T = c2f_tables<R>();
#ifdef EPPM_TOL
    const float* __restrict__ lg2 = L.gsp;          // log2(gs_j gs_i), load_patch_lut<true>
    const rgbf c1s = tol_scale_centre(c1), c2s[3] = {tol_scale_centre(c2[0]), tol_scale_centre(c2[1]), tol_scale_centre(c2[2])};
#endif
    float csA[3] = {0.0f, 0.0f, 0.0f}, wsA[3] = {0.0f, 0.0f, 0.0f}, csB[3] = {0.0f, 0.0f, 0.0f}, wsB[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int ii = 0; ii < S; ii++) {
        const int rdyA = (PA == 0) ? 2 * ii - R : T.rowdy[TA][ii];
        const int rdyB = (PB == 0) ? 2 * ii - R : T.rowdy[TB][ii];
EPPM_UNROLL(EPPM_C2F_UNROLL)
        for (int jj = 0; jj < S; jj++) {
            const float4 q1 = s_src[(ty + 2 * ii) * TW + tx + 2 * jj];
            const rgbf p1 = texel_rgb(q1);
            const uint32_t k1 = __float_as_uint(q1.w);
#ifdef EPPM_TOL
            const float lsrc = tol_exp_arg_scaled(c1s, p1, lg2[ii * S + jj]);                 // log2 of the source half of the weight: once per 6 terms
#else
            float a2 = max_abs_diff(c1, p1);
            a2 *= a2;
            const float gsp = L.gsp[ii * S + jj];
#endif
            const int soffA = (rdyA + ((PA != 0) ? T.off[TA][ii * S + jj].up : 0)) * (WW * 16) + ((PA == 0) ? (2 * jj - R) * 16 : T.off[TA][ii * S + jj].dx16);
            const int soffB = (rdyB + ((PB != 0) ? T.off[TB][ii * S + jj].up : 0)) * (WW * 16) + ((PB == 0) ? (2 * jj - R) * 16 : T.off[TB][ii * S + jj].dx16);
            const char* wa = reinterpret_cast<const char*>(s_win) + (wbase + soffA);
            const char* wb = reinterpret_cast<const char*>(s_win) + (wbase + soffB);
            float4 qa[3], qb[3];
#pragma unroll
            for (int n = 0; n < 3; n++) { qa[n] = *reinterpret_cast<const float4*>(wa + n * (WW * 16)); qb[n] = *reinterpret_cast<const float4*>(wb + n * (WW * 16)); }
#pragma unroll
            for (int n = 0; n < 3; n++) {
#ifdef EPPM_TOL
                {
                    const rgbf p2 = texel_rgb(qa[n]);
                    const float cost = tol_cost(L.tab(), p1, p2, k1, __float_as_uint(qa[n].w));
                    patch_accum(csA[n], wsA[n], cost, __builtin_amdgcn_exp2f(tol_exp_arg_scaled(c2s[n], p2, lsrc)));
                }
                {
                    const rgbf p2 = texel_rgb(qb[n]);
                    const float cost = tol_cost(L.tab(), p1, p2, k1, __float_as_uint(qb[n].w));
                    patch_accum(csB[n], wsB[n], cost, __builtin_amdgcn_exp2f(tol_exp_arg_scaled(c2s[n], p2, lsrc)));
                }
#else
                {
                    const rgbf p2 = texel_rgb(qa[n]);
                    float cost = max_abs_diff(p1, p2);
                    cost = EPPM_DELTA_PATCH ? delta_lookup(L.D, cost) : one_minus_fast_exp(div_ad2(-(cost * cost)));
                    cost += census_cost(L.cnx, k1, __float_as_uint(qa[n].w));
                    float temp = max_abs_diff(c2[n], p2);
                    temp *= temp;
                    float weight = fast_exp(div_ad2(-(a2 + temp)));
                    weight *= gsp;
                    cost *= weight;
                    csA[n] += cost;
                    wsA[n] += weight;
                }
                {
                    const rgbf p2 = texel_rgb(qb[n]);
                    float cost = max_abs_diff(p1, p2);
                    cost = EPPM_DELTA_PATCH ? delta_lookup(L.D, cost) : one_minus_fast_exp(div_ad2(-(cost * cost)));
                    cost += census_cost(L.cnx, k1, __float_as_uint(qb[n].w));
                    float temp = max_abs_diff(c2[n], p2);
                    temp *= temp;
                    float weight = fast_exp(div_ad2(-(a2 + temp)));
                    weight *= gsp;
                    cost *= weight;
                    csB[n] += cost;
                    wsB[n] += weight;
                }
#endif
            }
        }
    }
#pragma unroll
    for (int n = 0; n < 3; n++) { outA[n] = csA[n] / wsA[n]; outB[n] = csB[n] / wsB[n]; }
}

// Workgroup = 512 threads = the 256 pixels of the tile x 2 pass groups (threadIdx.z): group 0 evaluates the 4th and 3rd affine
// pass of every candidate, group 1 the 2nd and the 1st; both read the same source tile and target window, so the LDS footprint
// (78 KB) is shared by twice the waves: two workgroups per CU = 4 waves per SIMD (the one-group form ran at 2 and was 10 %
// slower than the gather kernel).  Group 0 hands its nested minimum of passes 4 and 3 to group 1 through LDS (the source
// tile's storage, after a barrier), which finishes __min(c1,__min(c2,.)) and the candidate loop.
template <int R>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(EPPM_C2F_WIN_WAVES, EPPM_C2F_WIN_WAVES)))
void k_c2f_refine_win(PlanesH Ph, float* __restrict__ flow_, const float* __restrict__ lut, size_t pstride)
{
    constexpr int TWU = kBlock + 2 * R;
    constexpr int TW = (TWU + 15) / 16 * 16;
    constexpr int WW = 64, WH = EPPM_C2F_WIN_H;                      // window: row stride 64 texels = 1 KiB (conflict-free ds_read_b128)
    constexpr int XLO = C2fWinGeom<R>::xlo(), XHI = C2fWinGeom<R>::xhi(), YLO = C2fWinGeom<R>::ylo(), YHI = C2fWinGeom<R>::yhi();
    // admissible spread of the candidate centres (max - min): a read lands at window column (cx + dx) - wx0 with cx in [mnx-1, mxx+1],
    // dx in [XLO, XHI] and wx0 = mnx - 1 + XLO, i.e. at most (mxx - mnx) + 2 + (XHI - XLO), which must stay <= WW - 1 (rows likewise)
    constexpr int SPAN_X = WW - 3 - (XHI - XLO), SPAN_Y = WH - 3 - (YHI - YLO);
    static_assert(SPAN_X >= kBlock - 1 && SPAN_Y >= kBlock - 1, "window too small for a constant-flow tile (spread kBlock - 1)");
    static_assert(TWU * TW * 16 >= 9 * 256 * 4, "the exchange buffer aliases the source tile");
    __shared__ EPPM_LUT_ALIGN PatchLutT<R + 1> L;
    __shared__ float4 s_src[TWU * TW];
    __shared__ float4 s_win[WH * WW];
    __shared__ int s_mm[4];                                          // min ccx, max ccx, min ccy, max ccy of the tile's pixels
    float* __restrict__ flow = pair_ptr(flow_, pstride, blockIdx.y);
    const int ptid = threadIdx.y * kBlock + threadIdx.x;            // pixel of the tile
    const int grp = threadIdx.z;                                     // pass group
    const int tid = grp * 256 + ptid;
    load_patch_lut<EPPM_C2F_LOG2>(L, lut, R, tid, 512);
    const Planes P = to_dev(Ph, pstride, blockIdx.y);
    const uint32_t* __restrict__ pc1 = pair_ptr_opt(Ph.pc1, pstride, blockIdx.y);
    const uint32_t* __restrict__ pc2 = pair_ptr_opt(Ph.pc2, pstride, blockIdx.y);
    const int tiles_x = (P.w + kBlock - 1) / kBlock, tiles = tiles_x * ((P.h + kBlock - 1) / kBlock);
    const int per_xcd = (tiles + 7) / 8;
    const int slot = blockIdx.x >> 3;
    const int tile = (blockIdx.x & 7) * per_xcd + slot;              // XCD-aware tile order, as k_c2f_refine_tiled
    if (slot >= per_xcd || tile >= tiles) return;
    const int x0 = (tile % tiles_x) * kBlock, y0 = (tile / tiles_x) * kBlock;
    if (tid == 0) { s_mm[0] = 0x7fffffff; s_mm[1] = -0x7fffffff; s_mm[2] = 0x7fffffff; s_mm[3] = -0x7fffffff; }
    for (int t = tid; t < TWU * TWU; t += 512) {
        const int ry = t / TWU, rx = t % TWU;
        const int sy = iclamp(y0 + ry - R, 0, P.h - 1), sx = iclamp(x0 + rx - R, 0, P.w - 1);
        s_src[ry * TW + rx] = stage_texel(pc1, P.pk1, (unsigned)(sy * P.pitch + sx));
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    const bool inimg = (x < P.w && y < P.h);
    float fvx = 0.0f, fvy = 0.0f;
    if (inimg) { fvx = flow[(y * P.w + x) * 2]; fvy = flow[(y * P.w + x) * 2 + 1]; }
    const bool known = inimg && !(fvx > kUnknownFlowThresh || fvy > kUnknownFlowThresh);
    const int ccx = (int)(int16_t)(f2short(fvx) + x);
    const int ccy = (int)(int16_t)(f2short(fvy) + y);
    if (known && grp == 0) {
        atomicMin(&s_mm[0], ccx); atomicMax(&s_mm[1], ccx);
        atomicMin(&s_mm[2], ccy); atomicMax(&s_mm[3], ccy);
    }
    __syncthreads();
    const int mnx = s_mm[0], mxx = s_mm[1], mny = s_mm[2], mxy = s_mm[3];
    const bool coherent = (mxx - mnx <= SPAN_X) && (mxy - mny <= SPAN_Y);       // workgroup-uniform (no known pixel: mxx < mnx, nobody reads)
    const int wx0 = mnx - 1 + XLO, wy0 = mny - 1 + YLO;
    if (coherent && mxx >= mnx) {
        for (int t = tid; t < WH * WW; t += 512) {
            const int sy = iclamp(wy0 + t / WW, 0, P.h - 1), sx = iclamp(wx0 + t % WW, 0, P.w - 1);
            s_win[t] = stage_texel(pc2, P.pk2, (unsigned)(sy * P.pitch + sx));
        }
    }
    __syncthreads();
    float res[9];                                   // group 0: __min(c3,c4) per candidate; group 1: c2 then the final cost
    float res1[9];                                  // group 1: c1 (raw)
    if (known) {
        const rgbf c1 = texel_rgb(s_src[(threadIdx.y + R) * TW + threadIdx.x + R]);
#pragma unroll
        for (int m = 0; m < 3; m++) {               // x offset outer, as the reference's candidate loop (kernel.cu:2028)
            const int cx = (int)(int16_t)(ccx + m - 1);
            float run[3] = {0.0f, 0.0f, 0.0f}, raw[3] = {0.0f, 0.0f, 0.0f};
            if (!(cx < 0 || cx >= P.w)) {            // else: every candidate of this column is skipped (:2030)
                rgbf c2[3];
                const int cx16 = cx << 4, wmax16 = (P.w - 1) << 4;
                if (coherent) {
                    const int wbase = ((ccy - 1 - wy0) * WW + (cx - wx0)) * 16;
#pragma unroll
                    for (int n = 0; n < 3; n++) c2[n] = texel_rgb(*reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_win) + wbase + n * (WW * 16)));
#if EPPM_C2F_PASS2
                    if (grp == 0) {
                        float c4[3], c3[3];
                        c2f_pass2_win<R, 3, 2, WW>(L, s_src, TW, threadIdx.x, threadIdx.y, c1, c2, s_win, wbase, c4, c3);
#pragma unroll
                        for (int n = 0; n < 3; n++) run[n] = (c3[n] < c4[n]) ? c3[n] : c4[n];       // __min(cost3, cost4)
                    } else {
                        c2f_pass2_win<R, 1, 0, WW>(L, s_src, TW, threadIdx.x, threadIdx.y, c1, c2, s_win, wbase, run, raw);
                    }
#else
                    if (grp == 0) {
                        c2f_pass<R, 3, false, true, WW>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run, s_win, wbase);
                        c2f_pass<R, 2, false, true, WW>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run, s_win, wbase);
                    } else {
                        c2f_pass<R, 1, true, true, WW>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run, s_win, wbase);
                        c2f_pass<R, 0, true, true, WW>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, raw, s_win, wbase);
                    }
#endif
                } else {
#pragma unroll
                    for (int n = 0; n < 3; n++) c2[n] = texel_rgb(tex_px(P.pk2, P.pitch, P.w, P.h, cx, ccy + n - 1));
                    if (grp == 0) {
                        c2f_pass<R, 3>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
                        c2f_pass<R, 2>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
                    } else {
                        c2f_pass<R, 1, true>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
                        c2f_pass<R, 0, true>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, raw);
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < 3; n++) { res[m * 3 + n] = run[n]; res1[m * 3 + n] = raw[n]; }
        }
    }
    __syncthreads();                                // every read of the source tile is done: its storage carries the exchange
    float* __restrict__ xch = reinterpret_cast<float*>(s_src);
    if (grp == 0 && known) {
#pragma unroll
        for (int k = 0; k < 9; k++) xch[k * 256 + ptid] = res[k];
    }
    __syncthreads();
    if (grp != 1 || !inimg) return;
    if (!known) {
        flow[(y * P.w + x) * 2] = 0.0f;
        flow[(y * P.w + x) * 2 + 1] = 0.0f;
        return;
    }
    int bx = ccx, by = ccy;
    float min_cost = 999999;
#pragma unroll
    for (int m = 0; m < 3; m++) {
        const int cx = (int)(int16_t)(ccx + m - 1);
        if (cx < 0 || cx >= P.w) continue;
#pragma unroll
        for (int n = 0; n < 3; n++) {
            const int cy = (int)(int16_t)(ccy + n - 1);
            if (cy < 0 || cy >= P.h) continue;
            const float m34 = xch[(m * 3 + n) * 256 + ptid];
            const float c_2 = res[m * 3 + n], c_1 = res1[m * 3 + n];
            const float m234 = (c_2 < m34) ? c_2 : m34;                // __min(cost1,__min(cost2,__min(cost3,cost4))), kernel.cu:512
            const float cv = (c_1 < m234) ? c_1 : m234;
            if (cv < min_cost) { min_cost = cv; bx = cx; by = cy; }
        }
    }
    flow[(y * P.w + x) * 2] = (float)(bx - x);
    flow[(y * P.w + x) * 2 + 1] = (float)(by - y);
}

// The same for large radii (PATCH_R 17: source tile 50x64 texels = 51 KB, target window 80x72 texels = 92 KB: one workgroup per
// CU): 1024 threads = 256 pixels x 4 pass groups, one affine pass each, so that the one resident workgroup still gives 4 waves
// per SIMD.  Groups 1..3 hand their raw pass costs to group 0 through the source tile's storage.
template <int R>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_c2f_refine_win4(PlanesH Ph, float* __restrict__ flow_, const float* __restrict__ lut, size_t pstride)
{
    constexpr int TWU = kBlock + 2 * R;
    constexpr int TW = (TWU + 15) / 16 * 16;
    constexpr int XLO = C2fWinGeom<R>::xlo(), XHI = C2fWinGeom<R>::xhi(), YLO = C2fWinGeom<R>::ylo(), YHI = C2fWinGeom<R>::yhi();
    constexpr int WW = (kBlock + 2 + (XHI - XLO) + 8 + 15) / 16 * 16;            // >= 8 px of admissible flow spread, row stride a multiple of 256 B
    constexpr int WH = kBlock + 2 + (YHI - YLO) + 7;
    constexpr int SPAN_X = WW - 3 - (XHI - XLO), SPAN_Y = WH - 3 - (YHI - YLO);          // see k_c2f_refine_win
    static_assert(SPAN_X >= kBlock - 1 && SPAN_Y >= kBlock - 1, "window too small for a constant-flow tile (spread kBlock - 1)");
    static_assert(TWU * TW * 16 >= 27 * 256 * 4, "the exchange buffer aliases the source tile");
    static_assert(sizeof(PatchLutT<R + 1>) + (TWU * TW + WH * WW) * 16 + 16 <= 160 * 1024, "LDS budget of one CU");
    __shared__ EPPM_LUT_ALIGN PatchLutT<R + 1> L;
    __shared__ float4 s_src[TWU * TW];
    __shared__ float4 s_win[WH * WW];
    __shared__ int s_mm[4];
    float* __restrict__ flow = pair_ptr(flow_, pstride, blockIdx.y);
    const int ptid = threadIdx.y * kBlock + threadIdx.x;
    const int grp = threadIdx.z;                                     // pass group: evaluates pass 3 - grp (0-based: 3 = the 4th pass)
    const int tid = grp * 256 + ptid;
    load_patch_lut<EPPM_C2F_LOG2>(L, lut, R, tid, 1024);
    const Planes P = to_dev(Ph, pstride, blockIdx.y);
    const uint32_t* __restrict__ pc1 = pair_ptr_opt(Ph.pc1, pstride, blockIdx.y);
    const uint32_t* __restrict__ pc2 = pair_ptr_opt(Ph.pc2, pstride, blockIdx.y);
    const int tiles_x = (P.w + kBlock - 1) / kBlock, tiles = tiles_x * ((P.h + kBlock - 1) / kBlock);
    const int per_xcd = (tiles + 7) / 8;
    const int slot = blockIdx.x >> 3;
    const int tile = (blockIdx.x & 7) * per_xcd + slot;
    if (slot >= per_xcd || tile >= tiles) return;
    const int x0 = (tile % tiles_x) * kBlock, y0 = (tile / tiles_x) * kBlock;
    if (tid == 0) { s_mm[0] = 0x7fffffff; s_mm[1] = -0x7fffffff; s_mm[2] = 0x7fffffff; s_mm[3] = -0x7fffffff; }
    for (int t = tid; t < TWU * TWU; t += 1024) {
        const int ry = t / TWU, rx = t % TWU;
        const int sy = iclamp(y0 + ry - R, 0, P.h - 1), sx = iclamp(x0 + rx - R, 0, P.w - 1);
        s_src[ry * TW + rx] = stage_texel(pc1, P.pk1, (unsigned)(sy * P.pitch + sx));
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    const bool inimg = (x < P.w && y < P.h);
    float fvx = 0.0f, fvy = 0.0f;
    if (inimg) { fvx = flow[(y * P.w + x) * 2]; fvy = flow[(y * P.w + x) * 2 + 1]; }
    const bool known = inimg && !(fvx > kUnknownFlowThresh || fvy > kUnknownFlowThresh);
    const int ccx = (int)(int16_t)(f2short(fvx) + x);
    const int ccy = (int)(int16_t)(f2short(fvy) + y);
    if (known && grp == 0) {
        atomicMin(&s_mm[0], ccx); atomicMax(&s_mm[1], ccx);
        atomicMin(&s_mm[2], ccy); atomicMax(&s_mm[3], ccy);
    }
    __syncthreads();
    const int mnx = s_mm[0], mxx = s_mm[1], mny = s_mm[2], mxy = s_mm[3];
    const bool coherent = (mxx - mnx <= SPAN_X) && (mxy - mny <= SPAN_Y);
    const int wx0 = mnx - 1 + XLO, wy0 = mny - 1 + YLO;
    if (coherent && mxx >= mnx) {
        for (int t = tid; t < WH * WW; t += 1024) {
            const int sy = iclamp(wy0 + t / WW, 0, P.h - 1), sx = iclamp(wx0 + t % WW, 0, P.w - 1);
            s_win[t] = stage_texel(pc2, P.pk2, (unsigned)(sy * P.pitch + sx));
        }
    }
    __syncthreads();
    float res[9];                                   // raw cost of this group's pass for the 9 candidates
    if (known) {
        const rgbf c1 = texel_rgb(s_src[(threadIdx.y + R) * TW + threadIdx.x + R]);
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const int cx = (int)(int16_t)(ccx + m - 1);
            float run[3] = {0.0f, 0.0f, 0.0f};
            if (!(cx < 0 || cx >= P.w)) {
                rgbf c2[3];
                const int cx16 = cx << 4, wmax16 = (P.w - 1) << 4;
                if (coherent) {
                    const int wbase = ((ccy - 1 - wy0) * WW + (cx - wx0)) * 16;
#pragma unroll
                    for (int n = 0; n < 3; n++) c2[n] = texel_rgb(*reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_win) + wbase + n * (WW * 16)));
                    if (grp == 0) c2f_pass<R, 3, true, true, WW>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run, s_win, wbase);
                    else if (grp == 1) c2f_pass<R, 2, true, true, WW>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run, s_win, wbase);
                    else if (grp == 2) c2f_pass<R, 1, true, true, WW>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run, s_win, wbase);
                    else c2f_pass<R, 0, true, true, WW>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run, s_win, wbase);
                } else {
#pragma unroll
                    for (int n = 0; n < 3; n++) c2[n] = texel_rgb(tex_px(P.pk2, P.pitch, P.w, P.h, cx, ccy + n - 1));
                    if (grp == 0) c2f_pass<R, 3, true>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
                    else if (grp == 1) c2f_pass<R, 2, true>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
                    else if (grp == 2) c2f_pass<R, 1, true>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
                    else c2f_pass<R, 0, true>(P, L, s_src, TW, threadIdx.x, threadIdx.y, cx16, wmax16, ccy, c1, c2, run);
                }
            }
#pragma unroll
            for (int n = 0; n < 3; n++) res[m * 3 + n] = run[n];
        }
    }
    __syncthreads();                                // every read of the source tile is done: its storage carries the exchange
    float* __restrict__ xch = reinterpret_cast<float*>(s_src);
    if (grp != 0 && known) {
#pragma unroll
        for (int k = 0; k < 9; k++) xch[((grp - 1) * 9 + k) * 256 + ptid] = res[k];
    }
    __syncthreads();
    if (grp != 0 || !inimg) return;
    if (!known) {
        flow[(y * P.w + x) * 2] = 0.0f;
        flow[(y * P.w + x) * 2 + 1] = 0.0f;
        return;
    }
    int bx = ccx, by = ccy;
    float min_cost = 999999;
#pragma unroll
    for (int m = 0; m < 3; m++) {
        const int cx = (int)(int16_t)(ccx + m - 1);
        if (cx < 0 || cx >= P.w) continue;
#pragma unroll
        for (int n = 0; n < 3; n++) {
            const int cy = (int)(int16_t)(ccy + n - 1);
            if (cy < 0 || cy >= P.h) continue;
            const int k = m * 3 + n;
            const float c_4 = res[k], c_3 = xch[k * 256 + ptid], c_2 = xch[(9 + k) * 256 + ptid], c_1 = xch[(18 + k) * 256 + ptid];
            const float m34 = (c_3 < c_4) ? c_3 : c_4;                // __min(cost1,__min(cost2,__min(cost3,cost4))), kernel.cu:512
            const float m234 = (c_2 < m34) ? c_2 : m34;
            const float cv = (c_1 < m234) ? c_1 : m234;
            if (cv < min_cost) { min_cost = cv; bx = cx; by = cy; }
        }
    }
    flow[(y * P.w + x) * 2] = (float)(bx - x);
    flow[(y * P.w + x) * 2 + 1] = (float)(by - y);
}

// the candidate loop of kernel.cu:2028-2040 over the 9 costs written by the split launch
template <int SPLIT>
__global__ __launch_bounds__(256) void k_c2f_select(float* __restrict__ flow_, const float* __restrict__ cost9_, int w, int h, size_t pstride)
{
    float* __restrict__ flow = pair_ptr(flow_, pstride, blockIdx.z);
    const float* __restrict__ cost9 = pair_ptr(cost9_, pstride, blockIdx.z);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const float fvx = flow[(y * w + x) * 2], fvy = flow[(y * w + x) * 2 + 1];
    if (fvx > kUnknownFlowThresh || fvy > kUnknownFlowThresh) {
        flow[(y * w + x) * 2] = 0.0f;
        flow[(y * w + x) * 2 + 1] = 0.0f;
        return;
    }
    const int ccx = (int)(int16_t)(f2short(fvx) + x);
    const int ccy = (int)(int16_t)(f2short(fvy) + y);
    int bx = ccx, by = ccy;
    float min_cost = 999999;
    for (int m = 0; m < 3; m++) {
        const int cx = (int)(int16_t)(ccx + m - 1);
        if (cx < 0 || cx >= w) continue;
        for (int n = 0; n < 3; n++) {
            const int cy = (int)(int16_t)(ccy + n - 1);
            if (cy < 0 || cy >= h) continue;
            const float* __restrict__ pc = cost9 + (size_t)(y * w + x) * 36 + m * 3 + n;
            float cv = pc[0];
            if (SPLIT == 4) {                    // __min(c1,__min(c2,__min(c3,c4))), kernel.cu:512: passes 4th to 1st
                cv = pc[27];
                cv = (pc[18] < cv) ? pc[18] : cv;
                cv = (pc[9] < cv) ? pc[9] : cv;
                cv = (pc[0] < cv) ? pc[0] : cv;
            }
            if (cv < min_cost) { min_cost = cv; bx = cx; by = cy; }
        }
    }
    flow[(y * w + x) * 2] = (float)(bx - x);
    flow[(y * w + x) * 2 + 1] = (float)(by - y);
}

// admissible spread (max - min) of a tile's candidate centres in the LDS-window kernels, for the tests that probe the boundary
bool c2f_window_span(int R, int* span_x, int* span_y)
{
    if (R == 9) {
        using G = C2fWinGeom<9>;
        *span_x = 64 - 3 - (G::xhi() - G::xlo());
        *span_y = EPPM_C2F_WIN_H - 3 - (G::yhi() - G::ylo());
        return true;
    }
    if (R == 17) {
        using G = C2fWinGeom<17>;
        constexpr int WW = (kBlock + 2 + (G::xhi() - G::xlo()) + 8 + 15) / 16 * 16, WH = kBlock + 2 + (G::yhi() - G::ylo()) + 7;
        *span_x = WW - 3 - (G::xhi() - G::xlo());
        *span_y = WH - 3 - (G::yhi() - G::ylo());
        return true;
    }
    return false;
}

// no_split (a context's "c2f_no_split" option): never split, so that small images go through the LDS-window kernels too
bool c2f_refine_wants_split(int w, int h, int R, int npairs, bool no_split)
{
    if (no_split) return false;
    const int tiles = ((w + kBlock - 1) / kBlock) * ((h + kBlock - 1) / kBlock) * npairs;
#ifndef EPPM_C2F_SPLIT_BELOW_WAVES
#define EPPM_C2F_SPLIT_BELOW_WAVES 1024              // fewer than 1 wave per SIMD on 256 CUs (at 256 threads per tile)
#endif
    // Round 2 split below 3 waves per SIMD, which sent level 1 of ONE 1024x436 pair (448 tiles) through the split gather kernel:
    // 0.367 ms and a 16 MB scratch plane of 36 costs per pixel; the LDS-window kernel does the same launch in 0.364 ms without it.
    return (R == 9 || R == 17) && tiles * 4 < EPPM_C2F_SPLIT_BELOW_WAVES;
}

// cost9: scratch of 36 floats per pixel, or NULL (never split)
void launch_c2f_refine(const PlanesH& P, float* flow, const float* lut, int R, float* cost9, hipStream_t s, Batch bt, bool no_split)
{
    dim3 grid((P.w + kBlock - 1) / kBlock, (P.h + kBlock - 1) / kBlock, bt.n), block(kBlock, kBlock);
    const int tiles = grid.x * grid.y;
    const int per_xcd = (tiles + 7) / 8;
    dim3 grid1(per_xcd * 8, bt.n);               // x: padded so every XCD gets the same number of slots; y: pair
    const bool table_ok = (P.w + R < 32764) && (P.h + R < 32764);     // range of the offset-table identity (c2f_pass)
    if (cost9 && table_ok && c2f_refine_wants_split(P.w, P.h, R, bt.n, no_split)) {
        // 3 or 4 workgroups per tile: the factor whose workgroup count divides more evenly over the 256 CUs
        auto imbalance = [&](int f) { const int wgs = tiles * f * bt.n; return (float)((wgs + 255) / 256) * 256.0f / (float)wgs; };
        const int f = (imbalance(4) < imbalance(3)) ? 4 : 3;
        dim3 gridf(per_xcd * f * 8, bt.n), gs((P.w + 63) / 64, (P.h + 3) / 4, bt.n), bs(64, 4);
        if (f == 3) {
            if (R == 9) hipLaunchKernelGGL((k_c2f_refine_tiled<9, 3>), gridf, block, 0, s, P, flow, lut, cost9, bt.stride);
            else hipLaunchKernelGGL((k_c2f_refine_tiled<17, 3>), gridf, block, 0, s, P, flow, lut, cost9, bt.stride);
            hipLaunchKernelGGL(k_c2f_select<3>, gs, bs, 0, s, flow, cost9, P.w, P.h, bt.stride);
        } else {
            if (R == 9) hipLaunchKernelGGL((k_c2f_refine_tiled<9, 4>), gridf, block, 0, s, P, flow, lut, cost9, bt.stride);
            else hipLaunchKernelGGL((k_c2f_refine_tiled<17, 4>), gridf, block, 0, s, P, flow, lut, cost9, bt.stride);
            hipLaunchKernelGGL(k_c2f_select<4>, gs, bs, 0, s, flow, cost9, P.w, P.h, bt.stride);
        }
        return;
    }
#ifndef EPPM_C2F_WINDOW
#define EPPM_C2F_WINDOW 1
#endif
    if (EPPM_C2F_WINDOW && R == 9 && table_ok) hipLaunchKernelGGL((k_c2f_refine_win<9>), grid1, dim3(kBlock, kBlock, 2), 0, s, P, flow, lut, bt.stride);
    else if (R == 9 && table_ok) hipLaunchKernelGGL((k_c2f_refine_tiled<9, 0>), grid1, block, 0, s, P, flow, lut, (float*)nullptr, bt.stride);
#ifndef EPPM_C2F_WINDOW17
#define EPPM_C2F_WINDOW17 1
#endif
    else if (EPPM_C2F_WINDOW17 && R == 17 && table_ok) hipLaunchKernelGGL((k_c2f_refine_win4<17>), grid1, dim3(kBlock, kBlock, 4), 0, s, P, flow, lut, bt.stride);
    else if (R == 17 && table_ok) hipLaunchKernelGGL((k_c2f_refine_tiled<17, 0>), grid1, block, 0, s, P, flow, lut, (float*)nullptr, bt.stride);
    else hipLaunchKernelGGL(k_c2f_refine, grid, block, 0, s, P, flow, lut, R, bt.stride);
}

// ---------------------------------------------------------------------------------------------------
// refine :764-799: 21x21 joint bilateral filter of the flow guided by image 1 (Jacobi).
//
// 32x16 output tile, 256 threads: a lane filters TWO vertically adjacent pixels, so each tap row it reads from
// LDS serves both (rows 0..20 the upper pixel, 1..21 the lower one) -- one pixel per lane reads 21 B of LDS per
// tap and is LDS-bandwidth bound.  The (32+20)x(16+20) halo tile holds {r, g, b, flow x} and {flow y} per texel.
// Taps the reference skips (outside the image, or unknown flow, refine :781) are stored with r = 100: their range
// distance is ~100, the exponent -2.5e7 and fast_exp returns exactly 0, so they add 0 * flow = 0 and 0 to the
// sums -- no validity flag, no divergent branch (unknown flows are the finite marker 1e10, never inf).
// ---------------------------------------------------------------------------------------------------
// PPL = pixels per lane: 2 (32x16 tile) for large launches, 1 (32x8 tile, twice the waves) otherwise, see launch_flow_blf.
constexpr int BT_W = 32, BR = kBlfRadius, BTW = BT_W + 2 * BR;

#ifndef EPPM_BLF_MIX
#define EPPM_BLF_MIX 3          // two pixels per lane: every M-th tap column's range weights by formula, the rest by table (LDS / VALU balance); 0: all by table
#endif
#ifndef EPPM_BLF_MIX1
#define EPPM_BLF_MIX1 2         // one pixel per lane (the smaller launches)
#endif
// which taps of the smoothing evaluate their range weight instead of reading it (pixel: 0 upper / only, 1 lower)
template <int PPL>
__device__ __forceinline__ constexpr bool blf_by_formula(int dx, int pixel)
{
    constexpr int M = (PPL == 1) ? EPPM_BLF_MIX1 : EPPM_BLF_MIX;
    return M == 1 ? (PPL == 2 ? pixel == 1 : (dx & 1)) : M >= 2 ? dx % M == 0 : false;
}
#ifdef EPPM_BLF_WAVES
#define EPPM_BLF_OCC __attribute__((amdgpu_waves_per_eu(EPPM_BLF_WAVES, EPPM_BLF_WAVES)))
#else
#define EPPM_BLF_OCC
#endif
template <int PPL>
__global__ __launch_bounds__(256) EPPM_BLF_OCC void k_flow_blf(float* __restrict__ out_, const float* __restrict__ in_,
                                                  const uint32_t* __restrict__ img_, int ipitch, int w, int h, int fpitch,
                                                  const float* __restrict__ blf_lut, size_t pstride)
{
    float* __restrict__ out = pair_ptr(out_, pstride, blockIdx.z);
    const float* __restrict__ in = pair_ptr(in_, pstride, blockIdx.z);
    const uint32_t* __restrict__ img = pair_ptr(img_, pstride, blockIdx.z);
    constexpr int BT_H = 8 * PPL, BTH = BT_H + 2 * BR;
    __shared__ float4 s_t[BTH * BTW];          // r, g, b (unorm), flow x
    __shared__ float s_fy[BTH * BTW];
    __shared__ float s_lut[BR + 1];
    __shared__ DeltaTab s_D;                   // exp(-d^2 / POSTPROC_BLF_SIG_R^2) by table: the same bits as the formula (eppm_device.cuh)
    const int x0 = blockIdx.x * BT_W, y0 = blockIdx.y * BT_H;
    const int tid = threadIdx.y * BT_W + threadIdx.x;
    if (tid <= BR) s_lut[tid] = blf_lut[tid];
    load_delta_tab<false>(s_D, blf_lut + BR + 1, tid, 256);
    for (int t = tid; t < BTW * BTH; t += 256) {
        const int cy = y0 + t / BTW - BR, cx = x0 + t % BTW - BR;
        float4 e = make_float4(100.0f, 0.0f, 0.0f, 0.0f);
        float fy = 0.0f;
        if (cx >= 0 && cy >= 0 && cx < w && cy < h) {
            e.w = in[(cy * fpitch + cx) * 2];
            fy = in[(cy * fpitch + cx) * 2 + 1];
            if (!(e.w > kUnknownFlowThresh || fy > kUnknownFlowThresh)) {     // refine :781
                const rgbf c = unpack_rgb(img[cy * ipitch + cx]);
                e.x = c.x; e.y = c.y; e.z = c.z;
            }
        }
        s_t[t] = e;
        s_fy[t] = fy;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, ya = y0 + PPL * threadIdx.y;        // pixels (x, ya) and, PPL = 2, (x, ya + 1)
    if (x >= w || ya >= h) return;
    const bool has_b = (PPL == 2) && (ya + 1 < h);
    const rgbf ca = unpack_rgb(img[ya * ipitch + x]);
    const rgbf cb = unpack_rgb(img[(has_b ? ya + 1 : ya) * ipitch + x]);
    float nxa = 0.f, nya = 0.f, wa = 0.f, nxb = 0.f, nyb = 0.f, wb = 0.f;
    const int base = (PPL * threadIdx.y) * BTW + threadIdx.x;
    // tap rows ya-10 .. ya+11: row 0 serves only the upper pixel, row 21 only the lower one, rows 1..20 both
    auto tap_row = [&](int r, auto use_a, auto use_b) {
        const float gya = use_a ? s_lut[abs(r - BR)] : 0.0f;
        const float gyb = use_b ? s_lut[abs(r - 1 - BR)] : 0.0f;
EPPM_UNROLL(EPPM_BLF_UNROLL)
        for (int dx = 0; dx <= 2 * BR; dx++) {
            const int ti = base + r * BTW + dx;
            const float4 tp = s_t[ti];
            const float tfy = s_fy[ti];
            const rgbf pix = {tp.x, tp.y, tp.z};
            const float gx = s_lut[abs(dx - BR)];
            if (use_a) {
                // (a skipped tap, r = 100, meets the entry of d = 1: exp(-2500) = 0 exactly, as the formula gives for d ~ 100)
                // one pixel per lane (PPL = 1): every other tap column by formula
                const bool tab_a = EPPM_DELTA_BLF && !blf_by_formula<PPL>(dx, 0);
                const float delta_r = tab_a ? __builtin_amdgcn_fmed3f(max_abs_diff(ca, pix), 0.0f, 1.0f) : max_abs_diff(ca, pix);
                const float coef_r = tab_a ? delta_lookup_off(s_D, delta_r) : fast_exp(div_wmf2(-(delta_r * delta_r)));
                const float coef_s = gx * gya;
                const float wgt = coef_r * coef_s;
                nxa += wgt * tp.w;
                nya += wgt * tfy;
                wa += wgt;
            }
            if (use_b) {
                // EPPM_BLF_MIX: the lower pixel EVALUATES the weight (the same bits: the table was filled by this formula; a skipped tap's
                // distance ~100 gives exp(-2.5e7) = 0 exactly) -- the table reads of both pixels made the LDS array the kernel's bound
                // (28 array cycles per tap against 14.5 issue cycles per CU); one of two by formula: 18 against 21.5
                const bool tab_b = EPPM_DELTA_BLF && !blf_by_formula<PPL>(dx, 1);
                const float delta_r = tab_b ? __builtin_amdgcn_fmed3f(max_abs_diff(cb, pix), 0.0f, 1.0f) : max_abs_diff(cb, pix);
                const float coef_r = tab_b ? delta_lookup_off(s_D, delta_r) : fast_exp(div_wmf2(-(delta_r * delta_r)));
                const float coef_s = gx * gyb;
                const float wgt = coef_r * coef_s;
                nxb += wgt * tp.w;
                nyb += wgt * tfy;
                wb += wgt;
            }
        }
    };
    if (PPL == 2) {
        tap_row(0, std::true_type{}, std::false_type{});
#pragma unroll 1
        for (int r = 1; r <= 2 * BR; r++) tap_row(r, std::true_type{}, std::true_type{});
        tap_row(2 * BR + 1, std::false_type{}, std::true_type{});
    } else {
#pragma unroll 1
        for (int r = 0; r <= 2 * BR; r++) tap_row(r, std::true_type{}, std::false_type{});
    }
    {
        const int ci = base + BR * BTW + BR;
        float ox = s_t[ci].w, oy = s_fy[ci];
        if (wa != 0) { ox = nxa / wa; oy = nya / wa; }
        out[(ya * fpitch + x) * 2] = ox;
        out[(ya * fpitch + x) * 2 + 1] = oy;
    }
    if (has_b) {
        const int ci = base + (BR + 1) * BTW + BR;
        float ox = s_t[ci].w, oy = s_fy[ci];
        if (wb != 0) { ox = nxb / wb; oy = nyb / wb; }
        out[((ya + 1) * fpitch + x) * 2] = ox;
        out[((ya + 1) * fpitch + x) * 2 + 1] = oy;
    }
}
void launch_flow_blf(float* out, const float* in, const uint32_t* img, int ipitch, int w, int h, int flow_pitch,
                     const float* blf_lut, hipStream_t s, Batch bt)
{
    dim3 block(BT_W, 8);
    const int wgs2 = ((w + BT_W - 1) / BT_W) * ((h + 15) / 16) * bt.n;
    // two pixels per lane halve the LDS traffic but double the work quantum: they pay from about 8 workgroups per CU
    // (1920x1080: 0.96 vs 1.03 ms); below that the finer quantum balances the 256 CUs better (1024x436: 0.25 vs 0.27 ms)
    if (wgs2 >= 8 * 256) {
        hipLaunchKernelGGL(k_flow_blf<2>, dim3((w + BT_W - 1) / BT_W, (h + 15) / 16, bt.n), block, 0, s, out, in, img, ipitch, w, h, flow_pitch, blf_lut, bt.stride);
    } else {
        hipLaunchKernelGGL(k_flow_blf<1>, dim3((w + BT_W - 1) / BT_W, (h + 7) / 8, bt.n), block, 0, s, out, in, img, ipitch, w, h, flow_pitch, blf_lut, bt.stride);
    }
}

}  // namespace eppm
