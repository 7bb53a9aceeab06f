// k_post.hip -- level-2 post-processing of the NNF (reference: bao_pmflow_refine_kernel.cu :53-92 left-right
// check, :149-193 outlier removal, :198-286 weighted median, :297-390 hole filling, :636-655 NNF->flow).
// Integer/short work on a quarter-resolution plane; every kernel is Jacobi (reads `in`, writes `out`).
#include "eppm_device.cuh"
#include "eppm_internal.h"

namespace eppm {

// ---------------------------------------------------------------------------------------------------
// refine :53-76.  In place on (nnf1,cost1): a thread touches only its own pixel; nnf2 is read-only.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lr_check(int16_t* __restrict__ nnf1_, float* __restrict__ cost1_,
                                                  const int16_t* __restrict__ nnf2_, int w, int h, int cpitch, int npitch, size_t pstride)
{
    int16_t* __restrict__ nnf1 = pair_ptr(nnf1_, pstride, blockIdx.z);
    float* __restrict__ cost1 = pair_ptr(cost1_, pstride, blockIdx.z);
    const int16_t* __restrict__ nnf2 = pair_ptr(nnf2_, pstride, blockIdx.z);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const int dx = nnf1[(y * npitch + x) * 2], dy = nnf1[(y * npitch + x) * 2 + 1];
    bool bad;
    if (dy < 0 || dy >= h || dx < 0 || dx >= w) bad = true;
    else {
        const int ex = nnf2[(dy * npitch + dx) * 2], ey = nnf2[(dy * npitch + dx) * 2 + 1];
        bad = (abs(ex - x) > 0 || abs(ey - y) > 0);      // DIFF_THRESH 0, refine :51
    }
    if (bad) {
        nnf1[(y * npitch + x) * 2] = (int16_t)kInvalid;
        nnf1[(y * npitch + x) * 2 + 1] = (int16_t)kInvalid;
        cost1[y * cpitch + x] = FLT_MAX;
    }
}
void launch_lr_check(int16_t* nnf1, float* cost1, const int16_t* nnf2, int w, int h, int cost_pitch, int nnf_pitch, hipStream_t s, Batch bt)
{
    dim3 block(64, 4), grid((w + 63) / 64, (h + 3) / 4, bt.n);
    hipLaunchKernelGGL(k_lr_check, grid, block, 0, s, nnf1, cost1, nnf2, w, h, cost_pitch, nnf_pitch, bt.stride);
}

// ---------------------------------------------------------------------------------------------------
// refine :149-182.  13x13 vote on flow similarity; 32x8 tile + 6-px halo of relative flows in LDS.
// ---------------------------------------------------------------------------------------------------
constexpr int OT_W = 32, OT_H = 8;
__global__ __launch_bounds__(256) void k_outlier(int16_t* __restrict__ nnf_out_, float* __restrict__ cost_,
                                                 const int16_t* __restrict__ nnf_in_, int w, int h, int cpitch, int npitch, size_t pstride)
{
    int16_t* __restrict__ nnf_out = pair_ptr(nnf_out_, pstride, blockIdx.z);
    float* __restrict__ cost = pair_ptr(cost_, pstride, blockIdx.z);
    const int16_t* __restrict__ nnf_in = pair_ptr(nnf_in_, pstride, blockIdx.z);
    constexpr int TW = OT_W + 2 * kStatRadius, TH = OT_H + 2 * kStatRadius;
    __shared__ int s_fx[TH * TW];
    __shared__ int s_fy[TH * TW];
    const int x0 = blockIdx.x * OT_W, y0 = blockIdx.y * OT_H;
    const int tid = threadIdx.y * OT_W + threadIdx.x;
    for (int t = tid; t < TW * TH; t += 256) {
        const int cy = y0 + t / TW - kStatRadius, cx = x0 + t % TW - kStatRadius;
        int fx = 0x40000000, fy = 0x40000000;            // out of image: never similar to anything
        if (cx >= 0 && cy >= 0 && cx < w && cy < h) {
            fx = (int)(int16_t)(nnf_in[(cy * npitch + cx) * 2] - cx);       // short arithmetic, refine :165-166
            fy = (int)(int16_t)(nnf_in[(cy * npitch + cx) * 2 + 1] - cy);
        }
        s_fx[t] = fx; s_fy[t] = fy;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= w || y >= h) return;
    const int ox = nnf_in[(y * npitch + x) * 2], oy = nnf_in[(y * npitch + x) * 2 + 1];
    int rx = ox, ry = oy;
    if (!(ox < 0 && oy < 0)) {                              // "skip occlusion", refine :156
        const int cfx = s_fx[(threadIdx.y + kStatRadius) * TW + threadIdx.x + kStatRadius];
        const int cfy = s_fy[(threadIdx.y + kStatRadius) * TW + threadIdx.x + kStatRadius];
        int count = 0;
        for (int dy = 0; dy <= 2 * kStatRadius; dy++)
            for (int dx = 0; dx <= 2 * kStatRadius; dx++) {
                const int nfx = s_fx[(threadIdx.y + dy) * TW + threadIdx.x + dx];
                const int nfy = s_fy[(threadIdx.y + dy) * TW + threadIdx.x + dx];
                if (abs(nfx - cfx) <= kStatSimThresh && abs(nfy - cfy) <= kStatSimThresh) count++;
            }
        if (count < kStatCountThresh) {
            rx = kInvalid; ry = kInvalid;
            cost[y * cpitch + x] = FLT_MAX;
        }
    }
    nnf_out[(y * npitch + x) * 2] = (int16_t)rx;
    nnf_out[(y * npitch + x) * 2 + 1] = (int16_t)ry;
}
void launch_outlier(int16_t* nnf_out, float* cost, const int16_t* nnf_in, int w, int h, int cost_pitch, int nnf_pitch, hipStream_t s, Batch bt)
{
    dim3 block(OT_W, OT_H), grid((w + OT_W - 1) / OT_W, (h + OT_H - 1) / OT_H, bt.n);
    hipLaunchKernelGGL(k_outlier, grid, block, 0, s, nnf_out, cost, nnf_in, w, h, cost_pitch, nnf_pitch, bt.stride);
}

// ---------------------------------------------------------------------------------------------------
// Weighted median by exhaustive candidate scoring (refine :206-286): num_iter Jacobi launches.
//
// The reference spends O(81 x 81) taps in ONE thread per pixel, and in the live call (20 iterations,
// occlusion only, driver :239) only the still-invalid pixels do any work.  Mapping here:
//  * a work list of the pixels that need work is built once and shrinks from launch to launch (a pixel
//    that became valid is carried one more launch as "copy only" so both ping-pong buffers stay equal
//    outside the list);
//  * ONE WAVE scores one pixel: the valid taps of the 9x9 window are compacted in row-major order into
//    LDS together with their bilateral weight (computed once per pixel, not once per candidate), the
//    candidates (the same taps) sit on the lanes, and every lane runs the reference's sequential sum over
//    the taps -- same order, same operations;
//  * the winner is the first minimum in candidate order (strict <): a lexicographic (cost, index) wave
//    reduction.
// ---------------------------------------------------------------------------------------------------
constexpr int WR = kWmfRadius, WN = (2 * WR + 1) * (2 * WR + 1);   // 4, 81
constexpr uint32_t kCopyOnly = 0x80000000u;
#ifndef EPPM_WMF_MAX_BLOCKS
#define EPPM_WMF_MAX_BLOCKS 1024
#endif
constexpr uint32_t kWmfBatch = 256;      // list entries a workgroup processes between two appends

__global__ __launch_bounds__(256) void k_wmf_build_list(const int16_t* __restrict__ nnf_, int npitch, int w, int h, int only_occ,
                                                        uint32_t* __restrict__ list_, uint32_t* __restrict__ count_, size_t pstride)
{
    const int16_t* __restrict__ nnf = pair_ptr(nnf_, pstride, blockIdx.z);
    uint32_t* __restrict__ list = pair_ptr(list_, pstride, blockIdx.z);
    uint32_t* __restrict__ count = pair_ptr(count_, pstride, blockIdx.z);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const int ox = nnf[(y * npitch + x) * 2], oy = nnf[(y * npitch + x) * 2 + 1];
    if (only_occ && ox >= 0 && oy >= 0) return;                // refine :213
    list[atomicAdd(count, 1u)] = ((uint32_t)y << 16) | (uint32_t)x;
}

// blockIdx.y = pair of the batch (work lists, counters and flags are per pair)
__global__ __launch_bounds__(256) void k_wmf_iter(int16_t* __restrict__ nnf_out_, const int16_t* __restrict__ nnf_in_,
                                                  const uint32_t* __restrict__ img_, int ipitch, int w, int h, int npitch,
                                                  const float* __restrict__ wmf_lut, int only_occ,
                                                  const uint32_t* __restrict__ list_in_, const uint32_t* __restrict__ count_in_,
                                                  uint32_t* __restrict__ list_out_, uint32_t* __restrict__ count_out_,
                                                  const uint32_t* __restrict__ changed_prev_, uint32_t* __restrict__ changed_cur_, size_t pstride)
{
    int16_t* __restrict__ nnf_out = pair_ptr(nnf_out_, pstride, blockIdx.y);
    const int16_t* __restrict__ nnf_in = pair_ptr(nnf_in_, pstride, blockIdx.y);
    const uint32_t* __restrict__ img = pair_ptr(img_, pstride, blockIdx.y);
    const uint32_t* __restrict__ list_in = pair_ptr(list_in_, pstride, blockIdx.y);
    const uint32_t* __restrict__ count_in = pair_ptr(count_in_, pstride, blockIdx.y);
    uint32_t* __restrict__ list_out = pair_ptr(list_out_, pstride, blockIdx.y);
    uint32_t* __restrict__ count_out = pair_ptr(count_out_, pstride, blockIdx.y);
    const uint32_t* __restrict__ changed_prev = pair_ptr_opt(changed_prev_, pstride, blockIdx.y);
    uint32_t* __restrict__ changed_cur = pair_ptr(changed_cur_, pstride, blockIdx.y);
    // Occlusion-only mode: if the previous launch filled no pixel, the field is at a fixed point -- the listed pixels
    // would be recomputed from unchanged neighbourhoods and stay invalid, and both ping-pong buffers already agree.
    // The remaining launches do nothing (their output count stays 0).
    if (only_occ && changed_prev && *changed_prev == 0u) return;
    __shared__ float s_lut[WR + 1];
    __shared__ DeltaTab s_D;              // exp(-d^2 / WMF_SIG_R^2) by table: the same bits as the formula (eppm_device.cuh)
    __shared__ float4 s_tap[4][WN];       // {bilateral weight, flow x, flow y, -} of the valid taps, row-major order
    __shared__ uint32_t s_keep[kWmfBatch], s_nkeep, s_base;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    if (tid <= WR) s_lut[tid] = wmf_lut[tid];
    load_delta_tab(s_D, wmf_lut + WR + 1, tid, blockDim.x);
    __syncthreads();
    // A workgroup owns a contiguous chunk of the list and works through it in batches; the entries that stay on
    // the list are collected in LDS and appended with ONE global atomic per batch (one returning atomic per
    // item on the single counter made the first launches atomic-bound: 17 k items took 208 us).
    const uint32_t n_items = *count_in;
    const uint32_t per_wg = (n_items + gridDim.x - 1) / gridDim.x;
    const uint32_t beg = blockIdx.x * per_wg, end = (beg + per_wg < n_items) ? beg + per_wg : n_items;
    for (uint32_t b0 = beg; b0 < end; b0 += kWmfBatch) {
        const uint32_t b1 = (b0 + kWmfBatch < end) ? b0 + kWmfBatch : end;
        if (tid == 0) s_nkeep = 0;
        __syncthreads();
        for (uint32_t item = b0 + wv; item < b1; item += 4) {
            const uint32_t e = list_in[item];
            const int x = (int)(e & 0xffffu), y = (int)((e >> 16) & 0x7fffu);
            const int pidx = (y * npitch + x) * 2;
            const int ox = nnf_in[pidx], oy = nnf_in[pidx + 1];
            if (e & kCopyOnly) {                                    // became valid in the previous launch
                if (lane == 0) { nnf_out[pidx] = (int16_t)ox; nnf_out[pidx + 1] = (int16_t)oy; }
                continue;
            }
            const rgbf center = unpack_rgb(img[y * ipitch + x]);
            // taps in row-major order (dy outer, dx inner): lanes 0..63 take taps 0..63, lanes 0..16 taps 64..80
            int nv = 0;
    #pragma unroll
            for (int rnd = 0; rnd < 2; rnd++) {
                const int t = rnd * 64 + lane;
                bool ok = false;
                int fx = 0, fy = 0;
                float wgt = 0.0f;
                if (t < WN) {
                    const int dy2 = t / 9 - WR, dx2 = t % 9 - WR;
                    const int cy = y + dy2, cx = x + dx2;
                    if (cx >= 0 && cy >= 0 && cx < w && cy < h) {
                        const int dx = nnf_in[(cy * npitch + cx) * 2], dy = nnf_in[(cy * npitch + cx) * 2 + 1];
                        if (!(dx < 0 || dy < 0)) {                  // "skip invalid disparity", refine :225,238
                            ok = true;
                            fx = (int)(int16_t)(dx - cx);
                            fy = (int)(int16_t)(dy - cy);
                            const rgbf pix = unpack_rgb(img[cy * ipitch + cx]);
                            const float delta_r = max_abs_diff(center, pix);
                            const float coef_r = EPPM_DELTA_BLF ? delta_lookup(s_D, delta_r) : fast_exp(div_wmf2(-(delta_r * delta_r)));     // the same bits either way
                            const float coef_s = s_lut[abs(dx2)] * s_lut[abs(dy2)];
                            wgt = coef_r * coef_s;                  // refine :198-204
                        }
                    }
                }
                const unsigned long long m = __ballot(ok);
                if (ok) {
                    const int pos = nv + __popcll(m & ((1ull << lane) - 1ull));
                    s_tap[wv][pos] = make_float4(wgt, (float)fx, (float)fy, 0.0f);     // |flow| < 2^16: exact in float
                }
                nv += __popcll(m);
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // Candidates lane and lane + 64 advance together through ONE pass over the taps (each tap is read once and
            // broadcast).  The flow differences are formed in float -- exact, so (float)max(|dx|,|dy|) is the reference's
            // value -- and weightSum, the same sequential sum for every candidate, is formed once.
            const bool has0 = lane < nv, has1 = lane + 64 < nv;
            const float4 e0 = has0 ? s_tap[wv][lane] : make_float4(0, 0, 0, 0);
            const float4 e1 = has1 ? s_tap[wv][lane + 64] : make_float4(0, 0, 0, 0);
            float cs0 = 0.0f, cs1 = 0.0f, weightSum = 0.0f;
    #pragma unroll 4
            for (int t = 0; t < nv; t++) {
                const float4 tp = s_tap[wv][t];
                cs0 += tp.x * fmaxf(fabsf(e0.y - tp.y), fabsf(e0.z - tp.z));
                cs1 += tp.x * fmaxf(fabsf(e1.y - tp.y), fabsf(e1.z - tp.z));
                weightSum += tp.x;
            }
            float bestc = FLT_MAX;
            int besti = 0x7fffffff;
            if (weightSum > 0.0f) {
                if (has0 && cs0 < FLT_MAX) { bestc = cs0; besti = lane; }
                if (has1 && cs1 < FLT_MAX && cs1 < bestc) { bestc = cs1; besti = lane + 64; }      // strict <: the lower index wins ties
            }
    #pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float oc = __shfl_xor(bestc, off, 64);
                const int oi = __shfl_xor(besti, off, 64);
                if (oc < bestc || (oc == bestc && oi < besti)) { bestc = oc; besti = oi; }
            }
            if (lane == 0) {
                int rx = ox, ry = oy;
                if (besti != 0x7fffffff) {
                    const int nx = (int)(int16_t)((int)s_tap[wv][besti].y + x), ny = (int)(int16_t)((int)s_tap[wv][besti].z + y);
                    if (!(nx < 0 || ny < 0)) { rx = nx; ry = ny; }          // refine :257
                }
                nnf_out[pidx] = (int16_t)rx;
                nnf_out[pidx + 1] = (int16_t)ry;
                const bool again = !(only_occ && rx >= 0 && ry >= 0);
                if (!again) *changed_cur = 1u;
                s_keep[atomicAdd(&s_nkeep, 1u)] = (e & 0x7fffffffu) | (again ? 0u : kCopyOnly);
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        __syncthreads();
        if (tid == 0) s_base = atomicAdd(count_out, s_nkeep);
        __syncthreads();
        for (uint32_t i = tid; i < s_nkeep; i += 256) list_out[s_base + i] = s_keep[i];
        __syncthreads();
    }
}

// start of a weighted-median run for every pair of the batch: counters and flags zeroed, buf_b = buf_a
__global__ __launch_bounds__(256) void k_wmf_begin(uint32_t* __restrict__ dst_, const uint32_t* __restrict__ src_, int n_words,
                                                   uint32_t* __restrict__ counts_, int n_counts, size_t pstride)
{
    uint32_t* __restrict__ dst = pair_ptr(dst_, pstride, blockIdx.z);
    const uint32_t* __restrict__ src = pair_ptr(src_, pstride, blockIdx.z);
    uint32_t* __restrict__ counts = pair_ptr(counts_, pstride, blockIdx.z);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) dst[i] = src[i];
    if (i < n_counts) counts[i] = 0u;
}

size_t wmf_workspace_words(int w, int h, int num_iter)
{
    return 2 * (size_t)w * h + 2 * (size_t)((num_iter > 0 ? num_iter : 0) + 2);
}

// Runs num_iter launches, ping-ponging between buf_a (input, holds the NNF) and buf_b.  ws: uint32 workspace of
// wmf_workspace_words() words (two lists, per-launch counters and "filled a pixel" flags).  Returns the buffer that
// holds the result.
int16_t* launch_wmf(int16_t* buf_a, int16_t* buf_b, const uint32_t* img, int ipitch, int w, int h, int nnf_pitch,
                    const float* wmf_lut, int num_iter, int only_occlusion, uint32_t* ws, hipStream_t s, Batch bt)
{
    if (num_iter <= 0) return buf_a;
    uint32_t* list0 = ws;
    uint32_t* list1 = ws + (size_t)w * h;
    uint32_t* counts = ws + 2 * (size_t)w * h;
    uint32_t* changed = counts + num_iter + 2;
    {
        const int n_words = nnf_pitch * h, n_counts = 2 * (num_iter + 2);
        const int n = n_words > n_counts ? n_words : n_counts;
        hipLaunchKernelGGL(k_wmf_begin, dim3((n + 255) / 256, 1, bt.n), dim3(256), 0, s, (uint32_t*)buf_b, (const uint32_t*)buf_a, n_words, counts,
                           n_counts, bt.stride);
    }
    dim3 block(64, 4), grid((w + 63) / 64, (h + 3) / 4, bt.n);
    hipLaunchKernelGGL(k_wmf_build_list, grid, block, 0, s, buf_a, nnf_pitch, w, h, only_occlusion, list0, counts, bt.stride);
    const int pixels = w * h;
    int nblocks0 = (pixels + 3) / 4;
    if (nblocks0 > EPPM_WMF_MAX_BLOCKS) nblocks0 = EPPM_WMF_MAX_BLOCKS;
    int16_t *in = buf_a, *out = buf_b;
    for (int i = 0; i < num_iter; i++) {
        // occlusion-only lists shrink fast (most pixels are filled by the first launches): later launches get a
        // smaller grid so that a (nearly) empty launch costs a launch, not 1024 workgroups reading the counter
        int nblocks = nblocks0;
        if (only_occlusion) nblocks = (nblocks0 >> i) > 64 ? (nblocks0 >> i) : (nblocks0 < 64 ? nblocks0 : 64);
        hipLaunchKernelGGL(k_wmf_iter, dim3(nblocks, bt.n), dim3(256), 0, s, out, in, img, ipitch, w, h, nnf_pitch, wmf_lut, only_occlusion,
                           (i & 1) ? list1 : list0, counts + i, (i & 1) ? list0 : list1, counts + i + 1,
                           i > 0 ? changed + i - 1 : nullptr, changed + i, bt.stride);
        int16_t* t = in; in = out; out = t;
    }
    return in;
}

// ---------------------------------------------------------------------------------------------------
// refine :297-371: nearest valid pixel in the four directions (left, right, up, down), closest colour.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fill_holes(int16_t* __restrict__ nnf_out_, const int16_t* __restrict__ nnf_in_,
                                                    const uint32_t* __restrict__ img_, int ipitch, int w, int h, int npitch, size_t pstride)
{
    int16_t* __restrict__ nnf_out = pair_ptr(nnf_out_, pstride, blockIdx.z);
    const int16_t* __restrict__ nnf_in = pair_ptr(nnf_in_, pstride, blockIdx.z);
    const uint32_t* __restrict__ img = pair_ptr(img_, pstride, blockIdx.z);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    int cx_ = nnf_in[(y * npitch + x) * 2], cy_ = nnf_in[(y * npitch + x) * 2 + 1];
    if (!(cx_ >= 0 && cy_ >= 0)) {
        int ndx[4] = {cx_, cx_, cx_, cx_}, ndy[4] = {cy_, cy_, cy_, cy_};
        int nx[4] = {x, x, x, x}, ny[4] = {y, y, y, y};
        for (int c = x - 1; c >= 0; c--) { ndx[0] = nnf_in[(y * npitch + c) * 2]; ndy[0] = nnf_in[(y * npitch + c) * 2 + 1]; if (ndx[0] >= 0 && ndy[0] >= 0) { nx[0] = c; break; } }
        for (int c = x + 1; c < w; c++)  { ndx[1] = nnf_in[(y * npitch + c) * 2]; ndy[1] = nnf_in[(y * npitch + c) * 2 + 1]; if (ndx[1] >= 0 && ndy[1] >= 0) { nx[1] = c; break; } }
        for (int c = y - 1; c >= 0; c--) { ndx[2] = nnf_in[(c * npitch + x) * 2]; ndy[2] = nnf_in[(c * npitch + x) * 2 + 1]; if (ndx[2] >= 0 && ndy[2] >= 0) { ny[2] = c; break; } }
        for (int c = y + 1; c < h; c++)  { ndx[3] = nnf_in[(c * npitch + x) * 2]; ndy[3] = nnf_in[(c * npitch + x) * 2 + 1]; if (ndx[3] >= 0 && ndy[3] >= 0) { ny[3] = c; break; } }
        const rgbf cur = unpack_rgb(tex_rgba(img, ipitch, w, h, x, y));
        float minPixDiff = FLT_MAX;
        int fx = cx_, fy = cy_;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const rgbf np = unpack_rgb(tex_rgba(img, ipitch, w, h, nx[i], ny[i]));
            const float pd = max_abs_diff(cur, np);
            if (pd < minPixDiff && ndx[i] >= 0 && ndy[i] >= 0) {
                minPixDiff = pd;
                fx = (int)(int16_t)(ndx[i] - nx[i]);
                fy = (int)(int16_t)(ndy[i] - ny[i]);
            }
        }
        cx_ = (int)(int16_t)(fx + x);
        cy_ = (int)(int16_t)(fy + y);
    }
    nnf_out[(y * npitch + x) * 2] = (int16_t)cx_;
    nnf_out[(y * npitch + x) * 2 + 1] = (int16_t)cy_;
}
void launch_fill_holes(int16_t* nnf_out, const int16_t* nnf_in, const uint32_t* img, int ipitch, int w, int h, int nnf_pitch,
                       hipStream_t s, Batch bt)
{
    dim3 block(64, 4), grid((w + 63) / 64, (h + 3) / 4, bt.n);
    hipLaunchKernelGGL(k_fill_holes, grid, block, 0, s, nnf_out, nnf_in, img, ipitch, w, h, nnf_pitch, bt.stride);
}

// refine :636-655
__global__ __launch_bounds__(256) void k_nnf2flow(float* __restrict__ flow_, int fpitch, const int16_t* __restrict__ nnf_, int npitch, int w, int h, size_t pstride)
{
    float* __restrict__ flow = pair_ptr(flow_, pstride, blockIdx.z);
    const int16_t* __restrict__ nnf = pair_ptr(nnf_, pstride, blockIdx.z);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    const int dx = nnf[(y * npitch + x) * 2], dy = nnf[(y * npitch + x) * 2 + 1];
    float fx, fy;
    if (dx <= kInvalid || dy <= kInvalid) { fx = kUnknownFlow; fy = kUnknownFlow; }
    else { fx = (float)(dx - x); fy = (float)(dy - y); }
    flow[(y * fpitch + x) * 2] = fx;
    flow[(y * fpitch + x) * 2 + 1] = fy;
}
void launch_nnf2flow(float* flow, int flow_pitch, const int16_t* nnf, int nnf_pitch, int w, int h, hipStream_t s, Batch bt)
{
    dim3 block(64, 4), grid((w + 63) / 64, (h + 3) / 4, bt.n);
    hipLaunchKernelGGL(k_nnf2flow, grid, block, 0, s, flow, flow_pitch, nnf, nnf_pitch, w, h, bt.stride);
}

// compute_flow's epilogue (driver :302-306) on the device: interleaved float2 flow -> planar u then v (h*w floats each), so
// that the host only copies two contiguous planes after the D2H instead of de-interleaving 2*h*w floats in a scalar loop
__global__ __launch_bounds__(256) void k_split_flow(float* __restrict__ uv_, const float2* __restrict__ flow_, int n, size_t pstride)
{
    float* __restrict__ uv = pair_ptr(uv_, pstride, blockIdx.y);
    const float2* __restrict__ flow = pair_ptr(flow_, pstride, blockIdx.y);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 f = flow[i];
    uv[i] = f.x;
    uv[n + i] = f.y;
}
void launch_split_flow(float* uv, const float* flow, int n, hipStream_t s, Batch bt)
{
    hipLaunchKernelGGL(k_split_flow, dim3((n + 255) / 256, bt.n), dim3(256), 0, s, uv, (const float2*)flow, n, bt.stride);
}

}  // namespace eppm
