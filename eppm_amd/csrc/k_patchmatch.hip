// k_patchmatch.hip -- PatchMatch at the coarsest level (reference: bao_pmflow_kernel.cu:50-109 random
// field, :636-645 cost field, :1049-1181 segmented propagation, :1519-1594 random search).
//
// Determinism: every kernel realises the "lockstep" order of the racy original --
// all threads read before any thread writes, segment seeds are read at step 0, and the doubly visited
// forward pixel L is visited by segment 1 before segment 0.
#include "eppm_device.cuh"
#include "eppm_internal.h"

namespace eppm {

__device__ __forceinline__ Planes to_dev(const PlanesH& h)
{
    Planes p;
    p.img1 = h.img1; p.img2 = h.img2; p.cen1 = h.cen1; p.cen2 = h.cen2;
    p.w = h.w; p.h = h.h; p.ipitch = h.ipitch; p.cpitch = h.cpitch;
    return p;
}

__device__ __forceinline__ Xorwow load_state(const uint32_t* p)
{
    Xorwow s;
    s.v0 = p[0]; s.v1 = p[1]; s.v2 = p[2]; s.v3 = p[3]; s.v4 = p[4]; s.d = p[5];
    return s;
}
__device__ __forceinline__ void store_state(uint32_t* p, const Xorwow& s)
{
    p[0] = s.v0; p[1] = s.v1; p[2] = s.v2; p[3] = s.v3; p[4] = s.v4; p[5] = s.d;
}

// ---------------------------------------------------------------------------------------------------
// Random initial NNF (d_setup_randgen + d_gen_rand_field, kernel.cu:50-109).  The reference lets thread
// (0,0) of each 16x16 block draw 2x256 numbers serially from the block's XORWOW stream; here the 64
// lanes of one wave each own 8 consecutive draws of the same stream (lane states precomputed on the
// host by walking the stream once, xorwow_host.cpp), so the numbers are identical and the draw is
// parallel.  Also rewinds the search states to the position after the 512 init draws.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_pm_init_field(PmRngDev rng, int16_t* __restrict__ nnf, int npitch, int w, int h)
{
    const int bx = blockIdx.x, by = blockIdx.y, lane = threadIdx.x;
    const int block_id = by * rng.gx + bx;
    const size_t so = ((size_t)block_id * 64 + lane) * 6;
    Xorwow st = load_state(rng.init_tab + so);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t r1 = xorwow_next(st);
        const uint32_t r2 = xorwow_next(st);
        const int t = lane * 4 + q;                // t = 16*i + j, row-major over the block (kernel.cu:90-101)
        const int x = bx * kBlock + (t & 15), y = by * kBlock + (t >> 4);
        if (x < w && y < h) {
            nnf[(y * npitch + x) * 2 + 0] = (int16_t)(r1 % (uint32_t)(w + 1));
            nnf[(y * npitch + x) * 2 + 1] = (int16_t)(r2 % (uint32_t)(h + 1));
        }
    }
    // search stream position = 512 draws in (states are re-initialised on every call, kernel.cu:160)
#pragma unroll
    for (int k = 0; k < 6; k++) rng.work[so + k] = rng.iter_tab[so + k];
}

void launch_pm_init_field(const PmRngDev& rng, int16_t* nnf, int nnf_pitch, int w, int h, hipStream_t s)
{
    hipLaunchKernelGGL(k_pm_init_field, dim3(rng.gx, rng.gy), dim3(64), 0, s, rng, nnf, nnf_pitch, w, h);
}

// ---------------------------------------------------------------------------------------------------
// Initial cost field (kernel.cu:636-645)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pm_cost_field(PlanesH Ph, float* __restrict__ cost, int cpitch,
                                                       const int16_t* __restrict__ nnf, int npitch,
                                                       const float* __restrict__ lut, int R)
{
    __shared__ PatchLut L;
    load_patch_lut(L, lut, R, threadIdx.y * kBlock + threadIdx.x, 256);
    __syncthreads();
    const Planes P = to_dev(Ph);
    const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y * kBlock + threadIdx.y;
    if (x >= P.w || y >= P.h) return;
    const int dx = nnf[(y * npitch + x) * 2], dy = nnf[(y * npitch + x) * 2 + 1];
    cost[y * cpitch + x] = patch_dist(P, L, R, x, y, dx, dy);
}

void launch_pm_cost_field(const PlanesH& P, float* cost, int cost_pitch, const int16_t* nnf, int nnf_pitch, const float* lut,
                          int R, hipStream_t s)
{
    dim3 grid((P.w + kBlock - 1) / kBlock, (P.h + kBlock - 1) / kBlock), block(kBlock, kBlock);
    hipLaunchKernelGGL(k_pm_cost_field, grid, block, 0, s, P, cost, cost_pitch, nnf, nnf_pitch, lut, R);
}

// ---------------------------------------------------------------------------------------------------
// Segmented scan-line propagation (kernel.cu:1049-1181).  One thread = (line, segment); all segments
// of a line live in one workgroup so that the two ordering points of the lockstep semantics are
// workgroup barriers: (a) every seed is read before any walk writes, (b) segment 1's first step
// (pixel L) precedes segment 0's last step (the same pixel) -- barrier after step 0.
// ---------------------------------------------------------------------------------------------------
template <bool IS_ROW, bool REVERSE>
__global__ __launch_bounds__(1024) void k_pm_seg_propagate(PlanesH Ph, float* __restrict__ cost, int cpitch,
                                                           int16_t* __restrict__ nnf, int npitch,
                                                           const float* __restrict__ lut, int R, int L_, int nseg,
                                                           int lines_per_block)
{
    __shared__ PatchLut L;
    load_patch_lut(L, lut, R, threadIdx.x, blockDim.x);
    const Planes P = to_dev(Ph);
    const int len = IS_ROW ? P.w : P.h, lines = IS_ROW ? P.h : P.w;
    const int lline = threadIdx.x / nseg, seg = threadIdx.x % nseg;
    const int line = blockIdx.x * lines_per_block + lline;
    const bool active = (lline < lines_per_block) && (line < lines);
    int start, count, i, step;
    if (!REVERSE) {
        start = (seg == 0) ? 0 : seg * L_ - 1;
        const int end = min(len - 1, start + L_);
        count = end - start;
        i = start + 1;
        step = 1;
    } else {
        start = (seg + 1) * L_;
        if (start >= len) start = len - 1;
        count = start - seg * L_;
        i = start - 1;
        step = -1;
    }
    int px = 0, py = 0;
    if (active) {
        const int sidx = IS_ROW ? (line * npitch + start) : (start * npitch + line);
        px = nnf[sidx * 2];
        py = nnf[sidx * 2 + 1];
    }
    __syncthreads();   // LUT ready; (a) all seeds read
    for (int s = 0; s < L_; s++) {
        if (active && s < count) {
            const int x = IS_ROW ? i : line, y = IS_ROW ? line : i;
            const int nidx = y * npitch + x, cidx = y * cpitch + x;
            const float cur_best = cost[cidx];
            if (IS_ROW) px = REVERSE ? max(px - 1, 0) : min(px + 1, P.w - 1);
            else        py = REVERSE ? max(py - 1, 0) : min(py + 1, P.h - 1);
            const float cv = patch_dist(P, L, R, x, y, px, py);
            if (cv < cur_best) {
                nnf[nidx * 2] = (int16_t)px;
                nnf[nidx * 2 + 1] = (int16_t)py;
                cost[cidx] = cv;
            } else {
                px = nnf[nidx * 2];
                py = nnf[nidx * 2 + 1];
            }
            i += step;
        }
        if (!REVERSE && s == 0) __syncthreads();   // (b)
    }
}

void launch_pm_seg_propagate(const PlanesH& P, float* cost, int cost_pitch, int16_t* nnf, int nnf_pitch, const float* lut,
                             int R, int seg_len, int dir, hipStream_t s)
{
    const bool is_row = (dir == 0 || dir == 2);
    const int len = is_row ? P.w : P.h, lines = is_row ? P.h : P.w;
    const int nseg = (len + seg_len - 1) / seg_len;
    if (nseg > 1024) return;   // caller validates (image wider than 10240*4 px)
    int lpb = 256 / nseg;
    if (lpb < 1) lpb = 1;
    int threads = ((nseg * lpb + 63) / 64) * 64;
    dim3 grid((lines + lpb - 1) / lpb), block(threads);
    switch (dir) {
        case 0: hipLaunchKernelGGL((k_pm_seg_propagate<true, false>), grid, block, 0, s, P, cost, cost_pitch, nnf, nnf_pitch, lut, R, seg_len, nseg, lpb); break;
        case 1: hipLaunchKernelGGL((k_pm_seg_propagate<false, false>), grid, block, 0, s, P, cost, cost_pitch, nnf, nnf_pitch, lut, R, seg_len, nseg, lpb); break;
        case 2: hipLaunchKernelGGL((k_pm_seg_propagate<true, true>), grid, block, 0, s, P, cost, cost_pitch, nnf, nnf_pitch, lut, R, seg_len, nseg, lpb); break;
        default: hipLaunchKernelGGL((k_pm_seg_propagate<false, true>), grid, block, 0, s, P, cost, cost_pitch, nnf, nnf_pitch, lut, R, seg_len, nseg, lpb); break;
    }
}

// ---------------------------------------------------------------------------------------------------
// Random search (kernel.cu:1519-1594): G guesses at radii search_range, /2, ... around the pre-search
// best, evaluated in order with strict <.  Random numbers: the block's XORWOW stream, 2x256 draws per
// guess in row-major pixel order; wave 0 produces the 512*G draws of this launch in parallel (lane l
// owns draws [per_lane*l, per_lane*(l+1)) ), then jumps its state over the other lanes' draws with the
// GF(2) skip matrix so that the next launch continues the same stream.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pm_random_search(PlanesH Ph, PmRngDev rng, float* __restrict__ cost, int cpitch,
                                                          int16_t* __restrict__ nnf, int npitch,
                                                          const float* __restrict__ lut, int R, int search_range, int G)
{
    __shared__ PatchLut L;
    __shared__ int16_t s_rand[8 * 512];
    const int tid = threadIdx.y * kBlock + threadIdx.x;
    const int block_id = blockIdx.y * rng.gx + blockIdx.x;
    load_patch_lut(L, lut, R, tid, 256);
    if (tid < 64) {
        const size_t so = ((size_t)block_id * 64 + tid) * 6;
        Xorwow st = load_state(rng.work + so);
        const int base = rng.per_lane * tid;
        for (int q = 0; q < rng.per_lane; q++) s_rand[base + q] = (int16_t)xorwow_next(st);   // short(rdn), :1550-1551
        // jump over the other 63 lanes' draws: v <- v * skip_mat over GF(2); Weyl counter by multiplication
        const uint32_t v[5] = {st.v0, st.v1, st.v2, st.v3, st.v4};
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
        for (int wd = 0; wd < 5; wd++) {
            const uint32_t vw = v[wd];
#pragma unroll 8
            for (int b = 0; b < 32; b++) {
                const uint32_t* row = rng.skip_mat + (wd * 32 + b) * 5;
                const uint32_t m = 0u - ((vw >> b) & 1u);
                a0 ^= m & row[0]; a1 ^= m & row[1]; a2 ^= m & row[2]; a3 ^= m & row[3]; a4 ^= m & row[4];
            }
        }
        st.v0 = a0; st.v1 = a1; st.v2 = a2; st.v3 = a3; st.v4 = a4;
        st.d += rng.skip_weyl;
        store_state(rng.work + so, st);
    }
    __syncthreads();
    const Planes P = to_dev(Ph);
    const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y * kBlock + threadIdx.y;
    if (x >= P.w || y >= P.h) return;
    const int nidx = y * npitch + x, cidx = y * cpitch + x;
    int bx = nnf[nidx * 2], by = nnf[nidx * 2 + 1];
    float best_cost = cost[cidx];
    int gxs[8], gys[8];
    int mag = search_range;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k < G) {
            const uint32_t rdn1 = (uint32_t)(int32_t)s_rand[512 * k + 2 * tid];       // short -> unsigned int
            const uint32_t rdn2 = (uint32_t)(int32_t)s_rand[512 * k + 2 * tid + 1];
            const int xmin = max(bx - mag, 0), xmax = min(bx + mag + 1, P.w + 1);
            const int ymin = max(by - mag, 0), ymax = min(by + mag + 1, P.h + 1);
            gxs[k] = (int)(int16_t)((uint32_t)xmin + rdn1 % (uint32_t)(xmax - xmin));
            gys[k] = (int)(int16_t)((uint32_t)ymin + rdn2 % (uint32_t)(ymax - ymin));
            if (mag / 2 >= 1) mag /= 2;
        }
    }
#pragma unroll 1
    for (int k = 0; k < G; k++) {
        const float cv = patch_dist(P, L, R, x, y, gxs[k], gys[k]);
        if (cv < best_cost) { bx = gxs[k]; by = gys[k]; best_cost = cv; }
    }
    nnf[nidx * 2] = (int16_t)bx;
    nnf[nidx * 2 + 1] = (int16_t)by;
    cost[cidx] = best_cost;
}

void launch_pm_random_search(const PlanesH& P, const PmRngDev& rng, float* cost, int cost_pitch, int16_t* nnf, int nnf_pitch,
                             const float* lut, int R, int search_range, int num_guess, hipStream_t s)
{
    dim3 grid(rng.gx, rng.gy), block(kBlock, kBlock);
    hipLaunchKernelGGL(k_pm_random_search, grid, block, 0, s, P, rng, cost, cost_pitch, nnf, nnf_pitch, lut, R, search_range,
                       num_guess);
}

}  // namespace eppm
