// k_patchmatch.hip -- PatchMatch at the coarsest level (reference: bao_pmflow_kernel.cu:50-109 random
// field, :636-645 cost field, :1049-1181 segmented propagation, :1519-1594 random search).
//
// Determinism: every kernel realises the "lockstep" order of the racy original --
// all threads read before any thread writes, segment seeds are read at step 0, and the doubly visited
// forward pixel L is visited by segment 1 before segment 0.
//
// MI355X mapping: the quarter-resolution level has only ~28 k pixels (436 waves of one pixel per lane on
// 1024 SIMDs) and the sweeps only ~2.8 k serial chains per direction, so the kernels spread ONE patch
// evaluation over 16 lanes (sweeps) or the six guesses of a pixel over separate lanes (search), and the
// forward and backward problems of a pair share every launch.
#include "eppm_device.cuh"
#include "eppm_internal.h"

#ifndef EPPM_LPC9
#ifdef EPPM_TOL
#define EPPM_LPC9 32      // tolerance library: a lane = a chunk of 5 samples (coop_chunk): 20 of 32 lanes work, whatever the launch size
#else
#define EPPM_LPC9 16      // lanes per sweep chain at patch radius 9 (100 samples); doubled for launches that cannot fill the chip, see launch_pm_sweep
#endif
#endif
#ifndef EPPM_SWEEP_GB
#define EPPM_SWEEP_GB 7   // sample gathers a sweep lane keeps in flight (per image)
#endif
#ifndef EPPM_LPC17
#define EPPM_LPC17 64     // ... at patch radius 17 (324 samples)
#endif
#ifndef EPPM_LPC9_SPEC
#ifdef EPPM_TOL
#define EPPM_LPC9_SPEC 32     // tolerance library: see EPPM_LPC9
#else
#define EPPM_LPC9_SPEC 16     // lanes per chain in phase B at radius 9 (4 were tried -- a quarter of the waves, one round of workgroups --
#endif
#endif                        // and lost: 50-56 vs 32-41 us per 8-pair launch, the evaluations after accepted candidates take four times as long)
#ifndef EPPM_LPC17_SPEC
#define EPPM_LPC17_SPEC 64    // ... at radius 17
#endif


namespace eppm {

__device__ __forceinline__ Planes to_dev(const PlanesH& h)
{
    Planes p;
    p.pk1 = (const float4*)h.pk1; p.pk2 = (const float4*)h.pk2;
    p.w = h.w; p.h = h.h; p.pitch = h.pitch;
    return p;
}

// problem q of a launch: direction q % n of pair q / n; pair k's planes lie k * stride bytes after pair 0's
__device__ __forceinline__ PmProblem pm_problem(const PmBatch& B, unsigned q)
{
    PmProblem p = B.p[q % (unsigned)B.n];
    const unsigned pair = q / (unsigned)B.n;
    p.P.pk1 = pair_ptr_opt(p.P.pk1, B.stride, pair);
    p.P.pk2 = pair_ptr_opt(p.P.pk2, B.stride, pair);
    p.P.pc1 = pair_ptr_opt(p.P.pc1, B.stride, pair);
    p.P.pc2 = pair_ptr_opt(p.P.pc2, B.stride, pair);
    p.P.pp1 = pair_ptr_opt(p.P.pp1, B.stride, pair);
    p.P.pp2 = pair_ptr_opt(p.P.pp2, B.stride, pair);
    p.cost = pair_ptr_opt(p.cost, B.stride, pair);
    p.nnf = pair_ptr_opt(p.nnf, B.stride, pair);
    p.nnf_alt = pair_ptr_opt(p.nnf_alt, B.stride, pair);
    p.spec = pair_ptr_opt(p.spec, B.stride, pair);
    p.scand = pair_ptr_opt(p.scand, B.stride, pair);
    p.wl = pair_ptr_opt(p.wl, B.stride, pair);
    p.seed = pair_ptr_opt(p.seed, B.stride, pair);
    p.rng_work = pair_ptr_opt(p.rng_work, B.stride, pair);
    p.rng_work_next = pair_ptr_opt(p.rng_work_next, B.stride, pair);
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
// host by walking the stream once, eppm_api.cpp rng_create), so the numbers are identical and the draw
// is parallel.  Also rewinds the search states to the position after the 512 init draws.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_pm_init_field(PmBatch B, PmRngDev rng)
{
    const PmProblem pr = pm_problem(B, blockIdx.z);
    const int bx = blockIdx.x, by = blockIdx.y, lane = threadIdx.x;
    const int w = pr.P.w, h = pr.P.h;
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
            pr.nnf[(y * B.npitch + x) * 2 + 0] = (int16_t)(r1 % (uint32_t)(w + 1));
            pr.nnf[(y * B.npitch + x) * 2 + 1] = (int16_t)(r2 % (uint32_t)(h + 1));
            if (pr.scand) {                      // new images: the sweeps' evaluation cache starts empty
#pragma unroll
                for (int d = 0; d < 4; d++) pr.scand[d * B.cache_plane + y * B.cpitch + x] = -1;
            }
        }
    }
    // search stream position = 512 draws in (states are re-initialised on every call, kernel.cu:160)
#pragma unroll
    for (int k = 0; k < 6; k++) pr.rng_work[so + k] = rng.iter_tab[so + k];
    // a new run: sweep numbers start at 0 again, so the work list's lengths and stamps do
    if (pr.wl)
        for (int wi = block_id * 64 + lane; wi < 16 + 6 * B.wl_units; wi += rng.gx * rng.gy * 64) pr.wl[wi] = 0u;
}

void launch_pm_init_field(const PmBatch& b, const PmRngDev& rng, hipStream_t s)
{
    hipLaunchKernelGGL(k_pm_init_field, dim3(rng.gx, rng.gy, b.n * b.npairs), dim3(64), 0, s, b, rng);
}

// ---------------------------------------------------------------------------------------------------
// Initial cost field (kernel.cu:636-645)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pm_cost_field(PmBatch B, const float* __restrict__ lut, int R)
{
    __shared__ EPPM_LUT_ALIGN PatchLut L;
    load_patch_lut(L, lut, R, threadIdx.y * kBlock + threadIdx.x, 256);
    __syncthreads();
    const PmProblem pr = pm_problem(B, blockIdx.z);
    const Planes P = to_dev(pr.P);
    const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y * kBlock + threadIdx.y;
    if (x >= P.w || y >= P.h) return;
    const int dx = pr.nnf[(y * B.npitch + x) * 2], dy = pr.nnf[(y * B.npitch + x) * 2 + 1];
    pr.cost[y * B.cpitch + x] = patch_dist(P, L, R, x, y, dx, dy);
}


// RT = 9 / 17: the source samples of the workgroup's 16x4 pixels come from an LDS tile (16+2R)x(4+2R), loaded once,
// clamped at load -- one LDS read per sample instead of a clamped address and a gather; RT = 0: any radius, source
// samples gathered from the plane.
template <int RT> struct SearchLut { using type = PatchLutT<RT + 1>; };
template <> struct SearchLut<0> { using type = PatchLut; };          // any radius the ABI accepts

__device__ __forceinline__ float patch_dist_any(const Planes& P, const PatchLut& L, int R, int x1, int y1, int x2, int y2) { return patch_dist(P, L, R, x1, y1, x2, y2); }
template <int M> __device__ __forceinline__ float patch_dist_any(const Planes&, const PatchLutT<M>&, int, int, int, int, int) { return 0.0f; }   // never called (RT != 0)

// PK = 1: the target texels are gathered from the 4-byte plane pc2 = {R, G, B, census} and converted at use (make_texel, the function
// that built the float4 plane: the same bits).  A 64-lane gather of 4 bytes costs the L1 38 clocks where one of 16 bytes costs 52-78
// (tools/ubench/gather_rate.hip), the conversion 12 VALU instructions per texel: for launches whose search runs at the L1's lane
// rate with VALU slots to spare -- radius 17, or one small pair per launch -- not for the batched radius-9 launches (VALU bound).
// PK = 2 (tolerance library): the S samples of a patch row are S consecutive words of the target's column-parity plane (PlanesH::pp2):
// 3 gathers per row of 10 samples (16 + 16 + 8 bytes) instead of 10, unpacked by unpack_texel.  With the patch term at a third of its
// exact instruction count these kernels run at the L1's lane rate; this divides their gathers by 3.3.
// (The source half of a sample's weight is the same for the six guesses of a pixel; forming it once per workgroup in LDS -- 26 KB,
// [sample][pixel] -- and a barrier LOSES here as it did in the exact library: search 215 -> 234 us per 8-pair launch, bench 304 -> 298,
// profiles/r06x_c_search_hoist.txt.)
#define EPPM_PM_PRAGMA_(x) _Pragma(#x)
#define EPPM_PM_UNROLL(n) EPPM_PM_PRAGMA_(unroll n)
template <int RT, int PK = 0, class LUT>
__device__ __forceinline__ float search_patch_dist(const Planes& P, const LUT& L, int R, const float4* __restrict__ s_src, int TW,
                                                   int tx, int ty, int x1, int y1, int x2, int y2, const PlanesH& PH)
{
    if (RT == 0) return patch_dist_any(P, L, R, x1, y1, x2, y2);
    constexpr int S = RT + 1;
    const int pitch16 = P.pitch << 4;
    const rgbf c1 = texel_rgb(s_src[(ty + RT) * TW + tx + RT]);
    const rgbf c2 = texel_rgb(texel_at(P.pk2, texel_off(pitch16, P.w, P.h, x2, y2)));
    PatchSum sum;
    constexpr int CS = tol_chunk(RT);          // tolerance library: chunk of the canonical summation order (PatchSum)
    static_assert(S % CS == 0, "whole chunks per row");
#ifdef EPPM_TOL
    if constexpr (PK == 2) {
        static_assert(S % 4 == 2, "a row = whole dwordx4 gathers + one dwordx2");
        // buffer loads: a dwordx4 at a 4-byte aligned per-lane offset in ONE instruction (a global load of that alignment is split by the
        // compiler), 32-bit offsets; the descriptor is built from workgroup-uniform values (the problem's plane, its size)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(PH.pp2), 0, 2 * P.h * PH.pp_pitch * 4, 0x00020000);
        const int xp = x2 - RT + PH.pp_pad;                       // padded column of the row's first sample, >= 1
        const int rb = ((xp & 1) * P.h) * PH.pp_pitch + (xp >> 1);
        uint32_t two = 2u;
        asm volatile("" : "+v"(two));
#pragma unroll 2
        for (int ii = 0; ii < S; ii++) {
            const int ro = (rb + iclamp(y2 + 2 * ii - RT, 0, P.h - 1) * PH.pp_pitch) * 4;      // byte offset of the row's first sample
            const float4* __restrict__ srow = s_src + (ty + 2 * ii) * TW + tx;
            uint32_t wq[S];
#pragma unroll
            for (int g = 0; g < S / 4; g++) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, ro + 16 * g, 0, 0);
                wq[4 * g] = v.x; wq[4 * g + 1] = v.y; wq[4 * g + 2] = v.z; wq[4 * g + 3] = v.w;
            }
            { const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, ro + 4 * (S - 2), 0, 0); wq[S - 2] = v.x; wq[S - 1] = v.y; }
            // terms of a row the compiler may interleave: a divisor of S that keeps the kernel at 65 VGPRs = 7 waves per SIMD (radius 9: 5 of
            // 10 -- all 10: 77 VGPRs = 6 waves, search 226 -> 213 us; radius 17: 9 of 18 -- all 18: 98 VGPRs)
#ifdef EPPM_SEARCH_TERM_UNROLL
            constexpr int TU = EPPM_SEARCH_TERM_UNROLL;
#else
            constexpr int TU = (S % 5 == 0) ? 5 : (S % 9 == 0) ? 9 : S;
#endif
EPPM_PM_UNROLL((TU))
            for (int jj = 0; jj < S; jj++) {
                float ct, wt;
                patch_terms(srow[2 * jj], unpack_texel(wq[jj], two), c1, c2, L.gsp[ii * S + jj], L.tab(), ct, wt);
                sum.add(ct, wt);
                if ((jj + 1) % CS == 0) sum.flush();
            }
        }
        return sum.result();
    }
#endif
    const uint32_t* __restrict__ pc2 = PH.pc2;
    for (int ii = 0; ii < S; ii++) {
        const int i = 2 * ii - RT;
        const unsigned r2 = __umul24((unsigned)iclamp(y2 + i, 0, P.h - 1), (unsigned)pitch16);
        const float4* __restrict__ srow = s_src + (ty + 2 * ii) * TW + tx;
        for (int j0 = 0; j0 < S; j0 += 5) {
            float4 q1[5], q2[5];
            uint32_t w2[PK == 1 ? 5 : 1];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int jj = min(j0 + k, S - 1);
                q1[k] = srow[2 * jj];
                const unsigned o2 = r2 + ((unsigned)iclamp(x2 + 2 * jj - RT, 0, P.w - 1) << 4);
                if (PK == 1) w2[k] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(pc2) + (o2 >> 2));
                else q2[k] = texel_at(P.pk2, o2);
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                if (j0 + k < S) {
                    float ct, wt;
                    if (PK == 1) q2[k] = make_texel(w2[k], w2[k] >> 24);
                    patch_terms(q1[k], q2[k], c1, c2, L.gsp[ii * S + j0 + k], L.tab(), ct, wt);
                    sum.add(ct, wt);
                    if ((j0 + k + 1) % CS == 0) sum.flush();
                }
            }
        }
    }
    return sum.result();
}

// the cost field with the source samples of the 16x16 block from an LDS tile, as in the search and in phase A of the sweeps (radius 9 / 17)
template <int RT, int PK = 0>
__global__ __launch_bounds__(256) void k_pm_cost_field_tile(PmBatch B, const float* __restrict__ lut, int R)
{
    using LUT = typename SearchLut<RT>::type;
    constexpr int TW = kBlock + 2 * RT;
    __shared__ float4 s_src[TW * TW];
    __shared__ EPPM_LUT_ALIGN LUT L;
    const int tid = threadIdx.y * kBlock + threadIdx.x;
    load_patch_lut(L, lut, R, tid, 256);
    const PmProblem pr = pm_problem(B, blockIdx.z);
    const Planes P = to_dev(pr.P);
    const int x0 = blockIdx.x * kBlock - RT, y0 = blockIdx.y * kBlock - RT;
    for (int t = tid; t < TW * TW; t += 256) {
        const int sy = iclamp(y0 + t / TW, 0, P.h - 1), sx = iclamp(x0 + t % TW, 0, P.w - 1);
        s_src[t] = P.pk1[(unsigned)(sy * P.pitch + sx)];
    }
    __syncthreads();
    const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y * kBlock + threadIdx.y;
    if (x >= P.w || y >= P.h) return;
    const int dx = pr.nnf[(y * B.npitch + x) * 2], dy = pr.nnf[(y * B.npitch + x) * 2 + 1];
    pr.cost[y * B.cpitch + x] = search_patch_dist<RT, PK>(P, L, R, s_src, TW, threadIdx.x, threadIdx.y, x, y, dx, dy, pr.P);
}

// tolerance library: every problem of the launch has its target's column-parity plane (PlanesH::pp2)
static bool pm_has_parity(const PmBatch& b)
{
#ifdef EPPM_TOL
    return b.p[0].P.pp2 && (b.n < 2 || b.p[1].P.pp2);
#else
    return false;
#endif
}

void launch_pm_cost_field(const PmBatch& b, const float* lut, int R, hipStream_t s)
{
    const int w = b.p[0].P.w, h = b.p[0].P.h;
    dim3 grid((w + kBlock - 1) / kBlock, (h + kBlock - 1) / kBlock, b.n * b.npairs), block(kBlock, kBlock);
#ifndef EPPM_COST_FIELD_TILE
#define EPPM_COST_FIELD_TILE 1
#endif
    if (EPPM_COST_FIELD_TILE && R == 9 && pm_has_parity(b)) hipLaunchKernelGGL((k_pm_cost_field_tile<9, 2>), grid, block, 0, s, b, lut, R);
    else if (EPPM_COST_FIELD_TILE && R == 17 && pm_has_parity(b)) hipLaunchKernelGGL((k_pm_cost_field_tile<17, 2>), grid, block, 0, s, b, lut, R);
    else if (EPPM_COST_FIELD_TILE && R == 9) hipLaunchKernelGGL(k_pm_cost_field_tile<9>, grid, block, 0, s, b, lut, R);
    else if (EPPM_COST_FIELD_TILE && R == 17) hipLaunchKernelGGL(k_pm_cost_field_tile<17>, grid, block, 0, s, b, lut, R);
    else hipLaunchKernelGGL(k_pm_cost_field, grid, block, 0, s, b, lut, R);
}

// ---------------------------------------------------------------------------------------------------
// Segmented scan-line propagation (kernel.cu:1049-1181), cooperative form.
//
// A chain = (line, segment): up to L sequential steps, each one patch evaluation whose candidate depends
// on the previous step's outcome.  The reference gives a chain ONE thread; at quarter resolution that is
// ~45 waves on the whole chip.  Here one chain owns a DPP row of 16 lanes.  Per step the S*S samples of
// the patch (row-major, the reference's order) are dealt to the lanes in contiguous chunks of CH; every
// lane computes the (cost*w, w) terms of its chunk, then the two running sums travel lane to lane
// (row_ror:1) while every lane adds its chunk in order: the sums are formed in exactly the reference's
// sequential order, only the expensive per-sample terms are computed in parallel.
//
// Ordering points of the lockstep semantics:
//  (a) seeds are read from nnf_in, all writes go to nnf_out (ping-pong) -- no cross-workgroup hazard;
//  (b) forward pixel L: segment 1's first step precedes segment 0's last step; segments 0 and 1 of a
//      line are always in one workgroup (chains per line padded to an even count, 16 chains per
//      workgroup) and a workgroup barrier follows step 0.
// cost is updated in place: a pixel's cost is touched only by its visitor(s).
// ---------------------------------------------------------------------------------------------------
// lane i receives lane i-1: inside a 16-lane DPP row (row_ror:1) or across the whole wave (wave_shr:1, gfx9)
template <int LPC>
__device__ __forceinline__ float dpp_prev_lane(float v)
{
    if (LPC == 4) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x93 /* quad_perm:[3,0,1,2] */, 0xf, 0xf, false));
    if (LPC == 16) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}

// Samples per lane of a cooperative evaluation.  Exact library: the S*S samples dealt evenly to the LPC lanes.  Tolerance library: a lane
// = a chunk of the canonical summation order (eppm_device.cuh: PatchSum) -- 5 samples at radius 9 (20 of 32 lanes work), 6 at radius 17
// (54 of 64) --, whatever LPC is.
template <int R, int LPC>
__host__ __device__ constexpr int coop_chunk()
{
#ifdef EPPM_TOL
    return tol_chunk(R);
#else
    return ((R + 1) * (R + 1) + LPC - 1) / LPC;
#endif
}
// The sums of a cooperative evaluation from the lanes' terms tc[], tw[] (zero past the last sample); complete in lane NL - 1 of the group.
// Exact library: the two running sums hop lane to lane while EVERY lane adds its whole chunk at every hop -- the additions happen in the
// reference's order (only the sum that travels through lane ln at hop ln is the real one).  Tolerance library: every lane first sums
// its own chunk from zero (fused multiply-adds), then the totals hop with one addition per hop: the canonical order of PatchSum, the
// same bits as a lane that evaluates alone.  (A DPP tree instead of the hops was 8 % cheaper and wrong: its result differed in the
// last bit from the serial kernels' and even between the lanes of one chain, and on images whose candidates TIE -- flat regions, saturated
// blocks -- strict "<" then accepted in one kernel what another had stored as equal: 0.34 px on a fuzz case)
template <int LPC, int CH, int NS, int NL>
__device__ __forceinline__ void coop_chain_sum(const float (&tc)[CH], const float (&tw)[CH], float& ac, float& aw)
{
#ifdef EPPM_TOL
    float pc = 0.0f, pw = 0.0f;
#pragma unroll
    for (int q = 0; q < CH; q++) patch_accum(pc, pw, tc[q], tw[q]);
#pragma unroll
    for (int ln = 0; ln < NL; ln++) {
        if (ln > 0) { ac = dpp_prev_lane<LPC>(ac); aw = dpp_prev_lane<LPC>(aw); }
        ac += pc; aw += pw;
    }
#else
#pragma unroll
    for (int ln = 0; ln < NL; ln++) {
        if (ln > 0) { ac = dpp_prev_lane<LPC>(ac); aw = dpp_prev_lane<LPC>(aw); }
#pragma unroll
        for (int q = 0; q < CH; q++) {
            if (ln * CH + q < NS) patch_accum(ac, aw, tc[q], tw[q]);
        }
    }
#endif
}

// One patch evaluation spread over the LPC lanes of a DPP row (16) or of a whole wave (64), for kernels whose source samples lie in
// the (kBlock + 2 RT)^2 LDS tile of a 16x16 block: the S*S samples are dealt to the lanes in contiguous chunks, each lane forms the
// terms of its chunk, and the two running sums hop lane to lane while every lane adds its chunk -- the reference's order of additions,
// as in the cooperative sweep.  r = lane within the group; every lane of the group returns the cost.
template <int RT, int LPC, class LUT>
__device__ __forceinline__ float coop_patch_dist(const Planes& P, const LUT& L, const float4* __restrict__ s_src, int TW, int tx, int ty,
                                                 int x2, int y2, int r)
{
    constexpr int S = RT + 1, NS = S * S, CH = coop_chunk<RT, LPC>(), NL = (NS + CH - 1) / CH;
    static_assert(NL <= LPC, "a lane per chunk");
    const int pitch16 = P.pitch << 4, wmax16 = (P.w - 1) << 4;
    const rgbf c1 = texel_rgb(s_src[(ty + RT) * TW + tx + RT]);
    const rgbf c2 = texel_rgb(texel_at(P.pk2, texel_off(pitch16, P.w, P.h, x2, y2)));
    const int t0 = r * CH;
    float tc[CH], tw[CH];
    float4 q2[CH];
    int so[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int t = min(t0 + k, NS - 1), ii = t / S, jj = t - ii * S;
        so[k] = (ty + 2 * ii) * TW + tx + 2 * jj;
        q2[k] = texel_at(P.pk2, texel_off16(pitch16, wmax16, P.h - 1, (x2 + 2 * jj - RT) << 4, y2 + 2 * ii - RT));
    }
#pragma unroll
    for (int k = 0; k < CH; k++) {
        tc[k] = 0.0f; tw[k] = 0.0f;
        if (t0 + k < NS) patch_terms(s_src[so[k]], q2[k], c1, c2, L.gsp[t0 + k], L.tab(), tc[k], tw[k]);
    }
    float ac = 0.0f, aw = 0.0f;
    coop_chain_sum<LPC, CH, NS, NL>(tc, tw, ac, aw);
    const int src = ((threadIdx.x & 63) / LPC) * LPC + (NL - 1);      // lane holding the complete sums (wave-relative)
    return __shfl(ac, src, 64) / __shfl(aw, src, 64);
}

// LPC = lanes per chain (16, 32 or 64)
#ifdef EPPM_SWEEP_WAVES
#define EPPM_SWEEP_OCC __attribute__((amdgpu_waves_per_eu(EPPM_SWEEP_WAVES, EPPM_SWEEP_WAVES)))
#else
#define EPPM_SWEEP_OCC
#endif
// TILE: the workgroup's CPB chains are SEGS consecutive segments of LINES lines of equal parity (y, y+2, ..): the sampled rows of
// neighbouring lines of one parity coincide (offsets -R, -R+2, .. R), so S + LINES - 1 rows of SEGS*L + 2R texels, plus the LINES
// lines themselves (patch centres), hold every source texel the workgroup's patches can touch.  They are staged in LDS once,
// clamped at load; the source half of the sample gathers then never reaches the L1 (which these launches load to 60 % at one
// 16-byte lane-fetch per clock, tools/ubench/gather_rate.hip).  TW = tile row length (odd: spreads a chain's ds_read_b128 over the banks).
// ---- merged form of the speculative sweeps: lists and stamps of all four directions live side by side --------------------------
// PmProblem::wl beyond the two-launch form's words: counters wl[8 + 4 * (iteration & 1) + d]; stamps [16 + 2U + d U, ..) = 1 + the
// iteration that listed the unit last; lists [16 + 6U + d U, ..).  d: 0 row forward, 1 column forward, 2 row reverse, 3 column reverse.
// Lists the unit (segments 2u, 2u + 1 of its line) whose chains visit pixel (px, py) in direction d, once per iteration.
__device__ __forceinline__ void merged_list_unit(uint32_t* __restrict__ wl, const PmBatch& B, int d, int px, int py)
{
    const bool row = (d & 1) == 0;
    const int along = row ? px : py, ln = row ? py : px, nseg = row ? B.nseg_row : B.nseg_col;
    const unsigned seg = (d < 2 && along < B.seg_len) ? 0u : (unsigned)(along / B.seg_len);
    const unsigned unit = (unsigned)ln * ((unsigned)(nseg + 1) >> 1) + (seg >> 1), U = (unsigned)B.wl_units, seq1 = (unsigned)B.merged_it + 1u;
    if (atomicMax(&wl[16 + 2 * U + d * U + unit], seq1) < seq1)
        wl[16 + 6 * U + d * U + atomicAdd(&wl[8 + 4 * (B.merged_it & 1) + d], 1u)] = unit;
}

template <int LPC> struct SweepTile { static constexpr int CPB = 256 / LPC, SEGS = (CPB >= 16) ? 4 : 2, LINES = CPB / SEGS; };

// SPEC: phase B of the speculative form (see k_pm_sweep_spec below): a step that follows a rejection takes its cost from
// pr.spec; the stored match, cost and speculative cost of a chain's pixels are fetched once, before the first step, and
// handed to the steps through LDS (they are the only memory a cheap step needs).  Requires L_ <= kSpecMaxSteps.
// PRE (always with SPEC): the per-step global loads happen before the first step.  The classic form uses it for launches that cannot
// fill the chip (one 1024x436 pair): once the field has converged nearly every step is answered by the evaluation cache, and what a
// step then waits for is the round trip of its own loads -- 1.3 us per step, ten steps deep; fetched up front they cost one round trip.
constexpr int kSpecMaxSteps = 16;
// MERGED (classic form, PRE, no tile): one of the four in-place launches of the merged speculative form (k_pm_spec_all below): the
// workgroup's chains are listed units of this direction, seeds come from PmProblem::seed (the field as it stood before this sweep), a
// step's cost comes from the evaluation cache when the cache holds its candidate and is evaluated otherwise, only accepted candidates are
// written (in place: a pixel is written by its own visitors only), and an accepted candidate is passed on to the later directions of the
// iteration: their seed planes, and the unit of the pixel whose candidate it changes.
template <int R, int LPC, bool IS_ROW, bool REVERSE, bool TILE, bool SPEC = false, bool PRE = SPEC, bool MERGED = false>
__global__ __launch_bounds__(256) EPPM_SWEEP_OCC void k_pm_sweep(PmBatch B, const float* __restrict__ lut, int L_, int nseg, int nseg_pad, int TW)
{
    static_assert(PRE || !SPEC, "phase B fetches its chains' pixels up front");
    static_assert(!MERGED || (PRE && !SPEC && !TILE), "the merged form's sweeps are the classic form with up-front fetches, without a tile");
    // SPEC with a work list (pr.wl): the workgroup's CPB chains are CPB / 2 listed units (a unit = segments 2u, 2u + 1 of a line, so
    // that segments 0 and 1 -- the two visitors of pixel L -- always sit in one workgroup); a workgroup past the end of the list
    // returns at once.  Chains that are not listed keep their pixels: phase A has copied the whole field to the output plane.
    constexpr int S = R + 1, NS = S * S, CH = coop_chunk<R, LPC>(), CPB = 256 / LPC;
    static_assert((NS + CH - 1) / CH <= LPC, "a lane per chunk");
    constexpr int SEGS = SweepTile<LPC>::SEGS, LINES = SweepTile<LPC>::LINES, TROWS = S + LINES - 1;
    static_assert(!(SPEC && TILE), "phase B evaluates rarely: it gathers its source samples");
    extern __shared__ float4 s_tile[];          // TILE: TROWS sample rows + LINES centre rows of TW texels
    __shared__ EPPM_LUT_ALIGN PatchLut L;
    __shared__ int s_own[PRE ? CPB * kSpecMaxSteps : 1];       // PRE: per chain and step, the pixel's stored match (x | y << 16),
    __shared__ float s_cst[PRE ? CPB * kSpecMaxSteps : 1];     //      its stored cost,
    __shared__ float s_spc[PRE ? CPB * kSpecMaxSteps : 1];     //      phase A's cost of the rejection-path candidate (classic form: the cached cost)
    __shared__ int s_ccd[(PRE && !SPEC) ? CPB * kSpecMaxSteps : 1];   // classic form: and the cached candidate
    load_patch_lut(L, lut, R, threadIdx.x, 256);
    // 1-D grid, problem = id mod nprob: workgroups are dealt to the 8 XCDs by id mod 8, so with 8 problems (4 pairs x 2 directions)
    // each problem's planes stay in ONE XCD's L2 instead of all problems' planes competing for every L2
    const unsigned nprob = B.n * B.npairs, bq = blockIdx.x % nprob, bxx = blockIdx.x / nprob;
    const PmProblem pr = pm_problem(B, bq);
    const Planes P = to_dev(pr.P);
    // (MERGED reads and writes one plane: no __restrict__ on either pointer then)
    typename std::conditional<MERGED, const int16_t*, const int16_t* __restrict__>::type nin = pr.nnf;
    typename std::conditional<MERGED, int16_t*, int16_t* __restrict__>::type nout = MERGED ? pr.nnf : pr.nnf_alt;
    float* __restrict__ cost = pr.cost;
    // this direction's planes of the evaluation cache (eppm_internal.h: PmProblem::spec / scand)
    constexpr int DIR = IS_ROW ? (REVERSE ? 2 : 0) : (REVERSE ? 3 : 1);
    const int16_t* __restrict__ seeds = MERGED ? pr.seed + DIR * B.seed_plane : nullptr;
    float* __restrict__ cval = pr.spec ? pr.spec + DIR * B.cache_plane : nullptr;
    int32_t* __restrict__ ccand = pr.scand ? pr.scand + DIR * B.cache_plane : nullptr;
    const int len = IS_ROW ? P.w : P.h, lines = IS_ROW ? P.h : P.w;
    const int grp = threadIdx.x / LPC, r = threadIdx.x % LPC;
    int line, seg, li = 0, line0 = 0, seg0 = 0;
    if (TILE) {
        // workgroup -> (group of 2*LINES lines, parity, group of SEGS segments); chain grp -> line li of the group, segment grp % SEGS
        const int nsg = nseg_pad / SEGS, sg = bxx % nsg, lp = bxx / nsg;
        line0 = (lp >> 1) * (2 * LINES) + (lp & 1);
        seg0 = sg * SEGS;
        li = grp / SEGS;
        line = line0 + 2 * li;
        seg = seg0 + grp % SEGS;
    } else {
        const int chain = bxx * CPB + grp;
        line = chain / nseg_pad;
        seg = chain % nseg_pad;
    }
    bool listed = true;
    if ((SPEC && pr.wl) || MERGED) {
        const uint32_t nlist = MERGED ? pr.wl[8 + 4 * (B.merged_it & 1) + DIR] : pr.wl[B.sweep_seq & 1];
        if (bxx * (CPB / 2) >= nlist) return;                         // (uniform over the workgroup, before any barrier)
        const unsigned slot = bxx * CPB + grp, upl = (unsigned)(nseg + 1) >> 1;
        listed = (slot >> 1) < nlist;
        const unsigned unit = listed ? pr.wl[(MERGED ? 16 + (6 + DIR) * B.wl_units : 16 + B.wl_units) + (slot >> 1)] : 0u;
        line = (int)(unit / upl);
        seg = (int)(unit % upl) * 2 + (int)(slot & 1);
    }
    const bool active = listed && (line < lines) && (seg < nseg);
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
    const int abase = seg0 * L_ - R;                                  // TILE: along-line coordinate of tile column 0
    if (TILE) {
        if (line0 >= lines) return;                                   // the whole workgroup is past the last line
        for (int t = threadIdx.x; t < (TROWS + LINES) * TW; t += 256) {
            const int row = t / TW, c = t - row * TW;
            const int across = iclamp(row < TROWS ? line0 - R + 2 * row : line0 + 2 * (row - TROWS), 0, lines - 1);
            const int along = iclamp(abase + c, 0, len - 1);
            s_tile[t] = P.pk1[(unsigned)(IS_ROW ? across * P.pitch + along : along * P.pitch + across)];
        }
    }
    int px = 0, py = 0;
    if (active) {
        const int sidx = IS_ROW ? (line * B.npitch + start) : (start * B.npitch + line);
        px = MERGED ? seeds[sidx * 2] : nin[sidx * 2];
        py = MERGED ? seeds[sidx * 2 + 1] : nin[sidx * 2 + 1];
        // the one pixel of the line no chain visits keeps its value
        const bool copier = (REVERSE ? (seg == nseg - 1) : (seg == 0)) && !(SPEC && pr.wl) && !MERGED;     // (work list: phase A copied every pixel; merged: in place)
        if (copier && r == 0) {
            const int u = REVERSE ? len - 1 : 0;
            const int uidx = IS_ROW ? (line * B.npitch + u) : (u * B.npitch + line);
            nout[uidx * 2] = nin[uidx * 2];
            nout[uidx * 2 + 1] = nin[uidx * 2 + 1];
        }
    }
    if (PRE) {
        // the lanes of a chain fetch what its steps need: lane r the steps r, r + LPC, ...
        for (int sr = r; sr < L_; sr += LPC) {
            int own = 0, ccd = -1;
            float cst = 0.0f, spc = 0.0f;
            if (active && sr < count) {
                const int ir = i + sr * step;
                const int xr = IS_ROW ? ir : line, yr = IS_ROW ? line : ir;
                own = (int)(uint16_t)nin[(yr * B.npitch + xr) * 2] | ((int)nin[(yr * B.npitch + xr) * 2 + 1] << 16);
                cst = cost[yr * B.cpitch + xr];
                if (cval) spc = cval[yr * B.cpitch + xr];
                if (!SPEC && ccand) ccd = ccand[yr * B.cpitch + xr];
            }
            const int sl = grp * kSpecMaxSteps + sr;
            s_own[sl] = own; s_cst[sl] = cst; s_spc[sl] = spc;
            if (!SPEC) s_ccd[sl] = ccd;
        }
    }
    bool from_nin = true;                  // SPEC: the chain carries a stored match (seed, or the own match of a pixel that rejected)
    float cost_L = 0.0f;                   // SPEC: see the barrier after step 0
    __syncthreads();   // LUT ready
    const int t0 = r * CH;
    const int pitch16 = P.pitch << 4, wmax16 = (P.w - 1) << 4;
    // this lane's sample offsets (column in bytes), the same at every step: kept in registers by the classic form; phase B, which
    // evaluates rarely and runs up to 25 samples per lane, recomputes them at use (registers are what limits its waves)
    constexpr int NOFF = SPEC ? 1 : CH;
    int dj16[NOFF], di_[NOFF];
    int lo_[NOFF];                         // TILE: tile index of the sample when the chain stands at along-line coordinate 0
#pragma unroll
    for (int k = 0; k < NOFF; k++) {
        const int t = min(t0 + k, NS - 1);
        di_[k] = 2 * (t / S) - R;
        dj16[k] = (2 * (t % S) - R) * 16;
        lo_[k] = IS_ROW ? (li + t / S) * TW + (2 * (t % S) - R) - abase : (li + t % S) * TW + (2 * (t / S) - R) - abase;
    }
    const int lc = (TROWS + li) * TW - abase;                          // TILE: the same for the patch centre
    for (int s = 0; s < L_; s++) {
        if (active && s < count) {
            const int x = IS_ROW ? i : line, y = IS_ROW ? line : i;
            const int nidx = y * B.npitch + x, cidx = y * B.cpitch + x;
            const bool second_visit = (!REVERSE) && (seg == 0) && (s == L_ - 1) && (nseg > 1);   // pixel L, after segment 1
            float cur_best;
            int ox, oy;
            if (PRE) {
                const int sl = grp * kSpecMaxSteps + s, e = s_own[sl];
                ox = (int)(int16_t)(e & 0xffff); oy = e >> 16;
                cur_best = second_visit ? cost_L : s_cst[sl];         // segment 1 may have lowered pixel L's cost at its first step
            } else {
                cur_best = cost[cidx];
                ox = nin[nidx * 2]; oy = nin[nidx * 2 + 1];           // the pixel's own match, needed on rejection: fetched with the rest
            }
            int hit_cand = -1;                                        // classic form: this pixel's cached evaluation, fetched with the rest
            float hit_val = 0.0f;
            if (!PRE && ccand) { hit_cand = ccand[cidx]; hit_val = cval[cidx]; }
            // (fetched up front: at pixel L's second visit the entry may predate segment 1's evaluation; a stale entry is still a
            // valid (candidate, cost) pair -- at worst this step evaluates what the entry written meanwhile would have answered)
            if (PRE && !SPEC) { hit_cand = s_ccd[grp * kSpecMaxSteps + s]; hit_val = s_spc[grp * kSpecMaxSteps + s]; }
            if (IS_ROW) px = REVERSE ? max(px - 1, 0) : min(px + 1, P.w - 1);
            else        py = REVERSE ? max(py - 1, 0) : min(py + 1, P.h - 1);
            // A candidate equal to the pixel's current match would reproduce the stored cost bit for bit
            // (every cost in the plane was produced by this same sum), so "cv < cur_best" is false: the
            // reference evaluates and rejects it, here the evaluation is skipped.  Converged regions --
            // neighbours sharing one offset -- make this the common case after the first iterations.
            float cv = cur_best;
            const bool differs = !(px == ox && py == oy);
            const int cpack = (px & 0xffff) | (py << 16);
            if (SPEC && differs && from_nin) cv = s_spc[grp * kSpecMaxSteps + s];           // phase A evaluated exactly this candidate (now or earlier)
            else if (!SPEC && differs && hit_cand == cpack) cv = hit_val;                   // evaluated in an earlier sweep of this direction
            else if (differs) {
            const rgbf c1 = texel_rgb(TILE ? s_tile[lc + i] : tex_px(P.pk1, P.pitch, P.w, P.h, x, y));
            const rgbf c2 = texel_rgb(tex_px(P.pk2, P.pitch, P.w, P.h, px, py));
            float tc[CH], tw[CH];
#ifndef EPPM_SWEEP_GB_SPEC
#define EPPM_SWEEP_GB_SPEC 3
#endif
            // gathers in flight per lane; phase B evaluates rarely and gains more from waves (one round of workgroups) than from depth
            constexpr int GBW = SPEC ? EPPM_SWEEP_GB_SPEC : EPPM_SWEEP_GB;
            constexpr int GB = (CH < GBW) ? CH : GBW;
#pragma unroll
            for (int q0 = 0; q0 < CH; q0 += GB) {
                float4 q1[GB], q2[GB];
#pragma unroll
                for (int k = 0; k < GB; k++) {
                    const int qk = (q0 + k < CH) ? q0 + k : CH - 1;       // (the last batch may be partial: its spare slots repeat the last sample)
                    int dj, di;
                    if (SPEC) {
                        int t0v = t0;
                        asm volatile("" : "+v"(t0v));          // keeps the 25 offset pairs from being hoisted out of the step loop into registers
                        const int t = min(t0v + qk, NS - 1);
                        di = 2 * (t / S) - R; dj = (2 * (t % S) - R) * 16;
                    } else { di = di_[qk]; dj = dj16[qk]; }
                    q1[k] = TILE ? s_tile[lo_[SPEC ? 0 : qk] + i]
                                 : texel_at(P.pk1, texel_off16(pitch16, wmax16, P.h - 1, (x << 4) + dj, y + di));
                    q2[k] = texel_at(P.pk2, texel_off16(pitch16, wmax16, P.h - 1, (px << 4) + dj, py + di));
                }
#pragma unroll
                for (int k = 0; k < GB; k++) {
                    const int q = q0 + k;
                    if (q < CH) {
                        const int t = t0 + q;
                        tc[q] = 0.0f; tw[q] = 0.0f;
                        if (t < NS) patch_terms(q1[k], q2[k], c1, c2, L.gsp[t], L.tab(), tc[q], tw[q]);
                    }
                }
            }
            // the sums in their defined order (coop_chain_sum): complete in the lane of the last chunk
            float ac = 0.0f, aw = 0.0f;
            constexpr int NL = (NS + CH - 1) / CH;       // lanes that own samples
            coop_chain_sum<LPC, CH, NS, NL>(tc, tw, ac, aw);
            const int src = ((threadIdx.x & 63) / LPC) * LPC + (NL - 1);   // lane holding the complete sums (wave-relative)
            const float cs = __shfl(ac, src, 64), ws = __shfl(aw, src, 64);
            cv = cs / ws;
            if (!SPEC && ccand && r == 0) { ccand[cidx] = cpack; cval[cidx] = cv; }
            }
            if (cv < cur_best) {
                if (r == 0) {
                    nout[nidx * 2] = (int16_t)px;
                    nout[nidx * 2 + 1] = (int16_t)py;
                    cost[cidx] = cv;
                    if (MERGED) {
                        // the later sweeps of this iteration start from the field this sweep leaves: their seeds, and the chain whose
                        // candidate at the next pixel in THEIR direction has just changed
#pragma unroll
                        for (int d2 = DIR + 1; d2 < 4; d2++) {
                            int16_t* sd = pr.seed + d2 * B.seed_plane;
                            sd[nidx * 2] = (int16_t)px; sd[nidx * 2 + 1] = (int16_t)py;
                            const int qx = (d2 == 2) ? x - 1 : x, qy = (d2 == 1) ? y + 1 : (d2 == 3) ? y - 1 : y;
                            if (qx >= 0 && qy >= 0 && qx < P.w && qy < P.h) merged_list_unit(pr.wl, B, d2, qx, qy);
                        }
                    }
                }
                from_nin = false;
            } else {
                if (r == 0 && !second_visit && !MERGED) {
                    nout[nidx * 2] = (int16_t)ox;
                    nout[nidx * 2 + 1] = (int16_t)oy;
                }
                px = ox; py = oy;
                from_nin = true;
            }
            i += step;
        }
        if (!REVERSE && s == 0) {
            __syncthreads();   // (b)
            // SPEC: pixel L's cost as segment 1's first step left it, for segment 0's last step (fetched here, by value: a select
            // between this global address and the LDS copy at the point of use would turn both loads into flat_load)
            if (PRE && active && seg == 0 && nseg > 1 && L_ < len) cost_L = cost[IS_ROW ? line * B.cpitch + L_ : L_ * B.cpitch + line];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Speculative form of a sweep, for the iterations in which few candidates are accepted (from the third iteration on fewer
// than one step in ten, tools/sweep_stats.py): phase A + phase B, same results bit for bit.
//
// The candidate a chain tries at pixel i is shift(p) where p is what the chain carries out of pixel i-1: the match it accepted
// there, or -- when pixel i-1 REJECTED its candidate (and at a segment's first step, whose seed is pixel i-1) -- pixel i-1's own
// match nin[i-1].  So for every visited pixel the candidate on the rejection path, shift(nin[i-1]), is known before the sweep
// starts and its cost does not depend on the chain's history (the patch cost is a pure function of (pixel, candidate)):
//  phase A (k_pm_sweep_spec): evaluates E(i, shift(nin[i-1])) for every visited pixel in parallel -- no dependent steps, one
//      evaluation per lane with the source samples from an LDS tile; pixels whose candidate equals their own match are skipped
//      (the skip rule), the rest are compacted inside the workgroup so that whole waves work or exit;
//  phase B (k_pm_sweep<.., SPEC>): the chains walk their pixels in the reference's order as before, but a step that follows a
//      rejection takes its cost from phase A's plane; only a step that follows an ACCEPTED candidate evaluates (cooperatively,
//      as in the classic form).  In the converged iterations phase B is ten compare-and-select steps.
// ---------------------------------------------------------------------------------------------------
//
// Work list (pr.wl): phase A also knows which chains can change anything.  A chain leaves the rejection path only where a
// rejection-path candidate is ACCEPTED, i.e. where E(i, shift(nin[i-1])) < cost[i]; a chain without such a pixel rejects at every
// step (by induction it never carries anything but stored matches) and writes back what it read.  So phase A copies the field to
// the output plane, tests every visited pixel (evaluated now, taken from the cache, or skipped by the skip rule: never accepted)
// and appends the UNIT of a pixel that would accept -- segments 2u and 2u + 1 of its line, so that segments 0 and 1, the two
// visitors of pixel L, are always listed together -- to the list, once (a stamp per unit holds the number of the last sweep that
// listed it).  Phase B walks the listed chains only: from the fifth iteration on that is one chain in ten.
template <int RT, bool IS_ROW, bool REVERSE, int PK = 0>
__global__ __launch_bounds__(256) void k_pm_sweep_spec(PmBatch B, const float* __restrict__ lut, int R, int gx, int L_, int nseg)
{
    using LUT = typename SearchLut<RT>::type;
    constexpr int TW = (RT == 0) ? 1 : kBlock + 2 * RT;
    constexpr int DIR = IS_ROW ? (REVERSE ? 2 : 0) : (REVERSE ? 3 : 1);
    __shared__ float4 s_src[TW * TW];
    __shared__ EPPM_LUT_ALIGN LUT L;
    __shared__ uint32_t s_list[256];       // compacted work: pixel index inside the block
    __shared__ int s_cand[256];            // its candidate, x | y << 16
    __shared__ int s_wcount[4];
    const unsigned nprob = B.n * B.npairs, bq = blockIdx.x % nprob, brest = blockIdx.x / nprob;
    const int bxx = brest % gx, byy = brest / gx;
    const PmProblem pr = pm_problem(B, bq);
    const Planes P = to_dev(pr.P);
    float* __restrict__ cval = pr.spec + DIR * B.cache_plane;
    int32_t* __restrict__ ccand = pr.scand ? pr.scand + DIR * B.cache_plane : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x = bxx * kBlock + (tid & 15), y = byy * kBlock + (tid >> 4);
    // the pixel the chain comes from: one step against the sweep direction; the first pixel of a line is never visited
    const int qx = IS_ROW ? (REVERSE ? x + 1 : x - 1) : x, qy = IS_ROW ? y : (REVERSE ? y + 1 : y - 1);
    bool need = false;
    int cpack = 0;
    uint32_t* __restrict__ wl = pr.wl;
    const unsigned upl = (unsigned)(nseg + 1) >> 1, seq1 = (unsigned)B.sweep_seq + 1u;
    // lists the unit of pixel (px, py) for this sweep, once
    auto list_unit = [&](int px, int py) {
        const int along = IS_ROW ? px : py, ln = IS_ROW ? py : px;
        const unsigned seg = (!REVERSE && along < L_) ? 0u : (unsigned)(along / L_);
        const unsigned unit = (unsigned)ln * upl + (seg >> 1);
        if (atomicMax(&wl[16 + unit], seq1) < seq1) wl[16 + B.wl_units + atomicAdd(&wl[B.sweep_seq & 1], 1u)] = unit;
    };
    if (wl && blockIdx.x < nprob && tid == 0) wl[(B.sweep_seq + 1) & 1] = 0u;        // the next sweep's list length (the previous sweep is done with it)
    if (x < P.w && y < P.h) {
        const int ni = (y * B.npitch + x) * 2;
        const int ox = pr.nnf[ni], oy = pr.nnf[ni + 1];
        if (wl) { pr.nnf_alt[ni] = (int16_t)ox; pr.nnf_alt[ni + 1] = (int16_t)oy; }   // a pixel no listed chain visits keeps its match
        if (qx >= 0 && qy >= 0 && qx < P.w && qy < P.h) {
            const int qi = (qy * B.npitch + qx) * 2;
            int cx = pr.nnf[qi], cy = pr.nnf[qi + 1];
            if (IS_ROW) cx = REVERSE ? max(cx - 1, 0) : min(cx + 1, P.w - 1);
            else        cy = REVERSE ? max(cy - 1, 0) : min(cy + 1, P.h - 1);
            cpack = (cx & 0xffff) | (cy << 16);
            need = !(cx == ox && cy == oy);                              // equal to the pixel's own match: rejected unevaluated
            if (need && ccand && ccand[y * B.cpitch + x] == cpack) {     // evaluated in an earlier sweep of this direction: the cost stands
                need = false;
                if (wl && cval[y * B.cpitch + x] < pr.cost[y * B.cpitch + x]) list_unit(x, y);
            }
        }
    }
    // compaction: wave-level ballot + prefix, then the four wave counts: whole waves work or exit
    const unsigned long long bal = __ballot(need);
    if (lane == 0) s_wcount[wv] = __popcll(bal);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const int c = s_wcount[k]; if (k < wv) base += c; total += c; }
    if (total == 0) return;                   // every candidate of the block is known (converged field): no tile, no table
    load_patch_lut(L, lut, R, tid, 256);
    if (RT != 0) {
        const int x0 = bxx * kBlock - RT, y0 = byy * kBlock - RT;
        for (int t = tid; t < TW * TW; t += 256) {
            const int sy = iclamp(y0 + t / TW, 0, P.h - 1), sx = iclamp(x0 + t % TW, 0, P.w - 1);
            s_src[t] = P.pk1[(unsigned)(sy * P.pitch + sx)];
        }
    }
    if (need) {
        const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
        s_list[slot] = (uint32_t)tid;
        s_cand[slot] = cpack;
    }
    __syncthreads();                          // LUT, tile, list
#ifndef EPPM_SPEC_COOP17_MAX
#define EPPM_SPEC_COOP17_MAX 32
#endif
    // Radius 17: one lane's evaluation is a serial chain of 324 samples (75 us), the 45 KB tile allows three workgroups per CU, and in
    // the converged iterations a block has a handful of evaluations left: then a whole wave takes one evaluation (6 samples per lane,
    // ordered 54-hop sum), four at a time.  (At radius 9 the same with 16 lanes per evaluation was measured and lost -- its registers cost the
    // early iterations, every block of the launch, more than the late ones gain -- and as a separate instantiation for the late
    // iterations only it changed nothing: profiles/r04x_c.)
    if (RT == 17 && total <= EPPM_SPEC_COOP17_MAX) {
        for (int slot = tid >> 6; slot < total; slot += 4) {
            const int pix = (int)s_list[slot], e = s_cand[slot];
            const int px = bxx * kBlock + (pix & 15), py = byy * kBlock + (pix >> 4);
            const float cv = coop_patch_dist<(RT == 17 ? 17 : 1), 64>(P, L, s_src, TW, pix & 15, pix >> 4, (int)(int16_t)(e & 0xffff), e >> 16, tid & 63);
            if ((tid & 63) == 0) {
                cval[py * B.cpitch + px] = cv;
                if (ccand) ccand[py * B.cpitch + px] = e;
                if (wl && cv < pr.cost[py * B.cpitch + px]) list_unit(px, py);
            }
        }
        return;
    }
    if (tid >= total) return;
    const int pix = (int)s_list[tid], e = s_cand[tid];
    const int tx = pix & 15, ty = pix >> 4;
    const int px = bxx * kBlock + tx, py = byy * kBlock + ty;
    // (radius 17: gathering the 4-byte target plane here as the search does changes nothing: 57.5 vs 57.6 ms PatchMatch at 3840x2160)
    const float cv = search_patch_dist<RT, PK>(P, L, R, s_src, TW, tx, ty, px, py, (int)(int16_t)(e & 0xffff), e >> 16, pr.P);
    cval[py * B.cpitch + px] = cv;
    if (ccand) ccand[py * B.cpitch + px] = e;
    if (wl && cv < pr.cost[py * B.cpitch + px]) list_unit(px, py);
}

// ---------------------------------------------------------------------------------------------------
// Merged form: ONE phase A for the four sweeps of an iteration (late iterations: nearly every candidate is answered by the cache, and
// what a phase-A launch then costs is the launch).  From the field F0 the iteration starts with, for every pixel and every direction d the
// rejection-path candidate shift_d(F0[i - 1_d]) is tested exactly as k_pm_sweep_spec tests it (skip rule, cache, evaluation) and the unit
// is listed when the candidate would be accepted against the pixel's cost.  The four sweeps then run as in-place launches over their lists
// (k_pm_sweep<.., MERGED>).  Why the lists stay complete although sweeps 1..3 see a field the earlier sweeps have changed:
//  * a chain leaves the rejection path first at a pixel i whose candidate shift_d(F[i - 1_d]) is accepted.  If pixel i - 1_d still holds its
//    F0 match, that candidate is the one tested here, against a cost that can only have fallen since: listed.  If an earlier sweep of the
//    iteration changed pixel i - 1_d, that sweep listed the unit of pixel i for direction d when it accepted (k_pm_sweep, MERGED);
//  * the cache answers by candidate, so an entry written here for a candidate the field no longer proposes is simply not used;
//  * seeds: PmProblem::seed[d] = F0 here, kept current by the earlier sweeps' accepted candidates, never by sweep d itself.
// ---------------------------------------------------------------------------------------------------
template <int RT, int PK = 0>
__global__ __launch_bounds__(256) void k_pm_spec_all(PmBatch B, const float* __restrict__ lut, int R, int gx)
{
    using LUT = typename SearchLut<RT>::type;
    constexpr int TW = kBlock + 2 * RT;
    __shared__ float4 s_src[TW * TW];
    __shared__ EPPM_LUT_ALIGN LUT L;
    __shared__ uint16_t s_list[1024];      // compacted work: pixel index inside the block | direction << 8
    __shared__ int s_cand[1024];           // its candidate, x | y << 16
    __shared__ int s_wcount[16];           // [direction][wave]
    const unsigned nprob = B.n * B.npairs, bq = blockIdx.x % nprob, brest = blockIdx.x / nprob;
    const int bxx = brest % gx, byy = brest / gx;
    const PmProblem pr = pm_problem(B, bq);
    const Planes P = to_dev(pr.P);
    uint32_t* __restrict__ wl = pr.wl;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x = bxx * kBlock + (tid & 15), y = byy * kBlock + (tid >> 4);
    if (blockIdx.x < nprob && tid < 4) wl[8 + 4 * ((B.merged_it + 1) & 1) + tid] = 0u;      // the next iteration's list lengths
    unsigned needmask = 0;
    int cpack[4] = {0, 0, 0, 0};
    if (x < P.w && y < P.h) {
        const int ni = (y * B.npitch + x) * 2, ci = y * B.cpitch + x;
        const uint32_t* __restrict__ nnf32 = reinterpret_cast<const uint32_t*>(pr.nnf);       // a match as one word: x | y << 16
        const uint32_t own = nnf32[ni >> 1];
        const int ox = (int)(int16_t)(own & 0xffffu), oy = (int)(int16_t)(own >> 16);
        const float c0 = pr.cost[ci];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            reinterpret_cast<uint32_t*>(pr.seed + d * B.seed_plane)[ni >> 1] = own;
            const int qx = (d == 0) ? x - 1 : (d == 2) ? x + 1 : x, qy = (d == 1) ? y - 1 : (d == 3) ? y + 1 : y;
            if (qx >= 0 && qy >= 0 && qx < P.w && qy < P.h) {
                const uint32_t qm = nnf32[qy * B.npitch + qx];
                int cx = (int)(int16_t)(qm & 0xffffu), cy = (int)(int16_t)(qm >> 16);
                if (d == 0) cx = min(cx + 1, P.w - 1);
                else if (d == 1) cy = min(cy + 1, P.h - 1);
                else if (d == 2) cx = max(cx - 1, 0);
                else cy = max(cy - 1, 0);
                const int cp = (cx & 0xffff) | (cy << 16);
                bool need = !(cx == ox && cy == oy);                             // equal to the pixel's own match: rejected unevaluated
                if (need && pr.scand[d * B.cache_plane + ci] == cp) {            // evaluated before: the cost stands
                    need = false;
                    if (pr.spec[d * B.cache_plane + ci] < c0) merged_list_unit(wl, B, d, x, y);
                }
                if (need) { needmask |= 1u << d; cpack[d] = cp; }
            }
        }
    }
    unsigned long long bal[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        bal[d] = __ballot((needmask >> d) & 1u);
        if (lane == 0) s_wcount[d * 4 + wv] = __popcll(bal[d]);
    }
    __syncthreads();
    int total = 0, base[4];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if ((k & 3) == 0) base[k >> 2] = total;              // start of direction k/4 ...
        const int c = s_wcount[k];
        if ((k & 3) < wv) base[k >> 2] += c;                 // ... plus the lower waves of that direction
        total += c;
    }
    if (total == 0) return;
    load_patch_lut(L, lut, R, tid, 256);
    {
        const int x0 = bxx * kBlock - RT, y0 = byy * kBlock - RT;
        for (int t = tid; t < TW * TW; t += 256) {
            const int sy = iclamp(y0 + t / TW, 0, P.h - 1), sx = iclamp(x0 + t % TW, 0, P.w - 1);
            s_src[t] = P.pk1[(unsigned)(sy * P.pitch + sx)];
        }
    }
#pragma unroll
    for (int d = 0; d < 4; d++)
        if ((needmask >> d) & 1u) {
            const int slot = base[d] + __popcll(bal[d] & ((1ull << lane) - 1ull));
            s_list[slot] = (uint16_t)(tid | (d << 8));
            s_cand[slot] = cpack[d];
        }
    __syncthreads();                          // table, tile, list
    if (RT == 17 && total <= EPPM_SPEC_COOP17_MAX) {          // few evaluations: a wave each (see k_pm_sweep_spec)
        for (int slot = tid >> 6; slot < total; slot += 4) {
            const int pix = (int)s_list[slot] & 255, d = (int)s_list[slot] >> 8, e = s_cand[slot];
            const int px = bxx * kBlock + (pix & 15), py = byy * kBlock + (pix >> 4), ci = py * B.cpitch + px;
            const float cv = coop_patch_dist<(RT == 17 ? 17 : 1), 64>(P, L, s_src, TW, pix & 15, pix >> 4, (int)(int16_t)(e & 0xffff), e >> 16, tid & 63);
            if ((tid & 63) == 0) {
                pr.spec[d * B.cache_plane + ci] = cv;
                pr.scand[d * B.cache_plane + ci] = e;
                if (cv < pr.cost[ci]) merged_list_unit(wl, B, d, px, py);
            }
        }
        return;
    }
#ifndef EPPM_MERGED_COOP9_MAX
#define EPPM_MERGED_COOP9_MAX 16
#endif
    // Radius 9, a block with at most 16 evaluations (the usual case in the iterations this kernel runs in): 16 lanes each, all at once --
    // the launch then waits for a 7-sample chain and 16 hops instead of one lane's 100 samples.  (In k_pm_sweep_spec the same lost: its
    // registers cost the early iterations, which every block of that kernel also serves; this kernel only runs late.)  PatchMatch
    // 0.747 -> 0.741 ms per pair in 8-pair launches, default bench +0.4 % (two interleaved rounds each), 1920x1080 unchanged.
    if (RT == 9 && total <= EPPM_MERGED_COOP9_MAX) {
        constexpr int CL = EPPM_LPC9_SPEC;            // lanes per evaluation: 16 (tolerance library: 32, a lane = a chunk)
        for (int slot = tid / CL; slot < total; slot += 256 / CL) {
            const int pix = (int)s_list[slot] & 255, d = (int)s_list[slot] >> 8, e = s_cand[slot];
            const int px = bxx * kBlock + (pix & 15), py = byy * kBlock + (pix >> 4), ci = py * B.cpitch + px;
            const float cv = coop_patch_dist<(RT == 9 ? 9 : 1), CL>(P, L, s_src, TW, pix & 15, pix >> 4, (int)(int16_t)(e & 0xffff), e >> 16, tid % CL);
            if ((tid % CL) == 0) {
                pr.spec[d * B.cache_plane + ci] = cv;
                pr.scand[d * B.cache_plane + ci] = e;
                if (cv < pr.cost[ci]) merged_list_unit(wl, B, d, px, py);
            }
        }
        return;
    }
    for (int slot = tid; slot < total; slot += 256) {
        const int pix = (int)s_list[slot] & 255, d = (int)s_list[slot] >> 8, e = s_cand[slot];
        const int tx = pix & 15, ty = pix >> 4;
        const int px = bxx * kBlock + tx, py = byy * kBlock + ty, ci = py * B.cpitch + px;
        const float cv = search_patch_dist<RT, PK>(P, L, R, s_src, TW, tx, ty, px, py, (int)(int16_t)(e & 0xffff), e >> 16, pr.P);
        pr.spec[d * B.cache_plane + ci] = cv;
        pr.scand[d * B.cache_plane + ci] = e;
        if (cv < pr.cost[ci]) merged_list_unit(wl, B, d, px, py);
    }
}

// Fallback for patch radii without a cooperative instantiation: the reference's one-thread-per-chain form,
// in place, all segments of a line in one workgroup (seed reads / pixel-L order by workgroup barriers).
template <bool IS_ROW, bool REVERSE>
__global__ __launch_bounds__(1024) void k_pm_seg_propagate(PmBatch B, const float* __restrict__ lut, int R, int L_, int nseg,
                                                           int lines_per_block)
{
    __shared__ EPPM_LUT_ALIGN PatchLut L;
    load_patch_lut(L, lut, R, threadIdx.x, blockDim.x);
    const PmProblem pr = pm_problem(B, blockIdx.y);
    const Planes P = to_dev(pr.P);
    int16_t* __restrict__ nnf = pr.nnf;
    float* __restrict__ cost = pr.cost;
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
        const int sidx = IS_ROW ? (line * B.npitch + start) : (start * B.npitch + line);
        px = nnf[sidx * 2];
        py = nnf[sidx * 2 + 1];
    }
    __syncthreads();   // LUT ready; (a) all seeds read
    for (int s = 0; s < L_; s++) {
        if (active && s < count) {
            const int x = IS_ROW ? i : line, y = IS_ROW ? line : i;
            const int nidx = y * B.npitch + x, cidx = y * B.cpitch + x;
            const float cur_best = cost[cidx];
            if (IS_ROW) px = REVERSE ? max(px - 1, 0) : min(px + 1, P.w - 1);
            else        py = REVERSE ? max(py - 1, 0) : min(py + 1, P.h - 1);
            const bool same = (px == nnf[nidx * 2]) && (py == nnf[nidx * 2 + 1]);   // would reproduce cur_best: rejected
            const float cv = same ? cur_best : patch_dist(P, L, R, x, y, px, py);
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

// phase A of the speculative form for one direction
template <int RT>
static void launch_sweep_spec(const PmBatch& b, const float* lut, int R, int dir, int seg_len, int nseg, hipStream_t s)
{
    const int w = b.p[0].P.w, h = b.p[0].P.h, gx = (w + kBlock - 1) / kBlock, gy = (h + kBlock - 1) / kBlock;
    dim3 grid(gx * gy * (b.n * b.npairs)), block(256);
    if (pm_has_parity(b)) {
        switch (dir) {
            case 0: hipLaunchKernelGGL((k_pm_sweep_spec<RT, true, false, 2>), grid, block, 0, s, b, lut, R, gx, seg_len, nseg); break;
            case 1: hipLaunchKernelGGL((k_pm_sweep_spec<RT, false, false, 2>), grid, block, 0, s, b, lut, R, gx, seg_len, nseg); break;
            case 2: hipLaunchKernelGGL((k_pm_sweep_spec<RT, true, true, 2>), grid, block, 0, s, b, lut, R, gx, seg_len, nseg); break;
            default: hipLaunchKernelGGL((k_pm_sweep_spec<RT, false, true, 2>), grid, block, 0, s, b, lut, R, gx, seg_len, nseg); break;
        }
        return;
    }
    switch (dir) {
        case 0: hipLaunchKernelGGL((k_pm_sweep_spec<RT, true, false>), grid, block, 0, s, b, lut, R, gx, seg_len, nseg); break;
        case 1: hipLaunchKernelGGL((k_pm_sweep_spec<RT, false, false>), grid, block, 0, s, b, lut, R, gx, seg_len, nseg); break;
        case 2: hipLaunchKernelGGL((k_pm_sweep_spec<RT, true, true>), grid, block, 0, s, b, lut, R, gx, seg_len, nseg); break;
        default: hipLaunchKernelGGL((k_pm_sweep_spec<RT, false, true>), grid, block, 0, s, b, lut, R, gx, seg_len, nseg); break;
    }
}
int pm_worklist_units(int w, int h, int seg_len)
{
    const int ur = h * (((w + seg_len - 1) / seg_len + 1) / 2), uc = w * (((h + seg_len - 1) / seg_len + 1) / 2);
    return ur > uc ? ur : uc;
}
// phase B
template <int R, int LPC>
static void launch_sweep_b(const PmBatch& b, const float* lut, int seg_len, int dir, int nseg, int lines, hipStream_t s)
{
    constexpr int CPB = 256 / LPC;
    const int nseg_pad = (nseg + 1) & ~1;
    const int wgs = (lines * nseg_pad + CPB - 1) / CPB;
    dim3 grid(wgs * (b.n * b.npairs)), block(256);
    switch (dir) {
        case 0: hipLaunchKernelGGL((k_pm_sweep<R, LPC, true, false, false, true>), grid, block, 0, s, b, lut, seg_len, nseg, nseg_pad, 0); break;
        case 1: hipLaunchKernelGGL((k_pm_sweep<R, LPC, false, false, false, true>), grid, block, 0, s, b, lut, seg_len, nseg, nseg_pad, 0); break;
        case 2: hipLaunchKernelGGL((k_pm_sweep<R, LPC, true, true, false, true>), grid, block, 0, s, b, lut, seg_len, nseg, nseg_pad, 0); break;
        default: hipLaunchKernelGGL((k_pm_sweep<R, LPC, false, true, false, true>), grid, block, 0, s, b, lut, seg_len, nseg, nseg_pad, 0); break;
    }
}

template <int R, int LPC, bool TILE, bool PRE = false>
static void launch_sweep_t(const PmBatch& b, const float* lut, int seg_len, int dir, int nseg, int lines, hipStream_t s)
{
    constexpr int CPB = 256 / LPC, SEGS = SweepTile<LPC>::SEGS, LINES = SweepTile<LPC>::LINES;
    const int nseg_pad = TILE ? (nseg + SEGS - 1) / SEGS * SEGS : (nseg + 1) & ~1;
    const int wgs = TILE ? ((lines + 2 * LINES - 1) / (2 * LINES)) * 2 * (nseg_pad / SEGS) : (lines * nseg_pad + CPB - 1) / CPB;
    const int TW = (SEGS * seg_len + 2 * R) | 1;
    const size_t lds = TILE ? (size_t)(R + 1 + 2 * LINES - 1) * TW * 16 : 0;
    dim3 grid(wgs * (b.n * b.npairs)), block(256);
    switch (dir) {
        case 0: hipLaunchKernelGGL((k_pm_sweep<R, LPC, true, false, TILE, false, PRE>), grid, block, lds, s, b, lut, seg_len, nseg, nseg_pad, TW); break;
        case 1: hipLaunchKernelGGL((k_pm_sweep<R, LPC, false, false, TILE, false, PRE>), grid, block, lds, s, b, lut, seg_len, nseg, nseg_pad, TW); break;
        case 2: hipLaunchKernelGGL((k_pm_sweep<R, LPC, true, true, TILE, false, PRE>), grid, block, lds, s, b, lut, seg_len, nseg, nseg_pad, TW); break;
        default: hipLaunchKernelGGL((k_pm_sweep<R, LPC, false, true, TILE, false, PRE>), grid, block, lds, s, b, lut, seg_len, nseg, nseg_pad, TW); break;
    }
}
#ifndef EPPM_SWEEP_TILE
#define EPPM_SWEEP_TILE 1
#endif
// source tile in LDS while it stays small (16 KiB at R = 9 and the default segment length of 10); very long segments gather
template <int R, int LPC, bool PRE = false>
static void launch_sweep_r(const PmBatch& b, const float* lut, int seg_len, int dir, int nseg, int lines, hipStream_t s)
{
    const size_t lds = (size_t)(R + 2 * SweepTile<LPC>::LINES) * ((SweepTile<LPC>::SEGS * seg_len + 2 * R) | 1) * 16;
    if (PRE && seg_len <= kSpecMaxSteps && b.p[0].scand) {
        if (EPPM_SWEEP_TILE && lds <= 32 * 1024) launch_sweep_t<R, LPC, true, PRE>(b, lut, seg_len, dir, nseg, lines, s);
        else launch_sweep_t<R, LPC, false, PRE>(b, lut, seg_len, dir, nseg, lines, s);
        return;
    }
    if (EPPM_SWEEP_TILE && lds <= 32 * 1024) launch_sweep_t<R, LPC, true>(b, lut, seg_len, dir, nseg, lines, s);
    else launch_sweep_t<R, LPC, false>(b, lut, seg_len, dir, nseg, lines, s);
}

bool launch_pm_sweep(PmBatch& b, const float* lut, int R, int seg_len, int dir, hipStream_t s, bool speculative)
{
    struct Count { PmBatch& b; ~Count() { b.sweep_seq++; } } count{b};      // every sweep of a run has its number (the work list's stamps)
    const PlanesH& P = b.p[0].P;
    const bool is_row = (dir == 0 || dir == 2);
    const int len = is_row ? P.w : P.h, lines = is_row ? P.h : P.w;
    const int nseg = (len + seg_len - 1) / seg_len;
    if (speculative && b.p[0].spec && (R == 9 || R == 17) && seg_len <= kSpecMaxSteps) {
        if (R == 9) { launch_sweep_spec<9>(b, lut, R, dir, seg_len, nseg, s); launch_sweep_b<9, EPPM_LPC9_SPEC>(b, lut, seg_len, dir, nseg, lines, s); }
        else { launch_sweep_spec<17>(b, lut, R, dir, seg_len, nseg, s); launch_sweep_b<17, EPPM_LPC17_SPEC>(b, lut, seg_len, dir, nseg, lines, s); }
        return true;
    }
    if (R == 9) {
        // 16 lanes per chain are the most instruction-efficient; when that leaves fewer than two waves per SIMD (the
        // quarter-resolution level of a 1024x436 pair: 1.4) the chip is latency bound and 32 lanes per chain shorten
        // the dependent step (PatchMatch 1.78 -> 1.61 ms, no change in throughput with pairs in flight)
        const int chains = lines * ((nseg + 1) & ~1) * b.n * b.npairs;
#ifndef EPPM_LPC_SWITCH_WAVES
#define EPPM_LPC_SWITCH_WAVES (2 * 1024)
#endif
#ifndef EPPM_SWEEP_PRE
#define EPPM_SWEEP_PRE 1      // the classic form fetches its chains' pixels up front: 0 never, 1 launches that cannot fill the chip, 2 always
#endif
#ifdef EPPM_TOL
        constexpr int LPC_SMALL = EPPM_LPC9;          // one dealing of the samples (coop_chunk): small launches only fetch up front
#else
        constexpr int LPC_SMALL = 2 * EPPM_LPC9;
#endif
        if (chains * 16 / 64 < EPPM_LPC_SWITCH_WAVES) launch_sweep_r<9, LPC_SMALL, (EPPM_SWEEP_PRE >= 1)>(b, lut, seg_len, dir, nseg, lines, s);
        else launch_sweep_r<9, EPPM_LPC9, (EPPM_SWEEP_PRE >= 2)>(b, lut, seg_len, dir, nseg, lines, s);
        return true;
    }
    if (R == 17) { launch_sweep_r<17, EPPM_LPC17>(b, lut, seg_len, dir, nseg, lines, s); return true; }
    if (nseg > 1024) return false;   // eppm_create / the launchers validate sizes
    int lpb = 256 / nseg;
    if (lpb < 1) lpb = 1;
    const int threads = ((nseg * lpb + 63) / 64) * 64;
    dim3 grid((lines + lpb - 1) / lpb, b.n * b.npairs), block(threads);
    switch (dir) {
        case 0: hipLaunchKernelGGL((k_pm_seg_propagate<true, false>), grid, block, 0, s, b, lut, R, seg_len, nseg, lpb); break;
        case 1: hipLaunchKernelGGL((k_pm_seg_propagate<false, false>), grid, block, 0, s, b, lut, R, seg_len, nseg, lpb); break;
        case 2: hipLaunchKernelGGL((k_pm_seg_propagate<true, true>), grid, block, 0, s, b, lut, R, seg_len, nseg, lpb); break;
        default: hipLaunchKernelGGL((k_pm_seg_propagate<false, true>), grid, block, 0, s, b, lut, R, seg_len, nseg, lpb); break;
    }
    return false;
}

template <int R, int LPC>
static void launch_sweep_merged(const PmBatch& b, const float* lut, int seg_len, int dir, hipStream_t s)
{
    const PlanesH& P = b.p[0].P;
    const bool is_row = (dir == 0 || dir == 2);
    const int len = is_row ? P.w : P.h, lines = is_row ? P.h : P.w;
    const int nseg = (len + seg_len - 1) / seg_len, nseg_pad = (nseg + 1) & ~1;
    constexpr int CPB = 256 / LPC;
    const int wgs = (lines * nseg_pad + CPB - 1) / CPB;
    dim3 grid(wgs * (b.n * b.npairs)), block(256);
    switch (dir) {
        case 0: hipLaunchKernelGGL((k_pm_sweep<R, LPC, true, false, false, false, true, true>), grid, block, 0, s, b, lut, seg_len, nseg, nseg_pad, 0); break;
        case 1: hipLaunchKernelGGL((k_pm_sweep<R, LPC, false, false, false, false, true, true>), grid, block, 0, s, b, lut, seg_len, nseg, nseg_pad, 0); break;
        case 2: hipLaunchKernelGGL((k_pm_sweep<R, LPC, true, true, false, false, true, true>), grid, block, 0, s, b, lut, seg_len, nseg, nseg_pad, 0); break;
        default: hipLaunchKernelGGL((k_pm_sweep<R, LPC, false, true, false, false, true, true>), grid, block, 0, s, b, lut, seg_len, nseg, nseg_pad, 0); break;
    }
}
bool launch_pm_sweeps_merged(PmBatch& b, const float* lut, int R, int seg_len, int iteration, hipStream_t s)
{
    for (int k = 0; k < b.n; k++)
        if (!b.p[k].seed || !b.p[k].wl || !b.p[k].spec || !b.p[k].scand) return false;
    if (!(R == 9 || R == 17) || seg_len > kSpecMaxSteps || seg_len < 1) return false;
    const int w = b.p[0].P.w, h = b.p[0].P.h, gx = (w + kBlock - 1) / kBlock, gy = (h + kBlock - 1) / kBlock;
    b.merged_it = iteration;
    b.seg_len = seg_len;
    b.nseg_row = (w + seg_len - 1) / seg_len;
    b.nseg_col = (h + seg_len - 1) / seg_len;
    dim3 grid(gx * gy * (b.n * b.npairs)), block(256);
    if (R == 9 && pm_has_parity(b)) hipLaunchKernelGGL((k_pm_spec_all<9, 2>), grid, block, 0, s, b, lut, R, gx);
    else if (pm_has_parity(b)) hipLaunchKernelGGL((k_pm_spec_all<17, 2>), grid, block, 0, s, b, lut, R, gx);
    else if (R == 9) hipLaunchKernelGGL(k_pm_spec_all<9>, grid, block, 0, s, b, lut, R, gx);
    else hipLaunchKernelGGL(k_pm_spec_all<17>, grid, block, 0, s, b, lut, R, gx);
    for (int dir = 0; dir < 4; dir++) {
        if (R == 9) launch_sweep_merged<9, EPPM_LPC9_SPEC>(b, lut, seg_len, dir, s);
        else launch_sweep_merged<17, EPPM_LPC17_SPEC>(b, lut, seg_len, dir, s);
        b.sweep_seq++;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------
// Jump-flood propagation (d_jump_propagate, kernel.cu:800-841; launcher :843-857, disabled in the reference).
// Each pixel tries the matches of its neighbours at distance `step` (left, right, up, down), shifted by that
// distance, in order with strict <; candidates outside the image are skipped.  Jacobi: reads nnf, writes
// nnf_alt.  No serial chains: workgroup = 64 pixels x 4 candidates, wave k = candidate k, costs meet in
// LDS and wave 0 replays the in-order selection.  A candidate equal to the pixel's own match is rejected
// without evaluation (it would reproduce the stored cost).
//
// NEIGHBOR = true is d_neighbor_propagate (kernel.cu:720-787; ten launches per iteration at the disabled call
// site :1804-1809): distance 1, order upper, lower, left, right, the neighbour's match is copied UNSHIFTED
// and unchecked, and a neighbour outside the image is the clamped border pixel.
// ---------------------------------------------------------------------------------------------------
template <bool NEIGHBOR>
__global__ __launch_bounds__(256) void k_pm_jump(PmBatch B, const float* __restrict__ lut, int R, int step)
{
    __shared__ EPPM_LUT_ALIGN PatchLut L;
    __shared__ float s_cost[4][64];
    __shared__ int s_cand[4][64];
    const PmProblem pr = pm_problem(B, blockIdx.z);
    const int tid = threadIdx.x, lane = tid & 63, k = tid >> 6;
    load_patch_lut(L, lut, R, tid, 256);
    __syncthreads();
    const Planes P = to_dev(pr.P);
    const int x = blockIdx.x * kBlock + (lane & 15), y = blockIdx.y * 4 + (lane >> 4);
    const bool inimg = (x < P.w && y < P.h);
    const int nidx = y * B.npitch + x, cidx = y * B.cpitch + x;
    int bx = 0, by = 0;
    float cv = FLT_MAX;
    int cand = -1;
    if (inimg) {
        bx = pr.nnf[nidx * 2]; by = pr.nnf[nidx * 2 + 1];
        if (NEIGHBOR) {
            const int nx = iclamp(x + ((k == 2) ? -1 : (k == 3) ? 1 : 0), 0, P.w - 1);
            const int ny = iclamp(y + ((k == 0) ? -1 : (k == 1) ? 1 : 0), 0, P.h - 1);
            const int dx = pr.nnf[(ny * B.npitch + nx) * 2], dy = pr.nnf[(ny * B.npitch + nx) * 2 + 1];
            if (!(dx == bx && dy == by)) {
                cand = (dx & 0xffff) | (dy << 16);
                cv = patch_dist(P, L, R, x, y, dx, dy);
            }
        }
        const int nx = x + ((k == 0) ? -step : (k == 1) ? step : 0);
        const int ny = y + ((k == 2) ? -step : (k == 3) ? step : 0);
        if (!NEIGHBOR && nx >= 0 && nx < P.w && ny >= 0 && ny < P.h) {
            int dx = pr.nnf[(ny * B.npitch + nx) * 2], dy = pr.nnf[(ny * B.npitch + nx) * 2 + 1];
            if (k == 0) dx = (int)(int16_t)(dx - step);
            else if (k == 1) dx = (int)(int16_t)(dx + step);
            else if (k == 2) dy = (int)(int16_t)(dy - step);
            else dy = (int)(int16_t)(dy + step);
            if (!(dx < 0 || dy < 0 || dx >= P.w || dy >= P.h)) {
                cand = (dx & 0xffff) | (dy << 16);
                if (!(dx == bx && dy == by)) cv = patch_dist(P, L, R, x, y, dx, dy);
                else cand = -1;                                   // equals the current match: never accepted
            }
        }
    }
    s_cost[k][lane] = cv;
    s_cand[k][lane] = cand;
    __syncthreads();
    if (k == 0 && inimg) {
        float best_cost = pr.cost[cidx];
        for (int g = 0; g < 4; g++) {
            const int e = s_cand[g][lane];
            if (e == -1) continue;
            const float c = s_cost[g][lane];
            if (c < best_cost) { bx = (int)(int16_t)(e & 0xffff); by = e >> 16; best_cost = c; }
        }
        pr.nnf_alt[nidx * 2] = (int16_t)bx;
        pr.nnf_alt[nidx * 2 + 1] = (int16_t)by;
        pr.cost[cidx] = best_cost;
    }
}

void launch_pm_jump(const PmBatch& b, const float* lut, int R, int step, hipStream_t s)
{
    const int w = b.p[0].P.w, h = b.p[0].P.h;
    dim3 grid((w + kBlock - 1) / kBlock, (h + 3) / 4, b.n * b.npairs), block(256);
    hipLaunchKernelGGL(k_pm_jump<false>, grid, block, 0, s, b, lut, R, step);
}

void launch_pm_neighbor(const PmBatch& b, const float* lut, int R, hipStream_t s)
{
    const int w = b.p[0].P.w, h = b.p[0].P.h;
    dim3 grid((w + kBlock - 1) / kBlock, (h + 3) / 4, b.n * b.npairs), block(256);
    hipLaunchKernelGGL(k_pm_jump<true>, grid, block, 0, s, b, lut, R, 1);
}

// ---------------------------------------------------------------------------------------------------
// Random search (kernel.cu:1519-1594): G guesses at radii search_range, /2, ... around the pre-search
// best, evaluated in order with strict <.
// Random numbers: the 16x16 block's XORWOW stream, 2x256 draws per guess in row-major pixel order.  One
// wave produces the 512*G draws of the launch in parallel (lane l owns draws [per_lane*l, per_lane*(l+1)) ),
// then jumps its state over the other lanes' draws with the GF(2) skip matrix so that the next launch
// continues the same stream.
// Evaluation: all guesses come from the pre-search best, so their costs are independent.  A workgroup
// covers a QUARTER of the reference's 16x16 block (4 rows, 64 pixels): wave k evaluates guess k of those 64
// pixels, the costs meet in LDS and wave 0 replays the reference's in-order strict-< selection.  The four
// quarter-workgroups of a block draw the same numbers (cheap); only quarter 0 advances the stored state.
// ---------------------------------------------------------------------------------------------------
// jump over the other 63 lanes' draws: v <- v * skip_mat over GF(2); Weyl counter by multiplication
__device__ __forceinline__ void xorwow_skip(Xorwow& st, const uint32_t* __restrict__ skip_mat, uint32_t skip_weyl)
{
    const uint32_t v[5] = {st.v0, st.v1, st.v2, st.v3, st.v4};
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
    for (int wd = 0; wd < 5; wd++) {
        const uint32_t vw = v[wd];
#pragma unroll 8
        for (int b = 0; b < 32; b++) {
            const uint32_t* row = skip_mat + (wd * 32 + b) * 5;
            const uint32_t m = 0u - ((vw >> b) & 1u);
            a0 ^= m & row[0]; a1 ^= m & row[1]; a2 ^= m & row[2]; a3 ^= m & row[3]; a4 ^= m & row[4];
        }
    }
    st.v0 = a0; st.v1 = a1; st.v2 = a2; st.v3 = a3; st.v4 = a4;
    st.d += skip_weyl;
}

// The draws of one search launch for every block, ahead of time (PmRngDev::rand_tab): one wave per 16x16 block does what wave G of
// the search does -- lane l draws numbers [per_lane*l, per_lane*(l+1)) of the block's stream as shorts, then jumps its state over the
// other lanes' draws -- and the states stay in `work` for the next launch's table.
__global__ __launch_bounds__(64) void k_pm_rand_table(PmRngDev rng, uint32_t* __restrict__ work, int16_t* __restrict__ tab, int G)
{
    const int block_id = blockIdx.x, lane = threadIdx.x;
    const size_t so = ((size_t)block_id * 64 + lane) * 6;
    Xorwow st = load_state(work + so);
    int16_t* __restrict__ out = tab + (size_t)block_id * 512 * G + rng.per_lane * lane;
    for (int q = 0; q < rng.per_lane; q++) out[q] = (int16_t)xorwow_next(st);
    xorwow_skip(st, rng.skip_mat, rng.skip_weyl);
    store_state(work + so, st);
}
void launch_pm_rand_table(const PmRngDev& rng, uint32_t* work, int16_t* tab, int G, hipStream_t s)
{
    hipLaunchKernelGGL(k_pm_rand_table, dim3(rng.gx * rng.gy), dim3(64), 0, s, rng, work, tab, G);
}

// TAB: the launch's random numbers come from PmRngDev::rand_tab (drawn ahead, see above): no drawing wave, no state, no LDS copy of the
// numbers -- a lane loads the two shorts of its pixel and guess -- and the workgroup is G waves instead of G + 1.
// ROWS = 2 (numbers drawn ahead only): a workgroup covers an EIGHTH of the block (2 rows, 32 pixels; a wave = two guesses of them) -- for
// launches of so few workgroups that their count per CU quantises badly (one 1024x436 pair: 872 quarter-workgroups on 256 CUs run as 4
// per CU where 3.4 are needed; the kernel runs at its CU's L1 rate, so the launch lasts as long as the fullest CU).
#ifdef EPPM_SEARCH_WAVES
#define EPPM_SEARCH_OCC __attribute__((amdgpu_waves_per_eu(EPPM_SEARCH_WAVES, EPPM_SEARCH_WAVES)))
#else
#define EPPM_SEARCH_OCC
#endif
template <int RT, int PK = 0, bool TAB = false, int ROWS = 4>
__global__ __launch_bounds__(576) EPPM_SEARCH_OCC void k_pm_random_search(PmBatch B, PmRngDev rng, const float* __restrict__ lut, int R,
                                                          int search_range, int G)
{
    static_assert(ROWS == 4 || (ROWS == 2 && TAB && RT != 0), "eighth-block workgroups read their numbers from the table");
    using LUT = typename SearchLut<RT>::type;
    constexpr int PIXW = 16 * ROWS, NSUB = kBlock / ROWS;        // pixels per workgroup, workgroups per 16x16 block
    constexpr int TW = (RT == 0) ? 1 : kBlock + 2 * RT, TH = (RT == 0) ? 1 : ROWS + 2 * RT;
    __shared__ float4 s_src[TW * TH];
    __shared__ EPPM_LUT_ALIGN LUT L;
    __shared__ int16_t s_rand[TAB ? 2 : 8 * 512];
    __shared__ float s_cost[8][64];
    __shared__ int s_guess[8][64];
    __shared__ uint32_t s_state[TAB ? 2 : 64 * 6];
    // problem = id mod nprob (one problem per XCD L2, see k_pm_sweep); the rest of the id walks the quarter-blocks row by row
    const unsigned nprob = B.n * B.npairs, bq = blockIdx.x % nprob, brest = blockIdx.x / nprob;
    const int bxx = brest % rng.gx, byy = brest / rng.gx;
    const PmProblem pr = pm_problem(B, bq);
    const int tid = threadIdx.x;
    const int tile_y = byy / NSUB, quarter = byy % NSUB;
    const int block_id = tile_y * rng.gx + bxx;
    load_patch_lut(L, lut, R, tid, blockDim.x);
    if (!TAB && tid < 64) {
        const size_t so = ((size_t)block_id * 64 + tid) * 6;
        Xorwow st = load_state(pr.rng_work + so);
        const int base = rng.per_lane * tid;
        for (int q = 0; q < rng.per_lane; q++) s_rand[base + q] = (int16_t)xorwow_next(st);   // short(rdn), :1550-1551
        if (quarter == 0) store_state(s_state + tid * 6, st);     // advanced by wave G while the guesses are evaluated
    }
    const Planes P = to_dev(pr.P);
    if (RT != 0) {
        const int x0 = bxx * kBlock - RT, y0 = tile_y * kBlock + quarter * ROWS - RT;
        for (int t = tid; t < TW * TH; t += blockDim.x) {
            const int sy = iclamp(y0 + t / TW, 0, P.h - 1), sx = iclamp(x0 + t % TW, 0, P.w - 1);
            s_src[t] = P.pk1[(unsigned)(sy * P.pitch + sx)];
        }
    }
    __syncthreads();
    const int lane = tid % PIXW, k = tid / PIXW;                 // ROWS = 4: wave k = guess k; wave G advances the RNG states
    if (!TAB && k == G) {
        if (quarter == 0) {
            // Off the critical path: this wave has nothing else to do, the other G waves are evaluating guesses.  (Round 3 tried
            // giving the jump to the last guess's wave after its evaluation -- six waves per workgroup, four workgroups per CU
            // instead of three: slower, 1.40 -> 1.46 ms PatchMatch for one pair, -0.4 % batched: the jump then ends the workgroup.)
            const size_t so = ((size_t)block_id * 64 + lane) * 6;
            Xorwow st = load_state(s_state + lane * 6);
            xorwow_skip(st, rng.skip_mat, rng.skip_weyl);
            store_state(pr.rng_work_next + so, st);
        }
    }
    const int pix = quarter * PIXW + lane;                       // row-major index inside the 16x16 block
    const int x = bxx * kBlock + (pix & 15), y = tile_y * kBlock + (pix >> 4);
    const bool inimg = (k < G) && (x < P.w && y < P.h);
    const int nidx = y * B.npitch + x, cidx = y * B.cpitch + x;
    int bx = 0, by = 0;
    // sampling window of guess k: mag = search_range halved k times while >= 1 (:1564)
    int mag = search_range;
    for (int q = 0; q < k; q++) if (mag / 2 >= 1) mag /= 2;
    if (inimg) { bx = pr.nnf[nidx * 2]; by = pr.nnf[nidx * 2 + 1]; }
    int gx = 0, gy = 0;
    bool evaluate = false;
    if (inimg) {
        uint32_t rdn1, rdn2;                                                     // short -> unsigned int, :1558-1559
        if (TAB) {
            const uint32_t two = *reinterpret_cast<const uint32_t*>(rng.rand_tab + ((size_t)block_id * G + k) * 512 + 2 * pix);
            rdn1 = (uint32_t)(int32_t)(int16_t)(two & 0xffffu);
            rdn2 = (uint32_t)(int32_t)(int16_t)(two >> 16);
        } else {
            rdn1 = (uint32_t)(int32_t)s_rand[512 * k + 2 * pix];
            rdn2 = (uint32_t)(int32_t)s_rand[512 * k + 2 * pix + 1];
        }
        const int xmin = max(bx - mag, 0), xmax = min(bx + mag + 1, P.w + 1);
        const int ymin = max(by - mag, 0), ymax = min(by + mag + 1, P.h + 1);
        gx = (int)(int16_t)((uint32_t)xmin + rdn1 % (uint32_t)(xmax - xmin));
        gy = (int)(int16_t)((uint32_t)ymin + rdn2 % (uint32_t)(ymax - ymin));
#ifndef EPPM_SEARCH_SKIP_SAME
#define EPPM_SEARCH_SKIP_SAME 1
#endif
        // A guess equal to the pixel's current match would reproduce the stored cost bit for bit (the skip rule of the sweeps): the
        // reference evaluates and rejects it ("<"), here the lane sits the evaluation out -- a ninth of the radius-1 guesses.
        evaluate = !(EPPM_SEARCH_SKIP_SAME && gx == bx && gy == by);
        s_guess[k][lane] = (gx & 0xffff) | (gy << 16);
    }
    if (inimg) {
        float cv = INFINITY;
        if (evaluate) {
            cv = search_patch_dist<RT, PK>(P, L, R, s_src, TW, lane & 15, lane >> 4, x, y, gx, gy, pr.P);
        }
        s_cost[k][lane] = cv;
    }
    __syncthreads();
    if (k == 0 && inimg) {
        float best_cost = pr.cost[cidx];
        for (int g = 0; g < G; g++) {
            const float cv = s_cost[g][lane];
            if (cv < best_cost) {
                const int e = s_guess[g][lane];
                bx = (int)(int16_t)(e & 0xffff); by = e >> 16; best_cost = cv;
            }
        }
        pr.nnf[nidx * 2] = (int16_t)bx;
        pr.nnf[nidx * 2 + 1] = (int16_t)by;
        pr.cost[cidx] = best_cost;
    }
}

void launch_pm_random_search(const PmBatch& b, const PmRngDev& rng, const float* lut, int R, int search_range, int num_guess,
                             hipStream_t s)
{
    dim3 grid(rng.gx * rng.gy * 4 * b.n * b.npairs), block(64 * (num_guess + 1));      // + the wave that advances the RNG states
    if (rng.rand_tab && (R == 9 || R == 17)) {                                         // numbers drawn ahead: G waves per workgroup
        dim3 blockt(64 * num_guess);
        const bool have_pc_t = b.p[0].P.pc2 && (b.n < 2 || b.p[1].P.pc2);
#ifndef EPPM_SEARCH_HALF_BELOW_WGS
#define EPPM_SEARCH_HALF_BELOW_WGS 1024       // under four quarter-workgroups per CU: eighth-block workgroups (PatchMatch of one 1024x436 pair 1.362 -> 1.330 ms;
#endif                                        // at 4080 workgroups, one 1920x1080 pair, they lose: 4.49 -> 4.57 ms)
        if (pm_has_parity(b)) {                                                        // tolerance library: column-parity target planes
            if (R == 9 && (int)grid.x < EPPM_SEARCH_HALF_BELOW_WGS)
                hipLaunchKernelGGL((k_pm_random_search<9, 2, true, 2>), dim3(grid.x * 2), dim3(32 * num_guess), 0, s, b, rng, lut, R, search_range, num_guess);
            else if (R == 9) hipLaunchKernelGGL((k_pm_random_search<9, 2, true>), grid, blockt, 0, s, b, rng, lut, R, search_range, num_guess);
            else hipLaunchKernelGGL((k_pm_random_search<17, 2, true>), grid, blockt, 0, s, b, rng, lut, R, search_range, num_guess);
            return;
        }
        if (R == 9 && (int)grid.x < EPPM_SEARCH_HALF_BELOW_WGS) {
            hipLaunchKernelGGL((k_pm_random_search<9, 0, true, 2>), dim3(grid.x * 2), dim3(32 * num_guess), 0, s, b, rng, lut, R, search_range, num_guess);
            return;
        }
        if (R == 9) hipLaunchKernelGGL((k_pm_random_search<9, 0, true>), grid, blockt, 0, s, b, rng, lut, R, search_range, num_guess);
        else if (have_pc_t) hipLaunchKernelGGL((k_pm_random_search<17, 1, true>), grid, blockt, 0, s, b, rng, lut, R, search_range, num_guess);
        else hipLaunchKernelGGL((k_pm_random_search<17, 0, true>), grid, blockt, 0, s, b, rng, lut, R, search_range, num_guess);
        return;
    }
    // (radius 9 gathers the float4 plane: the 4-byte plane's conversions cost it more than the narrower gathers save, at every size)
#ifndef EPPM_SEARCH_PK17
#define EPPM_SEARCH_PK17 1
#endif
    const bool have_pc = b.p[0].P.pc2 && (b.n < 2 || b.p[1].P.pc2);
    if (R == 17 && have_pc && EPPM_SEARCH_PK17) hipLaunchKernelGGL((k_pm_random_search<17, 1>), grid, block, 0, s, b, rng, lut, R, search_range, num_guess);
    else if (R == 9) hipLaunchKernelGGL(k_pm_random_search<9>, grid, block, 0, s, b, rng, lut, R, search_range, num_guess);
    else if (R == 17) hipLaunchKernelGGL(k_pm_random_search<17>, grid, block, 0, s, b, rng, lut, R, search_range, num_guess);
    else hipLaunchKernelGGL(k_pm_random_search<0>, grid, block, 0, s, b, rng, lut, R, search_range, num_guess);
}

}  // namespace eppm
