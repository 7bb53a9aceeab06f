// eppm_device.cuh -- device-side building blocks shared by every EPPM kernel (gfx950 / CDNA4).
//
// The float formulas here are the product's own statement of the arithmetic that the CPU oracle
// (oracle/eppm_oracle.c) restates from the reference; the two are written independently and must
// agree bit for bit (tests/test_parity_gpu.py: test_fast_exp_bits, test_div_const_bits and every stage test).  Everything is compiled with -ffp-contract=off:
// the only fused operations are the explicit __builtin_fmaf calls below.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <type_traits>

#include "eppm_internal.h"

namespace eppm {

// ---- batches of pairs: pair k's plane = pair 0's pointer + k * stride bytes (eppm_internal.h: Batch) ----------
// (byte arithmetic on the pointer itself: a round trip through an integer would lose the global address space the
// compiler infers for kernel arguments and turn every access into a flat_load / flat_store)
template <class T>
__device__ __forceinline__ T* pair_ptr(T* p, size_t stride, unsigned pair)
{
    using Byte = typename std::conditional<std::is_const<T>::value, const char, char>::type;
    return reinterpret_cast<T*>(reinterpret_cast<Byte*>(p) + stride * pair);
}
template <class T>
__device__ __forceinline__ T* pair_ptr_opt(T* p, size_t stride, unsigned pair)     // NULL stays NULL
{
    return p ? pair_ptr(p, stride, pair) : p;
}

// ---- constants (defs.h:31-76 and file-local #defines of the reference) -------------------------
constexpr float kLambdaAd2 = 0.1f * 0.1f;          // LAMBDA_AD*LAMBDA_AD, defs.h:51
constexpr float kPmSigR2 = 0.1f * 0.1f;            // PM_SIG_R*PM_SIG_R, defs.h:48
constexpr float kWmfSigR2 = 0.02f * 0.02f;         // WMF_SIG_R*WMF_SIG_R, defs.h:60
constexpr float kBlfSigR2 = 0.02f * 0.02f;         // POSTPROC_BLF_SIG_R^2, refine :752
constexpr int   kStatRadius = 6;                   // defs.h:68
constexpr int   kStatCountThresh = ((2 * kStatRadius + 1) * (2 * kStatRadius + 1) / 2);  // refine :146
constexpr int   kStatSimThresh = 2;                // refine :147
constexpr int   kInvalid = -10000;                 // INVALID_LOCATION, refine :46
constexpr float kUnknownFlowThresh = 1e9f;         // defs.h:85 (1e9 is exact in float)
constexpr float kUnknownFlow = 1e10f;              // defs.h:90

// ---- __expf restated -------------------------------------------------------------------------
// y = x*log2e; n = rint(y); 2^(y-n) by a degree-5 Horner polynomial in fmaf; ldexp (one correctly rounded
// scaling: gradual underflow, 0 below 2^-150 -- the reference's ex2.approx is built without .ftz).
// Arguments on this path are always <= 0 and >= -1e5 ((int)n is in range).  v_mul, v_rndne, v_sub, 5 x v_fma, v_cvt, v_ldexp.
__device__ __forceinline__ float fast_exp(float x)
{
    // v_mul, v_rndne, v_sub, 5 x v_fma, v_cvt, v_ldexp.  (Forming rint(y) with the 1.5*2^23 constant and reading the
    // exponent from the mantissa bits trades the half-rate v_rndne/v_cvt for three full-rate instructions; measured
    // 2.5 % slower in the refine kernel: the instruction count is what costs, tools/microbench/valu_rate.hip.)
    const float y = x * 0x1.715476p+0f;
    const float n = __builtin_rintf(y);
    const float f = y - n;
    float p = __builtin_fmaf(0x1.5bba14p-10f, f, 0x1.3cea88p-7f);
    p = __builtin_fmaf(p, f, 0x1.c6b752p-5f);
    p = __builtin_fmaf(p, f, 0x1.ebf9bcp-3f);
    p = __builtin_fmaf(p, f, 0x1.62e42ap-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}

// ---- x / c for a compile-time constant c, correctly rounded in 2 operations ---------------------
// With zh = fl(1/c) and zl = fl(1/c - zh) (1/c in double):  q = fma(x, zh, fl(x*zl)).  Equal to the IEEE quotient x / c, which is what
// the oracle computes, for EVERY float in [2^-30, 4] (and 0, and the negatives by symmetry) for c = .1f*.1f and c = .02f*.02f, and for
// the integers 0..255 for c = 255: verified exhaustively on the CPU by tests/csrc/verify_divconst.c, sampled on the GPU
// (test_div_const_bits).  Rounds 1-3 used a 3-operation form (q0 = x*rc; r = fma(-c,q0,x); q = fma(r,rc,q0)); the patch term has two of
// these divisions, so this form takes it from 44 to 42 instructions.  The operand order matters: fma(x, zl, fl(x*zh)) is wrong for 7 %
// of the inputs.
struct DivConst { float zh, zl; };
constexpr DivConst make_div_const(float c)
{
    const double rc = 1.0 / (double)c;
    const float zh = (float)rc;
    return DivConst{zh, (float)(rc - (double)zh)};
}
__device__ __forceinline__ float div_const(float x, const DivConst d) { return __builtin_fmaf(x, d.zh, x * d.zl); }
static_assert(kPmSigR2 == kLambdaAd2 && kBlfSigR2 == kWmfSigR2, "one helper per distinct constant");
__device__ __forceinline__ float div_ad2(float x) { constexpr DivConst d = make_div_const(kLambdaAd2); return div_const(x, d); }   // also PM_SIG_R^2
__device__ __forceinline__ float div_wmf2(float x) { constexpr DivConst d = make_div_const(kWmfSigR2); return div_const(x, d); }    // also POSTPROC_BLF_SIG_R^2

// unorm8 -> float exactly as c/255.0f (cudaReadModeNormalizedFloat)
__device__ __forceinline__ float unorm8(float c) { constexpr DivConst d = make_div_const(255.0f); return div_const(c, d); }

struct rgbf { float x, y, z; };

__device__ __forceinline__ rgbf unpack_rgb(uint32_t p)
{
    rgbf r;
    r.x = unorm8((float)(p & 0xffu));
    r.y = unorm8((float)((p >> 8) & 0xffu));
    r.z = unorm8((float)((p >> 16) & 0xffu));
    return r;
}

__device__ __forceinline__ int iclamp(int v, int lo, int hi) { return min(max(v, lo), hi); }
// v_med3_i32: one instruction where min(max()) compiles to two (same result for lo <= hi)
__device__ __forceinline__ int med3i(int v, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ float max_abs_diff(const rgbf a, const rgbf b)
{
    return fmaxf(fmaxf(fabsf(a.x - b.x), fabsf(a.y - b.y)), fabsf(a.z - b.z));
}

// ---- texture model: point sampling, clamp addressing -------------------------------
// The patch kernels read ONE "texel plane" per image: float4 { R/255, G/255, B/255, census bits } -- the
// unorm8 -> float conversion of cudaReadModeNormalizedFloat is done once per pixel when the plane is built
// (k_census / k_pack) instead of once per fetch, and colour + census arrive in a single 16-byte gather
// where the reference issues a uchar4 and a u8 texture fetch.
struct Planes {
    const float4* pk1;      // texel plane of the source image, pitch in pixels
    const float4* pk2;      // ... of the target image
    int w, h;
    int pitch;              // pixels
};

#ifndef EPPM_TOL
// The census byte is stored shifted left by 2, so that w1 ^ w2 is the byte offset of entry (c1 ^ c2) in a
// 256-entry table cnx[b] = cn[popcount(b)]: xor + LDS read + add (4 VALU cycles fewer per sample than
// xor + v_bcnt (half rate) + the 9-entry table).
#ifndef EPPM_CENSUS_POPCNT
#define EPPM_CENSUS_POPCNT 1      // 1: the census byte replicated into its word, cn[hamming] by popcount (9 entries, conflict free) instead of cnx[xor]
#endif
__device__ __forceinline__ float4 make_texel(uint32_t rgba, uint32_t census)
{
    const rgbf c = unpack_rgb(rgba);
    return make_float4(c.x, c.y, c.z, __uint_as_float(EPPM_CENSUS_POPCNT ? (census & 0xffu) * 0x01010101u : (census & 0xffu) << 2));
}
__device__ __forceinline__ float census_cost(const float* __restrict__ cnx, uint32_t w1, uint32_t w2)
{
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(cnx) + (EPPM_CENSUS_POPCNT ? (uint32_t)__builtin_popcount(w1 ^ w2) : (w1 ^ w2)));
}
#else
// ---- the tolerance library (libeppm_hip_tol.so, -DEPPM_TOL) ------------------------------------------------
// NOT bit-identical to the oracle: the two software exp of a patch term (and their exact divisions) leave the patch term.  A channel
// of a texel is u8/255, so the L-inf distance of two texels is k/255 with k the INTEGER L-inf distance of their bytes (up to the
// rounding of u8/255, 1 ulp), hence
//     1 - exp(-d^2/s) = td[k_d]          exp(-(a^2+b^2)/s) = ta[k_a] * ta[k_b]   or   exp2(-c (k_a^2 + k_b^2)),  c = log2(e) / (255^2 s)
// with the 256-entry tables formed on the host in double.  Perturbation: ~2e-7 relative on a patch cost -- the class of the
// reference's own 2-ulp __expf (bao_pmflow_kernel.cu:283,289); what it does to the flow is measured by tools/tolerance_envelope.py
// on the CPU (mean EPE <= 1.1e-5 px on frame10/frame11, 0 on the 1024x436 synthetic pair) and asserted on the GPU by the -m gpu suite.
//
// Texel = { 4R, 4G, 4B as INTEGER BIT PATTERNS, census * 0x01010101 }.  An integer below 2^23 read as a float is a denormal, and
// gfx950 subtracts denormals exactly at full rate (kernels run with float_denorm_mode_32 = preserve): v_sub_f32 x3 + v_max3_f32 |.|
// leave the bits of 4k -- the byte offset of table entry k -- in 4 instructions / 10 issue cycles, no conversion
// (tools/ubench/denormal_int_linf.hip: 0 mismatches; 2.3 / 4.0 cycles per instruction).  The census byte is replicated into the four
// bytes of its word: popcount(w1 ^ w2) = 4 * hamming = the byte offset of cn[hamming] (9 entries in 9 banks: conflict free, where the
// exact library's 256-entry cnx[] costs ~5 bank-conflict cycles per read).
constexpr unsigned kTolScale = 4u;                   // texel channel = kTolScale * byte value = byte offset into td[] / ta[]
__device__ __forceinline__ float4 make_texel(uint32_t rgba, uint32_t census)
{
    return make_float4(__uint_as_float((rgba & 0xffu) * kTolScale), __uint_as_float(((rgba >> 8) & 0xffu) * kTolScale),
                       __uint_as_float(((rgba >> 16) & 0xffu) * kTolScale), __uint_as_float((census & 0xffu) * 0x01010101u));
}
#endif
__device__ __forceinline__ float one_minus_fast_exp(float x) { return 1 - fast_exp(x); }

// ---- f(d) for d = the L-inf distance of two unorm8 texels, by table: bit-identical to evaluating f ------------------------------
// A channel is fl(k/255), so such a distance is |fl(a/255) - fl(b/255)| for two bytes a, b: one of only 598 distinct floats, and
// any function of it -- 1 - exp(-d^2/s) of the patch term, exp(-d^2/s') of the smoothing and weighted-median weights -- has at most 598
// values.  The table holds exactly what the formula returns for each of them (filled on the device BY that formula, k_delta_values,
// once per device), so reading it is the same bits for 6 issue cycles (v_fma, v_lshl_add) instead of 34 (mul, exact division, exp polynomial,
// ldexp).
// Two levels: kd = |a - b| = round(d * 255) names a group of neighbouring floats (the distances with the same byte
// difference lie within 64 ulp of each other), t1[kd] = byte offset of the group in t2 minus 4 x the bits of its smallest member, so
// the entry of d is at (bits(d) << 2) + t1[kd].  Built and checked exhaustively over the 65 536 byte pairs on the host
// (api_common.cpp: delta_index; tests/test_abi_cpu.py::test_delta_index_covers_every_byte_pair).
struct DeltaTab {
    int t1[256];
    float t2[kDeltaSlots];
};
#ifndef EPPM_DELTA_PATCH
#define EPPM_DELTA_PATCH 1        // the patch data term by table (0: evaluate the formula)
#endif
#ifndef EPPM_DELTA_BLF
#define EPPM_DELTA_BLF 1          // the smoothing / weighted-median range weight by table (0: evaluate the formula)
#endif
// a DeltaTab lives in LDS; its look-up forms 32-bit LDS ADDRESSES (no base to add to an offset whose range the compiler cannot know)
typedef const __attribute__((address_space(3))) float lds_cfloat;
typedef const __attribute__((address_space(3))) int lds_cint;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ float delta_lookup(const DeltaTab& D, float d)
{
    // The address of t1[kd] in ONE full-rate instruction: a float below 2^-125 has the bits of its value in units of 2^-149, so an fma whose
    // RESULT lies there is an integer -- fma(d, 1020 * 2^-149, &t1 * 2^-149) = &t1 + round(1020 d) = &t1 + 4 kd exactly, the distances of a
    // group lying within 4e-6 relative of kd / 255 (kernels run with float_denorm_mode_32 = preserve, as the tolerance library's texels
    // need too).  t1[kd] already contains &t2 (load_delta_tab), so (bits(d) << 2) + t1[kd] is the entry's address: v_fma, ds_read,
    // v_lshl_add, ds_read = 6 issue cycles (the offset forms cost v_fma, v_cvt_u32, v_and + an address add, and v_lshlrev, v_add3: 18).
    const uint32_t a1 = __float_as_uint(__builtin_fmaf(d, __uint_as_float(1020u), __uint_as_float(lds_addr(D.t1))));
    const uint32_t a2 = (__float_as_uint(d) << 2) + (uint32_t)*reinterpret_cast<lds_cint*>(a1);
    return *reinterpret_cast<lds_cfloat*>(a2);
}
// The same look-up by OFFSETS, for a table that is a __shared__ object of its own (the smoothing, the weighted median): its address is a
// compile-time constant the compiler folds into the reads' immediate offsets, and the address form only costs such a kernel registers
// (smoothing 155 -> 165 VGPRs, 565 -> 584 us).  Table staged with load_delta_tab<false>.
__device__ __forceinline__ float delta_lookup_off(const DeltaTab& D, float d)
{
    const uint32_t o1 = __float_as_uint(d * __uint_as_float(1020u));
    const uint32_t o2 = (__float_as_uint(d) << 2) + (uint32_t)*reinterpret_cast<const int*>(reinterpret_cast<const char*>(D.t1) + o1);
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(D.t2) + o2);
}
// global layout behind a look-up table's own entries: t1[256] (int bits), then t2[kDeltaSlots]
template <bool ADDRESSES = true>
__device__ __forceinline__ void load_delta_tab(DeltaTab& D, const float* __restrict__ src, int tid, int nthreads)
{
    const int t2_at = ADDRESSES ? (int)lds_addr(D.t2) : 0;
    for (int t = tid; t < 256; t += nthreads) D.t1[t] = __float_as_int(src[t]) + t2_at;          // modulo 2^32, as the look-up's sum
    for (int t = tid; t < kDeltaSlots; t += nthreads) D.t2[t] = src[256 + t];
}
__device__ __forceinline__ rgbf texel_rgb(const float4 t) { return rgbf{t.x, t.y, t.z}; }

__device__ __forceinline__ float4 tex_px(const float4* img, int pitch, int w, int h, int x, int y)
{
    x = iclamp(x, 0, w - 1);
    y = iclamp(y, 0, h - 1);
    return img[(unsigned)(y * pitch + x)];
}
// 32-bit byte offset form: global_load_dwordx4 with an SGPR base and one VGPR offset (no 64-bit address math).
// Planes are < 4 GiB (32767 x 32767 is rejected at eppm_create for 16-byte texels beyond that).
__device__ __forceinline__ float4 texel_at(const float4* base, unsigned byte_off)
{
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ unsigned texel_off(int pitch16, int w, int h, int x, int y)
{
    x = iclamp(x, 0, w - 1);
    y = iclamp(y, 0, h - 1);
    return __umul24((unsigned)y, (unsigned)pitch16) + ((unsigned)x << 4);
}
// the same with the column in byte units (x16 = 16*x, wmax16 = 16*(w-1)): add, 2 x v_med3, v_mad_u32_u24
__device__ __forceinline__ unsigned texel_off16(int pitch16, int wmax16, int hmax, int x16, int y)
{
    return __umul24((unsigned)med3i(y, 0, hmax), (unsigned)pitch16) + (unsigned)med3i(x16, 0, wmax16);
}
// same for the plain RGBA planes (guide image of the WMF / hole filling)
__device__ __forceinline__ uint32_t tex_rgba(const uint32_t* img, int pitch, int w, int h, int x, int y)
{
    x = iclamp(x, 0, w - 1);
    y = iclamp(y, 0, h - 1);
    return img[(unsigned)(y * pitch + x)];
}

// ---- one sample of the patch cost (bao_pmflow_kernel.cu:275-295), both texels already fetched ------
// gsp = gs[|j|]*gs[|i|] (the product is formed first in the reference too: "weight *= a*b").
#ifndef EPPM_TOL
struct ExactTables { const float* cnx; const DeltaTab& D; };      // what PatchLutT::tab() hands to the term: both tables live in LDS
__device__ __forceinline__ void patch_terms(const float4 q1, const float4 q2, const rgbf c1, const rgbf c2, float gsp,
                                            const ExactTables T, float& cost_term, float& weight_term)
{
    const float* __restrict__ cnx = T.cnx;
    const DeltaTab& D = T.D;
    const rgbf p1 = texel_rgb(q1);
    const rgbf p2 = texel_rgb(q2);
    float cost = max_abs_diff(p1, p2);
    cost = EPPM_DELTA_PATCH ? delta_lookup(D, cost) : one_minus_fast_exp(div_ad2(-(cost * cost)));     // the same bits either way
    cost += census_cost(cnx, __float_as_uint(q1.w), __float_as_uint(q2.w));
    float weight = max_abs_diff(c1, p1);
    weight *= weight;
    float temp = max_abs_diff(c2, p2);
    temp *= temp;
    weight = fast_exp(div_ad2(-(weight + temp)));
    weight *= gsp;
    cost *= weight;
    cost_term = cost;
    weight_term = weight;
}
// the running sums of a patch advance by one sample: the reference's two separate chains
__device__ __forceinline__ void patch_accum(float& cost_sum, float& weight_sum, float cost_term, float weight_term)
{
    cost_sum += cost_term;
    weight_sum += weight_term;
}
#else
struct TolTables {            // in LDS, first member of every kernel's table block (offsets below 64 KB: immediates of the ds_read)
    float td[256];            // 1 - exp(-(k/255)^2/s)
    float ta[256];            // exp(-(k/255)^2/s)
    float cn[16];             // cn[0..8], kernel.cu:670-687
};
// byte offset 4k of table entry k = the L-inf distance of two texels (see make_texel)
__device__ __forceinline__ uint32_t linf_off(const rgbf a, const rgbf b) { return __float_as_uint(max_abs_diff(a, b)); }
__device__ __forceinline__ float tol_at(const float* __restrict__ t, uint32_t byte_off)
{
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(t) + byte_off);
}
// 1 - exp(-d^2/s) + cn[hamming]
__device__ __forceinline__ float tol_cost(const TolTables& T, const rgbf p1, const rgbf p2, uint32_t w1, uint32_t w2)
{
    return tol_at(T.td, linf_off(p1, p2)) + tol_at(T.cn, (uint32_t)__builtin_popcount(w1 ^ w2));
}
// exp2(lsrc - c k^2) by the hardware v_exp_f32, for an L-inf distance given as its byte offset 4k; c = log2(e) / (255^2 s).  Results
// below 2^-126 are 0 (the instruction flushes; the exact library keeps them down to 2^-150).
constexpr float kTolExpC = (float)(1.4426950408889634 / (255.0 * 255.0 * (double)kPmSigR2) / (double)(kTolScale * kTolScale));
// Every weight of a patch may carry a common factor: it cancels in cost_sum / weight_sum.  The refine's exp2 argument starts at +24
// (load_patch_lut<true>: log2(gs_j gs_i) + kTolWeightBias): v_exp_f32 flushes results below 2^-126, so the weights that vanish are those
// below 2^-150 -- exactly the ones the exact formula rounds to 0 -- and everything above keeps full precision (the exact library's own
// weights between 2^-150 and 2^-126 are denormals with 1-23 bits).  The PatchMatch tables carry 2^24 per factor for the same reason.
constexpr float kTolWeightBias = 24.0f;
__device__ __forceinline__ float tol_exp_arg(uint32_t off4k, float lsrc)
{
    // the integer 4k as a float: its bit pattern is the denormal 4k * 2^-149, and one full-rate multiplication by 2^100 makes it the
    // normal number 4k * 2^-49 (exact) -- 2 issue cycles where v_cvt_f32_u32 takes 4; the scale is folded into the constant
    const float kf = __uint_as_float(off4k) * 0x1p100f;
    return __builtin_fmaf(kf * kf, -kTolExpC * 0x1p98f, lsrc);
}
// The same argument with the distance formed directly as the normal float 4k * 2^-49: the centre texel is kept scaled by 2^100
// (tol_scale_centre, once per candidate) and each channel difference is ONE fma, sample * 2^100 - centre (exact) -- the values of
// tol_exp_arg(linf_off(c, p), lsrc) bit for bit, one multiplication fewer per term
__device__ __forceinline__ rgbf tol_scale_centre(const rgbf c) { return rgbf{c.x * 0x1p100f, c.y * 0x1p100f, c.z * 0x1p100f}; }
__device__ __forceinline__ float tol_exp_arg_scaled(const rgbf cs, const rgbf p, float lsrc)
{
    const float kf = fmaxf(fmaxf(fabsf(__builtin_fmaf(p.x, 0x1p100f, -cs.x)), fabsf(__builtin_fmaf(p.y, 0x1p100f, -cs.y))), fabsf(__builtin_fmaf(p.z, 0x1p100f, -cs.z)));
    return __builtin_fmaf(kf * kf, -kTolExpC * 0x1p98f, lsrc);
}
// a word {R, G, B, census} of the 4-byte planes -> the texel make_texel builds from it: three SDWA shifts (byte k << 2) and a byte permute
// (census into all four bytes), 16 issue cycles; `two` = a register holding 2 (an SDWA operand cannot be an inline constant)
__device__ __forceinline__ float4 unpack_texel(uint32_t w, uint32_t two)
{
    uint32_t r, g, b;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(two), "v"(w));
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(g) : "v"(two), "v"(w));
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(b) : "v"(two), "v"(w));
    static_assert(kTolScale == 4u, "the shift above");
    return make_float4(__uint_as_float(r), __uint_as_float(g), __uint_as_float(b), __uint_as_float(__builtin_amdgcn_perm(w, w, 0x03030303u)));
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// cost_term = the sample's cost (NOT yet multiplied by its weight: patch_accum fuses that), weight_term = its weight
__device__ __forceinline__ void patch_terms(const float4 q1, const float4 q2, const rgbf c1, const rgbf c2, float gsp,
                                            const TolTables& T, float& cost_term, float& weight_term)
{
    const rgbf p1 = texel_rgb(q1);
    const rgbf p2 = texel_rgb(q2);
    cost_term = tol_cost(T, p1, p2, __float_as_uint(q1.w), __float_as_uint(q2.w));
    // ta[] holds 2^24 * exp(-(k/255)^2/s) (kTolWeightBias per factor): weights down to exp(-100) -- a saturated edge -- stay normal floats
    // with full precision where the unscaled product would be a denormal with a few bits; the common factor 2^48 cancels in cost / weight.
    weight_term = (tol_at(T.ta, linf_off(c1, p1)) * gsp) * tol_at(T.ta, linf_off(c2, p2));
}
__device__ __forceinline__ void patch_accum(float& cost_sum, float& weight_sum, float cost_term, float weight_term)
{
    cost_sum = __builtin_fmaf(cost_term, weight_term, cost_sum);
    weight_sum += weight_term;
}
#endif

template <class TAB>
__device__ __forceinline__ void patch_sample(const Planes& P, const rgbf c1, const rgbf c2, int sx1, int sy1, int sx2,
                                             int sy2, float gsp, const TAB& cnx, float& cost_term,
                                             float& weight_term)
{
    const float4 q1 = tex_px(P.pk1, P.pitch, P.w, P.h, sx1, sy1);
    const float4 q2 = tex_px(P.pk2, P.pitch, P.w, P.h, sx2, sy2);
    patch_terms(q1, q2, c1, c2, gsp, cnx, cost_term, weight_term);
}

// ---- the order in which a PatchMatch cost is summed --------------------------------------------------------------------------------
// Exact library: the reference's, one chain over the S*S samples (i outer, j inner).  Tolerance library: the same row-major sample order
// cut into CHUNKS -- half rows at radius 9 (5 samples), thirds of a row at radius 17 (6), halves in general --, each chunk summed from
// zero by fused multiply-adds, the chunk sums added one after the other.  Every kernel forms exactly this sum: a lane that evaluates
// alone closes a chunk every few samples (PatchSum::flush), a cooperative evaluation gives every lane one chunk and lets the total hop
// lane to lane with ONE addition per hop (the exact library adds a whole chunk per hop on every lane: 16 x 16 instructions against
// 20 x 4).  The cost of a (pixel, candidate) pair is thus the same bits whichever kernel evaluates it -- which strict "<" between
// equal-cost candidates requires.
__host__ __device__ constexpr int tol_chunk(int R) { return R == 17 ? 6 : (R + 2) / 2; }
struct PatchSum {
    float cs = 0.0f, ws = 0.0f;
#ifdef EPPM_TOL
    float cc = 0.0f, cw = 0.0f;
    __device__ __forceinline__ void add(float cost_term, float weight_term) { patch_accum(cc, cw, cost_term, weight_term); }
    __device__ __forceinline__ void flush() { cs += cc; ws += cw; cc = 0.0f; cw = 0.0f; }          // the end of a chunk
#else
    __device__ __forceinline__ void add(float cost_term, float weight_term) { patch_accum(cs, ws, cost_term, weight_term); }
    __device__ __forceinline__ void flush() {}
#endif
    __device__ __forceinline__ float result() const { return cs / ws; }
};

// LUTs staged in LDS by every patch kernel: gsp[i*S + j] = gs[|2j-R|]*gs[|2i-R|], and the table(s) of the per-sample terms -- tab():
// exact library cnx[b] = cn[popcount(b)]; tolerance library td[], ta[], cn[] (TolTables, FIRST: low LDS addresses)
template <int MAXS>
struct PatchLutT {
#ifndef EPPM_TOL
    float gsp[MAXS * MAXS];
    float cnx[EPPM_CENSUS_POPCNT ? 16 : 256];
    DeltaTab D;                       // 1 - exp(-d^2 / LAMBDA_AD^2) of the data term, by table (delta_lookup)
    __device__ __forceinline__ ExactTables tab() const { return ExactTables{cnx, D}; }
#else
    TolTables T;
    float gsp[MAXS * MAXS];
    __device__ __forceinline__ const TolTables& tab() const { return T; }
#endif
};
using PatchLut = PatchLutT<kMaxS>;     // any radius the ABI accepts; kernels instantiated per radius use PatchLutT<R + 1>
#ifdef EPPM_TOL
#define EPPM_LUT_ALIGN alignas(1024)   // the LDS allocator places the most aligned object first: the tables' offsets fit a ds_read's 16-bit immediate
#else
#define EPPM_LUT_ALIGN
#endif

// lut_src layout in global memory: gs[0..R] then cn[0..8]; exact library: then the data term's DeltaTab; tolerance library: then td[0..255] = 1 - exp(-(k/255)^2/s) and
// ta[0..255] = exp(-(k/255)^2/s), s = LAMBDA_AD^2 = PM_SIG_R^2, formed in double on the host (eppm_api.cpp: host_pm_lut)
// LOG2 (tolerance library's refine kernels): gsp[] holds log2 of the products -- the weight there is ONE exp2 of a summed argument
template <bool LOG2 = false, int MAXS>
__device__ __forceinline__ void load_patch_lut(PatchLutT<MAXS>& L, const float* __restrict__ lut_src, int R, int tid, int nthreads)
{
    const int S = R + 1;
    for (int t = tid; t < S * S; t += nthreads) {
        const int i = t / S, j = t % S;
        const int ai = abs(2 * i - R), aj = abs(2 * j - R);
        const float g = lut_src[aj] * lut_src[ai];
#ifdef EPPM_TOL
        L.gsp[i * S + j] = LOG2 ? log2f(g) + kTolWeightBias : g;
#else
        L.gsp[i * S + j] = LOG2 ? log2f(g) : g;
#endif
    }
#ifndef EPPM_TOL
    if (EPPM_CENSUS_POPCNT) { for (int t = tid; t < 16; t += nthreads) L.cnx[t] = (t < 9) ? lut_src[R + 1 + t] : 0.0f; }
    else for (int t = tid; t < 256; t += nthreads) L.cnx[t] = lut_src[R + 1 + __builtin_popcount(t)];
    load_delta_tab(L.D, lut_src + R + 10, tid, nthreads);
#else
    for (int t = tid; t < 256; t += nthreads) { L.T.td[t] = lut_src[R + 10 + t]; L.T.ta[t] = lut_src[R + 10 + 256 + t]; }
    for (int t = tid; t < 16; t += nthreads) L.T.cn[t] = (t < 9) ? lut_src[R + 1 + t] : 0.0f;
#endif
}

// ---- the patch cost, bao_pmflow_kernel.cu:255-301: sequential i-outer / j-inner accumulation ------
// The gathers of up to 5 consecutive samples are issued together before their terms are computed (the sums
// still advance in sample order).
__device__ __forceinline__ float patch_dist(const Planes& P, const PatchLut& L, int R, int x1, int y1, int x2, int y2)
{
    const int pitch16 = P.pitch << 4;
    const rgbf c1 = texel_rgb(texel_at(P.pk1, texel_off(pitch16, P.w, P.h, x1, y1)));
    const rgbf c2 = texel_rgb(texel_at(P.pk2, texel_off(pitch16, P.w, P.h, x2, y2)));
    PatchSum sum;
    const int S = R + 1, CS = tol_chunk(R);
    for (int ii = 0; ii < S; ii++) {
        const int i = 2 * ii - R;
        const unsigned r1 = __umul24((unsigned)iclamp(y1 + i, 0, P.h - 1), (unsigned)pitch16);
        const unsigned r2 = __umul24((unsigned)iclamp(y2 + i, 0, P.h - 1), (unsigned)pitch16);
        int left = CS;                         // samples to the end of the chunk (a row starts a chunk)
        for (int j0 = 0; j0 < S; j0 += 5) {
            float4 q1[5], q2[5];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int j = 2 * min(j0 + k, S - 1) - R;
                q1[k] = texel_at(P.pk1, r1 + ((unsigned)iclamp(x1 + j, 0, P.w - 1) << 4));
                q2[k] = texel_at(P.pk2, r2 + ((unsigned)iclamp(x2 + j, 0, P.w - 1) << 4));
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                if (j0 + k < S) {
                    float ct, wt;
                    patch_terms(q1[k], q2[k], c1, c2, L.gsp[ii * S + j0 + k], L.tab(), ct, wt);
                    sum.add(ct, wt);
                    if (--left == 0 || j0 + k == S - 1) { sum.flush(); left = CS; }
                }
            }
        }
    }
    return sum.result();
}

// plane-fitting coefficients: bao_pmflow_kernel.cu:319-332 (pass 0 = no warp)
// one affine pass of bao_pmflow_kernel.cu:334-513; target texel = floor of the float coordinate
template <int PASS>
__device__ __forceinline__ float patch_dist_pass(const Planes& P, const PatchLut& L, int R, int x1, int y1, float uu,
                                                 float vv, const rgbf c1, const rgbf c2)
{
    constexpr float kc[4][4] = {
        {0.0f, 0.0f, 0.0f, 0.0f},
        {0.177f, -0.011f, -0.003f, 0.301f},
        {0.125f, -0.357f, 0.009f, 0.308f},
        {0.205f, 0.370f, 0.011f, 0.296f},
    };
    float cost_sum = 0.0f, weight_sum = 0.0f;
    const int S = R + 1;
    for (int ii = 0; ii < S; ii++) {
        const int i = 2 * ii - R;
        for (int jj = 0; jj < S; jj++) {
            const int j = 2 * jj - R;
            const float cx1 = (float)(x1 + j);
            const float cy1 = (float)(y1 + i);
            float cx2, cy2;
            if (PASS == 0) {
                cx2 = cx1 + uu;
                cy2 = cy1 + vv;
            } else {
                cx2 = cx1 + uu + (float)(j)*kc[PASS][0] + (float)(i)*kc[PASS][1];
                cy2 = cy1 + vv + (float)(j)*kc[PASS][2] + (float)(i)*kc[PASS][3];
            }
            float ct, wt;
            patch_sample(P, c1, c2, x1 + j, y1 + i, (int)floorf(cx2), (int)floorf(cy2), L.gsp[ii * S + jj], L.tab(), ct, wt);
            patch_accum(cost_sum, weight_sum, ct, wt);
        }
    }
    return cost_sum / weight_sum;
}

__device__ __forceinline__ float patch_dist_planefit(const Planes& P, const PatchLut& L, int R, int x1, int y1, int x2, int y2)
{
    const rgbf c1 = texel_rgb(tex_px(P.pk1, P.pitch, P.w, P.h, x1, y1));
    const rgbf c2 = texel_rgb(tex_px(P.pk2, P.pitch, P.w, P.h, x2, y2));
    const float uu = (float)(x2 - x1);
    const float vv = (float)(y2 - y1);
    const float c_1 = patch_dist_pass<0>(P, L, R, x1, y1, uu, vv, c1, c2);
    const float c_2 = patch_dist_pass<1>(P, L, R, x1, y1, uu, vv, c1, c2);
    const float c_3 = patch_dist_pass<2>(P, L, R, x1, y1, uu, vv, c1, c2);
    const float c_4 = patch_dist_pass<3>(P, L, R, x1, y1, uu, vv, c1, c2);
    // __min(cost1,__min(cost2,__min(cost3,cost4))), __min(a,b) = (a<b)?a:b  (kernel.cu:512)
    const float m34 = (c_3 < c_4) ? c_3 : c_4;
    const float m234 = (c_2 < m34) ? c_2 : m34;
    return (c_1 < m234) ? c_1 : m234;
}

// ---- XORWOW (cuRAND default generator; Marsaglia 2003) -----------------------------------------
struct Xorwow {
    uint32_t v0, v1, v2, v3, v4, d;
};
__device__ __forceinline__ uint32_t xorwow_next(Xorwow& s)
{
    const uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1; s.v1 = s.v2; s.v2 = s.v3; s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v4 + s.d;
}

// float -> short as cvt.rzi.s16.f32 (truncate toward zero, saturate)
__device__ __forceinline__ int f2short(float f)
{
    if (!(f > -32768.0f)) return -32768;
    if (f > 32767.0f) return 32767;
    return (int)f;
}

}  // namespace eppm
