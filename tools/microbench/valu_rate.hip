// valu_rate -- issue rate of the VALU instructions the patch-cost kernels are made of (gfx950).
// Every kernel runs ITER x 32 independent instances of ONE instruction per wave, 8 waves per SIMD on every CU;
// the table gives cycles per wave64 instruction per SIMD at the clock rocm-smi reported under load (2.3 GHz).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096
#define R8(x) x x x x x x x x
#define R32(x) R8(x) R8(x) R8(x) R8(x)

#define KERNEL(name, asmtext)                                                                  \
    __global__ __launch_bounds__(256) void name(float* out, float a, float b, int n)          \
    {                                                                                          \
        float v0 = a + threadIdx.x, v1 = b, v2 = a * b, v3 = 1.0f;                             \
        int i0 = (int)threadIdx.x, i1 = n;                                                    \
        for (int it = 0; it < n; it++) {                                                      \
            asm volatile(R32(asmtext) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(i0), "+v"(i1));  \
        }                                                                                      \
        out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + (float)(i0 + i1);          \
    }

// %0..%3 float regs, %4,%5 int regs.  Each line is ONE instruction; consecutive copies are dependent on their own
// result (latency is hidden by the 8 waves per SIMD).
KERNEL(k_fma,      "v_fma_f32 %0, %1, %2, %0\n")
KERNEL(k_fmac,     "v_fmac_f32 %0, %1, %2\n")
KERNEL(k_fmaak,    "v_fmaak_f32 %0, %1, %0, 0x3f317218\n")
KERNEL(k_add,      "v_add_f32 %0, %1, %0\n")
KERNEL(k_sub,      "v_sub_f32 %0, %1, %0\n")
KERNEL(k_mul,      "v_mul_f32 %0, %1, %0\n")
KERNEL(k_max3abs,  "v_max3_f32 %0, |%1|, |%2|, |%0|\n")
KERNEL(k_ldexp,    "v_ldexp_f32 %0, %0, %4\n")
KERNEL(k_cvt_i32,  "v_cvt_i32_f32 %4, %0\n")
KERNEL(k_rndne,    "v_rndne_f32 %0, %0\n")
KERNEL(k_bcnt,     "v_bcnt_u32_b32 %4, %5, %4\n")
KERNEL(k_xor,      "v_xor_b32 %4, %5, %4\n")
KERNEL(k_cndmask,  "v_cndmask_b32 %0, %1, %0, vcc\n")
KERNEL(k_med3i,    "v_med3_i32 %4, %4, %5, 0\n")
KERNEL(k_mul24,    "v_mul_u32_u24 %4, %4, %5\n")
KERNEL(k_addu,     "v_add_u32 %4, %5, %4\n")
KERNEL(k_lshl_add, "v_lshl_add_u32 %4, %4, 4, %5\n")
KERNEL(k_exp,      "v_exp_f32 %0, %0\n")
KERNEL(k_cndmask64,"v_cndmask_b32_e64 %0, %1, %0, s[10:11]\n")
KERNEL(k_max_abs,  "v_max_f32_e64 %0, |%1|, |%0|\n")
KERNEL(k_max_e32,  "v_max_f32_e32 %0, %1, %0\n")
KERNEL(k_max3,     "v_max3_f32 %0, %1, %2, %0\n")
KERNEL(k_med3f,    "v_med3_f32 %0, %1, %2, %0\n")
KERNEL(k_min_i32,  "v_min_i32_e32 %4, %5, %4\n")
KERNEL(k_lshl,     "v_lshlrev_b32_e32 %4, 4, %4\n")
KERNEL(k_sub_u32,  "v_sub_u32_e32 %4, %5, %4\n")
KERNEL(k_and,      "v_and_b32_e32 %4, %5, %4\n")
KERNEL(k_bfe,      "v_bfe_u32 %4, %4, 3, 8\n")
KERNEL(k_perm,     "v_perm_b32 %4, %4, %5, %4\n")
KERNEL(k_cmp,      "v_cmp_lt_f32_e32 vcc, %1, %0\n")
KERNEL(k_cvt_f_i,  "v_cvt_f32_i32_e32 %0, %4\n")
KERNEL(k_cvt_ub,   "v_cvt_f32_ubyte0_e32 %0, %4\n")
KERNEL(k_mad24,    "v_mad_u32_u24 %4, %4, %5, %4\n")
KERNEL(k_floor,    "v_floor_f32_e32 %0, %0\n")
KERNEL(k_fract,    "v_fract_f32_e32 %0, %0\n")
KERNEL(k_mul_lo,   "v_mul_lo_u32 %4, %4, %5\n")
KERNEL(k_add3,     "v_add3_u32 %4, %4, %5, %4\n")
KERNEL(k_sub_e64,  "v_sub_f32_e64 %0, %1, |%0|\n")
KERNEL(k_mov,      "v_mov_b32_e32 %0, %1\n")
// round 6: the integer-domain patch term of the tolerance library
KERNEL(k_sad_u32,  "v_sad_u32 %4, %5, %4, 0\n")
KERNEL(k_sad_u8,   "v_sad_u8 %4, %5, %4, 0\n")
KERNEL(k_max3_u32, "v_max3_u32 %4, %5, %4, %4\n")
KERNEL(k_max3_i32, "v_max3_i32 %4, %5, %4, %4\n")
KERNEL(k_cvt_u32,  "v_cvt_u32_f32 %4, %0\n")
KERNEL(k_mul24_sdwa, "v_mul_u32_u24_sdwa %4, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n")
KERNEL(k_lshl_sdwa, "v_lshlrev_b32_sdwa %4, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n")
KERNEL(k_sub_sdwa, "v_sub_u32_sdwa %4, %5, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n")
KERNEL(k_pk_max_u16, "v_pk_max_u16 %4, %5, %4\n")
KERNEL(k_pk_sub_i16, "v_pk_sub_i16 %4, %5, %4\n")
KERNEL(k_max_u32,  "v_max_u32_e32 %4, %5, %4\n")
KERNEL(k_and_or,   "v_and_or_b32 %4, %4, %5, %4\n")

// packed fp32: two independent IEEE operations per lane in one instruction (64-bit register pairs)
#define KERNEL_PK(name, asmtext)                                                               \
    __global__ __launch_bounds__(256) void name(float* out, float a, float b, int n)          \
    {                                                                                          \
        double d0 = a + threadIdx.x, d1 = b, d2 = a * b;                                       \
        for (int it = 0; it < n; it++) {                                                      \
            asm volatile(R32(asmtext) : "+v"(d0), "+v"(d1), "+v"(d2));                        \
        }                                                                                      \
        out[blockIdx.x * 256 + threadIdx.x] = (float)(d0 + d1 + d2);                          \
    }
KERNEL_PK(k_pk_add, "v_pk_add_f32 %0, %1, %0\n")
KERNEL_PK(k_pk_mul, "v_pk_mul_f32 %0, %1, %0\n")
KERNEL_PK(k_pk_fma, "v_pk_fma_f32 %0, %1, %2, %0\n")

// round 6 (late): table offsets formed in the float pipeline -- a product whose RESULT is denormal is the integer round(d * 1020)
KERNEL(k_mul_den,  "v_mul_f32 %0, 0x3fc, %1\n")
KERNEL(k_fma_den,  "v_fmaak_f32 %0, %1, %3, 0x3fc\n")

typedef void (*kern_t)(float*, float, float, int);

int main()
{
    int dev = 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8;            // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    struct { const char* name; kern_t k; } tab[] = {
        {"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac}, {"v_fmaak_f32", k_fmaak}, {"v_add_f32", k_add}, {"v_sub_f32", k_sub}, {"v_mul_f32", k_mul},
        {"v_max3_f32 |abs|", k_max3abs}, {"v_ldexp_f32", k_ldexp}, {"v_cvt_i32_f32", k_cvt_i32}, {"v_rndne_f32", k_rndne},
        {"v_bcnt_u32_b32", k_bcnt}, {"v_xor_b32", k_xor}, {"v_cndmask_b32", k_cndmask}, {"v_med3_i32", k_med3i},
        {"v_mul_u32_u24", k_mul24}, {"v_add_u32", k_addu}, {"v_lshl_add_u32", k_lshl_add}, {"v_exp_f32", k_exp},
        {"v_cndmask_b32_e64 sgpr", k_cndmask64}, {"v_max_f32_e64 |abs|", k_max_abs}, {"v_max_f32_e32", k_max_e32}, {"v_max3_f32", k_max3},
        {"v_med3_f32", k_med3f}, {"v_min_i32", k_min_i32}, {"v_lshlrev_b32", k_lshl}, {"v_sub_u32", k_sub_u32}, {"v_and_b32", k_and},
        {"v_bfe_u32", k_bfe}, {"v_perm_b32", k_perm}, {"v_cmp_lt_f32 vcc", k_cmp}, {"v_cvt_f32_i32", k_cvt_f_i}, {"v_cvt_f32_ubyte0", k_cvt_ub},
        {"v_mad_u32_u24", k_mad24}, {"v_floor_f32", k_floor}, {"v_fract_f32", k_fract}, {"v_mul_lo_u32", k_mul_lo}, {"v_add3_u32", k_add3},
        {"v_sub_f32_e64 |abs|", k_sub_e64}, {"v_mov_b32", k_mov},
        {"v_sad_u32", k_sad_u32}, {"v_sad_u8", k_sad_u8}, {"v_max3_u32", k_max3_u32}, {"v_max3_i32", k_max3_i32}, {"v_cvt_u32_f32", k_cvt_u32},
        {"v_mul_u32_u24_sdwa", k_mul24_sdwa}, {"v_lshlrev_b32_sdwa", k_lshl_sdwa}, {"v_sub_u32_sdwa", k_sub_sdwa}, {"v_pk_max_u16", k_pk_max_u16},
        {"v_pk_sub_i16", k_pk_sub_i16}, {"v_max_u32", k_max_u32}, {"v_and_or_b32", k_and_or},
        {"v_mul_f32 -> denormal", k_mul_den}, {"v_fmaak_f32 ~denormal", k_fma_den},
        {"v_pk_add_f32 (2 adds)", k_pk_add}, {"v_pk_mul_f32 (2 muls)", k_pk_mul}, {"v_pk_fma_f32 (2 fmas)", k_pk_fma},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%d CUs; clock for the cycle column: 2.3 GHz\n", cus);
    for (int w = 0; w < 40; w++) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.999f, ITER);   // clock ramp
    hipDeviceSynchronize();
    for (auto& t : tab) {
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.999f, 64);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.999f, ITER);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_simd = (double)ITER * 32 * 8;          // 8 waves per SIMD
        const double cyc = ms * 1e-3 * 2.3e9 / insts_per_simd;
        printf("%-20s %8.3f ms  %6.2f cycles per wave64 instruction per SIMD\n", t.name, ms, cyc);
    }
    return 0;
}
