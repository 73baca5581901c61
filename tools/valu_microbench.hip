// valu_microbench.hip — measures the f32 VALU issue rate of gfx950 that bounds the trace kernel:
// wave-instructions per cycle per SIMD for independent / dependent v_fma_f32, v_pk_fma_f32,
// compare+select and transcendental streams at 1..8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o gpurun_out/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 32768

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, float b, unsigned long long *cyc) {
    unsigned long long c0 = __builtin_readcyclecounter();  // s_memtime: shader-clock ticks
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float x0b = threadIdx.x * 3e-3f;
    for (int i = 0; i < ITER; ++i) {
        if (MODE == 40) {
            asm volatile("v_perm_b32 %0, %9, %0, %10\n v_perm_b32 %1, %9, %1, %10\n v_perm_b32 %2, %9, %2, %10\n v_perm_b32 %3, %9, %3, %10\n v_perm_b32 %4, %9, %4, %10\n v_perm_b32 %5, %9, %5, %10\n v_perm_b32 %6, %9, %6, %10\n v_perm_b32 %7, %9, %7, %10\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "v"(b), "v"(x0b));
        }
        if (MODE == 41) {
            asm volatile("v_and_or_b32 %0, %0, %9, %10\n v_and_or_b32 %1, %1, %9, %10\n v_and_or_b32 %2, %2, %9, %10\n v_and_or_b32 %3, %3, %9, %10\n v_and_or_b32 %4, %4, %9, %10\n v_and_or_b32 %5, %5, %9, %10\n v_and_or_b32 %6, %6, %9, %10\n v_and_or_b32 %7, %7, %9, %10\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "v"(b), "v"(x0b));
        }
        if (MODE == 42) {
            asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %5, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %6, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %7, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "v"(b), "v"(x0b));
        }
        if (MODE == 43) {
            asm volatile("v_cvt_f32_ubyte1 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte1 %2, %2\n v_cvt_f32_ubyte1 %3, %3\n v_cvt_f32_ubyte1 %4, %4\n v_cvt_f32_ubyte1 %5, %5\n v_cvt_f32_ubyte1 %6, %6\n v_cvt_f32_ubyte1 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "v"(b), "v"(x0b));
        }
        if (MODE == 44) {
            asm volatile("v_cvt_f32_f16 %0, %0\n v_cvt_f32_f16 %1, %1\n v_cvt_f32_f16 %2, %2\n v_cvt_f32_f16 %3, %3\n v_cvt_f32_f16 %4, %4\n v_cvt_f32_f16 %5, %5\n v_cvt_f32_f16 %6, %6\n v_cvt_f32_f16 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "v"(b), "v"(x0b));
        }
        if (MODE == 45) {
            asm volatile("v_perm_b32 %0, %9, %0, %8\n v_perm_b32 %1, %9, %1, %8\n v_perm_b32 %2, %9, %2, %8\n v_perm_b32 %3, %9, %3, %8\n v_perm_b32 %4, %9, %4, %8\n v_perm_b32 %5, %9, %5, %8\n v_perm_b32 %6, %9, %6, %8\n v_perm_b32 %7, %9, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "v"(b), "v"(x0b));
        }
        if (MODE == 46) {  // v_fma_mix_f32, first source a half (low word) -- the wide-node box decode
            asm volatile("v_fma_mix_f32 %0, %10, %8, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %10, %8, %1 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %10, %8, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %10, %8, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %10, %8, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %10, %8, %5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %10, %8, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %10, %8, %7 op_sel_hi:[1,0,0]\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(x0b));
        }
        if (MODE == 47) {  // the same, high word, negated
            asm volatile("v_fma_mix_f32 %0, -%10, %8, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, -%10, %8, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, -%10, %8, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, -%10, %8, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, -%10, %8, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, -%10, %8, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, -%10, %8, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, -%10, %8, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(x0b));
        }
        if (MODE == 0) {  // 8 independent fma chains
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 1) {  // one dependent chain, 8 long
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(x0) : "v"(a), "v"(b));
        } else if (MODE == 2) {  // mul/add mix, independent
            asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                         "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 3) {  // compare + select pairs
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 4) {  // transcendental (rcp / sqrt), independent
            asm volatile("v_rcp_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_rcp_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_rcp_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (MODE == 5) {  // integer xor/shift/rotate (RNG-like), independent
            asm volatile("v_xor_b32 %0, %0, %8\n v_alignbit_b32 %1, %1, %1, 6\n v_lshlrev_b32 %2, 9, %2\n v_add_u32 %3, %3, %9\n"
                         "v_xor_b32 %4, %4, %8\n v_alignbit_b32 %5, %5, %5, 19\n v_lshrrev_b32 %6, 8, %6\n v_cvt_f32_u32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 7) {  // packed fma: 4 independent v_pk_fma_f32 on register pairs (= 8 scalar fma)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
        } else if (MODE == 8) {  // packed mul/add
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
        } else if (MODE == 9) {  // v_cmp into vcc only
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 10) {  // v_cmp (VOP3) into distinct SGPR pairs
            asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n"
                         "v_cmp_lt_f32 s[28:29], %4, %8\n v_cmp_lt_f32 s[30:31], %5, %8\n v_cmp_lt_f32 s[32:33], %6, %8\n v_cmp_lt_f32 s[34:35], %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
        } else if (MODE == 11) {  // v_min / v_max
            asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %9\n"
                         "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 12) {  // v_min3 / v_max3 / v_med3
            asm volatile("v_min3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                         "v_max3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 13) {  // v_cndmask only (vcc constant)
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 14) {  // VOP3 encodings with |abs| / neg modifiers
            asm volatile("v_mul_f32 %0, |%0|, %8\n v_sub_f32 %1, -%1, %9\n v_max_f32 %2, |%2|, |%8|\n v_fma_f32 %3, -%3, %8, %9\n"
                         "v_mul_f32 %4, |%4|, %8\n v_sub_f32 %5, -%5, %9\n v_max_f32 %6, |%6|, |%8|\n v_fma_f32 %7, -%7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 15) {  // 32-bit literal operands (v_fmamk / v_add with literal)
            asm volatile("v_fmamk_f32 %0, %0, 0x3951b717, %8\n v_add_f32 %1, 0x38d1b717, %1\n v_fmamk_f32 %2, %2, 0x3951b717, %8\n v_add_f32 %3, 0x38d1b717, %3\n"
                         "v_fmamk_f32 %4, %4, 0x3951b717, %8\n v_add_f32 %5, 0x38d1b717, %5\n v_fmamk_f32 %6, %6, 0x3951b717, %8\n v_add_f32 %7, 0x38d1b717, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 16) {  // v_fmac (VOP2) independent
            asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                         "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 20) {
            asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 21) {
            asm volatile("v_alignbit_b32 %0, %0, %0, 6\n v_alignbit_b32 %1, %1, %1, 6\n v_alignbit_b32 %2, %2, %2, 6\n v_alignbit_b32 %3, %3, %3, 6\n v_alignbit_b32 %4, %4, %4, 6\n v_alignbit_b32 %5, %5, %5, 6\n v_alignbit_b32 %6, %6, %6, 6\n v_alignbit_b32 %7, %7, %7, 6\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 22) {
            asm volatile("v_lshlrev_b32 %0, 9, %0\n v_lshlrev_b32 %1, 9, %1\n v_lshlrev_b32 %2, 9, %2\n v_lshlrev_b32 %3, 9, %3\n v_lshlrev_b32 %4, 9, %4\n v_lshlrev_b32 %5, 9, %5\n v_lshlrev_b32 %6, 9, %6\n v_lshlrev_b32 %7, 9, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 23) {
            asm volatile("v_add_u32 %0, %0, %9\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %9\n v_add_u32 %3, %3, %9\n v_add_u32 %4, %4, %9\n v_add_u32 %5, %5, %9\n v_add_u32 %6, %6, %9\n v_add_u32 %7, %7, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 24) {
            asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 25) {
            asm volatile("v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\n v_bitop3_b32 %4, %4, %8, %9 bitop3:0x96\n v_bitop3_b32 %5, %5, %8, %9 bitop3:0x96\n v_bitop3_b32 %6, %6, %8, %9 bitop3:0x96\n v_bitop3_b32 %7, %7, %8, %9 bitop3:0x96\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 26) {
            asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 27) {
            asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 28) {
            asm volatile("v_lshl_add_u32 %0, %0, 3, %9\n v_lshl_add_u32 %1, %1, 3, %9\n v_lshl_add_u32 %2, %2, 3, %9\n v_lshl_add_u32 %3, %3, 3, %9\n v_lshl_add_u32 %4, %4, 3, %9\n v_lshl_add_u32 %5, %5, 3, %9\n v_lshl_add_u32 %6, %6, 3, %9\n v_lshl_add_u32 %7, %7, 3, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 29) {
            asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 30) {
            asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 31) {
            asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 32) {
            asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 33) {
            asm volatile("v_cmp_lt_f32 vcc, |%0|, %8\n v_cmp_lt_f32 vcc, |%1|, %8\n v_cmp_lt_f32 vcc, |%2|, %8\n v_cmp_lt_f32 vcc, |%3|, %8\n v_cmp_lt_f32 vcc, |%4|, %8\n v_cmp_lt_f32 vcc, |%5|, %8\n v_cmp_lt_f32 vcc, |%6|, %8\n v_cmp_lt_f32 vcc, |%7|, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 34) {
            asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n v_fract_f32 %4, %4\n v_fract_f32 %5, %5\n v_fract_f32 %6, %6\n v_fract_f32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 35) {
            asm volatile("v_ldexp_f32 %0, %0, %8\n v_ldexp_f32 %1, %1, %8\n v_ldexp_f32 %2, %2, %8\n v_ldexp_f32 %3, %3, %8\n v_ldexp_f32 %4, %4, %8\n v_ldexp_f32 %5, %5, %8\n v_ldexp_f32 %6, %6, %8\n v_ldexp_f32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 36) {
            asm volatile("v_lshrrev_b32 %0, 8, %0\n v_lshrrev_b32 %1, 8, %1\n v_lshrrev_b32 %2, 8, %2\n v_lshrrev_b32 %3, 8, %3\n v_lshrrev_b32 %4, 8, %4\n v_lshrrev_b32 %5, 8, %5\n v_lshrrev_b32 %6, 8, %6\n v_lshrrev_b32 %7, 8, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 37) {
            asm volatile("v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 38) {
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 39) {
            asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3\n v_sin_f32 %4, %4\n v_sin_f32 %5, %5\n v_sin_f32 %6, %6\n v_sin_f32 %7, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 6) {  // 4 x (fma with one SGPR operand), independent
            asm volatile("v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                         "v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "v"(b));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    unsigned long long c1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int MODE>
void run(const char *name, float *d, unsigned long long *dc) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    const int wlist[] = {8};
    for (int waves : wlist) {
        int blocks = cus * waves;  // one 256-thread block = 1 wave on each of the CU's 4 SIMDs
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, dc);
        hipEventRecord(a);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, dc);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        double insts_per_simd = 5.0 * waves * ITER * 8.0;  // wave-instructions issued on one SIMD
        double sec = ms * 1e-3;
        // in-kernel view: mean shader-clock ticks one wave spent, shared by `waves` waves on its SIMD
        std::vector<unsigned long long> hc(blocks * 4);
        hipMemcpy(hc.data(), dc, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mean = 0, mx = 0;
        for (auto v : hc) { mean += (double)v; if ((double)v > mx) mx = (double)v; }
        mean /= hc.size();
        printf("%-30s waves/SIMD=%d  %.3f wave-inst/ns/SIMD (events)  | s_memtime: mean %.0f max %.0f ticks per wave => %.2f ticks/inst/SIMD (max / insts on the SIMD); "
               "kernel %.1f us\n", name, waves, insts_per_simd / (sec * 1e9), mean, mx, mx / (double(ITER) * 8.0 * waves), ms * 1e3 / 5);
    }
}

int main() {
    float *d;
    unsigned long long *dc;
    hipMalloc(&d, 256 * 8 * 2048 * sizeof(float));
    hipMalloc(&dc, 256 * 8 * 4 * sizeof(unsigned long long));
    run<40>("v_perm_b32 (vgpr selector)", d, dc);
    run<45>("v_perm_b32 (sgpr selector)", d, dc);
    run<41>("v_and_or_b32", d, dc);
    run<42>("v_cvt_f32_u32_sdwa WORD_1", d, dc);
    run<43>("v_cvt_f32_ubyte1", d, dc);
    run<44>("v_cvt_f32_f16", d, dc);
    run<46>("v_fma_mix_f32 (half src0, lo)", d, dc);
    run<47>("v_fma_mix_f32 (-half src0, hi)", d, dc);
    run<0>("fma independent", d, dc);
    run<24>("v_cvt_f32_u32", d, dc);
    return 0;
    run<1>("fma dependent chain", d, dc);
    run<2>("mul/add independent", d, dc);
    run<3>("cmp+cndmask", d, dc);
    run<4>("rcp/sqrt", d, dc);
    run<5>("int xor/shift/cvt", d, dc);
    run<6>("fma chain w/ sgpr operand", d, dc);
    run<7>("v_pk_fma_f32 (8 inst/iter)", d, dc);
    run<8>("v_pk_mul/add_f32 (8 inst/iter)", d, dc);
    run<9>("v_cmp -> vcc", d, dc);
    run<10>("v_cmp (VOP3) -> sgpr pairs", d, dc);
    run<11>("v_min/v_max", d, dc);
    run<12>("v_min3/max3/med3", d, dc);
    run<13>("v_cndmask (vcc fixed)", d, dc);
    run<14>("VOP3 abs/neg modifiers", d, dc);
    run<15>("32-bit literal operands", d, dc);
    run<16>("v_fmac_f32 (VOP2)", d, dc);
    run<20>("v_xor_b32", d, dc);
    run<21>("v_alignbit_b32", d, dc);
    run<22>("v_lshlrev_b32", d, dc);
    run<23>("v_add_u32", d, dc);
    run<24>("v_cvt_f32_u32", d, dc);
    run<25>("v_bitop3_b32", d, dc);
    run<26>("v_and_b32", d, dc);
    run<27>("v_mov_b32", d, dc);
    run<28>("v_lshl_add_u32", d, dc);
    run<29>("v_mul_lo_u32", d, dc);
    run<30>("v_mul_u32_u24", d, dc);
    run<31>("v_mad_u32_u24", d, dc);
    run<32>("v_sub_f32", d, dc);
    run<33>("v_cmp |x|", d, dc);
    run<34>("v_fract_f32", d, dc);
    run<35>("v_ldexp_f32", d, dc);
    run<36>("v_lshrrev_b32", d, dc);
    run<37>("v_or_b32", d, dc);
    run<38>("v_exp_f32", d, dc);
    run<39>("v_sin_f32", d, dc);
    hipFree(d);
    return 0;
}
