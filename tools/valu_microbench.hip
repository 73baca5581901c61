// valu_microbench.hip — measures the f32 VALU issue rate of gfx950 that bounds the trace kernel:
// wave-instructions per cycle per SIMD for independent / dependent v_fma_f32, v_pk_fma_f32,
// compare+select and transcendental streams at 1..8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o gpurun_out/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, float b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITER; ++i) {
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
        } else if (MODE == 6) {  // 4 x (fma with one SGPR operand), independent
            asm volatile("v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                         "v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a), "v"(b));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
void run(const char *name, float *d) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    for (int waves = 1; waves <= 8; waves *= 2) {
        int blocks = cus * waves;  // one 256-thread block = 1 wave on each of the CU's 4 SIMDs
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
        hipEventRecord(a);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        double insts_per_simd = 5.0 * waves * ITER * 8.0;  // wave-instructions issued on one SIMD
        double sec = ms * 1e-3;
        printf("%-28s waves/SIMD=%d  %.3f wave-inst/ns/SIMD  => %.2f cycles/inst at 2.4 GHz  (%.1f Ginst/s chip)\n", name, waves,
               insts_per_simd / (sec * 1e9), 2.4 / (insts_per_simd / (sec * 1e9)), insts_per_simd * cus * 4 / sec / 1e9);
    }
}

int main() {
    float *d;
    hipMalloc(&d, 256 * 8 * 2048 * sizeof(float));
    run<0>("fma independent", d);
    run<1>("fma dependent chain", d);
    run<2>("mul/add independent", d);
    run<3>("cmp+cndmask", d);
    run<4>("rcp/sqrt", d);
    run<5>("int xor/shift/cvt", d);
    run<6>("fma chain w/ sgpr operand", d);
    run<7>("v_pk_fma_f32 (8 inst/iter)", d);
    run<8>("v_pk_mul/add_f32 (8 inst/iter)", d);
    hipFree(d);
    return 0;
}
