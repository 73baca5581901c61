// gather_microbench.hip — how fast can every lane of a wave chase its OWN pointer through a table of 16-byte rows?
// (the access pattern of the LBVH traversal kernels: one dependent 16-byte gather per lane and step). Rows come from a
// 3.2 MB table in global memory (L2-resident, partly L1), from a 32 KB table in LDS, or half and half.
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_microbench.hip -o tools/gather_microbench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

template <int MODE>  // 0: global, 1: LDS, 2: alternate
__global__ __launch_bounds__(256) void chase(const uint4 *__restrict__ tbl, uint32_t n, uint32_t n_lds, int steps, uint32_t *out) {
    __shared__ uint4 s_tbl[2048];  // 32 KB
    for (uint32_t i = threadIdx.x; i < 2048; i += 256) s_tbl[i] = tbl[i];
    __syncthreads();
    uint32_t idx = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        uint4 v;
        bool lds = MODE == 1 || (MODE == 2 && (s & 1));
        if (lds) v = s_tbl[idx % n_lds];
        else v = tbl[idx % n];
        acc += v.y;
        idx = v.x + acc;  // dependent
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const uint32_t n = 200000;  // 3.2 MB of 16-byte rows
    std::vector<uint4> h(n);
    uint32_t x = 12345;
    for (auto &r : h) {
        x = x * 1664525u + 1013904223u; r.x = x;
        x = x * 1664525u + 1013904223u; r.y = x;
        r.z = r.w = 0;
    }
    uint4 *d; uint32_t *o;
    hipMalloc(&d, n * 16); hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    const int blocks = 256 * 8, steps = 2000;
    hipMalloc(&o, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(blocks), dim3(256), 0, 0, d, n, 2048u, steps, o);
            if (mode == 1) hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(256), 0, 0, d, n, 2048u, steps, o);
            if (mode == 2) hipLaunchKernelGGL(chase<2>, dim3(blocks), dim3(256), 0, 0, d, n, 2048u, steps, o);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            double loads = (double)blocks * 256 * steps;
            if (rep) printf("mode %d (%s): %.2f ms, %.1f G lane-loads/s, %.3f lane-loads per clock per CU (2.4 GHz, 256 CUs)\n", mode,
                            mode == 0 ? "global 3.2 MB" : mode == 1 ? "LDS 32 KB" : "alternating", ms, loads / ms / 1e6,
                            loads / (ms * 1e-3) / (2.4e9 * 256));
        }
    }
    return 0;
}
