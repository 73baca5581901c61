// gather_occupancy_microbench.hip — is the per-lane 32-byte node fetch (LBVH traversal) bound by latency or by a rate?
// Every lane chases its own pointer through 100 000 nodes of 32 bytes (3.2 MB, L2-resident); the launch keeps 1..8
// blocks of 256 threads resident per CU (LDS padding limits residency). If node visits per clock grow with the number
// of resident waves the fetch is latency-bound, if they saturate it is a rate limit of the L1 / L2 path.
// LOADS: 0 = plain, 1 = nontemporal (slc), 2 = 16 bytes only (first half), 3 = both halves through one 128-bit load each but volatile glc-like (__builtin_amdgcn_raw_buffer_load with glc)
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_occupancy_microbench.hip -o tools/gather_occupancy_microbench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

template <int LOADS>
__global__ __launch_bounds__(256) void chase(const uint4 *__restrict__ tbl, uint32_t n, int steps, uint32_t *out) {
    extern __shared__ uint32_t pad[];
    uint32_t idx = ((blockIdx.x * 256 + threadIdx.x) * 2654435761u) % n;
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        uint4 a, b;
        if (LOADS == 1) {
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            const v4u *p = reinterpret_cast<const v4u *>(tbl) + idx * 2;
            v4u va = __builtin_nontemporal_load(p), vb = __builtin_nontemporal_load(p + 1);
            a = uint4{va.x, va.y, va.z, va.w}, b = uint4{vb.x, vb.y, vb.z, vb.w};
        } else if (LOADS == 2) {
            a = tbl[idx * 2 + 0];
            b = a;
        } else {
            a = tbl[idx * 2 + 0], b = tbl[idx * 2 + 1];
        }
        acc += a.y ^ b.z;
        idx = (a.x + b.w + acc) % n;
    }
    if (acc == 0x12345678u) pad[threadIdx.x] = acc;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const uint32_t n = 100000;
    std::vector<uint4> h(2 * n);
    uint32_t x = 12345;
    for (auto &r : h) {
        x = x * 1664525u + 1013904223u; r.x = x;
        x = x * 1664525u + 1013904223u; r.y = x;
        r.z = x >> 3; r.w = x >> 7;
    }
    uint4 *d; uint32_t *o;
    (void)hipMalloc(&d, 2 * n * 16); (void)hipMemcpy(d, h.data(), 2 * n * 16, hipMemcpyHostToDevice);
    const int steps = 2000;
    (void)hipMalloc(&o, 256 * 8 * 256 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipFuncSetAttribute((const void *)chase<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)chase<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)chase<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int loads = 0; loads < 3; ++loads)
        for (int per_cu : {1, 2, 4, 6, 8}) {
            const int blocks = 256 * per_cu;
            const size_t lds = (size_t)(160 * 1024 / per_cu) - 1024;  // so that exactly per_cu blocks fit
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(a);
                if (loads == 0) hipLaunchKernelGGL(chase<0>, dim3(blocks), dim3(256), lds, 0, d, n, steps, o);
                if (loads == 1) hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(256), lds, 0, d, n, steps, o);
                if (loads == 2) hipLaunchKernelGGL(chase<2>, dim3(blocks), dim3(256), lds, 0, d, n, steps, o);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                (void)hipEventElapsedTime(&ms, a, b);
            }
            double visits = (double)blocks * 256 * steps;
            printf("%s, %d waves per CU: %.2f ms, %.3f node visits per clock per CU, %.0f clocks per dependent visit\n",
                   loads == 0 ? "plain 2 x 16 B" : loads == 1 ? "nontemporal   " : "16 B only     ", per_cu * 4, ms,
                   visits / (ms * 1e-3) / (2.4e9 * 256), ms * 1e-3 * 2.4e9 / steps);
        }
    return 0;
}
