// gather_pair_microbench.hip — the node fetch of the LBVH traversal: every lane chases its own pointer through a table
// of 32-byte nodes. Variant A: each lane loads its node's two 16-byte halves itself (two instructions, 128 distinct
// 32-byte blocks per wave and visit). Variant B: lanes work in pairs — one instruction fetches the even lane's node
// (its two halves by the two lanes, 32 contiguous bytes per pair), the next one the odd lane's node, and the halves are
// swapped with DPP. Same bytes, half as many distinct blocks per instruction.
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_pair_microbench.hip -o tools/gather_pair_microbench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

__device__ __forceinline__ uint32_t swap1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true); }

template <int MODE>
__global__ __launch_bounds__(256) void chase(const uint4 *__restrict__ tbl, uint32_t n, int steps, uint32_t *out) {
    uint32_t idx = ((blockIdx.x * 256 + threadIdx.x) * 2654435761u) % n;
    uint32_t acc = 0;
    const uint32_t odd = threadIdx.x & 1u;
    for (int s = 0; s < steps; ++s) {
        uint4 a, b;
        if (MODE == 0 || MODE == 2) {
            a = tbl[idx * 2 + 0], b = tbl[idx * 2 + 1];
        } else {
            const uint32_t other = swap1(idx);
            const uint32_t ie = odd ? other : idx, io = odd ? idx : other;
            const uint4 r1 = tbl[ie * 2 + odd];  // the even lane's node, half = lane parity
            const uint4 r2 = tbl[io * 2 + odd];  // the odd lane's node
            // even lane: own half 0 = r1, needs own half 1 = the odd lane's r1; odd lane: own half 1 = r2, needs half 0 = the even lane's r2
            const uint4 send = odd ? r1 : r2;
            uint4 got{swap1(send.x), swap1(send.y), swap1(send.z), swap1(send.w)};
            a = odd ? got : r1;
            b = odd ? r2 : got;
        }
        acc += a.y ^ b.z;
        idx = (a.x + b.w + acc) % n;  // dependent
        if (MODE == 2 && (s & 1)) idx = ((a.x & 3u) == (idx & 3u) ? idx ^ 1u : (idx & ~3u) | (a.x & 3u)) % n;  // every other visit stays in the 128-byte line
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const uint32_t sizes[3] = {256, 4096, 100000};  // 8 KB (L1), 128 KB, 3.2 MB (L2) of 32-byte nodes
    for (uint32_t n : sizes) {
        std::vector<uint4> h(2 * n);
        uint32_t x = 12345;
        for (auto &r : h) {
            x = x * 1664525u + 1013904223u; r.x = x;
            x = x * 1664525u + 1013904223u; r.y = x;
            r.z = x >> 3; r.w = x >> 7;
        }
        uint4 *d; uint32_t *o;
        hipMalloc(&d, 2 * n * 16); hipMemcpy(d, h.data(), 2 * n * 16, hipMemcpyHostToDevice);
        const int blocks = 256 * 6, steps = 2000;
        hipMalloc(&o, blocks * 256 * 4);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        uint32_t chk[3] = {0, 0, 0};
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(blocks), dim3(256), 0, 0, d, n, steps, o);
                if (mode == 1) hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(256), 0, 0, d, n, steps, o);
                if (mode == 2) hipLaunchKernelGGL(chase<2>, dim3(blocks), dim3(256), 0, 0, d, n, steps, o);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                double visits = (double)blocks * 256 * steps;
                if (rep) printf("%7u nodes, %s: %.2f ms, %.3f node visits per clock per CU (2.4 GHz, 256 CUs)\n", n,
                                mode == 0 ? "own halves   " : mode == 1 ? "pair + DPP   " : "own halves, every other visit in the same line", ms, visits / (ms * 1e-3) / (2.4e9 * 256));
            }
            std::vector<uint32_t> ho(blocks * 256);
            hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost);
            for (uint32_t v : ho) chk[mode] = chk[mode] * 31u + v;
        }
        printf("   results %s\n", chk[0] == chk[1] ? "equal" : "DIFFERENT");
        hipFree(d); hipFree(o);
    }
    return 0;
}
