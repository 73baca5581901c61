// pt_wfqueue.h — path-state queues of the wavefront pipelines (pt_wavefront.hip, pt_wavefront_bvh.hip).
#pragma once
#include "pt_device.h"

namespace pt {

// A queue carries the path state of its entries. Layout (round 3): TILES of 64 entries — one wave's worth — each
// holding all NF + NU planes of its entries back to back, plane k of entry i at word
//     (i / 64) * 64 * (NF + NU) + k * 64 + i % 64,
// so a wave reads or appends its 64 entries as NF + NU consecutive 256-byte rows of one 6 KB block. Rounds 1-2 kept
// whole planes (plane k at k * cap + i): with up to 2^27 entries per queue the 24 planes of one wave's entries lay
// 512 MB apart, i.e. 24 DRAM pages and 24 address translations per wave and kernel, and the logic / shading kernels
// of the LBVH pipeline streamed the queues at under a quarter of the HBM rate (profiles/r03_c4_*). Kernels read
// entry i with lane-consecutive i and append to output queues at base + rank-in-wave (coalesced per wave).
enum { F_RO = 0, F_RD = 3, F_T = 6, F_SUM = 9, F_X = 12, F_N = 15, NF = 18 };
enum { U_PIX = 0, U_RNG0, U_RNG1, U_META, U_NEXT, U_MAT, NU };
// U_META: bit 0 alive, bits 1-5 depth, bit 6 pending vertex is a scatter, bits 8.. media+1
// U_NEXT: next global sample index of this pixel; U_MAT: material of the pending vertex | (next_media+1) << 16
constexpr uint32_t kQueueTile = 64;
constexpr uint32_t kQueueWords = NF + NU;
// bytes of a queue of n entries
__host__ __device__ inline size_t queue_bytes(size_t n) { return ((n + kQueueTile - 1) / kQueueTile) * kQueueTile * kQueueWords * 4; }
struct WfQueue {
    float *f;          // the queue's words (float planes 0..NF-1, then the uint planes)
    uint32_t *count;
    uint32_t cap;
    __device__ __forceinline__ size_t at(int k, uint32_t i) const {
        return (size_t)(i / kQueueTile) * (kQueueTile * kQueueWords) + (uint32_t)k * kQueueTile + (i % kQueueTile);
    }
    __device__ __forceinline__ F3 ld3(int k, uint32_t i) const {
        const float *p = f + at(k, i);
        return F3{p[0], p[kQueueTile], p[2 * kQueueTile]};
    }
    __device__ __forceinline__ void st3(int k, uint32_t i, F3 v) const {
        float *p = f + at(k, i);
        p[0] = v.x;
        p[kQueueTile] = v.y;
        p[2 * kQueueTile] = v.z;
    }
    __device__ __forceinline__ uint32_t ldu(int k, uint32_t i) const { return __float_as_uint(f[at(NF + k, i)]); }
    __device__ __forceinline__ void stu(int k, uint32_t i, uint32_t v) const { f[at(NF + k, i)] = __uint_as_float(v); }
};

// Reserves queue positions for the lanes of a 256-thread block: wave ballot + prefix popcount,
// wave totals combined through LDS, ONE atomicAdd per block and queue (a per-wave atomic on the
// single queue counter serialises: 65k same-address atomics per launch took ~0.5 ms).
// Every thread of the block must call it. Returns this lane's position in queue 0 / queue 1.
__device__ __forceinline__ void queue_reserve2(bool want0, uint32_t *count0, uint32_t &pos0, bool want1, uint32_t *count1,
                                               uint32_t &pos1, uint32_t (*s_cnt)[4], uint32_t *s_base) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long m0 = __ballot(want0), m1 = __ballot(want1);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) {
        s_cnt[0][wave] = (uint32_t)__popcll(m0);
        s_cnt[1][wave] = (uint32_t)__popcll(m1);
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        uint32_t *cnt = threadIdx.x == 0 ? count0 : count1;
        uint32_t tot = s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1] + s_cnt[threadIdx.x][2] + s_cnt[threadIdx.x][3];
        s_base[threadIdx.x] = (tot && cnt) ? atomicAdd(cnt, tot) : 0u;
    }
    __syncthreads();
    uint32_t off0 = 0, off1 = 0;
    for (uint32_t w = 0; w < wave; ++w) {
        off0 += s_cnt[0][w];
        off1 += s_cnt[1][w];
    }
    pos0 = s_base[0] + off0 + (uint32_t)__popcll(m0 & below);
    pos1 = s_base[1] + off1 + (uint32_t)__popcll(m1 & below);
    __syncthreads();  // s_cnt / s_base are reused by the next trip
}

}  // namespace pt
