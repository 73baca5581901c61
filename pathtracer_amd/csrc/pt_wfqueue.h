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
    // a uint kept in one of the float planes (F_SUM is free in the LBVH pipeline's shade entries)
    __device__ __forceinline__ uint32_t ldw(int k, uint32_t i) const { return __float_as_uint(f[at(k, i)]); }
    __device__ __forceinline__ void stw(int k, uint32_t i, uint32_t v) const { f[at(k, i)] = __uint_as_float(v); }
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

// ---- appends through per-block slabs (LBVH pipeline) ---------------------------------------------------------------
// queue_reserve2 costs one atomic on the queue's counter per block trip, i.e. per 256 entries, and same-address
// atomics complete one every ~8-9 ns chip-wide: with 2^27 entries per round the logic kernel's two counters alone
// were ~150 ms of a C4 step (profiles/r03_c4_wide64_*: the queue kernels streamed at < 2 TB/s). Here a block takes
// queue positions a SLAB at a time (one atomic per slab of 256..2048 entries) and hands them out from LDS; a trip
// that outgrows the current slab fills its tail and continues in a new one, so the only unused positions are the
// tail of each block's last slab, which the block marks dead (kill_tail) before it leaves: consumers skip entries
// whose alive bit is clear. A queue's `count` is therefore its number of POSITIONS, a few per cent above its entries.
struct SlabState {
    uint32_t pos[3], end[3];      // the block's current slab per output queue ([2]: the shadow-ray id list)
    uint32_t a[3], split[3], b[3];  // this trip: ranks below split sit at a + rank, the others at b + rank - split
    uint32_t cnt[3][4];
};
constexpr uint32_t kSlabMin = 256, kSlabMax = 2048;
// slab size for a kernel whose input queue holds `count_in` positions: about an eighth of a block's share
__host__ __device__ inline uint32_t slab_size(uint32_t count_in, uint32_t blocks) {
    uint32_t s = kSlabMin;
    while (s < kSlabMax && (uint64_t)s * 2u * blocks * 8u <= count_in) s *= 2u;
    return s;
}
// positions a queue may need beyond its entries: every block of the (at most two) kernels that append to it leaves one slab tail
// (a round's input may itself hold up to 3 n positions, hence the slab size of 3 n)
__host__ __device__ inline size_t slab_slack(size_t n, uint32_t blocks) {
    return 2 * (size_t)blocks * slab_size((uint32_t)(n * 3 < 0xffffffffull ? n * 3 : 0xffffffffull), blocks);
}

// slab of the shadow-ray list (list_reserve_slab): a block trip appends up to 256 x 4 L ids
__host__ __device__ inline uint32_t ids_slab_size(uint32_t slab, uint32_t n_light) {
    return slab * 4u > 1024u * n_light ? slab * 4u : 1024u * (n_light ? n_light : 1u);
}
__device__ __forceinline__ void slab_init(SlabState *S) {
    if (threadIdx.x < 3) S->pos[threadIdx.x] = S->end[threadIdx.x] = 0u;
    __syncthreads();
}
// Third allocator, for lists in which a lane appends n >= 0 consecutive items (the shadow-ray ids of its vertex):
// returns the position of the lane's first item; item j of the lane sits at slab_item(S, first + j).
// `slab` must be at least the largest total a trip can ask for. Every thread of the block must call it.
__device__ __forceinline__ uint32_t list_reserve_slab(uint32_t n, uint32_t *count, SlabState *S, uint32_t slab) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = n;  // inclusive prefix sum over the wave
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += v;
    }
    if (lane == 63u) S->cnt[2][wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = S->cnt[2][0] + S->cnt[2][1] + S->cnt[2][2] + S->cnt[2][3];
        const uint32_t pos = S->pos[2], room = S->end[2] - pos;
        S->a[2] = pos;
        if (tot <= room) {
            S->split[2] = tot;
            S->b[2] = 0u;
            S->pos[2] = pos + tot;
        } else {
            const uint32_t nb = atomicAdd(count, slab);
            S->split[2] = room;
            S->b[2] = nb;
            S->pos[2] = nb + (tot - room);
            S->end[2] = nb + slab;
        }
    }
    __syncthreads();
    uint32_t first = incl - n;
    for (uint32_t w = 0; w < wave; ++w) first += S->cnt[2][w];
    return first;  // a RANK within this trip's items of the block; slab_item() turns ranks into positions
}
__device__ __forceinline__ uint32_t slab_item(const SlabState *S, uint32_t rank) {
    return rank < S->split[2] ? S->a[2] + rank : S->b[2] + (rank - S->split[2]);
}
// Every thread of the block must call it. Returns this lane's position in queue 0 / queue 1.
__device__ __forceinline__ void queue_reserve_slab(bool want0, uint32_t *count0, uint32_t &pos0, bool want1, uint32_t *count1,
                                                   uint32_t &pos1, SlabState *S, uint32_t slab) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long m0 = __ballot(want0), m1 = __ballot(want1);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) {
        S->cnt[0][wave] = (uint32_t)__popcll(m0);
        S->cnt[1][wave] = (uint32_t)__popcll(m1);
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const uint32_t q = threadIdx.x;
        uint32_t *cnt = q == 0 ? count0 : count1;
        const uint32_t tot = S->cnt[q][0] + S->cnt[q][1] + S->cnt[q][2] + S->cnt[q][3];
        const uint32_t pos = S->pos[q], room = S->end[q] - pos;
        S->a[q] = pos;
        if (tot <= room || !cnt) {
            S->split[q] = tot;
            S->b[q] = 0u;
            S->pos[q] = pos + tot;
        } else {  // tot <= 256 <= slab: one new slab is enough
            const uint32_t nb = atomicAdd(cnt, slab);
            S->split[q] = room;
            S->b[q] = nb;
            S->pos[q] = nb + (tot - room);
            S->end[q] = nb + slab;
        }
    }
    __syncthreads();
    uint32_t r0 = (uint32_t)__popcll(m0 & below), r1 = (uint32_t)__popcll(m1 & below);
    for (uint32_t w = 0; w < wave; ++w) {
        r0 += S->cnt[0][w];
        r1 += S->cnt[1][w];
    }
    pos0 = r0 < S->split[0] ? S->a[0] + r0 : S->b[0] + (r0 - S->split[0]);
    pos1 = r1 < S->split[1] ? S->a[1] + r1 : S->b[1] + (r1 - S->split[1]);
    __syncthreads();  // the trip's fields are reused by the next trip
}
// The unused tail of the block's last slab of output queue q: alive bit clear (the logic / shading kernels skip the
// entry), zero direction (the closest-hit traversal skips it) and zero normal (every light sample of the any-hit
// traversal "faces away" and is dropped).
__device__ __forceinline__ void kill_tail(const WfQueue &Q, const SlabState *S, int q) {
    for (uint32_t i = S->pos[q] + threadIdx.x; i < S->end[q]; i += blockDim.x) {
        Q.stu(U_META, i, 0u);
        Q.st3(F_RD, i, F3{0.0f, 0.0f, 0.0f});
        Q.st3(F_N, i, F3{0.0f, 0.0f, 0.0f});
    }
}

}  // namespace pt
