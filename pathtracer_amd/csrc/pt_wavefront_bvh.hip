// pt_wavefront_bvh.hip — queue-based pipeline for many-sphere scenes (BASELINE config C4).
//
// On the 7-object scenes the megakernel wins. With 100 000 spheres behind an LBVH the
// picture turns: traversal lengths are long-tailed (a ray skimming the ground visits hundreds of nodes,
// its wave neighbours a dozen), and a lane that owns a path must sit through its wave's longest
// traversal — rocprofv3 counts 8 of 64 lanes active per VALU instruction in the BVH megakernel
// (profiles/r01_c4_lbvh_pmc_summary.txt). Here rays live in queues and the two traversal kernels are
// PERSISTENT: a lane that finishes its ray takes the next one from the queue (wave-aggregated ranges,
// one atomic per 256 rays), so a wave is refilled while it runs, and the traversal kernels are lean
// enough for 8 waves per SIMD. Per round:
//
//   wfb_traverse<false> : closest sphere through the LBVH for every entry of the ray queue
//   wfb_logic           : planes + winner refinement (pt_device.h: refine_hit), medium free flight,
//                         classification; glass / miss handled here; opaque / scatter vertices go to the
//                         shade queue; a finished path is replaced by the pixel's next camera path
//   wfb_traverse<true>  : any sphere hit for the 4 L light samples of every shade-queue entry (renderer.rs:35-46):
//                         the lane that takes a sample regenerates it from the entry's stream state, drops it if
//                         it faces away, traces it otherwise (there is no sampling kernel and no ray list)
//   wfb_shade           : plane occlusion, BRDF evaluation (renderer.rs:49-82), Russian roulette, lobe
//                         sampling (renderer.rs:323-349) or phase-function scatter (:269-286)
//
// SAMPLE STREAMS. A queue entry is a (pixel, stream) pair: the samples of a launch are cut into 2^k contiguous parts
// per pixel (P.wf_streams_log2, chosen by the host so that up to 2^28 paths are in flight) and every part is a stream with
// its own entry and its own 12 bytes of a partial-sum buffer that hold its running sum; a pixel's sums are added up in a
// fixed order at the end.
// Why: a round is five dependent launches, and each traversal launch lasts at least as long as its LONGEST ray — a
// ray that skims the sphere slab visits several hundred nodes at ~1 us per dependent visit, as long as the other two
// million rays of a 1080p round take together. Measured with one stream per pixel: 160 / 364 / 495 / 562 Mpaths/s at
// 0.5 / 2 / 8 / 33 M pixels, i.e. 43 % of a 1080p round was that fixed cost. More paths per round amortise it, and
// there are 2^k times fewer rounds.
//
// The arithmetic is the device code of the megakernels (pt_device.h, pt_bvh.h) with the same RNG stream
// per (pixel, sample) and the same draw order, so the image equals the BVH megakernel's.
#include <cstdlib>

#include "pt_kernels.h"
#include "pt_trace.h"
#include "pt_wfqueue.h"

namespace pt {

namespace {

#ifndef WFB_FETCH_MIN
#define WFB_FETCH_MIN 16
#endif
#ifndef WFB_RESERVE
#define WFB_RESERVE 2048
#endif
constexpr uint32_t kFetchChunk = 256;          // rays of one range of a wave (one slot of result bits)
// Rays a wave reserves with ONE atomic on the launch's fetch head, consumed range by range. Same-address atomics
// complete one every ~8 ns chip-wide, so with 256 rays per atomic no traversal launch could take more than ~33 G rays
// per second however fast the traversal itself was, and the shadow-ray launches (4 L "rays" per vertex, a third of
// them dropped at once) ran close to that.
constexpr uint32_t kReserve = WFB_RESERVE;
constexpr uint32_t kFetchMin = WFB_FETCH_MIN;  // idle lanes that trigger a refill (C4, final pipeline: 4: -12 %, 8: -4 %, 16: best, 24: -1 %)

// Entry id v = pixel << k | stream. Stream j of 2^k renders the j-th part of the launch's samples (parts differ by at
// most one sample; the host keeps 2^k <= sample_count, so no part is empty).
__device__ __forceinline__ void stream_range(const Params &P, uint32_t v, uint32_t &begin, uint32_t &end) {
    const uint32_t k = 1u << P.wf_streams_log2, j = v & (k - 1u);
    const uint32_t q = P.sample_count >> P.wf_streams_log2, r = P.sample_count & (k - 1u);
    begin = P.sample_begin + j * q + min(j, r);
    end = begin + q + (j < r ? 1u : 0u);
}
// A stream's running sum lives in the partial-sum buffer (zeroed by the host before the launch), not in
// the queue entries: only a vertex that receives light touches it (read-modify-write of 12 bytes; a stream has one
// live entry at a time, so there is no race, and the additions happen in the same order as a carried sum's would).
// The sums are laid out by entry id (pixel-major, the streams of a pixel side by side): the queues keep the entries
// roughly in id order, so the 12-byte updates of neighbouring lanes fall into the same lines. (Stream-major, as the
// megakernel's slices are, every lane touched a line of its own — 25 MB apart — and wfb_shade moved 250 GB per C4 step
// through HBM for 12 GB of updates.)
__device__ __forceinline__ void stream_add(const Params &, float *__restrict__ out, uint32_t v, F3 c) {
    float *o = out + (size_t)v * 3;
    o[0] += c.x, o[1] += c.y, o[2] += c.z;
}
// accum[pixel] += stream 0 + stream 1 + ... of the pixel, in that order (as pt_add_partials adds the megakernel's slices)
__global__ __launch_bounds__(256) void wfb_add_streams(float *__restrict__ accum, const float *__restrict__ sums, uint32_t streams_log2,
                                                       size_t n_floats) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n_floats) return;
    const size_t pix = i / 3u, c = i - pix * 3u;
    const float *p = sums + ((pix << streams_log2) * 3u + c);
    float v = p[0];
    for (uint32_t j = 1; j < (1u << streams_log2); ++j) v += p[(size_t)j * 3u];
    accum[i] += v;
}

// renderer.rs:170-178: the pixel's next camera path
__device__ __forceinline__ void new_camera_path(const TblGlobal &tbl, const Params &P, uint32_t seed_mixed, uint32_t v, uint32_t &next, Rng &rng,
                                                F3 &ro, F3 &rd, F3 &T, uint32_t &depth, int &media) {
    const uint32_t pix = v >> P.wf_streams_log2;
    uint32_t px = pix % P.width, py = pix / P.width;
    rng.seed(seed_mixed, mix32(pix + seed_mixed), next);
    float jx = rng.gen(), jy = rng.gen(), ds = rng.gen(), dt = rng.gen();
    camera_ray(tbl, P, px, py, jx, jy, ds, dt, ro, rd);
    T = F3{1, 1, 1};
    depth = 0;
    media = -1;
    ++next;
}

__device__ __forceinline__ void push_ray(const WfQueue &Q, uint32_t pos, uint32_t pix, uint32_t depth, int media, uint32_t next,
                                         F3 ro, F3 rd, F3 T, const Rng &rng) {
    Q.stu(U_PIX, pos, pix);
    Q.stu(U_META, pos, 1u | (depth << 1) | ((uint32_t)(media + 1) << 8));
    Q.stu(U_NEXT, pos, next);
    Q.st3(F_RO, pos, ro);
    Q.st3(F_RD, pos, rd);
    Q.st3(F_T, pos, T);
    Q.stu(U_RNG0, pos, rng.s0);
    Q.stu(U_RNG1, pos, rng.s1);
}

}  // namespace

// One light sample of renderer.rs:35-44 from its two uniforms: unit direction, distance limit of the shadow ray
// (|lvec| sqrt(0.999), renderer.rs:46) and 1 / dist^2. wfb_traverse<ANY> (for the shadow rays) and wfb_shade (for the
// evaluation) both call it with the same uniforms instead of passing the samples through HBM; the operations
// are spelt out so that both kernels round alike.
__device__ __forceinline__ void light_sample(float4 l0, float4 l1, float4 l2, F3 x, float e1, float e2, F3 &l, float &tmax, float &inv_d2) {
    const F3 lvec = F3{fmaf(l2.x, e2, fmaf(l1.x, e1, l0.x)) - x.x, fmaf(l2.y, e2, fmaf(l1.y, e1, l0.y)) - x.y,
                       fmaf(l2.z, e2, fmaf(l1.z, e1, l0.z)) - x.z};
    const float dist2 = fmaf(lvec.z, lvec.z, fmaf(lvec.y, lvec.y, lvec.x * lvec.x));
    const float inv = rsq(dist2);
    l = F3{lvec.x * inv, lvec.y * inv, lvec.z * inv};
    tmax = (dist2 * inv) * 0.99949987493746f;
    inv_d2 = inv * inv;
}

// renderer.rs:49-55: a light sample contributes only if the surface faces it and it faces the surface. wfb_logic
// (which lists the shadow rays to trace) and wfb_shade (which looks their results up) must decide alike.
__device__ __forceinline__ bool sample_faces(F3 nn, F3 l, float4 l4) { return (dot(nn, l) > 0.0f) & (-dot(l, f3(l4)) > 0.0f); }
// The 4 L light samples of a vertex in draw order (renderer.rs:35-39): f(k, l, tmax, inv_d2, faces) for k = 0 .. 4L-1.
// `rng` advances by the 8 L draws.
template <class F>
__device__ __forceinline__ void for_each_light_sample(const TblGlobal &tbl, const Params &P, Rng &rng, F3 x, F3 nn, F f) {
    for (uint32_t li = 0; li < P.n_light; ++li) {
        const uint32_t o = P.light_off + li * LIGHT_F4;
        const float4 l0 = tbl.ld(o), l1 = tbl.ld(o + 1), l2 = tbl.ld(o + 2), l4 = tbl.ld(o + 4);
#pragma unroll
        for (uint32_t s = 0; s < 4u; ++s) {
            const float e1 = rng.gen() - 0.5f;
            const float e2 = rng.gen() - 0.5f;
            F3 l;
            float tmax, inv_d2;
            light_sample(l0, l1, l2, x, e1, e2, l, tmax, inv_d2);
            f(li * 4u + s, l, tmax, inv_d2, sample_faces(nn, l, l4));
        }
    }
}

__device__ __forceinline__ uint32_t slab_item_of(uint32_t a, uint32_t split, uint32_t b, uint32_t rank) {
    return rank < split ? a + rank : b + (rank - split);
}

// Every stream starts with the camera path of its first sample; entries of the ray queue are always live rays.
// The streams of a launch are dealt to 2^group_bits groups (the host's concurrent pipelines); group g owns the streams
// whose top group_bits bits are g, and entry j of its queue is (pixel, stream) = (j >> (k - group_bits), g's j-th stream).
__global__ __launch_bounds__(256) void wfb_init(Params P, const float4 *__restrict__ blob, WfQueue Q, uint32_t n, uint32_t group_bits,
                                                uint32_t group, uint32_t *__restrict__ jump) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0) {  // LCG jump-ahead constants of the light samples (wfb_traverse<ANY>): after j draws the state is A_j s + C_j inc
        uint32_t A = 1u, C = 0u;
        for (uint32_t k = 0; k < 4u * P.n_light; ++k) {
            jump[2 * k] = A, jump[2 * k + 1] = C;
            for (int d = 0; d < 2; ++d) C = C * 2891336453u + 1u, A *= 2891336453u;
        }
    }
    if (i >= n) return;
    TblGlobal tbl{blob};
    const uint32_t seed_mixed = mix32(P.seed ^ 0x85ebca6bu);
    const uint32_t low_bits = P.wf_streams_log2 - group_bits;
    const uint32_t v = ((i >> low_bits) << P.wf_streams_log2) | (group << low_bits) | (i & ((1u << low_bits) - 1u));
    uint32_t next, s_end, depth;
    stream_range(P, v, next, s_end);
    int media;
    Rng rng{0u, 1u};
    F3 ro, rd, T;
    new_camera_path(tbl, P, seed_mixed, v, next, rng, ro, rd, T, depth, media);
    push_ray(Q, i, v, depth, media, next, ro, rd, T, rng);
    if (i == 0) *Q.count = n;
}

// Persistent traversal of the 4-wide tree (pt_bvh.h: "Wide nodes") over a ray array. ANY = false: closest sphere of
// ray i = entry i of Q (same result as pt_bvh.h: bvh_closest); out: hit_leaf[i] = sorted position of the sphere, -1:
// none (wfb_logic forms t again from the sphere, so the 4 bytes are all a ray writes). ANY = true: shadow ray r is
// light sample k = r % 4L of shade-queue entry e = r / 4L — vertex-major, so the samples of one vertex are
// neighbouring lanes that walk the same nodes (+5.3 % on C4 over a sample-major order) — origin Q.F_X[e] + 1e-4 l,
// direction l and tmax from light_sample(); out: bit r of occ_bits (same result as bvh_any; 1 also for samples that
// need no ray).
//
// A lane works on ITEMS: a key [entry distance, truncated | unified index] names a wide node or a sphere. A trip
// fetches the item in hand — 64 bytes of node (LDS for the first n_top nodes of the breadth-first array) or 16
// bytes of sphere, all loads of a trip issued before the first is used — and then either tests the sphere or tests
// the node's four child boxes, sorts the children it enters by distance (7 integer min / max / med3 on the keys),
// stacks all but the nearest and goes on with that one; when nothing is entered it pops, skipping (closest) every
// stacked item that begins beyond the best hit so far. The truncated distance in a key always rounds DOWN, so
// skipping is conservative, and boxes are entered inclusively (t <= best), so a sphere that ties is still tested
// and the scene-order rule of renderer.rs:359-361 decides.
//
// RESULTS ARE PACKED (round 3). A wave reserves rays 256 at a time; for shadow rays it keeps, per reserved range, 256
// result bits and a count of unfinished rays in LDS (kSlots ranges at a time) and writes the 32 bytes once, when the
// range is complete — round 2 stored one byte per ray when the ray ended, scattered over 4 L planes, and the
// traversal kernels wrote 5x their algorithmic output to HBM (partial lines evicted again and again).
//
// STATS: also counts what the rays fetch (stats[0..3] = rays, wide nodes visited (64 B each), sphere fetches (16 B
// each), aux rows read on exact ties (16 B)), wave-reduced, for the memory roofline of bench.py --config c4.
#ifndef WFB_INFLIGHT_LOG2
#define WFB_INFLIGHT_LOG2 28  // paths in flight the host aims for (sample streams per pixel); C4: 2^24 -5 %, 2^25 0, 2^26 +3.2 %, 2^27 +5.3 %, 2^28 +7 %
#endif
#ifndef WFB_MAX_STREAMS_LOG2
#define WFB_MAX_STREAMS_LOG2 7u  // at most 128 sample streams per pixel
#endif
#ifndef WFB_QUEUE_CAP_GIB
#define WFB_QUEUE_CAP_GIB 96  // queues of at most this many GiB of the GPU's 288 (C4 at 1080p: 75 GiB for 2^28 entries; hippt_api.hip falls back to fewer streams when the allocation fails)
#endif
#ifndef WFB_TRAV_THREADS
#define WFB_TRAV_THREADS 256
#endif
#ifndef WFB_WIDE_TOP
#define WFB_WIDE_TOP 85  // wide nodes in LDS: four full levels
#endif
constexpr uint32_t kTravThreads = WFB_TRAV_THREADS;  // traversal blocks: one LDS copy of the tree's top per 4 waves
constexpr uint32_t kWideTop = WFB_WIDE_TOP;
constexpr uint32_t kSlots = 4;        // shadow rays: 256-ray ranges a wave may have unfinished at a time
constexpr uint32_t kSlotWords = 12;   // 8 words of result bits, rays left, first ray of the range, 2 unused
constexpr uint32_t kSlotFree = 0xffffffffu;
constexpr uint32_t kSlotBytes = (kTravThreads / 64u) * kSlots * kSlotWords * 4u;
constexpr uint32_t kKeyNone = 0xffffffffu;
#ifndef WFB_TRAV_MIN_BLOCKS
#define WFB_TRAV_MIN_BLOCKS 1
#endif
template <bool ANY, bool STATS = false>
__global__ __launch_bounds__(kTravThreads, WFB_TRAV_MIN_BLOCKS) void wfb_traverse(Bvh bvh, WfQueue Q, const float4 *__restrict__ blob,
                                                    const uint32_t *__restrict__ jump,
                                                    uint32_t *__restrict__ head,
                                                    int *__restrict__ hit_leaf,
                                                    uint32_t *__restrict__ occ_bits, const uint32_t *__restrict__ ray_ids,
                                                    const uint32_t *__restrict__ ray_count, unsigned long long *__restrict__ stats,
                                                    uint32_t lds_entries, uint32_t *__restrict__ spill, uint32_t idx_bits, uint32_t n_top,
                                                    uint32_t light_off = 0, uint32_t n_light = 0) {
    uint32_t st_rays = 0, st_nodes = 0, st_leaves = 0, st_aux = 0, st_top = 0;
    extern __shared__ uint4 s_top[];  // the first n_top wide nodes (4 uint4 each), the waves' range slots, then lds_entries x kTravThreads stack words
    for (uint32_t i = threadIdx.x; i < 4u * n_top; i += kTravThreads) s_top[i] = bvh.wide[i];
    uint32_t *const slots_all = reinterpret_cast<uint32_t *>(s_top + 4u * n_top);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *const wsa = slots_all + (threadIdx.x >> 6) * (kSlots * kSlotWords);  // this wave's slots
    volatile uint32_t *const ws = wsa;
    if (lane < kSlots) {
        ws[lane * kSlotWords + 8] = 0u;
        ws[lane * kSlotWords + 9] = kSlotFree;
    }
    __syncthreads();
    uint32_t *stack = slots_all + kSlotBytes / 4u + threadIdx.x;
    const uint32_t stride = kTravThreads;
    // Stack entries beyond the first lds_entries live in this thread's column of a global spill area. The host sizes the
    // LDS part so that 8 blocks fit a CU beside the copy of the tree's first levels — few rays defer more items than
    // that, so the spill path costs little.
    uint32_t *const spill_col = spill + (size_t)blockIdx.x * kTravThreads + threadIdx.x;
    const size_t spill_stride = (size_t)gridDim.x * kTravThreads;
    const uint32_t idx_mask = (1u << idx_bits) - 1u;
    // ANY: the "rays" are the 4 L light samples of every shade-queue entry, vertex-major (sample k of entry e is ray
    // e * 4 L + k): a lane that takes one regenerates the sample from the entry's stream state (LCG jump-ahead by 2 k
    // draws, table `jump` = {a^(2k), 1 + a + ... + a^(2k-1)}), drops it if it faces away from the surface or the
    // light (renderer.rs:49: no shadow ray needed, result bit 1) and otherwise traces it. There is no sampling
    // kernel, no sample buffer and no ray list.
    const uint32_t n4 = 4u * n_light;
    const uint32_t n4_shift = (n4 & (n4 - 1u)) == 0u ? (uint32_t)__builtin_ctz(n4 | 0x80000000u) : 0xffffffffu;
    const uint32_t n_rays = ANY ? *ray_count : *Q.count;

    uint32_t r_next = 0, r_end = 0;  // ray indices this wave has reserved (wave-uniform)
    uint32_t w_next = 0, w_end = 0;  // the range of them it is handing out to its lanes
    uint32_t w_base = 0, cur_slot = 0;  // ANY: first ray and LDS slot of that range
    bool exhausted = false;          // the global queue has no ray left (wave-uniform)
    bool has = false;
    uint32_t out_idx = 0;  // closest: ray index; ANY: slot << 8 | position in the slot's range
    RayK r{};
    BoxRay br{};
    float best_t = __builtin_inff(), t_max = 0.0f;
    int best = -1;
    uint32_t cur = 0;  // the item in hand
    int sp = 0;

    // ANY: a ray's result goes into its range's bits; the range's count of unfinished rays goes down by one
    auto settle = [&](uint32_t oi, bool occluded) {
        asm volatile("" : "+v"(oi));  // (or the slot addresses are computed when the ray starts and held all along)
        const uint32_t s = oi >> 8, bit = oi & 255u;
        if (occluded) atomicOr(&wsa[s * kSlotWords + (bit >> 5)], 1u << (bit & 31u));
        atomicSub(&wsa[s * kSlotWords + 8], 1u);
    };

    // Two nested loops: the outer one hands rays to idle lanes, the inner one runs trips until kFetchMin lanes are idle
    // again. (As ONE loop with the refill under `if (n_idle >= kFetchMin)` the compiler copied the ~22 registers of
    // per-ray state through temporaries on every pass through that block: a third of the kernel's VALU instructions.)
    for (;;) {
        const unsigned long long idle = ballot64(!has);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        {
            unsigned long long free_slots = 1ull;
            if (ANY) {  // complete ranges go out as 32 bytes of bits and give their slot back
                uint32_t s_rem = 1u, s_first = kSlotFree;
                if (lane < kSlots) {
                    s_rem = ws[lane * kSlotWords + 8];
                    s_first = ws[lane * kSlotWords + 9];
                }
                unsigned long long done = __ballot(s_first != kSlotFree && s_rem == 0u);
                while (done) {
                    const uint32_t s = (uint32_t)__builtin_ctzll(done);
                    done &= done - 1ull;
                    const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)s_first, (int)s);
                    if (lane < 8u) occ_bits[(first >> 5) + lane] = ws[s * kSlotWords + lane];
                    if (lane == s) {
                        ws[s * kSlotWords + 9] = kSlotFree;
                        s_first = kSlotFree;
                    }
                }
                free_slots = __ballot(lane < kSlots && s_first == kSlotFree);
            }
            if (w_next == w_end && !exhausted && free_slots != 0ull) {
                if (r_next == r_end) {
                    uint32_t got = 0;
                    if (lane == 0) got = atomicAdd(head, kReserve);
                    got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                    if (got >= n_rays) {
                        exhausted = true;
                    } else {
                        r_next = got;
                        r_end = min(got + kReserve, n_rays);
                    }
                }
                if (r_next != r_end) {
                    const uint32_t base = r_next;
                    w_next = base;
                    w_end = min(base + kFetchChunk, r_end);
                    r_next = w_end;
                    if (ANY) {
                        cur_slot = (uint32_t)__builtin_ctzll(free_slots);
                        w_base = base;
                        if (lane < 8u) ws[cur_slot * kSlotWords + lane] = 0u;
                        if (lane == 8u) ws[cur_slot * kSlotWords + 8] = w_end - w_next;
                        if (lane == 9u) ws[cur_slot * kSlotWords + 9] = base;
                    }
                }
            }
            const uint32_t avail = w_end - w_next;
            if (avail != 0u) {
                uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (!has && rank < avail) {
                    uint32_t ridx = w_next + rank;
                    bool take = true;
                    if (ANY) {
                        // shadow ray `ridx` of the list wfb_logic wrote: sample id e * 4 L + k of a sample that faces its light
                        const uint32_t sid = ray_ids[ridx];
                        take = sid != 0xffffffffu;  // (the unused tail of a block's last slab of the list)
                        const uint32_t sidc = take ? sid : 0u;
                        const uint32_t e = n4_shift != 0xffffffffu ? sidc >> n4_shift : sidc / n4, k = sidc - e * n4;
                        const float4 *lrow = blob + light_off + (k >> 2) * LIGHT_F4;
                        const float4 l0 = lrow[0], l1 = lrow[1], l2 = lrow[2];
                        const uint32_t ja = jump[2u * k], jc = jump[2u * k + 1u];
                        const uint32_t s0 = Q.ldu(U_RNG0, e), inc = Q.ldu(U_RNG1, e);
                        Rng rg{s0 * ja + inc * jc, inc};
                        const float e1 = rg.gen() - 0.5f;
                        const float e2 = rg.gen() - 0.5f;
                        const F3 x = Q.ld3(F_X, e);
                        F3 l;
                        float tmax, inv_d2;
                        light_sample(l0, l1, l2, x, e1, e2, l, tmax, inv_d2);
                        out_idx = (cur_slot << 8) | (ridx - w_base);
                        if (!take) settle(out_idx, true);
                        r = make_ray(madd(x, l, 1e-4f), l);
                        t_max = tmax;  // a sphere occludes when t * t < tmax * tmax, so t < tmax: the boxes need no more than tmax
                    } else {
                        // (streaming loads here, so that the rays do not displace the result lines a wave is still filling from
                        // the L2, were measured: the closest-hit launches' writes fall from 2.2x to 1.4x their algorithmic 4 bytes per
                        // ray, but their fetches double — a wave takes ~20 rays of a 64-ray row per refill and the rest of the line is
                        // fetched again — and the step time is the same: not adopted, profiles/r03_c4_stream_loads_traffic.txt)
                        const F3 qd = Q.ld3(F_RD, ridx);
                        take = (qd.x != 0.0f) | (qd.y != 0.0f) | (qd.z != 0.0f);  // a zero direction marks a position without an entry (pt_wfqueue.h: kill_tail)
                        r = make_ray(Q.ld3(F_RO, ridx), qd);
                        best_t = __builtin_inff();
                        best = -1;
                        out_idx = ridx;
                    }
                    if (take) {
                        br = make_box_ray(bvh, r.ro, r.rd);
                        cur = bvh.n;  // the root: wide node 0, entry distance 0
                        sp = 0;
                        has = true;
                        if (STATS) ++st_rays;
                    }
                }
                w_next += min(n_idle, avail);
            } else if (n_idle == 64u) {
                // exhausted and every lane is done (then every range is complete and was written above). With rays left in
                // the queue an all-idle wave always has a free slot; should the bookkeeping ever be wrong, leaving is
                // a wrong image, which the tests see, where spinning would hang the GPU.
                break;
            }
        }
        do {
        if (has) {
        // ---- one item: fetch ----
#ifdef WFB_DUMMY_VALU
        {   // sensitivity experiment (profiles/r03_c4_ab_log.txt): -DWFB_DUMMY_VALU=n adds n independent full-rate instructions per trip
            float d0 = t_max, d1 = t_max;
#pragma unroll
            for (int i = 0; i < WFB_DUMMY_VALU / 2; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1" : "+v"(d0), "+v"(d1));
            asm volatile("" ::"v"(d0), "v"(d1));
        }
#endif
        const uint32_t u = cur & idx_mask;
        const bool is_leaf = u < bvh.n;
        const uint32_t w = u - bvh.n;
        uint4 q0, q1, q2, q3;  // (q1..q3 are read on the node path only)
        if (!is_leaf && w < n_top) {
            q0 = s_top[w * 4u + 0u], q1 = s_top[w * 4u + 1u], q2 = s_top[w * 4u + 2u], q3 = s_top[w * 4u + 3u];
            if (STATS) ++st_top;
        } else {
            const uint4 *p = is_leaf ? reinterpret_cast<const uint4 *>(bvh.sph) + u : bvh.wide + (size_t)w * 4u;
            q0 = p[0];
            if (!is_leaf) q1 = p[1], q2 = p[2], q3 = p[3];
        }
        bool finished = false, need_pop = false;
        if (is_leaf) {
            // ---- a sphere ----
            if (STATS) ++st_leaves;
            const float t = sphere_t(r, float4{__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w)});
            if (ANY) {
                float tm = t_max;  // (behind a barrier, or tmax * tmax is kept in a register for the whole traversal)
                asm volatile("" : "+v"(tm));
                if ((t > 0.0f) & (t * t < tm * tm)) {
                    settle(out_idx, true);
                    finished = true;
                } else {
                    need_pop = true;
                }
            } else {
                if (t > 0.0f && t <= best_t) {
                    if (t < best_t) {
                        best_t = t;
                        best = (int)u;
                    } else if (best >= 0) {  // exact tie: renderer.rs:359-361, first minimum in scene order
                        const uint32_t ord_new = __float_as_uint(bvh.aux[u].z), ord_old = __float_as_uint(bvh.aux[best].z);
                        if (STATS) st_aux += 2u;
                        if (ord_new < ord_old) best = (int)u;
                    }
                }
                need_pop = true;
            }
        } else {
            // ---- a wide node: enter the child boxes, nearest first ----
            if (STATS) ++st_nodes;
            const float bound = ANY ? t_max : best_t;  // inclusive: a sphere that ties must still be visited
            // An item's key: the bits of its entry distance (>= 0, so they order as integers) with the low idx_bits replaced
            // by the child index: one v_and_or_b32 and one select per slot. (The compiler, left alone, selects the distance
            // against a sentinel first and the key against that sentinel again — two compares and two selects per slot,
            // all half-rate instructions — hence the barrier.)
            auto key = [&](uint4 q) {
                float t;
                const bool hit = slot_hit(br, q, bound, t);
                uint32_t k = (__float_as_uint(t) & (~idx_mask & 0x7fffffffu)) | q.w;  // (t >= 0; the sign is masked so that a -0 cannot pose as the farthest key)
                asm volatile("" : "+v"(k));
                return hit ? k : kKeyNone;
            };
            const uint32_t k0 = key(q0), k1 = key(q1), k2 = key(q2), k3 = key(q3);
            // sorted ascending (misses, kKeyNone, last): min3 / med3 / max3 of three, then the fourth put in place
            const uint32_t lo3 = min(min(k0, k1), k2), hi3 = max(max(k0, k1), k2);
            const uint32_t md3 = max(min(k0, k1), min(max(k0, k1), k2));
            const uint32_t a0 = min(lo3, k3), a3 = max(hi3, k3);
            const uint32_t a1 = max(min(lo3, md3), min(max(lo3, md3), k3));  // med3(lo3, md3, k3)
            const uint32_t a2 = max(min(md3, hi3), min(max(md3, hi3), k3));  // med3(md3, hi3, k3)
            auto push = [&](uint32_t item) {
                if ((uint32_t)sp < lds_entries) stack[sp * stride] = item;
                else spill_col[((uint32_t)sp - lds_entries) * spill_stride] = item;
                ++sp;
            };
            if (__builtin_expect((uint32_t)sp + 3u <= lds_entries, 1)) {
                // without branches: each candidate is stored at the running top, which moves on only if it was a real one
                // (a miss is overwritten by the next store or stays above the top, unused)
                stack[sp * stride] = a3;
                sp += a3 != kKeyNone ? 1 : 0;
                stack[sp * stride] = a2;
                sp += a2 != kKeyNone ? 1 : 0;
                stack[sp * stride] = a1;
                sp += a1 != kKeyNone ? 1 : 0;
            } else {
                if (a3 != kKeyNone) push(a3);
                if (a2 != kKeyNone) push(a2);
                if (a1 != kKeyNone) push(a1);
            }
            if (a0 != kKeyNone) cur = a0;
            else need_pop = true;
        }
        if (need_pop) {
            const uint32_t best_key = __float_as_uint(best_t);
            for (;;) {
                if (sp == 0) {
                    finished = true;
                    if (ANY) settle(out_idx, false);
                    else hit_leaf[out_idx] = best;
                    break;
                }
                --sp;
                if (__builtin_expect((uint32_t)sp < lds_entries, 1)) {
                    cur = stack[sp * stride];
                } else {  // (the barrier keeps the two loads apart: merged, they become one flat load with a selected address)
                    cur = spill_col[((uint32_t)sp - lds_entries) * spill_stride];
                    asm volatile("" : "+v"(cur));
                }
                if (ANY || (cur & ~idx_mask) <= best_key) break;  // an item that begins beyond the best hit cannot improve it
            }
        }
        if (finished) has = false;
        }
        } while ((uint32_t)__popcll(ballot64(!has)) < kFetchMin);
    }
    if (STATS) {
        uint32_t v[5] = {st_rays, st_nodes, st_leaves, st_aux, st_top};
        for (int k = 0; k < 5; ++k) {
            unsigned long long w = v[k];
            for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off);
            if (lane == 0 && w) atomicAdd(&stats[k < 4 ? (ANY ? 4 : 0) + k : (ANY ? 9 : 8)], w);
        }
    }
}

// Everything of one path vertex that follows the closest-sphere query and precedes next-event estimation.
#ifndef WFB_LOGIC_WAVES
#define WFB_LOGIC_WAVES 1
#endif
#ifndef WFB_SHADE_WAVES
#define WFB_SHADE_WAVES 1
#endif
__global__ __launch_bounds__(256, WFB_LOGIC_WAVES) void wfb_logic(Params P, const float4 *__restrict__ blob, Bvh bvh, WfQueue Qin,
                                                 const int *__restrict__ hit_leaf,
                                                 WfQueue Qnext, WfQueue Qshade, float *__restrict__ accum,
                                                 uint32_t *__restrict__ ray_ids, uint32_t *__restrict__ ray_count) {
    __shared__ SlabState s_slab;
    slab_init(&s_slab);
    TblGlobal tbl{blob};
    Tracer<TblGlobal, false, true> tr(P, tbl, bvh, nullptr);
    const uint32_t count = *Qin.count;
    const uint32_t slab = slab_size(count, gridDim.x);
    const uint32_t slab_ids = ids_slab_size(slab, P.n_light);
    const uint32_t seed_mixed = mix32(P.seed ^ 0x85ebca6bu);
    for (uint32_t base = blockIdx.x * 256u; base < count; base += gridDim.x * 256u) {
        uint32_t qi = base + threadIdx.x;
        bool act = qi < count;
        bool to_next = false, to_shade = false;
        uint32_t pix = 0, next = 0, depth = 0, mat_word = 0;
        int media = -1;
        bool scatter = false;
        Rng rng{0u, 1u};
        F3 ro = F3{0, 0, 0}, rd = F3{0, 0, 1}, T = F3{1, 1, 1}, x = F3{0, 0, 0}, nn = F3{0, 1, 0};
        // The whole entry and its hit are requested together, before the entry's flag is known (a position without an
        // entry holds stale words that are not used): one memory round trip instead of two at the head of a chain of
        // four (entry, sphere, material, queue writes) that this kernel spends two thirds of its time waiting for.
        uint32_t meta = 0u;
        int best = -1;
        if (act) {
            meta = Qin.ldu(U_META, qi);
            pix = Qin.ldu(U_PIX, qi);
            next = Qin.ldu(U_NEXT, qi);
            ro = Qin.ld3(F_RO, qi);
            rd = Qin.ld3(F_RD, qi);
            T = Qin.ld3(F_T, qi);
            rng.s0 = Qin.ldu(U_RNG0, qi);
            rng.s1 = Qin.ldu(U_RNG1, qi);
            best = hit_leaf[qi];
        }
        act = act && (meta & 1u);  // a position without an entry (the tail of some block's last slab)
        if (act) {
            depth = (meta >> 1) & 31u;
            media = (int)(meta >> 8) - 1;
            bool alive = true;

            // pt_trace.h: Tracer::closest after bvh_closest — planes, then the winner with the reference's roundings
            RayK ray = make_ray(ro, rd);
            // the traversal hands over the sphere only; its t is formed again here, by the same sphere_t() on the same ray
            float best_t = best >= 0 ? sphere_t(ray, bvh.sph[best]) : __builtin_inff();
            closest_planes(tbl, P, ray, best_t, best, [&](int id) { return __float_as_uint(bvh.aux[id].z); });
            HitRec h{best_t, best};
            refine_hit(tbl, P, ray.ro, ray.rd, h, [&](int id) { return bvh.sph[id]; });

            F3 A = F3{1, 1, 1};
            float t_media = __builtin_inff();
            if (media >= 0) {
                Mat mm = load_mat(tbl, P.mat_off, (uint32_t)media);
                t_media = -fln(rng.gen()) * rcp(mm.density);
                float k = -mm.density * fminf(t_media, h.t);
                A = F3{fexp(fmaxf(1.0f - mm.color.x, 0.0f) * k), fexp(fmaxf(1.0f - mm.color.y, 0.0f) * k),
                       fexp(fmaxf(1.0f - mm.color.z, 0.0f) * k)};
            }
            if (t_media < h.t) {
                T = T * A;
                x = point_at_ref(ro, rd, t_media);
                nn = F3{0.0f, 1.0f, 0.0f};
                mat_word = (uint32_t)media | ((uint32_t)(media + 1) << 16);
                scatter = true;
                to_shade = true;
            } else if (h.id < 0) {
                alive = false;
            } else {
                uint32_t mid, order;
                tr.finish(ray, h, x, nn, mid, order);
                Mat m = load_mat(tbl, P.mat_off, mid);
                int next_media = media;
                if (m.density > 0.0f) next_media = (dot(-rd, nn) > 0.0f) ? (int)mid : -1;
                T = T * A;
                if (m.ior > 1.0f && m.metallic < 0.1f) {
                    F3 nd = glass_dir(rd, nn, m.ior, rng.gen());
                    ro = madd(x, nd, 1e-4f);
                    rd = nd;
                    media = next_media;
                    ++depth;
                    if (depth >= P.max_depth) alive = false;
                } else {
                    mat_word = mid | ((uint32_t)(next_media + 1) << 16);
                    to_shade = true;
                }
            }
            if (!to_shade) {
                uint32_t s_begin, s_end;
                stream_range(P, pix, s_begin, s_end);
                if (!alive && next < s_end) {  // the stream's next sample takes the finished path's place
                    new_camera_path(tbl, P, seed_mixed, pix, next, rng, ro, rd, T, depth, media);
                    alive = true;
                }
                to_next = alive;  // (a finished stream's sum is already in its slice)
            }
        }
        uint32_t pn, ps;
        queue_reserve_slab(to_next, Qnext.count, pn, to_shade, Qshade.count, ps, &s_slab, slab);
        // The shadow rays of the vertex: the light samples that face their light (renderer.rs:49-55; half of C4's do
        // not), listed by sample id for wfb_traverse<ANY>, vertex by vertex. This kernel waits on memory, so drawing
        // the samples here costs nothing, whereas a traversal lane that drew a sample only to drop it stood idle
        // until its wave's next refill. The samples are drawn twice (count, then write) rather than kept.
        uint32_t n_face = 0u;
        if (to_shade) {
            Rng rc = rng;
            for_each_light_sample(tbl, P, rc, x, nn, [&](uint32_t, F3, float, float, bool faces) { n_face += faces ? 1u : 0u; });
        }
        const uint32_t first_ray = list_reserve_slab(n_face, ray_count, &s_slab, slab_ids);
        if (n_face) {
            Rng rc = rng;
            uint32_t j = first_ray;
            const uint32_t sid0 = ps * (4u * P.n_light);
            for_each_light_sample(tbl, P, rc, x, nn, [&](uint32_t k, F3, float, float, bool faces) {
                if (faces) ray_ids[slab_item(&s_slab, j++)] = sid0 + k;
            });
        }
        const uint32_t s_slab_a = s_slab.a[2], s_slab_split = s_slab.split[2], s_slab_b = s_slab.b[2];
        __syncthreads();  // the list allocator's fields are reused by the next trip
        if (to_next) push_ray(Qnext, pn, pix, depth, media, next, ro, rd, T, rng);
        if (to_shade) {
            // where the results of the vertex's shadow rays will be: list positions p_first .. p_first + seg - 1, then
            // (a vertex whose rays straddle two slabs) p_second ..
            const uint32_t split = s_slab_split, seg = first_ray >= split ? n_face : min(n_face, split - first_ray);
            Qshade.stw(F_SUM + 0, ps, slab_item_of(s_slab_a, split, s_slab_b, first_ray));
            Qshade.stw(F_SUM + 1, ps, seg);
            Qshade.stw(F_SUM + 2, ps, s_slab_b);
            Qshade.stu(U_PIX, ps, pix);
            Qshade.stu(U_META, ps, 1u | (depth << 1) | (scatter ? 64u : 0u) | ((uint32_t)(media + 1) << 8));
            Qshade.stu(U_NEXT, ps, next);
            Qshade.stu(U_MAT, ps, mat_word);
            Qshade.stu(U_RNG0, ps, rng.s0);
            Qshade.stu(U_RNG1, ps, rng.s1);
            Qshade.st3(F_RD, ps, rd);
            Qshade.st3(F_T, ps, T);
            Qshade.st3(F_X, ps, x);
            Qshade.st3(F_N, ps, nn);
        }
    }
    kill_tail(Qnext, &s_slab, 0);
    kill_tail(Qshade, &s_slab, 1);
    for (uint32_t i = s_slab.pos[2] + threadIdx.x; i < s_slab.end[2]; i += blockDim.x) ray_ids[i] = 0xffffffffu;  // no ray
}

__global__ __launch_bounds__(256, WFB_SHADE_WAVES) void wfb_shade(Params P, const float4 *__restrict__ blob, WfQueue Qshade,
                                                 const uint32_t *__restrict__ occ_bits,
                                                 WfQueue Qnext, float *__restrict__ accum) {
    __shared__ SlabState s_slab;
    slab_init(&s_slab);
    TblGlobal tbl{blob};
    const uint32_t count = *Qshade.count;
    const uint32_t slab = slab_size(count, gridDim.x);
    const uint32_t seed_mixed = mix32(P.seed ^ 0x85ebca6bu);
    for (uint32_t base = blockIdx.x * 256u; base < count; base += gridDim.x * 256u) {
        uint32_t qi = base + threadIdx.x;
        bool act = qi < count;
        bool to_next = false;
        uint32_t pix = 0, next = 0, depth = 0;
        int media = -1;
        Rng rng{0u, 1u};
        F3 ro = F3{0, 0, 0}, rd = F3{0, 0, 1}, T = F3{1, 1, 1};
        // (the whole entry is requested before its flag is known, as in wfb_logic)
        uint32_t meta = 0u, mw = 0u, p_first = 0u, seg = 0u, p_second = 0u;
        F3 x = F3{0, 0, 0}, nn = F3{0, 1, 0};
        if (act) {
            meta = Qshade.ldu(U_META, qi);
            pix = Qshade.ldu(U_PIX, qi);
            next = Qshade.ldu(U_NEXT, qi);
            mw = Qshade.ldu(U_MAT, qi);
            rng.s0 = Qshade.ldu(U_RNG0, qi);
            rng.s1 = Qshade.ldu(U_RNG1, qi);
            x = Qshade.ld3(F_X, qi), nn = Qshade.ld3(F_N, qi);
            rd = Qshade.ld3(F_RD, qi);
            T = Qshade.ld3(F_T, qi);
            p_first = Qshade.ldw(F_SUM + 0, qi), seg = Qshade.ldw(F_SUM + 1, qi), p_second = Qshade.ldw(F_SUM + 2, qi);
        }
        act = act && (meta & 1u);  // a position without an entry
        if (act) {
            depth = (meta >> 1) & 31u;
            bool scatter = meta & 64u;
            media = (int)(meta >> 8) - 1;
            uint32_t mid = mw & 0xffffu;
            int next_media = (int)(mw >> 16) - 1;
            Mat m = load_mat(tbl, P.mat_off, mid);
            ShadeK sk = make_shade(nn, -rd, m);

            // pt_trace.h: direct_light, with the samples and their sphere occlusion already computed
            uint32_t n_seen = 0u;  // facing samples so far: the j-th one's result is the j-th of the vertex's list positions
            F3 total = F3{0.0f, 0.0f, 0.0f};
            for (uint32_t li = 0; li < P.n_light; ++li) {
                uint32_t o = P.light_off + li * LIGHT_F4;
                float4 l0 = tbl.ld(o), l1 = tbl.ld(o + 1), l2 = tbl.ld(o + 2), l3 = tbl.ld(o + 3), l4 = tbl.ld(o + 4);
                F3 l[4];
                float id2[4], tmax[4];
                bool occ[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {  // the same draws as wfb_traverse<ANY> made for the shadow rays
                    float e1 = rng.gen() - 0.5f;
                    float e2 = rng.gen() - 0.5f;
                    light_sample(l0, l1, l2, x, e1, e2, l[s], tmax[s], id2[s]);
                    occ[s] = true;  // a sample that faces away needs no ray and adds nothing (renderer.rs:49-55)
                    if (sample_faces(nn, l[s], l4)) {
                        const uint32_t rbit = n_seen < seg ? p_first + n_seen : p_second + (n_seen - seg);
                        ++n_seen;
                        occ[s] = ((occ_bits[rbit >> 5] >> (rbit & 31u)) & 1u) != 0u;
                    }
                }
                uint32_t c0 = 0, c1 = 0;
                occluded_multi<TblGlobal, 4, false>(tbl, P, x, 1e-4f, l, tmax, occ, c0, c1, false);
                F3 contrib = F3{0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (!occ[s]) {
                        bool lit;
                        F3 c = nee_eval(sk, l[s], id2[s], f3(l4), l0.w, f3(l3), lit);
                        contrib = contrib + c;
                    }
                }
                total = madd(total, contrib, 0.25f);  // :82
            }
            const F3 gain = T * total;
            if (gain.x != 0.0f || gain.y != 0.0f || gain.z != 0.0f) stream_add(P, accum, pix, gain);

            bool alive = true;
            if (scatter) {
                float ct = sample_phase_function(m.aniso, rng.gen());
                float up = rng.gen();
                ro = x;
                rd = scatter_dir(rd, ct, up);
            } else {
                float p = fmaxf(fmaxf(m.color.x, m.color.y), m.color.z);
                bool cont = depth < 2u;
                if (!cont) cont = rng.gen() < p;
                if (cont) {
                    F3 nd, f;
                    if (rng.gen() < (1.0f - m.metallic)) {
                        float up = rng.gen(), r2 = rng.gen();
                        nd = diffuse_dir(nn, up, r2);
                        f = m.color * 0.31830988618379067154f;
                    } else {
                        float r1 = rng.gen(), r2 = rng.gen();
                        nd = reflect(rd, sample_ggx_h(nn, m.roughness, r1, r2));
                        f = F3{1.0f, 1.0f, 1.0f};
                    }
                    float c = dot(nd, nn);
                    if (c > 0.0f) {
                        float w = depth >= 2u ? c * rcp(p) : c;
                        T = T * f * w;
                        ro = madd(x, nd, 1e-4f);
                        rd = nd;
                        media = next_media;
                    } else {
                        alive = false;
                    }
                } else {
                    alive = false;
                }
            }
            ++depth;
            if (depth >= P.max_depth) alive = false;
            uint32_t s_begin, s_end;
            stream_range(P, pix, s_begin, s_end);
            if (!alive && next < s_end) {
                new_camera_path(tbl, P, seed_mixed, pix, next, rng, ro, rd, T, depth, media);
                alive = true;
            }
            to_next = alive;
        }
        uint32_t pn, unused;
        queue_reserve_slab(to_next, Qnext.count, pn, false, nullptr, unused, &s_slab, slab);
        if (to_next) push_ray(Qnext, pn, pix, depth, media, next, ro, rd, T, rng);
    }
    kill_tail(Qnext, &s_slab, 0);
}

#define WFB_TRAV wfb_traverse

// ---- host driver -----------------------------------------------------------------------
namespace {
struct WfbLayout {
    size_t per_queue, hits, occ, ids, total;
};
#ifndef WFB_TRAV_BLOCKS_PER_CU
#define WFB_TRAV_BLOCKS_PER_CU 7u
#endif
#ifndef WFB_ANY_BLOCKS_PER_CU
#define WFB_ANY_BLOCKS_PER_CU 7u
#endif
// Persistent traversal blocks. Seven per CU, not the eight that fit: the any-hit kernel's 66 registers allow seven waves
// per SIMD anyway (an eighth block per CU would start when the others end, and find nothing), and the closest-hit
// kernel leaves a wave slot per SIMD to the other pipeline's logic / shading kernels (C4: 8 / 8, 8 / 7 and 7 / 7 are
// equal within the +-0.4 % of repeated runs; 6 / 6 -1.4 %, 5 / 5 -2.6 %).
constexpr unsigned kGridTrav = 256u * 8u * 256u / kTravThreads;  // (sizes the spill columns)
constexpr unsigned kGridClosest = 256u * WFB_TRAV_BLOCKS_PER_CU * 256u / kTravThreads, kGridAny = 256u * WFB_ANY_BLOCKS_PER_CU * 256u / kTravThreads;
// stack entries in LDS with 8 blocks per CU: what is left of 20 KB beside the first kWideTop wide nodes and the range slots (13 for 85 nodes)
constexpr uint32_t kLdsEntriesMax = (160u * 1024u / 8u - kWideTop * 64u - kSlotBytes) / (kTravThreads * 4u);
// a ray stacks at most three items per level of the wide tree; deeper trees than this stay with the one-kernel LBVH variant
constexpr uint32_t kWideStackCap = 3u * 48u + 1u;
// stack entries that do not fit the LDS part, one column per traversal thread, shared by all traversal launches of a stream
size_t wfb_spill_bytes(uint32_t stack_entries) {
    return stack_entries > kLdsEntriesMax ? (size_t)(stack_entries - kLdsEntriesMax) * kGridTrav * kTravThreads * 4 : 256;
}
// blocks of the logic / shading kernels for n entries, and the queue positions those entries may occupy (pt_wfqueue.h: slabs)
unsigned wfb_queue_blocks(uint32_t n) { return std::min((n + 255u) / 256u, 256u * 8u); }
size_t wfb_positions(uint32_t n) { return (size_t)n + slab_slack(n, wfb_queue_blocks(n)); }
// positions the list may occupy: every sample of every entry plus one slab tail per block
size_t wfb_ray_positions(uint32_t n, uint32_t n_light) {
    const unsigned blocks = wfb_queue_blocks(n);
    return wfb_positions(n) * 4 * std::max(1u, n_light) + (size_t)blocks * ids_slab_size(kSlabMax, n_light);
}
WfbLayout wfb_layout(uint32_t n, uint32_t n_light) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    WfbLayout L;
    const size_t np = wfb_positions(n);
    L.per_queue = al(queue_bytes(np));
    L.hits = al(np * 4);                                          // closest rays: the sphere, 4 bytes
    const size_t nr = wfb_ray_positions(n, n_light);
    L.occ = al(nr / 8 + 64);                                      // shadow rays: one bit each, written 32 bytes at a time
    L.ids = al(nr * 4);                                           // the list of shadow rays (sample ids)
    L.total = 4096 + 3 * L.per_queue + L.hits + L.occ + L.ids;
    return L;
}
}  // namespace

// The queue pipeline takes trees whose unified index (spheres + wide nodes) leaves at least 8 bits of a key for the
// entry distance and whose depth fits the stack cap; hippt_api.hip keeps other scenes on the one-kernel LBVH variant.
bool wfb_supports(const BvhBuffers &b) {
    return b.wide && b.n_wide > 0 && (uint64_t)b.n + b.n_wide <= (1ull << 24) && 3u * b.wide_levels + 1u <= kWideStackCap;
}
uint32_t wfb_stack_entries(const BvhBuffers &b) { return std::max(8u, 3u * b.wide_levels + 1u); }

// ray ids are 32-bit, and the fetch head runs past the last ray by up to one reservation per wave
constexpr uint64_t kMaxRayIds = (1ull << 32) - (1ull << 27);
uint64_t wfb_max_ray_ids() { return kMaxRayIds; }

// room for one pipeline over n entries or two over n / 2 each
size_t wfb_workspace_bytes(uint32_t n_pixels, uint32_t n_light) {
    return std::max(wfb_layout(n_pixels, n_light).total, 2 * wfb_layout((n_pixels + 1) / 2, n_light).total) +
           2 * ((wfb_spill_bytes(kWideStackCap) + 255) & ~(size_t)255);
}

uint32_t wfb_streams_log2(uint32_t n_pixels, uint32_t sample_count, uint32_t n_light) {
    uint32_t lg = 0;
    while (lg < WFB_MAX_STREAMS_LOG2 && ((uint64_t)n_pixels << (lg + 1)) <= (1ull << WFB_INFLIGHT_LOG2) && (2u << lg) <= sample_count &&
           wfb_layout(n_pixels << (lg + 1), n_light).total <= ((unsigned long long)WFB_QUEUE_CAP_GIB << 30) &&
           wfb_ray_positions(n_pixels << (lg + 1), n_light) < kMaxRayIds)  // shadow-ray ids e * 4 L + k are 32-bit
        ++lg;
    return lg;
}

// P.wf_streams_log2 streams per pixel; `partial` = width*height*2^that*3 floats: the streams' running sums, by entry id.
//
// TWO PIPELINES. With a second stream from the caller (`st2`) and at least two sample streams per pixel, the streams are
// dealt to two groups, each with its own queues and counters, and the groups' rounds run concurrently: group 0 on `st`,
// group 1 on `st2` (forked from and joined to `st` with the two events). The traversal kernels wait on node fetches
// for most of their cycles and leave HBM idle; the logic / shading kernels stream the queues through HBM and
// leave the vector L1s idle; and every traversal launch ends with a tail in which a few long rays keep a few waves
// busy. Two independent pipelines let the hardware fill each of these gaps with the other group's kernels.
// Round 3: the gain depends on the kernels of the day. With the first 4-wide traversal kernels one pipeline over all
// streams was 3.2 % faster (1 087 vs 1 053 Mpaths/s on C4, interleaved A/B: the second pipeline's kernels only competed
// for issue slots); after the later changes (cheaper item keys, fewer registers, whole-entry loads in the logic /
// shading kernels, pixel-major sums) two pipelines are 1.5 % faster again (1 143 vs 1 125). ON.
#ifndef WFB_GROUPS
#define WFB_GROUPS 2
#endif
hipError_t wavefront_bvh_render(const Params &P, const float4 *blob, const Bvh &bvh, uint32_t stack_entries, float *accum,
                                void *workspace, hipStream_t st, uint32_t *iterations_out, unsigned long long *stats, float *partial,
                                hipStream_t st2, hipEvent_t ev_fork, hipEvent_t ev_join) {
    const uint32_t n_all = (P.width * P.height) << P.wf_streams_log2;
    float *const sums = partial;
    if (!partial) return hipErrorInvalidValue;
    const uint32_t group_bits = (WFB_GROUPS > 1 && st2 && ev_fork && ev_join && P.wf_streams_log2 >= 1u) ? 1u : 0u;
    const uint32_t n_groups = 1u << group_bits;
    const uint32_t n = n_all >> group_bits;
    const WfbLayout L = wfb_layout(n, P.n_light);
    struct Group {
        hipStream_t st;
        uint32_t *counters;  // [0],[1]: ray queue counts (ping-pong), [2]: shade count, [3],[4]: fetch heads; [256..]: the jump table
        WfQueue Q[3];
        int *hit_leaf;
        uint32_t *occ;
        uint32_t *ids;
        uint32_t *spill;
        bool done;
    } G[2];
    uint32_t lds_entries = std::min(stack_entries, kLdsEntriesMax);
    if (const char *dbg = getenv("HIPPT_WFB_LDS_STACK")) {  // test hook: a small LDS part forces the spill path (tests/test_gpu_bvh.py)
        const long v = strtol(dbg, nullptr, 10);
        if (v >= 1 && (uint32_t)v < lds_entries && stack_entries - (uint32_t)v <= kWideStackCap - kLdsEntriesMax) lds_entries = (uint32_t)v;
    }
    const size_t spill_bytes = (wfb_spill_bytes(kWideStackCap) + 255) & ~(size_t)255;
    for (uint32_t g = 0; g < n_groups; ++g) {
        uint8_t *w = (uint8_t *)workspace + (size_t)g * (L.total + spill_bytes);
        G[g].spill = (uint32_t *)(w + L.total);
        G[g].st = g ? st2 : st;
        G[g].counters = (uint32_t *)w;
        w += 4096;
        for (int k = 0; k < 3; ++k) {
            G[g].Q[k].f = (float *)(w + L.per_queue * k);
            G[g].Q[k].count = &G[g].counters[k];
            G[g].Q[k].cap = n;
        }
        w += 3 * L.per_queue;
        G[g].hit_leaf = (int *)w;
        w += L.hits;
        G[g].occ = (uint32_t *)w;
        w += L.occ;
        G[g].ids = (uint32_t *)w;
        G[g].done = false;
    }

    const unsigned blocks_all = (n + 255) / 256;
    const unsigned grid = wfb_queue_blocks(n);
    // the shading kernel needs 98 VGPRs (5 blocks per CU): launched with the 8 blocks per CU of `grid`, three eighths of
    // its blocks would start only when the first ones end; its grid is what is resident at once
    unsigned grid_shade = grid;
    {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)wfb_shade, 256, 0) == hipSuccess && per_cu > 0 &&
            hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            grid_shade = std::min(grid, (unsigned)(per_cu * cus));
    }
    if (wfb_ray_positions(n, P.n_light) >= kMaxRayIds) return hipErrorInvalidValue;
    const unsigned grid_closest = kGridClosest, grid_any = kGridAny;
    if (stack_entries > kWideStackCap || !bvh.wide || (uint64_t)bvh.n + bvh.n_wide > (1ull << 24)) return hipErrorInvalidValue;  // wfb_supports()
    const uint32_t n_top = std::min(bvh.n_wide, kWideTop);
    uint32_t idx_bits = 1;
    while ((1ull << idx_bits) < (uint64_t)bvh.n + bvh.n_wide) ++idx_bits;
    const size_t stack_bytes = (size_t)n_top * 64 + kSlotBytes + (size_t)lds_entries * kTravThreads * 4;
    hipError_t e;
    if (P.sample_count == 0) return hipSuccess;
    if (stack_bytes > 64 * 1024) {  // more dynamic LDS than a launch may ask for by default
        const int b = (int)stack_bytes;
        if ((e = hipFuncSetAttribute((const void *)WFB_TRAV<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, b)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void *)WFB_TRAV<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, b)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void *)WFB_TRAV<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, b)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void *)WFB_TRAV<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, b)) != hipSuccess) return e;
    }
    const size_t image_floats = (size_t)P.width * P.height * 3;
    if ((e = hipMemsetAsync(partial, 0, (image_floats * sizeof(float)) << P.wf_streams_log2, st)) != hipSuccess) return e;
    if (n_groups > 1) {
        if ((e = hipEventRecord(ev_fork, st)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(st2, ev_fork, 0)) != hipSuccess) return e;
    }
    // whatever happens after the fork, `st` waits for `st2` again before this function returns
    auto join = [&](hipError_t err) {
        if (n_groups > 1) {
            hipError_t j = hipEventRecord(ev_join, st2);
            if (j == hipSuccess) j = hipStreamWaitEvent(st, ev_join, 0);
            if (err == hipSuccess) err = j;
        }
        return err;
    };
    if (P.n_light > 96u) return join(hipErrorInvalidValue);  // the jump table lives in the counters block (the API keeps such scenes off this pipeline)
    for (uint32_t g = 0; g < n_groups; ++g) {
        if ((e = hipMemsetAsync(G[g].counters, 0, 4096, G[g].st)) != hipSuccess) return join(e);
        hipLaunchKernelGGL(wfb_init, dim3(blocks_all), dim3(256), 0, G[g].st, P, blob, G[g].Q[0], n, group_bits, g, G[g].counters + 256);
    }
    const uint32_t max_iter = ((P.sample_count >> P.wf_streams_log2) + 1u) * P.max_depth + 1;
    uint32_t it = 0;
    int cur = 0;
    for (; it < max_iter; ++it) {
        for (uint32_t g = 0; g < n_groups; ++g) {
            Group &A = G[g];
            if (A.done) continue;
            hipStream_t s = A.st;
            uint32_t *counters = A.counters;
            if ((e = hipMemsetAsync(&counters[1 - cur], 0, 4, s)) != hipSuccess) return join(e);
            if ((e = hipMemsetAsync(&counters[2], 0, 16, s)) != hipSuccess) return join(e);  // shade count, both fetch heads, shadow-ray count
            if (stats)
                hipLaunchKernelGGL((WFB_TRAV<false, true>), dim3(grid_closest), dim3(kTravThreads), stack_bytes, s, bvh, A.Q[cur], (const float4 *)nullptr,
                                   (const uint32_t *)nullptr, &counters[3], A.hit_leaf, (uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, stats, lds_entries, A.spill, idx_bits, n_top);
            else
                hipLaunchKernelGGL((WFB_TRAV<false, false>), dim3(grid_closest), dim3(kTravThreads), stack_bytes, s, bvh, A.Q[cur], (const float4 *)nullptr,
                                   (const uint32_t *)nullptr, &counters[3], A.hit_leaf, (uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                                   (unsigned long long *)nullptr, lds_entries, A.spill, idx_bits, n_top);
            hipLaunchKernelGGL(wfb_logic, dim3(grid), dim3(256), 0, s, P, blob, bvh, A.Q[cur],
                               (const int *)A.hit_leaf, A.Q[1 - cur], A.Q[2], sums, A.ids, &counters[5]);
            if (stats)
                hipLaunchKernelGGL((WFB_TRAV<true, true>), dim3(grid_any), dim3(kTravThreads), stack_bytes, s, bvh, A.Q[2], blob,
                                   (const uint32_t *)(counters + 256), &counters[4], (int *)nullptr, A.occ, (const uint32_t *)A.ids, (const uint32_t *)&counters[5],
                                   stats, lds_entries, A.spill, idx_bits, n_top, P.light_off, P.n_light);
            else
                hipLaunchKernelGGL((WFB_TRAV<true, false>), dim3(grid_any), dim3(kTravThreads), stack_bytes, s, bvh, A.Q[2], blob,
                                   (const uint32_t *)(counters + 256), &counters[4], (int *)nullptr, A.occ, (const uint32_t *)A.ids, (const uint32_t *)&counters[5],
                                   (unsigned long long *)nullptr, lds_entries, A.spill, idx_bits, n_top, P.light_off, P.n_light);
            hipLaunchKernelGGL(wfb_shade, dim3(grid_shade), dim3(256), 0, s, P, blob, A.Q[2], (const uint32_t *)A.occ,
                               A.Q[1 - cur], sums);
        }
        cur = 1 - cur;
        // The host looks at the queue counts every 16 rounds at first (a look drains the launch pipeline), every 4 once a
        // 5-deep path's rounds are through: an empty round still costs 0.25 ms of launches (C4: 16 rounds per step).
        if ((it & 15u) == 15u || (it >= 11u && (it & 3u) == 3u)) {
            uint32_t left[2] = {0, 0};
            for (uint32_t g = 0; g < n_groups; ++g)
                if (!G[g].done && (e = hipMemcpyAsync(&left[g], &G[g].counters[cur], 4, hipMemcpyDeviceToHost, G[g].st)) != hipSuccess) return join(e);
            bool all_done = true;
            for (uint32_t g = 0; g < n_groups; ++g) {
                if (G[g].done) continue;
                if ((e = hipStreamSynchronize(G[g].st)) != hipSuccess) return join(e);
                if (left[g] == 0) G[g].done = true;
                all_done = all_done && G[g].done;
            }
            if (all_done) {
                ++it;
                break;
            }
        }
    }
    if (iterations_out) *iterations_out = it;
    if ((e = join(hipGetLastError())) != hipSuccess) return e;
    hipLaunchKernelGGL(wfb_add_streams, dim3((unsigned)((image_floats + 255) / 256)), dim3(256), 0, st, accum, (const float *)partial,
                       P.wf_streams_log2, image_floats);
    return hipGetLastError();
}

}  // namespace pt
