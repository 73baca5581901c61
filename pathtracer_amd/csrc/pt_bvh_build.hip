// pt_bvh_build.hip — GPU construction of the LBVH over the sphere table
// (north star "Optional BVH"; BASELINE config C4). Not in the reference.
//
//   1. bounds of the sphere centres            (block reduce + ordered-int atomics)
//   2. 30-bit Morton code per centre            (10 bits per axis)
//   3. stable radix sort of (code, index)       (rocPRIM device radix sort)
//   4. Karras 2012 binary radix tree over keys (code << 32 | sorted position)
//      (or, opt-in, PLOC over the same order: HIPPT_BVH_BUILDER=ploc, below)
//   5. leaf boxes, then bottom-up refit, one launch per level (a kernel boundary is the
//      only inter-workgroup ordering relied upon: no in-launch flags, cf. guide G16)
//   6. pack 32-byte traversal nodes {left box, right box, children} (half-float centre / half-size boxes)
//   7. collapse the tree into 4-wide nodes, breadth-first, one level per pass (expand, prefix sum, emit) for the
//      queue pipeline's traversal kernels (pt_bvh.h: "Wide nodes")
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>  // rocprim's texture iterator needs ::memset on the host side
#include <rocprim/rocprim.hpp>

#include "pt_bvh_build.h"
#include "pt_bvh.h"

namespace pt {
namespace {

__device__ __forceinline__ uint32_t f2o(float f) {  // order-preserving float -> uint
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float o2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

__global__ void k_bounds(const float4 *__restrict__ sph, uint32_t n, uint32_t *bounds /*lo3, hi3*/) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (; i < n; i += gridDim.x * blockDim.x) {
        float4 s = sph[i];
        lo[0] = fminf(lo[0], s.x), lo[1] = fminf(lo[1], s.y), lo[2] = fminf(lo[2], s.z);
        hi[0] = fmaxf(hi[0], s.x), hi[1] = fmaxf(hi[1], s.y), hi[2] = fmaxf(hi[2], s.z);
    }
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&bounds[k], f2o(lo[k]));
            atomicMax(&bounds[3 + k], f2o(hi[k]));
        }
    }
}

__device__ __forceinline__ uint32_t expand10(uint32_t v) {  // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void k_morton(const float4 *__restrict__ sph, uint32_t n, const uint32_t *__restrict__ bounds, uint32_t *codes,
                         uint32_t *idx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 s = sph[i];
    float c[3] = {s.x, s.y, s.z};
    uint32_t q[3];
    // one cell size for the three axes (cubic Morton cells): with per-axis scaling a flat scene such as the C4 slab
    // (80 x 12 x 80) is cut as often across its thin axis as along the others, which gives sliver boxes
    float ext = 1e-30f;
    for (int k = 0; k < 3; ++k) ext = fmaxf(ext, o2f(bounds[3 + k]) - o2f(bounds[k]));
    for (int k = 0; k < 3; ++k) {
        float lo = o2f(bounds[k]);
        float u = fminf(fmaxf((c[k] - lo) / ext * 1024.0f, 0.0f), 1023.0f);
        q[k] = (uint32_t)u;
    }
    codes[i] = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
    idx[i] = i;
}

// delta(i, j) of Karras 2012 on the unique keys (code << 32 | position); -1 outside [0, n).
__device__ __forceinline__ int delta(const uint32_t *__restrict__ codes, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    uint64_t a = ((uint64_t)codes[i] << 32) | (uint32_t)i, b = ((uint64_t)codes[j] << 32) | (uint32_t)j;
    return __clzll((long long)(a ^ b));
}

// Internal node i in [0, n-2]; leaves are numbered n-1+k in the parent array.
__global__ void k_hierarchy(const uint32_t *__restrict__ codes, int n, int *left, int *right, int *parent) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (delta(codes, n, i, i + 1) - delta(codes, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(codes, n, i, i - d);
    int lmax = 2;
    while (delta(codes, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(codes, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(codes, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (delta(codes, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    int lc = (lo == gamma) ? ~gamma : gamma;          // leaf children are stored as ~position
    int rc = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    left[i] = lc;
    right[i] = rc;
    parent[lc < 0 ? (n - 1 + ~lc) : lc] = i;
    parent[rc < 0 ? (n - 1 + ~rc) : rc] = i;
    if (i == 0) parent[0] = -1;
}

// Conservative sphere box: the traversal must never prune a sphere the brute-force loop would hit.
__global__ void k_leaf_boxes(const float4 *__restrict__ sph_sorted, const float4 *__restrict__ aux_sorted, int n, float4 *lo,
                             float4 *hi, int *ready) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    float4 s = sph_sorted[k];
    float r = aux_sorted[k].w;  // radius
    float pad = r * 1e-4f + 1e-5f * fmaxf(fmaxf(fabsf(s.x), fabsf(s.y)), fmaxf(fabsf(s.z), 1.0f));
    float e = r + pad;
    lo[n - 1 + k] = make_float4(s.x - e, s.y - e, s.z - e, 0);
    hi[n - 1 + k] = make_float4(s.x + e, s.y + e, s.z + e, 0);
    ready[n - 1 + k] = 1;
}

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) ------------------------------------------
// A better tree over the same Morton order than the radix tree: the clusters (at first the spheres) stand in Morton
// order; every cluster looks at its `radius` neighbours on either side for the one whose union box with its own has
// the smallest surface, mutual nearest neighbours are merged into a new node, the rest stay, the array is compacted
// and the pass repeats until one cluster is left. The smallest pair of all is always mutual (k_ploc_nearest orders the
// pairs totally), so every pass merges at least one pair. Nodes are numbered downwards from
// n - 2 in the order they are made, so that the last one, the root, is node 0 as in the radix tree.
__device__ __forceinline__ float union_area(float4 alo, float4 ahi, float4 blo, float4 bhi) {
    const float dx = fmaxf(ahi.x, bhi.x) - fminf(alo.x, blo.x), dy = fmaxf(ahi.y, bhi.y) - fminf(alo.y, blo.y),
                dz = fmaxf(ahi.z, bhi.z) - fminf(alo.z, blo.z);
    return dx * dy + dy * dz + dz * dx;
}
__global__ void k_ploc_nearest(int count, int radius, const int *__restrict__ cl, const float4 *__restrict__ lo,
                               const float4 *__restrict__ hi, int *__restrict__ nearest) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int bi = cl[i];
    const float4 alo = lo[bi], ahi = hi[bi];
    float best = __builtin_inff();
    int best_j = -1;
    const int j0 = max(0, i - radius), j1 = min(count - 1, i + radius);
    for (int j = j0; j <= j1; ++j) {
        if (j == i) continue;
        const int bj = cl[j];
        const float a = union_area(alo, ahi, lo[bj], hi[bj]);  // symmetric in its arguments, bit for bit
        // Pairs are ordered by (surface, not an aligned pair {2k, 2k+1}, lower position, upper position) — a total
        // order on unordered pairs, so the smallest pair of all is the choice of both its members. The aligned-pair
        // rule matters for runs of identical boxes (coincident spheres): they pair off and the subtree stays log-deep,
        // where "lowest position wins" merged one pair per pass into a chain (depth 70 for 250 coincident spheres).
        bool better = a < best;
        if (a == best) {
            const int t_new = (i ^ 1) == j ? 0 : 1, t_old = (i ^ 1) == best_j ? 0 : 1;
            const int lo_new = min(i, j), hi_new = max(i, j), lo_old = min(i, best_j), hi_old = max(i, best_j);
            better = t_new != t_old ? t_new < t_old : (lo_new != lo_old ? lo_new < lo_old : hi_new < hi_old);
        }
        if (better) best = a, best_j = j;
    }
    nearest[i] = best_j;
}
// keep[i] = 0 for the upper cluster of a mutual pair (it disappears into the lower one), merge[i] = 1 for the lower
__global__ void k_ploc_flags(int count, const int *__restrict__ nearest, int *__restrict__ keep, int *__restrict__ merge) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int j = nearest[i];
    const bool mutual = j >= 0 && nearest[j] == i;
    keep[i] = (mutual && j < i) ? 0 : 1;
    merge[i] = (mutual && i < j) ? 1 : 0;
}
__global__ void k_ploc_merge(int count, int n, int node_top, const int *__restrict__ cl, const int *__restrict__ nearest,
                             const int *__restrict__ keep, const int *__restrict__ keep_off, const int *__restrict__ merge,
                             const int *__restrict__ merge_off, float4 *lo, float4 *hi, int *__restrict__ left,
                             int *__restrict__ right, int *__restrict__ cl_next) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count || !keep[i]) return;
    int c = cl[i];
    if (merge[i]) {
        const int other = cl[nearest[i]];
        const int node = node_top - merge_off[i];  // this pass makes nodes node_top, node_top - 1, ...
        const float4 alo = lo[c], ahi = hi[c], blo = lo[other], bhi = hi[other];
        lo[node] = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), 0);
        hi[node] = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), 0);
        left[node] = c >= n - 1 ? ~(c - (n - 1)) : c;  // leaf children are stored as ~position
        right[node] = other >= n - 1 ? ~(other - (n - 1)) : other;
        c = node;
    }
    cl_next[keep_off[i]] = c;
}
__global__ void k_ploc_init(int n, int *cl) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) cl[k] = n - 1 + k;
}

// One refit level: an internal node whose children were ready BEFORE this launch gets its box.
__global__ void k_refit_level(int n, const int *__restrict__ left, const int *__restrict__ right, float4 *lo, float4 *hi,
                              const int *__restrict__ ready_in, int *ready_out, int *n_done) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    if (ready_in[i]) return;
    int l = left[i], r = right[i];
    int li = l < 0 ? n - 1 + ~l : l, ri = r < 0 ? n - 1 + ~r : r;
    if (!ready_in[li] || !ready_in[ri]) return;
    float4 a = lo[li], b = lo[ri], c = hi[li], d = hi[ri];
    lo[i] = make_float4(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), 0);
    hi[i] = make_float4(fmaxf(c.x, d.x), fmaxf(c.y, d.y), fmaxf(c.z, d.z), 0);
    ready_out[i] = 1;
    atomicAdd(n_done, 1);
}

__global__ void k_copy_ready(int n_nodes, const int *__restrict__ src, int *dst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_nodes) dst[i] = src[i];
}

// 32-byte node (pt_bvh.h): per child {c.x | c.y << 16, c.z | h.x << 16, h.y | h.z << 16, child}, half floats in node
// units u = (world - ctr) * inv_unit. The exact box is first widened by `margin` node units (f32 rounding of the
// build-side (v - ctr) * inv and of the traversal-side t = fma(u, slope, off), see bvh_build), the centre is rounded to
// the nearest half and the half-size is then rounded UP from what that centre needs to cover the widened box, so
// that the decoded box always contains the exact one: boxes may only prune.
struct Grid {
    float ctr[3], unit[3], inv_unit[3];
    float margin;
};
__device__ __forceinline__ uint32_t half_bits(_Float16 h) { return (uint32_t)__builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ void pack_axis(float blo, float bhi, float ctr, float inv, float margin, uint32_t &c_bits, uint32_t &h_bits) {
    const float ulo = (blo - ctr) * inv - margin, uhi = (bhi - ctr) * inv + margin;
    const _Float16 c = (_Float16)(0.5f * (ulo + uhi));  // round to nearest
    const float cf = (float)c;
    const float need = fmaxf(uhi - cf, cf - ulo) * 1.000001f;  // the two subtractions round, too
    _Float16 h = (_Float16)fminf(need, 60000.0f);
    uint32_t hb = half_bits(h);
    if ((float)h < need) ++hb;  // next half up (positive, finite: the bit pattern is monotonic)
    c_bits = half_bits(c);
    h_bits = hb;
}
__global__ void k_pack(int n, const int *__restrict__ left, const int *__restrict__ right, const float4 *__restrict__ lo,
                       const float4 *__restrict__ hi, Grid G, uint4 *nodes) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int l = left[i], r = right[i];
    int li = l < 0 ? n - 1 + ~l : l, ri = r < 0 ? n - 1 + ~r : r;
    float4 a = lo[li], b = hi[li], c = lo[ri], d = hi[ri];
    auto pack = [&](float4 blo, float4 bhi, int child) {
        uint32_t cx, cy, cz, hx, hy, hz;
        pack_axis(blo.x, bhi.x, G.ctr[0], G.inv_unit[0], G.margin, cx, hx);
        pack_axis(blo.y, bhi.y, G.ctr[1], G.inv_unit[1], G.margin, cy, hy);
        pack_axis(blo.z, bhi.z, G.ctr[2], G.inv_unit[2], G.margin, cz, hz);
        return make_uint4(cx | (cy << 16), cz | (hx << 16), hy | (hz << 16), (uint32_t)child);
    };
    nodes[i * 2 + 0] = pack(a, b, l);
    nodes[i * 2 + 1] = pack(c, d, r);
}

// ---- 4-wide collapse (pt_bvh.h: "Wide nodes") ---------------------------------------------------------------------
constexpr int kNoChild = INT32_MIN;  // unused slot of a wide node

// One level. Thread i turns binary node queue[i] into a wide node: its two children, then again and again the
// internal child with the largest box surface replaced by ITS two children, until there are four children or only
// leaves are left (children stay in Morton order). children[4i+k]: >= 0 a binary internal node (it becomes a wide
// node of the next level), < 0 the leaf ~position, kNoChild unused; flags[4i+k] = 1 for the internal ones, whose
// exclusive prefix sum is their place in the next level.
__global__ void k_wide_expand(int count, const int *__restrict__ queue, const int *__restrict__ left, const int *__restrict__ right,
                              const float4 *__restrict__ lo, const float4 *__restrict__ hi, int *__restrict__ children,
                              int *__restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int b = queue[i];
    int c[4] = {left[b], right[b], kNoChild, kNoChild};
    int m = 2;
    while (m < 4) {
        int best = -1;
        float best_area = -1.0f;
        for (int k = 0; k < m; ++k) {
            if (c[k] < 0) continue;
            const float4 a = lo[c[k]], z = hi[c[k]];
            const float dx = z.x - a.x, dy = z.y - a.y, dz = z.z - a.z;
            const float area = dx * dy + dy * dz + dz * dx;
            if (area > best_area) best_area = area, best = k;
        }
        if (best < 0) break;
        const int nd = c[best];
        for (int k = m; k > best + 1; --k) c[k] = c[k - 1];
        c[best] = left[nd];
        c[best + 1] = right[nd];
        ++m;
    }
    for (int k = 0; k < 4; ++k) {
        children[4 * i + k] = c[k];
        flags[4 * i + k] = c[k] >= 0 ? 1 : 0;
    }
}

// Writes the wide nodes of the level (wide_base + i) and the next level's queue. Child references are unified
// indices: sphere position for a leaf, n + wide index for a node (pt_bvh.h).
__global__ void k_wide_emit(int count, int n, int wide_base, int next_base, const int *__restrict__ children,
                            const int *__restrict__ offsets, const float4 *__restrict__ lo, const float4 *__restrict__ hi, Grid G,
                            uint4 *__restrict__ wide, int *__restrict__ queue_next) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    for (int k = 0; k < 4; ++k) {
        const int c = children[4 * i + k];
        uint4 slot;
        if (c == kNoChild) {
            slot = make_uint4(0u, 0xBC00u << 16, 0xBC00u | (0xBC00u << 16), 0u);  // centre 0, half-size -1: no ray enters
        } else {
            const int bi = c < 0 ? n - 1 + ~c : c;
            const float4 blo = lo[bi], bhi = hi[bi];
            uint32_t cx, cy, cz, hx, hy, hz;
            pack_axis(blo.x, bhi.x, G.ctr[0], G.inv_unit[0], G.margin, cx, hx);
            pack_axis(blo.y, bhi.y, G.ctr[1], G.inv_unit[1], G.margin, cy, hy);
            pack_axis(blo.z, bhi.z, G.ctr[2], G.inv_unit[2], G.margin, cz, hz);
            uint32_t ref;
            if (c < 0) {
                ref = (uint32_t)~c;
            } else {
                const int pos = offsets[4 * i + k];
                ref = (uint32_t)(n + next_base + pos);
                queue_next[pos] = c;
            }
            slot = make_uint4(cx | (cy << 16), cz | (hx << 16), hy | (hz << 16), ref);
        }
        wide[(size_t)(wide_base + i) * 4 + k] = slot;
    }
}

__global__ void k_gather(const float4 *__restrict__ sph, const float4 *__restrict__ aux, const uint32_t *__restrict__ idx,
                         uint32_t n, float4 *sph_sorted, float4 *aux_sorted) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t i = idx[k];
    sph_sorted[k] = sph[i];
    aux_sorted[k] = aux[i];
}

struct Tmp {
    void *p = nullptr;
    ~Tmp() {
        if (p) (void)hipFree(p);
    }
};

}  // namespace

#define BVH_TRY(expr)                    \
    do {                                 \
        hipError_t e_ = (expr);          \
        if (e_ != hipSuccess) return e_; \
    } while (0)

// Heap copy of the first kBvhTopLevels levels (pt_bvh.h). One block; level by level.
__global__ __launch_bounds__(1024) void k_top(const uint4 *__restrict__ nodes, uint4 *__restrict__ top) {
    __shared__ int src[kBvhTopSlots];  // node of `nodes` behind each heap slot, -1: none
    const int tid = (int)threadIdx.x;
    if (tid == 0) src[0] = 0;
    __syncthreads();
    for (int level = 0; level < kBvhTopLevels; ++level) {
        const int base = (1 << level) - 1, cnt = 1 << level;
        if (tid < cnt) {
            const int e = base + tid, g = src[e];
            uint4 a = uint4{0, 0, 0, 0}, b = uint4{0, 0, 0, 0};
            int l = -1, r = -1;
            if (g >= 0) {
                a = nodes[g * 2 + 0];
                b = nodes[g * 2 + 1];
                if (level + 1 < kBvhTopLevels) {  // internal children move into the heap as well
                    if ((int)a.w >= 0) {
                        l = (int)a.w;
                        a.w = (uint32_t)(kBvhTopFlag | (2 * e + 1));
                    }
                    if ((int)b.w >= 0) {
                        r = (int)b.w;
                        b.w = (uint32_t)(kBvhTopFlag | (2 * e + 2));
                    }
                }
            }
            if (level + 1 < kBvhTopLevels) {
                src[2 * e + 1] = l;
                src[2 * e + 2] = r;
            }
            top[e * 2 + 0] = a;
            top[e * 2 + 1] = b;
        }
        __syncthreads();
    }
}

// HIPPT_BVH_BUILDER = "lbvh" (default) | "ploc"; HIPPT_BVH_PLOC_RADIUS = neighbours searched on either side (default 8).
// Measured on C4 (100 000 uniformly placed spheres, where Morton splits are already spatial medians): node visits
// per closest / any-hit ray 15.6 / 19.2 with the radix tree, 16.0 / 19.6 with PLOC at radius 4-8, 17.4-18.4 / 20.3-20.6
// at radius 16-100; step time equal at radius 8, 4 % longer at 16 and above. Hence opt-in.
static bool bvh_builder_is_ploc() {
    const char *e = getenv("HIPPT_BVH_BUILDER");
    return e && strcmp(e, "ploc") == 0;
}
static int bvh_ploc_radius() {
    const char *e = getenv("HIPPT_BVH_PLOC_RADIUS");
    const int r = e ? atoi(e) : 8;
    return r < 1 ? 1 : (r > 256 ? 256 : r);
}

hipError_t bvh_build(const float4 *d_sph, const float4 *d_aux, uint32_t n, BvhBuffers &out, hipStream_t st, int *depth_out) {
    if (n < 2) return hipErrorInvalidValue;
    const int N = (int)n, n_nodes = 2 * N - 1;
    Tmp bounds, codes, codes_s, idx, idx_s, sort_tmp, left, right, parent, lo, hi, ready_a, ready_b, done;
    BVH_TRY(hipMalloc(&bounds.p, 6 * 4));
    BVH_TRY(hipMalloc(&codes.p, (size_t)n * 4));
    BVH_TRY(hipMalloc(&codes_s.p, (size_t)n * 4));
    BVH_TRY(hipMalloc(&idx.p, (size_t)n * 4));
    BVH_TRY(hipMalloc(&idx_s.p, (size_t)n * 4));
    BVH_TRY(hipMalloc(&left.p, (size_t)n * 4));
    BVH_TRY(hipMalloc(&right.p, (size_t)n * 4));
    BVH_TRY(hipMalloc(&parent.p, (size_t)n_nodes * 4));
    BVH_TRY(hipMalloc(&lo.p, (size_t)n_nodes * 16));
    BVH_TRY(hipMalloc(&hi.p, (size_t)n_nodes * 16));
    BVH_TRY(hipMalloc(&ready_a.p, (size_t)n_nodes * 4));
    BVH_TRY(hipMalloc(&ready_b.p, (size_t)n_nodes * 4));
    BVH_TRY(hipMalloc(&done.p, 4));
    BVH_TRY(hipMalloc((void **)&out.nodes, (size_t)(N - 1) * 32));
    BVH_TRY(hipMalloc((void **)&out.top, (size_t)kBvhTopSlots * 32));
    BVH_TRY(hipMalloc((void **)&out.sph_sorted, (size_t)n * 16));
    BVH_TRY(hipMalloc((void **)&out.aux_sorted, (size_t)n * 16));
    out.n = n;

    const unsigned bs = 256, g = (n + bs - 1) / bs, gn = (n_nodes + bs - 1) / bs;
    uint32_t init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    BVH_TRY(hipMemcpyAsync(bounds.p, init, sizeof init, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_bounds, dim3(std::min(g, 1024u)), dim3(bs), 0, st, d_sph, n, (uint32_t *)bounds.p);
    hipLaunchKernelGGL(k_morton, dim3(g), dim3(bs), 0, st, d_sph, n, (const uint32_t *)bounds.p, (uint32_t *)codes.p,
                       (uint32_t *)idx.p);
    size_t tmp_bytes = 0;
    BVH_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint32_t *)codes.p, (uint32_t *)codes_s.p, (uint32_t *)idx.p,
                                      (uint32_t *)idx_s.p, n, 0, 30, st));
    BVH_TRY(hipMalloc(&sort_tmp.p, tmp_bytes ? tmp_bytes : 4));
    BVH_TRY(rocprim::radix_sort_pairs(sort_tmp.p, tmp_bytes, (uint32_t *)codes.p, (uint32_t *)codes_s.p, (uint32_t *)idx.p,
                                      (uint32_t *)idx_s.p, n, 0, 30, st));
    hipLaunchKernelGGL(k_gather, dim3(g), dim3(bs), 0, st, d_sph, d_aux, (const uint32_t *)idx_s.p, n, out.sph_sorted,
                       out.aux_sorted);
    // The tree (radix tree, or PLOC when asked for), the leaf boxes and the bottom-up refit. A PLOC tree can come out
    // deeper than the traversal stacks allow (concentric spheres of different radii chain: each pass merges the two
    // smallest); the radix tree, whose depth is bounded by the key length, is built instead then.
    int depth = 0;
    for (int attempt = 0;; ++attempt) {
        const bool ploc = bvh_builder_is_ploc() && attempt == 0;
        hipLaunchKernelGGL(k_hierarchy, dim3(g), dim3(bs), 0, st, (const uint32_t *)codes_s.p, N, (int *)left.p, (int *)right.p,
                           (int *)parent.p);
        BVH_TRY(hipMemsetAsync(ready_a.p, 0, (size_t)n_nodes * 4, st));
        hipLaunchKernelGGL(k_leaf_boxes, dim3(g), dim3(bs), 0, st, out.sph_sorted, out.aux_sorted, N, (float4 *)lo.p, (float4 *)hi.p,
                           (int *)ready_a.p);
        if (ploc) {
            // the radix tree's left / right arrays are replaced; everything after this (refit, packing, collapse) is unchanged
            Tmp cl_a, cl_b, nearest, keep, keep_off, merge, merge_off, scan_tmp;
            for (Tmp *t : {&cl_a, &cl_b, &nearest, &keep, &keep_off, &merge, &merge_off}) BVH_TRY(hipMalloc(&t->p, (size_t)n * 4));
            size_t scan_bytes = 0;
            BVH_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, (int *)keep.p, (int *)keep_off.p, 0, (size_t)n, rocprim::plus<int>(), st));
            BVH_TRY(hipMalloc(&scan_tmp.p, scan_bytes ? scan_bytes : 4));
            hipLaunchKernelGGL(k_ploc_init, dim3(g), dim3(bs), 0, st, N, (int *)cl_a.p);
            const int radius = bvh_ploc_radius();
            int count = N, node_top = N - 2;
            for (int pass = 0; count > 1; ++pass) {
                if (pass > N) return hipErrorUnknown;  // (every pass merges at least one pair)
                const unsigned gc = ((unsigned)count + bs - 1) / bs;
                hipLaunchKernelGGL(k_ploc_nearest, dim3(gc), dim3(bs), 0, st, count, radius, (const int *)cl_a.p, (const float4 *)lo.p,
                                   (const float4 *)hi.p, (int *)nearest.p);
                hipLaunchKernelGGL(k_ploc_flags, dim3(gc), dim3(bs), 0, st, count, (const int *)nearest.p, (int *)keep.p, (int *)merge.p);
                BVH_TRY(rocprim::exclusive_scan(scan_tmp.p, scan_bytes, (int *)keep.p, (int *)keep_off.p, 0, (size_t)count,
                                                rocprim::plus<int>(), st));
                BVH_TRY(rocprim::exclusive_scan(scan_tmp.p, scan_bytes, (int *)merge.p, (int *)merge_off.p, 0, (size_t)count,
                                                rocprim::plus<int>(), st));
                hipLaunchKernelGGL(k_ploc_merge, dim3(gc), dim3(bs), 0, st, count, N, node_top, (const int *)cl_a.p, (const int *)nearest.p,
                                   (const int *)keep.p, (const int *)keep_off.p, (const int *)merge.p, (const int *)merge_off.p,
                                   (float4 *)lo.p, (float4 *)hi.p, (int *)left.p, (int *)right.p, (int *)cl_b.p);
                int last[4] = {0, 0, 0, 0};
                BVH_TRY(hipMemcpyAsync(&last[0], (int *)keep_off.p + count - 1, 4, hipMemcpyDeviceToHost, st));
                BVH_TRY(hipMemcpyAsync(&last[1], (int *)keep.p + count - 1, 4, hipMemcpyDeviceToHost, st));
                BVH_TRY(hipMemcpyAsync(&last[2], (int *)merge_off.p + count - 1, 4, hipMemcpyDeviceToHost, st));
                BVH_TRY(hipMemcpyAsync(&last[3], (int *)merge.p + count - 1, 4, hipMemcpyDeviceToHost, st));
                BVH_TRY(hipStreamSynchronize(st));
                const int kept = last[0] + last[1], merged = last[2] + last[3];
                if (merged < 1 || kept != count - merged || node_top - merged < -1) return hipErrorUnknown;
                count = kept;
                node_top -= merged;
                std::swap(cl_a.p, cl_b.p);
            }
            if (node_top != -1) return hipErrorUnknown;  // n - 1 nodes made, the last of them node 0
            BVH_TRY(hipStreamSynchronize(st));
        }
        // bottom-up, one launch per level; the number of levels is the tree depth
        int total = 0;
        depth = 0;
        for (; total < N - 1; ++depth) {
            if (ploc && depth > BVH_STACK) break;         // too deep already: no need to finish the refit
            if (depth > 4 * 64) return hipErrorUnknown;  // a level that completes no node: malformed tree
            hipLaunchKernelGGL(k_copy_ready, dim3(gn), dim3(bs), 0, st, n_nodes, (const int *)ready_a.p, (int *)ready_b.p);
            BVH_TRY(hipMemsetAsync(done.p, 0, 4, st));
            hipLaunchKernelGGL(k_refit_level, dim3(g), dim3(bs), 0, st, N, (const int *)left.p, (const int *)right.p, (float4 *)lo.p,
                               (float4 *)hi.p, (const int *)ready_a.p, (int *)ready_b.p, (int *)done.p);
            int d = 0;
            BVH_TRY(hipMemcpyAsync(&d, done.p, 4, hipMemcpyDeviceToHost, st));
            BVH_TRY(hipStreamSynchronize(st));
            if (d == 0) return hipErrorUnknown;
            total += d;
            std::swap(ready_a.p, ready_b.p);
        }
        if (ploc && depth > BVH_STACK) continue;
        break;
    }
    // node units over the root box (internal node 0): its centre is the origin, its half-extents are 1024 units
    float4 root_lo, root_hi;
    BVH_TRY(hipMemcpyAsync(&root_lo, lo.p, 16, hipMemcpyDeviceToHost, st));
    BVH_TRY(hipMemcpyAsync(&root_hi, hi.p, 16, hipMemcpyDeviceToHost, st));
    BVH_TRY(hipStreamSynchronize(st));
    Grid G;
    const float rl[3] = {root_lo.x, root_lo.y, root_lo.z}, rh[3] = {root_hi.x, root_hi.y, root_hi.z};
    double max_abs = 0.0, min_unit = INFINITY;
    for (int k = 0; k < 3; ++k) {
        double half = std::max(0.5 * ((double)rh[k] - (double)rl[k]), 1e-30);
        double unit = half / 1024.0;
        G.ctr[k] = (float)(0.5 * ((double)rh[k] + (double)rl[k]));
        G.unit[k] = (float)unit;
        G.inv_unit[k] = (float)(1.0 / unit);
        max_abs = std::max({max_abs, std::fabs((double)rl[k]), std::fabs((double)rh[k])});
        min_unit = std::min(min_unit, unit);
    }
    if (!std::isfinite(max_abs) || !std::isfinite(min_unit)) return hipErrorInvalidValue;
    // f32 rounding of (v - ctr) * inv at build time: a few ulps of the largest coordinate, in node units; plus 1/32 unit
    // for the ray-side rounding of t(u) = fma(u, slope, off) and of the face offsets (pt_bvh.h), which covers ray
    // origins up to ~250 root half-extents away from the root's centre
    G.margin = (float)(1.0 / 32.0 + std::min(64.0, 8.0 * max_abs * 1.2e-7 / min_unit));
    for (int k = 0; k < 3; ++k) {
        out.grid_lo[k] = G.ctr[k];
        out.grid_cell[k] = G.unit[k];
    }
    hipLaunchKernelGGL(k_pack, dim3(g), dim3(bs), 0, st, N, (const int *)left.p, (const int *)right.p, (const float4 *)lo.p,
                       (const float4 *)hi.p, G, out.nodes);
    hipLaunchKernelGGL(k_top, dim3(1), dim3(1024), 0, st, (const uint4 *)out.nodes, out.top);
    {   // 4-wide collapse, breadth-first; the level sizes come back to the host (one small copy per level)
        Tmp queue_a, queue_b, children, flags, offsets, scan_tmp;
        const size_t slots = (size_t)(N - 1) * 4;
        BVH_TRY(hipMalloc(&queue_a.p, (size_t)(N - 1) * 4));
        BVH_TRY(hipMalloc(&queue_b.p, (size_t)(N - 1) * 4));
        BVH_TRY(hipMalloc(&children.p, slots * 4));
        BVH_TRY(hipMalloc(&flags.p, slots * 4));
        BVH_TRY(hipMalloc(&offsets.p, slots * 4));
        BVH_TRY(hipMalloc((void **)&out.wide, (size_t)(N - 1) * 64));
        size_t scan_bytes = 0;
        BVH_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, (int *)flags.p, (int *)offsets.p, 0, slots, rocprim::plus<int>(), st));
        BVH_TRY(hipMalloc(&scan_tmp.p, scan_bytes ? scan_bytes : 4));
        BVH_TRY(hipMemsetAsync(queue_a.p, 0, 4, st));  // the root
        int count = 1, base = 0, levels = 0;
        while (count > 0) {
            if (levels > depth + 1 || base + count > N - 1) return hipErrorUnknown;
            const unsigned gc = ((unsigned)count + bs - 1) / bs;
            hipLaunchKernelGGL(k_wide_expand, dim3(gc), dim3(bs), 0, st, count, (const int *)queue_a.p, (const int *)left.p,
                               (const int *)right.p, (const float4 *)lo.p, (const float4 *)hi.p, (int *)children.p, (int *)flags.p);
            BVH_TRY(rocprim::exclusive_scan(scan_tmp.p, scan_bytes, (int *)flags.p, (int *)offsets.p, 0, (size_t)count * 4,
                                            rocprim::plus<int>(), st));
            int last[2] = {0, 0};
            BVH_TRY(hipMemcpyAsync(&last[0], (int *)offsets.p + (size_t)count * 4 - 1, 4, hipMemcpyDeviceToHost, st));
            BVH_TRY(hipMemcpyAsync(&last[1], (int *)flags.p + (size_t)count * 4 - 1, 4, hipMemcpyDeviceToHost, st));
            BVH_TRY(hipStreamSynchronize(st));
            const int next_count = last[0] + last[1];
            hipLaunchKernelGGL(k_wide_emit, dim3(gc), dim3(bs), 0, st, count, N, base, base + count, (const int *)children.p,
                               (const int *)offsets.p, (const float4 *)lo.p, (const float4 *)hi.p, G, out.wide, (int *)queue_b.p);
            base += count;
            count = next_count;
            std::swap(queue_a.p, queue_b.p);
            ++levels;
        }
        out.n_wide = (uint32_t)base;
        out.wide_levels = (uint32_t)levels;
        BVH_TRY(hipStreamSynchronize(st));  // the temporaries are freed when this block ends
    }
    BVH_TRY(hipStreamSynchronize(st));
    BVH_TRY(hipGetLastError());
    if (depth_out) *depth_out = depth;
    return hipSuccess;
}

void bvh_free(BvhBuffers &b) {
    if (b.nodes) (void)hipFree(b.nodes);
    if (b.top) (void)hipFree(b.top);
    if (b.wide) (void)hipFree(b.wide);
    if (b.sph_sorted) (void)hipFree(b.sph_sorted);
    if (b.aux_sorted) (void)hipFree(b.aux_sorted);
    b = BvhBuffers{};
}

}  // namespace pt
