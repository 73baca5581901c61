// pt_bvh.h — device-side traversal of the GPU-built LBVH over the sphere table.
//
// Used only for many-sphere scenes (BASELINE config C4); the reference has no
// acceleration structure: its closest hit is the brute-force loop of
// src/renderer.rs:356-362, and the traversal below must return exactly that
// loop's answer (same t, same object, first-in-scene-order on ties). It does so
// by construction: the BVH only prunes — every sphere that can win is tested with
// the same sphere_t() the brute-force kernels use, boxes are conservative, and the
// winner is chosen by (t, scene order).
#pragma once
#include "pt_device.h"

namespace pt {

// Node = 2 x uint4 (32 B): per child {c.x | c.y << 16, c.z | h.x << 16, h.y | h.z << 16, child} — the child box as
// CENTRE c and HALF-SIZE h, six IEEE half floats in node units (u = (world - g_lo) / g_cell, g_lo = centre of the
// root box, |u| <= 1024). Why this format: the traversal kernels are bound by VALU issue (~80 % busy on C4), and
//   * v_fma_mix_f32 reads a half straight out of either half of a register, so decoding costs nothing, where a
//     16-bit integer grid cost one half-rate v_cvt per face (12 per node);
//   * with t_c = c * slope + off the two faces of an axis are t_c -+ h * |slope|: three full-rate FMAs per axis
//     instead of two FMAs plus a half-rate min and max to order the faces.
// The centre is rounded to nearest (<= 1/4 node unit at the far ends of the root box, less towards its centre, i.e.
// exactly: half an ulp of a half float) and the half-size is rounded UP
// after that error and the build / traversal rounding margins have been added to it (pt_bvh_build.hip: k_pack), so a
// decoded box always contains the exact one: boxes may only prune.
// child >= 0: internal node index; child < 0: leaf ~child = position in the
// Morton-sorted sphere arrays.
// The first kBvhTopLevels levels of the tree are ALSO kept as a complete binary heap (slot 0 = root, children of
// slot e at 2e+1 / 2e+2, unused slots zero) so that the queue pipeline's traversal kernels can hold them in LDS: a
// child pointer with kBvhTopFlag set is a heap slot, any other pointer is a node of `nodes` (>= 0) or a leaf (< 0) as
// before. Every lane of a traversal wave chases its own pointer, and 16-byte gathers cost the vector L1 about one
// lane per clock and CU (tools/gather_microbench.hip: 0.43 lane-loads per clock per CU from a 3 MB table, 4.7 from
// LDS). Measured on C4: round 2 began with 6 levels (63 nodes, 2 KB of LDS per block, 19 % of the node
// visits; +2.7 %) because more levels cost resident blocks; since the traversal stack keeps only what fits beside the
// heap in LDS and spills the rest (pt_wavefront_bvh.hip), 8 levels (255 nodes, 8 KB, 24 % of the visits, 12 stack
// entries in LDS) are worth another +1.2 %; 9 levels leave 4 stack entries: -12 %.
#ifndef PT_BVH_TOP_LEVELS
#define PT_BVH_TOP_LEVELS 8
#endif
constexpr int kBvhTopLevels = PT_BVH_TOP_LEVELS;
constexpr int kBvhTopSlots = (1 << kBvhTopLevels) - 1;  // 255 nodes, 8 160 bytes
constexpr int kBvhTopFlag = 1 << 30;

// WIDE NODES (round 3; queue pipeline only, pt_wavefront_bvh.hip). The same radix tree collapsed into nodes with up to
// four children: a node keeps absorbing its child with the largest surface area until it has four (or only leaves
// are left), so a ray makes about half as many dependent node visits. Node = 4 x uint4 (64 B, one half of a 128-byte
// line): slot k = {c.x | c.y << 16, c.z | h.x << 16, h.y | h.z << 16, child} with the same half-float centre /
// half-size boxes as above. `child` is a UNIFIED index u: u < n is the sphere at sorted position u, u >= n is wide
// node u - n; an unused slot has a box no ray can enter (half-size -1). Nodes are numbered breadth-first and the
// children of a node are consecutive, so (i) the first `wide_top` nodes, which the traversal kernels keep in LDS, are
// simply a prefix of the array, and (ii) siblings share cache lines.
struct Bvh {
    const uint4 *__restrict__ nodes;
    const uint4 *__restrict__ top;   // kBvhTopSlots x 2 uint4
    const float4 *__restrict__ sph;  // sorted {c, r*r}
    const float4 *__restrict__ aux;  // sorted {1/r, material, scene order, 0}
    uint32_t n;                      // number of spheres (>= 2)
    float g_lo[3], g_cell[3];        // world = g_lo + u * g_cell (u: node units of the child boxes)
    const uint4 *__restrict__ wide;  // n_wide x 4 uint4
    uint32_t n_wide;
};

// Child boxes stay in NODE units; the ray is brought into node units once, when it starts its traversal:
// world = g_lo + u * g_cell, so the slab distance (world - ro) / rd is
//     t(u) = u * (cell / rd) + (g_lo - ro) / rd = fma(u, slope, off).
// Rounding: each t is off by a few 1e-7 (|u| + |ro - g_lo| / cell) node units, far below the margin added to the
// half-sizes (>= 1/32 node unit, pt_bvh_build.hip), so a box still never rejects a hit its sphere test accepts.
struct NodeBoxes {
    F3 lc, lh, rc, rh;  // centre and half-size of the left / right child box, node units
    int left, right;
};
__device__ __forceinline__ NodeBoxes decode_node(uint4 a, uint4 b) {
    // (float) of a half taken from the low / high 16 bits: folded into v_fma_mix_f32's operand selection
    auto lo16 = [](uint32_t v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v & 0xffffu)); };
    auto hi16 = [](uint32_t v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v >> 16)); };
    NodeBoxes nb;
    nb.lc = F3{lo16(a.x), hi16(a.x), lo16(a.y)};
    nb.lh = F3{hi16(a.y), lo16(a.z), hi16(a.z)};
    nb.rc = F3{lo16(b.x), hi16(b.x), lo16(b.y)};
    nb.rh = F3{hi16(b.y), lo16(b.z), hi16(b.z)};
    nb.left = (int)a.w;
    nb.right = (int)b.w;
    return nb;
}
__device__ __forceinline__ NodeBoxes load_node(const Bvh &B, int node) {
    return decode_node(B.nodes[node * 2 + 0], B.nodes[node * 2 + 1]);
}

constexpr int BVH_STACK = 48;  // entries per lane (LDS); deeper trees raise an error at build time

struct BoxRay {
    F3 slope, off, aslope;  // t(u) = fma(u, slope, off) per axis; aslope = |slope|
};
__device__ __forceinline__ BoxRay make_box_ray(const Bvh &B, F3 ro, F3 rd) {
    auto safe = [](float d) { return fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d; };
    const F3 inv = F3{rcp(safe(rd.x)), rcp(safe(rd.y)), rcp(safe(rd.z))};
    const F3 slope = F3{B.g_cell[0] * inv.x, B.g_cell[1] * inv.y, B.g_cell[2] * inv.z};
    return BoxRay{slope, F3{(B.g_lo[0] - ro.x) * inv.x, (B.g_lo[1] - ro.y) * inv.y, (B.g_lo[2] - ro.z) * inv.z},
                  F3{fabsf(slope.x), fabsf(slope.y), fabsf(slope.z)}};
}
// Entry distance of the ray into the box (centre c, half-size h, node units), +inf if it misses or lies beyond
// t_max / behind the origin.
__device__ __forceinline__ float box_entry(const BoxRay &b, F3 c, F3 h, float t_max) {
    const float cx = fmaf(c.x, b.slope.x, b.off.x), cy = fmaf(c.y, b.slope.y, b.off.y), cz = fmaf(c.z, b.slope.z, b.off.z);
    const float nx = fmaf(-h.x, b.aslope.x, cx), fx = fmaf(h.x, b.aslope.x, cx);
    const float ny = fmaf(-h.y, b.aslope.y, cy), fy = fmaf(h.y, b.aslope.y, cy);
    const float nz = fmaf(-h.z, b.aslope.z, cz), fz = fmaf(h.z, b.aslope.z, cz);
    float tmin = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
    float tmax = fminf(fminf(fx, fy), fminf(fz, t_max));
    return tmin <= tmax ? tmin : __builtin_inff();
}

// One slot of a wide node: entry distance of the ray into the child box, +inf on a miss (an unused slot always misses).
__device__ __forceinline__ float slot_entry(const BoxRay &b, uint4 q, float t_max) {
    auto lo16 = [](uint32_t v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v & 0xffffu)); };
    auto hi16 = [](uint32_t v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v >> 16)); };
    return box_entry(b, F3{lo16(q.x), hi16(q.x), lo16(q.y)}, F3{hi16(q.y), lo16(q.z), hi16(q.z)}, t_max);
}

// The same test, returning whether the box is entered and the entry distance separately (one compare and no select
// fewer than going through +inf).
__device__ __forceinline__ bool slot_hit(const BoxRay &b, uint4 q, float t_max, float &t_entry) {
    auto lo16 = [](uint32_t v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v & 0xffffu)); };
    auto hi16 = [](uint32_t v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v >> 16)); };
    const F3 c = F3{lo16(q.x), hi16(q.x), lo16(q.y)}, h = F3{hi16(q.y), lo16(q.z), hi16(q.z)};
    const float cx = fmaf(c.x, b.slope.x, b.off.x), cy = fmaf(c.y, b.slope.y, b.off.y), cz = fmaf(c.z, b.slope.z, b.off.z);
    // |slope| as a source modifier of the instruction rather than three more registers of per-ray state
    const float ax = fabsf(b.slope.x), ay = fabsf(b.slope.y), az = fabsf(b.slope.z);
    const float nx = fmaf(-h.x, ax, cx), fx = fmaf(h.x, ax, cx);
    const float ny = fmaf(-h.y, ay, cy), fy = fmaf(h.y, ay, cy);
    const float nz = fmaf(-h.z, az, cz), fz = fmaf(h.z, az, cz);
    t_entry = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
    return t_entry <= fminf(fminf(fx, fy), fminf(fz, t_max));
}

// Closest sphere hit. best_t/best_leaf/best_ord are updated in place (start: inf, -1, ~0u).
// stack: this lane's LDS column, element k at stack[k * stride].
__device__ __forceinline__ void bvh_closest(const Bvh &B, const RayK &r, float &best_t, int &best_leaf,
                                            uint32_t &best_ord, uint32_t *stack, uint32_t stride, uint32_t &n_tests) {
    BoxRay br = make_box_ray(B, r.ro, r.rd);
    auto leaf = [&](int lf, float4 sph) {
        float t = sphere_t(r, sph);
        ++n_tests;
        if (t > 0.0f && t <= best_t) {
            uint32_t ord = __float_as_uint(B.aux[lf].z);
            if (t < best_t || ord < best_ord) {  // renderer.rs:359-361: first minimum in scene order
                best_t = t;
                best_leaf = lf;
                best_ord = ord;
            }
        }
    };
    int node = 0, sp = 0;
    for (;;) {
        NodeBoxes nb = load_node(B, node);
        int left = nb.left, right = nb.right;
        // boxes are tested against best_t inclusively: a sphere that ties must still be visited
        float tl = box_entry(br, nb.lc, nb.lh, best_t);
        float tr = box_entry(br, nb.rc, nb.rh, best_t);
        bool hl = tl < __builtin_inff(), hr = tr < __builtin_inff();
        const bool ll = hl && left < 0, rl = hr && right < 0;
        if (ll || rl) {  // both leaf spheres are fetched at once (one memory latency, not two)
            const int il = ll ? ~left : ~right, ir = rl ? ~right : ~left;
            const float4 sl = B.sph[il], sr = B.sph[ir];
            leaf(il, sl);
            if (ll && rl) leaf(ir, sr);
            if (ll) hl = false;
            if (rl) hr = false;
        }
        if (hl && hr) {
            bool lfirst = tl <= tr;
            stack[sp * stride] = (uint32_t)(lfirst ? right : left);
            ++sp;
            node = lfirst ? left : right;
        } else if (hl || hr) {
            node = hl ? left : right;
        } else {
            if (sp == 0) break;
            --sp;
            node = (int)stack[sp * stride];
        }
    }
}

// Any sphere with t > 0 and t*t < lim (renderer.rs:46).
__device__ __forceinline__ bool bvh_any(const Bvh &B, const RayK &r, float lim, uint32_t *stack, uint32_t stride,
                                        uint32_t &n_tests) {
    BoxRay br = make_box_ray(B, r.ro, r.rd);
    float t_max = fsqrt(lim) * 1.000001f;
    bool occ = false;
    int node = 0, sp = 0;
    for (;;) {
        NodeBoxes nb = load_node(B, node);
        int left = nb.left, right = nb.right;
        bool hl = box_entry(br, nb.lc, nb.lh, t_max) < __builtin_inff();
        bool hr = box_entry(br, nb.rc, nb.rh, t_max) < __builtin_inff();
        const bool ll = hl && left < 0, rl = hr && right < 0;
        if (ll || rl) {  // both leaf spheres at once; a lone leaf is read twice and counted once
            const int il = ll ? ~left : ~right, ir = rl ? ~right : ~left;
            const float4 sl = B.sph[il], sr = B.sph[ir];
            const float t0 = sphere_t(r, sl), t1 = sphere_t(r, sr);
            n_tests += (ll && rl) ? 2u : 1u;
            occ = occ | ((t0 > 0.0f) & (t0 * t0 < lim)) | ((t1 > 0.0f) & (t1 * t1 < lim));
            if (ll) hl = false;
            if (rl) hr = false;
        }
        if (occ) break;
        if (hl && hr) {
            stack[sp * stride] = (uint32_t)right;
            ++sp;
            node = left;
        } else if (hl || hr) {
            node = hl ? left : right;
        } else {
            if (sp == 0) break;
            --sp;
            node = (int)stack[sp * stride];
        }
    }
    return occ;
}

}  // namespace pt
