// pt_trace.h — per-vertex building blocks shared by the megakernel (pt_kernels.hip) and the
// wavefront pipeline (pt_wavefront.hip): closest hit dispatch (table loops or LBVH) and
// next-event estimation (renderer.rs:19-85).
#pragma once
#include "pt_bvh.h"
#include "pt_device.h"

namespace pt {

// BVH = true: spheres through the LBVH (pt_bvh.h), planes through the table loops.
template <class Tbl, bool COUNT, bool BVH>
struct Tracer {
    const Params &P;
    Tbl tbl;
    Bvh bvh;
    uint32_t *stack;  // this lane's LDS traversal stack column (BVH only), stride 256
    uint32_t cnt[HIPPT_N_COUNTERS];

    __device__ __forceinline__ Tracer(const Params &p, Tbl t, Bvh b, uint32_t *stk) : P(p), tbl(t), bvh(b), stack(stk) {
        if (COUNT)
            for (int i = 0; i < HIPPT_N_COUNTERS; ++i) cnt[i] = 0;
    }

    // renderer.rs:356-362
    __device__ __forceinline__ HitRec closest(const RayK &ray) {
        if constexpr (BVH) {
            float best_t = __builtin_inff();
            int best = -1;
            uint32_t best_ord = 0xffffffffu;
            uint32_t nt = 0;
            bvh_closest(bvh, ray, best_t, best, best_ord, stack, 256u, nt);
            bump(HIPPT_C_EXTEND_SPHERE_TESTS, nt);
            closest_planes(tbl, P, ray, best_t, best, [&](int) { return best_ord; });
            HitRec h{best_t, best};
            refine_hit(tbl, P, ray.ro, ray.rd, h, [&](int id) { return bvh.sph[id]; });
            return h;
        } else {
            return closest_hit(tbl, P, ray);
        }
    }
    __device__ __forceinline__ void finish(const RayK &ray, HitRec h, F3 &hit, F3 &n, uint32_t &mat, uint32_t &order) {
        if constexpr (BVH) {
            finish_hit(tbl, P, ray, h, hit, n, mat, order, [&](int id, float4 &s, float4 &aux) {
                s = bvh.sph[id];
                aux = bvh.aux[id];
            });
        } else {
            finish_hit(tbl, P, ray, h, hit, n, mat, order);
        }
    }
    __device__ __forceinline__ void bump(int k, uint32_t v = 1) {
        if (COUNT) cnt[k] += v;
    }

    // renderer.rs:19-85. All lights, 4 shadow samples each, uniforms drawn
    // before the visibility test (quirk Q10).
    __device__ __forceinline__ F3 direct_light(F3 x, const ShadeK &sk, Rng &rng) {
        F3 total = F3{0.0f, 0.0f, 0.0f};
        bump(HIPPT_C_NEE_VERTICES);
        for (uint32_t li = 0; li < P.n_light; ++li) {
            uint32_t o = P.light_off + li * LIGHT_F4;
            float4 l0 = tbl.ld(o), l1 = tbl.ld(o + 1), l2 = tbl.ld(o + 2), l3 = tbl.ld(o + 3), l4 = tbl.ld(o + 4);
            F3 contrib = F3{0.0f, 0.0f, 0.0f};
            // the light's 4 shadow samples (:35-47) are drawn first, then tested together
            F3 l[4];
            float id2[4], tmax[4];  // 1 / dist2; sqrt(dist2 * 0.999) (:46)
            bool occ[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float e1 = rng.gen() - 0.5f;
                float e2 = rng.gen() - 0.5f;
                F3 lp = f3(l0) + f3(l1) * e1 + f3(l2) * e2;
                F3 lvec = lp - x;
                float dist2 = dot(lvec, lvec);
                float inv = rsq(dist2);
                l[s] = lvec * inv;
                id2[s] = inv * inv;
                tmax[s] = (dist2 * inv) * 0.99949987493746f;  // |lvec| sqrt(0.999)
            }
            bump(HIPPT_C_SHADOW_RAYS, 4);
            if constexpr (BVH) {
#pragma unroll 1
                for (int s = 0; s < 4; ++s) {
                    // a sample that faces away contributes nothing (:50,:55) whether occluded or not: no traversal
                    bool facing = (dot(sk.n, l[s]) > 0.0f) & (-dot(l[s], f3(l4)) > 0.0f);
                    uint32_t nt = 0;
                    occ[s] = facing ? bvh_any(bvh, make_ray(madd(x, l[s], 1e-4f), l[s]), tmax[s] * tmax[s], stack, 256u, nt) : true;
                    bump(HIPPT_C_SHADOW_SPHERE_TESTS, nt);
                }
            }
            occluded_multi<Tbl, 4, COUNT>(tbl, P, x, 1e-4f, l, tmax, occ, cnt[HIPPT_C_SHADOW_SPHERE_TESTS],
                                          cnt[HIPPT_C_SHADOW_PLANE_TESTS], !BVH);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (!occ[s]) {
                    bool lit;
                    F3 c = nee_eval(sk, l[s], id2[s], f3(l4), l0.w, f3(l3), lit);
                    if (lit) bump(HIPPT_C_LIT_EVALS);
                    contrib = contrib + c;
                }
            }
            total = madd(total, contrib, 0.25f);  // :82
        }
        return total;
    }
};

}  // namespace pt
