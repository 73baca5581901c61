// pt_trace.h — per-vertex building blocks shared by the megakernel (pt_kernels.hip) and the
// wavefront pipeline (pt_wavefront.hip): closest hit dispatch (table loops or LBVH) and
// next-event estimation (renderer.rs:19-85) in two forms: Tracer::direct_light, the lights loop as the reference
// orders it (one lane, one vertex), and Tracer::direct_light_shared, the same work shared across the lanes of a wave
// (default of the table-scene megakernel; DESIGN.md §3 "Wave-shared NEE").
#pragma once
#include "pt_bvh.h"
#include "pt_device.h"

namespace pt {

// Per-wave LDS of Tracer::direct_light_shared.
#ifndef PT_HINT_SLACK
#define PT_HINT_SLACK 128u
#endif
// 0: shadow loops test every plane row (A/B builds, tools/build_variant.sh)
#ifndef PT_CULL
#define PT_CULL 1
#endif
constexpr uint32_t kHintLights = 8;  // lights that keep a remembered occluder
constexpr uint32_t kNoHint = 0xffffffffu;
constexpr uint32_t kPairCap = 64;    // pending shadow rays (a light whose rays do not fit beside them has them tested first)
constexpr uint32_t kPairBatch = 64;
constexpr uint32_t kPairWords = 18;  // x, l, tmax, 1/dist2, n, v, T, {material | pixel slot << 23 | light << 29}
struct NeeShare {
    uint32_t hint[kHintLights];  // per light: the object that last blocked all 4 samples of most lanes
    uint32_t pending;            // rays in the list
    float pair[kPairWords * kPairCap];  // word-major list of shadow rays not yet tested against the whole scene
};

__device__ __forceinline__ uint32_t rank_in(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// BVH = true: spheres through the LBVH (pt_bvh.h), planes through the table loops.
// EXEC (with COUNT): the counters of the shadow tests and BRDF evaluations report what THIS code executes — the
// wave-shared lights loop with its remembered occluders, culled plane rows and deferred evaluations — instead of what
// the reference's loop would (SURVEY §8d credits shadow tests "as executed"; bench.py prints both fractions).
template <class Tbl, bool COUNT, bool BVH, bool EXEC = false>
struct Tracer {
    const Params &P;
    Tbl tbl;
    Bvh bvh;
    uint32_t *stack;  // this lane's LDS traversal stack column (BVH only), stride 256
    uint32_t cnt[HIPPT_N_COUNTERS];
#ifdef PT_NEE_DIAG
    unsigned long long *diag = nullptr;
#endif

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
            occluded_multi<Tbl, 4, COUNT, false, PT_CULL != 0>(tbl, P, x, 1e-4f, l, tmax, occ, cnt[HIPPT_C_SHADOW_SPHERE_TESTS],
                                                       cnt[HIPPT_C_SHADOW_PLANE_TESTS], !BVH, nullptr, nullptr,
                                                       P.cull_off + li * P.cull_stride);
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

    // direct_light for a whole wave (every lane of the wave that is still in the trace loop calls it; `nee` =
    // this lane has a vertex to light). Same samples and same tests as direct_light, in a different order: on
    // scenes like scene.example most shading points have all 4 samples of one light blocked by the SAME object
    // (the glass pane in front of the second light), but almost never all lanes of a wave at once, so a
    // wave-uniform early out does not fire. Here the wave remembers, per light, the object that blocked
    // everything for most lanes (learnt in the ordinary loop, LEARN) and tests that object first. A
    // (lane, sample) ray it does not block is evaluated at once (T * brdf * light / 4, the radiance it carries
    // if it arrives) and put on a list in LDS; whenever the list holds 64 rays the wave tests them against the
    // whole scene, one ray per lane, and adds what arrives to the pixel sums `acc` (pixel slot `pix`).
    // Occlusion is an OR over objects, so every ray gets the decision direct_light gives it; only the order of
    // the f32 additions into the pixel sum differs. The part that is not deferred is added to `psum`, the path's sum.
    __device__ __forceinline__ void process_pairs(NeeShare *sh, float *acc, uint32_t first, uint32_t count) {
        const unsigned long long act = ballot64(true);
        const uint32_t n_act = (uint32_t)__popcll(act);
        uint32_t d0 = 0, d1 = 0;
        wave_lds_sync();
        for (uint32_t i = rank_in(act); i < count; i += n_act) {
            const float *pr = sh->pair + first + i;
            F3 px = F3{pr[0], pr[kPairCap], pr[2 * kPairCap]};
            F3 pl[1] = {F3{pr[3 * kPairCap], pr[4 * kPairCap], pr[5 * kPairCap]}};
            float pt[1] = {pr[6 * kPairCap]};
            bool po[1];
            occluded_multi<Tbl, 1, false>(tbl, P, px, 1e-4f, pl, pt, po, d0, d1);
            bump(HIPPT_C_SHADOW_SPHERE_TESTS, P.n_sph);
            bump(HIPPT_C_SHADOW_PLANE_TESTS, P.n_pl);
            if (!po[0]) {
                // the ray arrives: renderer.rs:49-77 for it, here rather than in the lane that drew it — on the list
                // the wave's few survivors of a mostly blocked light sit side by side
                const uint32_t w = __float_as_uint(pr[17 * kPairCap]);
                const Mat m = load_mat(tbl, P.mat_off, w & 0x7fffffu);
                const ShadeK sk = make_shade(F3{pr[8 * kPairCap], pr[9 * kPairCap], pr[10 * kPairCap]},
                                             F3{pr[11 * kPairCap], pr[12 * kPairCap], pr[13 * kPairCap]}, m);
                const uint32_t o = P.light_off + (w >> 29) * LIGHT_F4;
                const float4 l0 = tbl.ld(o), l3 = tbl.ld(o + 3), l4 = tbl.ld(o + 4);
                bool lit;
                F3 c = nee_eval(sk, pl[0], pr[7 * kPairCap], f3(l4), l0.w, f3(l3), lit);
                if (lit) bump(HIPPT_C_LIT_EVALS);
                c = F3{pr[14 * kPairCap], pr[15 * kPairCap], pr[16 * kPairCap]} * c * 0.25f;  // :82, and the path throughput (:285 / :325)
                float *a = acc + ((w >> 23) & 63u) * 3u;
                __hip_atomic_fetch_add(a + 0, c.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(a + 1, c.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(a + 2, c.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        wave_lds_sync();
    }
    // after the trace loop, all lanes of the wave: the rays still on the list
    __device__ __forceinline__ void flush_pairs(NeeShare *sh, float *acc) {
        wave_lds_sync();
        uint32_t n = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sh->pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (n) process_pairs(sh, acc, 0u, n);
    }
    __device__ __forceinline__ void direct_light_shared(bool nee, F3 x, F3 n, F3 v, uint32_t mat_id, Rng &rng, F3 T,
                                                        F3 &psum, uint32_t pix, NeeShare *sh, float *acc) {
        static_assert((!COUNT || EXEC) && !BVH, "the reference-counting and LBVH kernels use direct_light");
        static_assert(kHintLights <= 8, "the list word keeps the light in 3 bits");
        const uint32_t n_nee = (uint32_t)__popcll(ballot64(nee));
        if (n_nee == 0u) return;
        if (nee) {
            bump(HIPPT_C_NEE_VERTICES);
            bump(HIPPT_C_SHADOW_RAYS, 4u * P.n_light);
        }
        // not kept in a register across trips: a lane whose pool item lies outside the image skips the trip
        uint32_t pending = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sh->pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        uint32_t dummy0 = 0, dummy1 = 0;
#ifdef PT_NEE_DIAG
        const bool diag_lane = rank_in(ballot64(true)) == 0u;
#define NEE_DIAG(k, v) if (diag_lane && li < 3u) atomicAdd(&diag[k], (unsigned long long)(v))
#else
#define NEE_DIAG(k, v)
#endif
        for (uint32_t li = 0; li < P.n_light; ++li) {
            uint32_t o = P.light_off + li * LIGHT_F4;
            float4 l0 = tbl.ld(o), l1 = tbl.ld(o + 1), l2 = tbl.ld(o + 2), l3 = tbl.ld(o + 3), l4 = tbl.ld(o + 4);
            NEE_DIAG(li * 4u, 1);
            F3 l[4];
            float id2[4], tmax[4];
            if (nee) {
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
            }
            uint32_t h = kNoHint;
            if (li < kHintLights)
                h = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sh->hint[li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            bool full = true;
            uint32_t surv = 0;  // bit s: evaluate sample s (hinted: lit from the front and not blocked by the remembered object)
            if (h != kNoHint) {
                if (nee) {
                    bool hit[4];
                    object_shadow<Tbl, 4>(tbl, P, h, x, 1e-4f, l, tmax, hit);
                    bump(h < P.n_sph ? HIPPT_C_SHADOW_SPHERE_TESTS : HIPPT_C_SHADOW_PLANE_TESTS, 4u);
#pragma unroll
                    for (int s = 0; s < 4; ++s)  // a sample that faces away contributes nothing (:50,:55)
                        if (!hit[s] & (dot(n, l[s]) > 0.0f) & (-dot(l[s], f3(l4)) > 0.0f)) surv |= 1u << s;
                }
                const unsigned long long b0 = ballot64((surv & 1u) != 0u), b1 = ballot64((surv & 2u) != 0u),
                                         b2 = ballot64((surv & 4u) != 0u), b3 = ballot64((surv & 8u) != 0u);
                const uint32_t n0 = (uint32_t)__popcll(b0), n1 = n0 + (uint32_t)__popcll(b1), n2 = n1 + (uint32_t)__popcll(b2),
                               n_surv = n2 + (uint32_t)__popcll(b3);
                if (n_surv <= kPairBatch) {
                    full = false;
                    if (n_surv == 0u) NEE_DIAG(li * 4u + 1u, 1);
                    if (n_surv != 0u) {
                        NEE_DIAG(li * 4u + 2u, 1);
                        NEE_DIAG(li * 4u + 3u, n_surv);
                        if (pending + n_surv > kPairCap) {  // no room beside the rays already parked: test those now
                            process_pairs(sh, acc, 0u, pending);
                            pending = 0u;
                        }
                        const uint32_t base[4] = {pending, pending + n0, pending + n1, pending + n2};
                        const unsigned long long bs[4] = {b0, b1, b2, b3};
                        const float word = __uint_as_float(mat_id | (pix << 23) | (li << 29));  // li < kHintLights = 8, materials < 2^23 (host check)
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            if (surv & (1u << s)) {
                                float *pr = sh->pair + base[s] + rank_in(bs[s]);
                                pr[0] = x.x, pr[kPairCap] = x.y, pr[2 * kPairCap] = x.z;
                                pr[3 * kPairCap] = l[s].x, pr[4 * kPairCap] = l[s].y, pr[5 * kPairCap] = l[s].z;
                                pr[6 * kPairCap] = tmax[s], pr[7 * kPairCap] = id2[s];
                                pr[8 * kPairCap] = n.x, pr[9 * kPairCap] = n.y, pr[10 * kPairCap] = n.z;
                                pr[11 * kPairCap] = v.x, pr[12 * kPairCap] = v.y, pr[13 * kPairCap] = v.z;
                                pr[14 * kPairCap] = T.x, pr[15 * kPairCap] = T.y, pr[16 * kPairCap] = T.z;
                                pr[17 * kPairCap] = word;
                            }
                        }
                        pending += n_surv;
                        if (pending == kPairBatch) {
                            process_pairs(sh, acc, 0u, kPairBatch);
                            pending = 0u;
                        }
                        __hip_atomic_store(&sh->pending, pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    surv = 0;  // nothing left to evaluate in this lane
                } else {  // the remembered object no longer blocks most of the wave's samples
                    NEE_DIAG(12u + li, 1);
                    __hip_atomic_store(&sh->hint[li], kNoHint, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    surv = 0;
                }
            }
            if (full && nee) {
                bool occ[4];
                uint32_t dom, dom_lanes;
                uint32_t rows_run = 0u;
                occluded_multi<Tbl, 4, false, true, PT_CULL != 0>(tbl, P, x, 1e-4f, l, tmax, occ, dummy0, dummy1, true, &dom, &dom_lanes,
                                                          P.cull_off + li * P.cull_stride, true, COUNT ? &rows_run : nullptr);
                bump(HIPPT_C_SHADOW_SPHERE_TESTS, 4u * P.n_sph);
                bump(HIPPT_C_SHADOW_PLANE_TESTS, 4u * rows_run);
                // worth remembering if the lanes it leaves over fit a batch: at most 4 rays each
                if (li < kHintLights && dom_lanes != 0u && (n_nee - dom_lanes) * 4u <= PT_HINT_SLACK)
                    __hip_atomic_store(&sh->hint[li], dom, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (!occ[s]) surv |= 1u << s;
            }
            // the ordinary loop's survivors are evaluated where they stand
            if (surv != 0u) {
                // the shading constants are formed here, per light, rather than carried through the shadow tests in
                // 10 registers the kernel does not have (5 waves per SIMD: 96 VGPRs). The empty asm statements make
                // their inputs opaque, or the optimiser hoists the whole of make_shade out of the lights loop again
                // (loop-invariant code motion) and spills to make room for it.
                // The material is read again from the tables by its id (two wave-wide LDS reads), so that its 8
                // registers are free during the shadow tests.
                F3 n_here = n;
                uint32_t id_here = mat_id;
                asm volatile("" : "+v"(n_here.x), "+v"(id_here));
                const Mat m_here = load_mat(tbl, P.mat_off, id_here);
                const ShadeK sk = make_shade(n_here, v, m_here);
                F3 contrib = F3{0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (surv & (1u << s)) {
                        bool lit;
                        contrib = contrib + nee_eval(sk, l[s], id2[s], f3(l4), l0.w, f3(l3), lit);
                        if (lit) bump(HIPPT_C_LIT_EVALS);
                    }
                }
                psum = madd(psum, T * contrib, 0.25f);  // :82 and :285 / :325, straight into the path's sum
            }
        }
    }
};

}  // namespace pt
