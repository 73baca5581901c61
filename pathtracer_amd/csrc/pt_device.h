// pt_device.h — device-side math of the gfx950 path-trace kernels.
//
// Written from the behaviour of the reference CPU renderer
// (eliasstepanik/Pathtracer src/renderer.rs, ggx.rs, sphere.rs, plane.rs,
// algebra.rs, tonemap.rs); each function cites the lines whose result it must
// reproduce. The code is free to re-associate and fuse (FMA contraction, fast
// v_rcp/v_rsq/v_sqrt/v_sin/v_cos/v_log/v_exp) as long as results stay within
// the 1e-5 relative parity bound; per-ray and per-light invariants are hoisted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hippt.h"

namespace pt {

// ---- scene blob layout (float4 units), built by hippt_api.hip ---------------
//   spheres  : n_sph x 1  {cx,cy,cz,r*r}
//   planes   : n_pl  x 3  {px,py,pz,nx} {ny,nz,u'x,u'y} {u'z,v'x,v'y,v'z}, u' = u/(u.u), v' = v/(v.v)
//   aux      : (n_sph + n_pl) x 1  {1/r, material, scene order, 0}
//   materials: n_mat x 2  {r,g,b,metallic} {roughness,ior,density,anisotropy}
//   lights   : n_light x 5 {pos,area} {u,0} {v,0} {intensity,0} {normal,0}
//   camera   : 5  {pos,aspect} {right,scale} {up,aperture} {forward,focus} {1/width,1/height,0,0}
//              (in the blob, not in Params: read once per path, and 22 fewer SGPRs live across the trace loop)
//   cull     : n_light x n_pl  {side,0,0,0}: +1 / -1 if the whole rectangle of light i lies strictly on the
//              positive / negative side of plane j ((y - p).n > 0 / < 0 for its four corners), 0 if it straddles it
constexpr int MAT_F4 = 2;
constexpr int LIGHT_F4 = 5;
constexpr int CAM_F4 = 5;

struct Params {
    uint32_t width, height;
    uint32_t n_sph, n_pl, n_mat, n_light;
    uint32_t sph_off, pl_off, aux_off, mat_off, light_off, cam_off, cull_off;  // float4 offsets into the blob
    uint32_t cull_stride;         // rows of the cull table per light: n_pl, or 0 when all lights share one all-zero row (table too large to pay)
    uint32_t blob_f4;             // blob size in float4
    uint32_t sample_begin, sample_count, max_depth, seed;
    uint32_t tiles_x;             // 16x16 pixel tiles per row
    uint32_t wf_streams_log2;     // LBVH queue pipeline: every pixel's samples run as 2^this concurrent streams (pt_wavefront_bvh.hip)
};

// ---- float3 -------------------------------------------------------------------
struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 f3(float4 v) { return F3{v.x, v.y, v.z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return F3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ F3 operator-(F3 a) { return F3{-a.x, -a.y, -a.z}; }
// the contraction is spelt out, so that every inlined copy of a test rounds the same way
__device__ __forceinline__ float dot(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ F3 cross(F3 a, F3 b) {
    return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// a + b*s
__device__ __forceinline__ F3 madd(F3 a, F3 b, float s) {
    return F3{fmaf(b.x, s, a.x), fmaf(b.y, s, a.y), fmaf(b.z, s, a.z)};
}

// lanes of the wave (among the active ones) for which p holds; unlike __ballot(), which goes through a 0/1
// VGPR (v_cndmask + v_cmp), this is the compare mask itself
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// sin / cos of 2*pi*rev (v_sin_f32 / v_cos_f32 take revolutions)
__device__ __forceinline__ float sin2pi(float rev) { return __builtin_amdgcn_sinf(rev); }
__device__ __forceinline__ float cos2pi(float rev) { return __builtin_amdgcn_cosf(rev); }
__device__ __forceinline__ float fln(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// algebra.rs:19  self.scale(1.0/self.norm())
__device__ __forceinline__ F3 normalize(F3 a) { return a * rsq(dot(a, a)); }
// algebra.rs:21-27
__device__ __forceinline__ F3 any_orthonormal(F3 a) {
    return (fabsf(a.z) < 0.9999999f) ? F3{a.y, -a.x, 0.0f} : F3{0.0f, -a.z, a.y};
}
// f32::clamp (NaN passes through)
__device__ __forceinline__ float rclamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ float pow5(float x) {
    float x2 = x * x;
    return x * (x2 * x2);
}

// ---- uniform stream -------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
// One stream per (seed, pixel, global sample): a 32-bit LCG s <- s * A + inc whose odd increment is part of the
// stream's key (A = 2891336453, L'Ecuyer 1999 table 4); a uniform is the top 24 bits of the new state scaled by
// 2^-24, i.e. the distribution of rand's Standard f32. One v_mul_lo_u32 + one v_add_u32 per draw (6 issue cycles)
// where the xoroshiro64+ of round 1 cost 18 (two rotates and a left shift at half rate): -5 % kernel time on C2.
// The key is 63 bits wide (round 3): the start state and the increment are two DIFFERENT hashes of (pixel, sample)
// - an xor-combination and a multiply-add combination, so that no (pixel, sample) difference cancels in both -
// where round 2 derived the increment from the start state (2^32 streams for the 2^31 paths of C2).
struct Rng {
    uint32_t s0, s1;  // state, increment
    __device__ __forceinline__ void seed(uint32_t seed_mixed, uint32_t pixel_mixed_base, uint32_t sample) {
        // seed_mixed = mix32(seed ^ 0x85ebca6b) and pixel_mixed_base = mix32(pixel + seed_mixed)
        // are per-launch / per-pixel invariants, hoisted by the caller.
        (void)seed_mixed;
        s0 = mix32(sample ^ pixel_mixed_base);
        const uint32_t d = mix32(pixel_mixed_base ^ 0x5bd1e995u);
        s1 = mix32(sample * 0x9e3779b1u + d) | 1u;
    }
    __device__ __forceinline__ float gen() {
        s0 = s0 * 2891336453u + s1;
        return (float)(s0 >> 8) * 5.9604644775390625e-08f;  // 2^-24: rand's Standard f32, [0,1)
    }
};

// ---- table access policies ------------------------------------------------------
struct TblLds {
    const float4 *p;  // points into __shared__
    __device__ __forceinline__ float4 ld(uint32_t i) const { return p[i]; }
};
struct TblGlobal {
    const float4 *__restrict__ p;
    __device__ __forceinline__ float4 ld(uint32_t i) const { return p[i]; }
};

// ---- intersection ---------------------------------------------------------------
struct RayK {  // per-ray invariants of the sphere quadratic (sphere.rs:27,33)
    F3 ro, rd;
    float a, inv_a;
};
__device__ __forceinline__ RayK make_ray(F3 ro, F3 rd) {
    float a = dot(rd, rd);
    return RayK{ro, rd, a, rcp(a)};
}

// sphere.rs:26-34 in half-b form: b = 2h, disc = 4(h^2 - a c), t = (-h - sqrt(h^2 - a c))/a
// (exactly the same quantities: scaling by 2 and 4 is exact in binary fp).
// disc < 0 -> sqrt = NaN -> t NaN -> every later comparison is false (miss).
__device__ __forceinline__ float sphere_t(const RayK &r, float4 c_r2) {
    F3 oc = r.ro - f3(c_r2);
    float h = dot(oc, r.rd);
    float c = fmaf(oc.z, oc.z, fmaf(oc.y, oc.y, fmaf(oc.x, oc.x, -c_r2.w)));  // oc.oc - r*r
    float disc = h * h - r.a * c;
    return (-h - fsqrt(disc)) * r.inv_a;
}

// plane.rs:25-52, branch-free. q0={p,nx} q1={ny,nz,u'x,u'y} q2={u'z,v'x,v'y,v'z} with
// u' = u/(u.u), v' = v/(v.v), so the bounds |d.u| <= u.u become |d.u'| <= 1.
// Returns t, or a value that fails `t > 0` (NaN or -1) on a miss.
__device__ __forceinline__ float plane_t(const RayK &r, float4 q0, float4 q1, float4 q2) {
    F3 p = f3(q0);
    F3 n = F3{q0.w, q1.x, q1.y};
    float denom = dot(n, r.rd);
    float t = dot(p - r.ro, n) * rcp(denom);
    F3 d = madd(r.ro, r.rd, t) - p;
    float du = dot(d, F3{q1.z, q1.w, q2.x});
    float dv = dot(d, F3{q2.y, q2.z, q2.w});
    bool ok = (fabsf(denom) >= 1e-6f) & (t > 1e-4f) & (fmaxf(fabsf(du), fabsf(dv)) <= 1.0f);
    return ok ? t : -1.0f;
}

struct HitRec {
    float t;
    int id;  // index into the aux table (spheres first, then planes), -1 = miss
};

// The WINNING object's t, hit point and normal are recomputed once per ray with the reference's own
// operation order and roundings (no FMA contraction, sqrt and divide corrected to the rounded result):
// sphere.rs:26-37, plane.rs:25-28, renderer.rs:295. The candidate tests above may fuse and use v_rcp
// (they only rank the objects); but t of a small far sphere is a cancellation (disc = b^2 - 4ac with
// b^2 ~ 10^3, disc ~ r^2), so any re-association moves t by ~1e-4 and the normal by ~1e-2, after which
// a path among many spheres takes a different course. With the reference's rounding the hit record is
// bit-identical to the CPU renderer's in all but threshold cases, at ~30 instructions per ray.
// n / d and sqrt(x) from v_rcp / v_rsq plus one residual correction (Markstein): the correctly rounded
// result in all but a vanishing fraction of cases, at 4-5 instructions instead of the ~12 of hipcc's
// IEEE expansions (which also handle denormal scaling that cannot occur here). x <= 0 gives NaN.
__device__ __forceinline__ float div_cr(float n, float d) {
    float rc = rcp(d);
    float q = n * rc;
    return fmaf(fmaf(-q, d, n), rc, q);
}
__device__ __forceinline__ float sqrt_cr(float x) {
    float y = rsq(x);
    float s = x * y;
    return fmaf(fmaf(-s, s, x), 0.5f * y, s);
}
__device__ __forceinline__ float sphere_t_ref(F3 ro, F3 rd, float4 c_r2) {
#pragma clang fp contract(off)
    float ox = ro.x - c_r2.x, oy = ro.y - c_r2.y, oz = ro.z - c_r2.z;
    float a = rd.x * rd.x + rd.y * rd.y + rd.z * rd.z;
    float b = 2.0f * (ox * rd.x + oy * rd.y + oz * rd.z);
    float c = (ox * ox + oy * oy + oz * oz) - c_r2.w;
    float disc = b * b - 4.0f * a * c;
    return div_cr(-b - sqrt_cr(disc), 2.0f * a);
}
__device__ __forceinline__ float plane_t_ref(F3 ro, F3 rd, float4 q0, float4 q1) {
#pragma clang fp contract(off)
    float nx = q0.w, ny = q1.x, nz = q1.y;
    float denom = nx * rd.x + ny * rd.y + nz * rd.z;
    float num = (q0.x - ro.x) * nx + (q0.y - ro.y) * ny + (q0.z - ro.z) * nz;
    return div_cr(num, denom);
}
// ro + rd * t with two roundings per component (renderer.rs:272,295; sphere.rs:36)
__device__ __forceinline__ F3 point_at_ref(F3 ro, F3 rd, float t) {
#pragma clang fp contract(off)
    return F3{ro.x + rd.x * t, ro.y + rd.y * t, ro.z + rd.z * t};
}
// Replaces h.t by the reference-rounded value (kept if the refined value fails the hit condition, which
// can only happen on a threshold).
template <class Tbl, class SphRow>
__device__ __forceinline__ void refine_hit(const Tbl &tbl, const Params &P, F3 ro, F3 rd, HitRec &h, SphRow sphere_row) {
    if (h.id < 0) return;
    float te;
    if (h.id < (int)P.n_sph) {
        te = sphere_t_ref(ro, rd, sphere_row(h.id));
    } else {
        uint32_t o = P.pl_off + ((uint32_t)h.id - P.n_sph) * 3u;
        te = plane_t_ref(ro, rd, tbl.ld(o), tbl.ld(o + 1));
    }
    if (te > 0.0f && te < __builtin_inff()) h.t = te;
}


// renderer.rs:356-362: closest hit, first minimum in scene order. The object index is
// wave-uniform, so every table read is a uniform-address read (scalar loads / LDS
// broadcast) and the loops carry no divergence. Within one table ascending index is
// ascending scene order, so strict '<' keeps the first minimum; an exact tie between a
// plane and the current best sphere is resolved by the objects' scene order.
// Plane part of the closest-hit loop. (best_t, best) hold the best sphere so far; sph_order(best)
// gives that sphere's scene order and is only evaluated on an exact tie.
template <class Tbl, class OrdFn>
__device__ __forceinline__ void closest_planes(const Tbl &tbl, const Params &P, const RayK &r, float &best_t, int &best,
                                               OrdFn sph_order) {
#pragma unroll 2
    for (uint32_t j = 0; j < P.n_pl; ++j) {
        uint32_t o = P.pl_off + j * 3u;
        float t = plane_t(r, tbl.ld(o), tbl.ld(o + 1), tbl.ld(o + 2));
        bool b = (t > 0.0f) & (t < best_t);
        bool tie = (t == best_t) & (best >= 0) & (best < (int)P.n_sph);
        if (__builtin_expect(__any(tie), 0)) {  // coincident surfaces only
            uint32_t ord_p = __float_as_uint(tbl.ld(P.aux_off + P.n_sph + j).z);
            uint32_t ord_s = tie ? sph_order(best) : 0u;
            b = b | (tie & (ord_p < ord_s));
        }
        best_t = b ? t : best_t;
        best = b ? (int)(P.n_sph + j) : best;
    }
}

template <class Tbl>
__device__ __forceinline__ HitRec closest_hit(const Tbl &tbl, const Params &P, const RayK &r) {
    float best_t = __builtin_inff();
    int best = -1;
#pragma unroll 4
    for (uint32_t i = 0; i < P.n_sph; ++i) {
        float t = sphere_t(r, tbl.ld(P.sph_off + i));
        bool b = (t > 0.0f) & (t < best_t);
        best_t = b ? t : best_t;
        best = b ? (int)i : best;
    }
    closest_planes(tbl, P, r, best_t, best,
                   [&](int id) { return __float_as_uint(tbl.ld(P.aux_off + (uint32_t)id).z); });
    HitRec h{best_t, best};
    refine_hit(tbl, P, r.ro, r.rd, h, [&](int id) { return tbl.ld(P.sph_off + (uint32_t)id); });
    return h;
}

// Normal, material id and scene-order index of the winning object (sphere.rs:36-37, plane.rs:32).
template <class Tbl, class SphFn>
__device__ __forceinline__ void finish_hit(const Tbl &tbl, const Params &P, const RayK &r, HitRec h, F3 &hit, F3 &n,
                                           uint32_t &mat, uint32_t &order, SphFn sphere_rows) {
    hit = point_at_ref(r.ro, r.rd, h.t);
    float4 aux;  // {1/r, mat, order, r}
    if (h.id < (int)P.n_sph) {
        float4 s;
        sphere_rows(h.id, s, aux);
        n = (hit - f3(s)) * aux.x;
    } else {
        aux = tbl.ld(P.aux_off + (uint32_t)h.id);
        uint32_t o = P.pl_off + ((uint32_t)h.id - P.n_sph) * 3u;
        float4 q0 = tbl.ld(o), q1 = tbl.ld(o + 1);
        F3 pn = F3{q0.w, q1.x, q1.y};
        n = dot(pn, r.rd) < 0.0f ? pn : -pn;
    }
    mat = __float_as_uint(aux.y);
    order = __float_as_uint(aux.z);
}
template <class Tbl>
__device__ __forceinline__ void finish_hit(const Tbl &tbl, const Params &P, const RayK &r, HitRec h, F3 &hit, F3 &n,
                                           uint32_t &mat, uint32_t &order) {
    finish_hit(tbl, P, r, h, hit, n, mat, order, [&](int id, float4 &s, float4 &aux) {
        s = tbl.ld(P.sph_off + (uint32_t)id);
        aux = tbl.ld(P.aux_off + (uint32_t)id);
    });
}

// a - b that is never contracted into an FMA with the product that formed a: with a literal b the
// subtraction is a full-rate VOP2, while the fused form needs the constant in an SGPR, and VALU
// instructions with an SGPR source issue at half rate on gfx950.
__device__ __forceinline__ float sub_unfused(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}

// renderer.rs:44-46 for NS shadow rays that leave the same point x: origin x + l*eps, unit
// direction l[s], occluded iff some object has t > 0 (plane: t > 1e-4) and t*t < dist2*0.999,
// i.e. t < tmax[s] = sqrt(0.999 dist2); both bounds are checked with one compare, |t - mid| < half
// (compares issue at half rate on gfx950). Every object is tested, branch-free (on scene.example the reference's
// first-occluder early out skips only 8 % of the tests, and a per-lane exit costs more in
// divergence than it saves). The loops are object-major so that each table row is read once
// per NS rays and the terms that depend only on (x, object) are formed once:
//   sphere: oc = X + eps l, X = x - c  =>  oc.l = X.l + eps             (l.l = 1 to rounding)
//           (oc.l)^2 - (oc.oc - r^2) = (X.l)^2 - (X.X - r^2)         (the eps terms cancel)
//   plane : (p - o).n = Dn - eps n.l, Dn = (p - x).n  =>  t = Dn/(n.l) - eps
//           hit - p = (x - p) + l (t + eps)            =>  d.u' = Du + (t + eps) l.u'
// which is the arithmetic of sphere.rs:26-33 / plane.rs:25-52 re-associated (differences
// ~1e-7 relative in t; they can only move an occlusion decision that sits on its threshold).
// One sphere row against the NS rays: hit[s] = the ray is blocked by this sphere.
template <int NS>
__device__ __forceinline__ void sphere_shadow_row(float4 c, F3 x, float eps, const F3 (&l)[NS], const float (&q)[NS],
                                                  bool (&hit)[NS]) {
#pragma clang fp contract(off)  // explicit fmaf only: the copies of this test (NS = 1, NS = 4, object_shadow) must agree bit for bit
    F3 X = x - f3(c);
    float C0 = fmaf(X.z, X.z, fmaf(X.y, X.y, fmaf(X.x, X.x, -c.w)));  // X.X - r^2
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        // the discriminant does not depend on where along the ray the origin sits: with o = x + eps l,
        // (oc.l)^2 - (oc.oc - r^2) = (X.l)^2 - (X.X - r^2); the near root from x is t0 = -X.l - sqrt(disc)
        float d1 = dot(X, l[s]);
        float t0 = -d1 - fsqrt(fmaf(d1, d1, -C0));
        hit[s] = fabsf(t0 - q[s]) < q[s] - eps;  // NaN: false
    }
}
// One plane (3 rows) against the NS rays.
template <int NS>
__device__ __forceinline__ void plane_shadow_row(float4 q0, float4 q1, float4 q2, F3 x, float eps, const F3 (&l)[NS],
                                                 const float (&q)[NS], bool (&hit)[NS]) {
#pragma clang fp contract(off)
    F3 n = F3{q0.w, q1.x, q1.y}, u = F3{q1.z, q1.w, q2.x}, v = F3{q2.y, q2.z, q2.w};
    F3 Xp = x - f3(q0);
    float Dn = -dot(Xp, n), Du = dot(Xp, u), Dv = dot(Xp, v);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float denom = dot(n, l[s]);
        float tp = Dn * rcp(denom);
        float du = fmaf(tp, dot(l[s], u), Du);
        float dv = fmaf(tp, dot(l[s], v), Dv);
        // 1e-4 < t < tmax with t = tp - eps:  |tp - 5e-5 - q| < q - eps - 5e-5
        float dt = sub_unfused(tp, 5e-5f) - q[s];
        bool ok = (fabsf(denom) >= 1e-6f) & (fmaxf(fabsf(du), fabsf(dv)) <= 1.0f);
        hit[s] = ok & (fabsf(dt) < q[s] - (eps + 5e-5f));
    }
}
// q = eps + tmax/2: with t = t0 - eps (t0 measured from x), 0 < t < tmax  <=>  |t0 - q| < q - eps
template <int NS>
__device__ __forceinline__ void shadow_mid(float eps, const float (&tmax)[NS], float (&q)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) q[s] = fmaf(0.5f, tmax[s], eps);
}
// One object (wave-uniform index: spheres first, then planes) against the NS rays, same arithmetic as the loops below.
template <class Tbl, int NS>
__device__ __forceinline__ void object_shadow(const Tbl &tbl, const Params &P, uint32_t obj, F3 x, float eps,
                                              const F3 (&l)[NS], const float (&tmax)[NS], bool (&hit)[NS]) {
    float q[NS];
    shadow_mid<NS>(eps, tmax, q);
    if (obj < P.n_sph) {
        sphere_shadow_row<NS>(tbl.ld(P.sph_off + obj), x, eps, l, q, hit);
    } else {
        uint32_t o = P.pl_off + (obj - P.n_sph) * 3u;
        plane_shadow_row<NS>(tbl.ld(o), tbl.ld(o + 1), tbl.ld(o + 2), x, eps, l, q, hit);
    }
}

// LEARN: also report the object that blocks ALL NS rays of the most lanes (`dominant`, an object index as
// object_shadow takes it, and on how many of the calling lanes, `dominant_lanes`); both are wave-uniform and
// formed with scalar instructions from the compare masks.
//
// CULL (cull_row = float4 offset of this light's rows of the cull table, `active` = this lane's x / l are real): a
// plane whose positive side holds the whole light can only block a ray that starts on its negative side, so the wave
// skips the plane's row unless some lane stands there. Exactly the decisions of the full test: a hit needs
// tp = Dn / (n.l) > eps + 1e-4 (plane_shadow_row), i.e. Dn and n.l of one sign and |Dn| > (eps + 1e-4) |n.l|;
//  - a lane on the light's side (Dn side <= 0) has tp <= 0, or meets the plane BEHIND the light point (the zero of
//    (y - p).n along x -> light lies beyond the light when both ends are on one side), which fails t < 0.9995 dist;
//  - a lane on the far side is a candidate when |Dn| > 0.95 (eps + 1e-4) |n.l| for one of its rays (conservative:
//    5 % against the ~1e-7 of v_rcp_f32). Lanes that stand ON the plane with Dn = +-1e-7 of rounding noise — half of
//    all lanes on a floor or a wall — are candidates only at |n.l| < ~2e-3, so the skip still fires.
// On scene.example the four room planes never block the light inside the room: 3/4 of the plane rows of the open
// light's loop go.
template <class Tbl, int NS, bool COUNT, bool LEARN = false, bool CULL = false>
__device__ __forceinline__ void occluded_multi(const Tbl &tbl, const Params &P, F3 x, float eps, const F3 (&l)[NS],
                                               const float (&tmax)[NS], bool (&occ)[NS], uint32_t &n_sph_tests,
                                               uint32_t &n_pl_tests, bool spheres = true, uint32_t *dominant = nullptr,
                                               uint32_t *dominant_lanes = nullptr, uint32_t cull_row = 0u, bool active = true,
                                               uint32_t *plane_rows_run = nullptr) {
    const uint32_t n_sph = spheres ? P.n_sph : 0u;  // false: the caller has tested the spheres (BVH) and set occ[]
    if (spheres) {
#pragma unroll
        for (int s = 0; s < NS; ++s) occ[s] = false;
    }
    float q[NS];
    shadow_mid<NS>(eps, tmax, q);
    uint32_t best = 0xffffffffu, best_lanes = 0;
    uint32_t first[NS];  // COUNT: scene order of the ray's first occluder
#pragma unroll
    for (int s = 0; s < NS; ++s) first[s] = 0xffffffffu;
    auto fold = [&](const bool (&hit)[NS], uint32_t obj) {
        bool all = true;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            occ[s] = occ[s] | hit[s];
            all = all & hit[s];
        }
        if (COUNT && spheres) {
            const uint32_t ord = __float_as_uint(tbl.ld(P.aux_off + obj).z);
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (hit[s]) first[s] = min(first[s], ord);
        }
        if (LEARN) {
            uint32_t c = (uint32_t)__popcll(ballot64(all));
            if (c > best_lanes) {
                best_lanes = c;
                best = obj;
            }
        }
    };
#pragma unroll 1
    for (uint32_t i = 0; i < n_sph; ++i) {
        bool hit[NS];
        sphere_shadow_row<NS>(tbl.ld(P.sph_off + i), x, eps, l, q, hit);
        fold(hit, i);
    }
#pragma unroll 1
    for (uint32_t j = 0; j < P.n_pl; ++j) {
        uint32_t o = P.pl_off + j * 3u;
        const float4 q0 = tbl.ld(o), q1 = tbl.ld(o + 1);
        if constexpr (CULL) {
            const float side = tbl.ld(cull_row + j).x;  // wave-uniform
            if (side != 0.0f) {
                const F3 n = F3{q0.w, q1.x, q1.y};
                const float Dn = -dot(x - f3(q0), n);
                const float thr = -0.95f * (eps + 1e-4f);
                float m = fmaf(fabsf(dot(n, l[0])), thr, fabsf(Dn));
#pragma unroll
                for (int s = 1; s < NS; ++s) m = fmaxf(m, fmaf(fabsf(dot(n, l[s])), thr, fabsf(Dn)));
                const bool cand = active & (Dn * side > 0.0f) & (m > 0.0f);
                if (ballot64(cand) == 0ull) continue;
            }
        }
        bool hit[NS];
        plane_shadow_row<NS>(q0, q1, tbl.ld(o + 2), x, eps, l, q, hit);
        if (plane_rows_run) ++*plane_rows_run;  // (executed-test counting build only)
        fold(hit, P.n_sph + j);
    }
    if (COUNT) {
        // The counters report the tests the REFERENCE executes (renderer.rs:46 walks the objects in scene order and
        // leaves at the first occluder), not the ones this loop runs: they feed the algorithmic FLOP count
        // (SURVEY §8d), which must not depend on how the kernel orders or skips its work.
        if (spheres) {
            for (uint32_t i = 0; i < P.n_sph + P.n_pl; ++i) {
                const uint32_t ord = __float_as_uint(tbl.ld(P.aux_off + i).z);
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (ord <= first[s]) (i < P.n_sph ? n_sph_tests : n_pl_tests) += 1u;
            }
        } else {  // LBVH kernels: sphere tests are counted by the traversal, plane tests as run
            n_pl_tests += NS * P.n_pl;
        }
    }
    if (LEARN) {
        *dominant = best;
        *dominant_lanes = best_lanes;
    }
}

// ---- BSDF -------------------------------------------------------------------------
// ggx.rs:5
__device__ __forceinline__ F3 reflect(F3 v, F3 n) { return madd(v, n, -2.0f * dot(v, n)); }
// ggx.rs:7-9
__device__ __forceinline__ F3 fresnel_schlick(float cos_theta, F3 f0) {
#pragma clang fp contract(off)  // explicit fmaf only: every inlined copy rounds the same way (see nee_eval)
    float w = pow5(1.0f - cos_theta);
    return F3{fmaf(1.0f - f0.x, w, f0.x), fmaf(1.0f - f0.y, w, f0.y), fmaf(1.0f - f0.z, w, f0.z)};
}
// ggx.rs:10-12 (a = roughness, un-squared: quirk Q7)
// q = nh^2 (a^2 - 1) + 1 with the reference's roundings (product rounded, then the sum): at roughness 0.01 and
// nh -> 1 it is a cancellation down to a^2 = 1e-4, so a fused product moves D by up to 1e-3 relative.
// (a^2 - 1 likewise: fused into fma(a, a, -1) it differs from the reference's by half an ulp of a^2, which q amplifies.)
__device__ __forceinline__ float ggx_a2m1(float a2) {
#pragma clang fp contract(off)
    return a2 - 1.0f;
}
__device__ __forceinline__ float ggx_q(float nh, float a2m1) {
#pragma clang fp contract(off)
    return (nh * nh) * a2m1 + 1.0f;
}
__device__ __forceinline__ float d_term(float nh, float a) {
    float a2 = a * a;
    float q = ggx_q(nh, ggx_a2m1(a2));
    return a2 * rcp(3.14159265358979323846f * (q * q));
}
// ggx.rs:13-18
__device__ __forceinline__ float g_term(float nv, float nl, float a) {
    float k = a * a * 0.5f;
    float g1 = nv * rcp(fmaf(nv, 1.0f - k, k));
    float g2 = nl * rcp(fmaf(nl, 1.0f - k, k));
    return g1 * g2;
}
// ggx.rs:20-45 with explicit uniforms. cos_theta/sin_theta are evaluated with the
// reference's exact operation order and IEEE-rounded ops: at roughness 0.01
// (a2 = 1e-8) both are cancellations whose result jumps by ~3e-4 on a one-ulp change,
// and this runs once per sampled specular lobe, so exactness costs nothing.
__device__ __forceinline__ void ggx_angles_exact(float roughness, float r2, float &cos_theta, float &sin_theta) {
#pragma clang fp contract(off)
    float a = roughness * roughness;
    float a2 = a * a;
    float t = (a2 - 1.0f) * r2;
    float den = 1.0f + t;
    cos_theta = sqrtf((1.0f - r2) / den);  // correctly rounded div and sqrt (hipcc default)
    float c2 = cos_theta * cos_theta;
    sin_theta = sqrtf(fmaxf(1.0f - c2, 0.0f));
}
__device__ __forceinline__ F3 sample_ggx_h(F3 n, float roughness, float r1, float r2) {
    float cos_theta, sin_theta;
    ggx_angles_exact(roughness, r2, cos_theta, sin_theta);
    float hx = cos2pi(r1) * sin_theta, hy = sin2pi(r1) * sin_theta;
    F3 u = normalize(any_orthonormal(n));
    F3 v = cross(n, u);
    return u * hx + v * hy + n * cos_theta;
}
// renderer.rs:330-335
// Both surface lobes in one piece of code (renderer.rs:330-336 and ggx.rs:20-45 share the frame around n, the
// azimuth and the combination; they differ in the polar angle and in what happens to the vector afterwards): a wave
// whose lanes picked different lobes runs the common part once instead of twice. Same operations per lobe as
// diffuse_dir / reflect(rd, sample_ggx_h()), hence the same bits.
__device__ __forceinline__ F3 lobe_dir(bool diffuse, F3 n, F3 rd, float roughness, float ra, float rb) {
    float cos_theta, sin_theta;
    if (diffuse) {
        sin_theta = fsqrt(rb);
        cos_theta = fsqrt(1.0f - rb);
    } else {
        ggx_angles_exact(roughness, rb, cos_theta, sin_theta);
    }
    const float hx = cos2pi(ra) * sin_theta, hy = sin2pi(ra) * sin_theta;
    const F3 u = normalize(any_orthonormal(n));
    const F3 v = cross(n, u);
    const F3 d = u * hx + v * hy + n * cos_theta;
    return diffuse ? normalize(d) : reflect(rd, d);
}
__device__ __forceinline__ F3 diffuse_dir(F3 n, float u_phi, float r2) {
    F3 u = normalize(any_orthonormal(n));
    F3 v = cross(n, u);
    float sr = fsqrt(r2);
    return normalize(u * (cos2pi(u_phi) * sr) + v * (sin2pi(u_phi) * sr) + n * fsqrt(1.0f - r2));
}
// renderer.rs:364-369
__device__ __forceinline__ bool refract(F3 v, F3 n, float eta_ratio, F3 &out) {
    float cos_theta = fminf(dot(-v, n), 1.0f);
    F3 perp = madd(v, n, cos_theta) * eta_ratio;
    float pp = dot(perp, perp);
    out = madd(perp, n, -fsqrt(fabsf(1.0f - pp)));
    return pp < 1.0f;
}
// renderer.rs:309-317. rd = incoming direction.
__device__ __forceinline__ F3 glass_dir(F3 rd, F3 n, float ior, float u) {
    float cosi = rclamp(-dot(rd, n), -1.0f, 1.0f);
    bool enter = cosi > 0.0f;
    float etai = enter ? 1.0f : ior, etat = enter ? ior : 1.0f;
    F3 hn = enter ? n : -n;
    float q = (etai - etat) * rcp(etai + etat);
    float r0 = q * q;
    float reflectance = fmaf(1.0f - r0, pow5(1.0f - fabsf(cosi)), r0);
    F3 refl = reflect(rd, hn);
    F3 refr;
    bool ok = refract(rd, hn, etai * rcp(etat), refr);
    return (u < reflectance || !ok) ? refl : refr;
}
// renderer.rs:194-202
__device__ __forceinline__ float sample_phase_function(float g, float r) {
    if (fabsf(g) < 1e-3f) return 1.0f - 2.0f * r;
    float g2 = g * g;
    float term = (1.0f - g2) * rcp(1.0f - g + 2.0f * g * r);
    return (1.0f + g2 - term * term) * rcp(2.0f * g);
}
// renderer.rs:276-282
__device__ __forceinline__ F3 scatter_dir(F3 rd, float cos_theta, float u_phi) {
    F3 u = normalize(any_orthonormal(rd));
    F3 v = cross(rd, u);
    float sin_theta = fsqrt(1.0f - cos_theta * cos_theta);
    return normalize(u * (cos2pi(u_phi) * sin_theta) + v * (sin2pi(u_phi) * sin_theta) + rd * cos_theta);
}

struct Mat {  // material.rs:4-11 after the clamps of renderer.rs:297-298
    F3 color;
    float metallic, roughness, ior, density, aniso;
};
template <class Tbl>
__device__ __forceinline__ Mat load_mat(const Tbl &tbl, uint32_t mat_off, uint32_t id) {
    float4 m0 = tbl.ld(mat_off + id * MAT_F4);
    float4 m1 = tbl.ld(mat_off + id * MAT_F4 + 1);
    Mat m;
    m.color = f3(m0);
    m.metallic = rclamp(m0.w, 0.0f, 1.0f);
    m.roughness = rclamp(m1.x, 0.01f, 1.0f);
    m.ior = m1.y;
    m.density = m1.z;
    m.aniso = m1.w;
    return m;
}

struct ShadeK {  // per-vertex invariants of renderer.rs:59,63,65-66,72
    F3 n, v, f0, diffuse_over_pi;
    float a2m1;     // a^2 - 1, a = roughness (un-squared, quirk Q7)
    float k;        // a^2 / 2 (ggx.rs:14)
    float four_nv;  // 4 n.v (:69)
    float spec_v;   // a^2 n.v / (pi (n.v (1 - k) + k)): the n.v-only factors of D G
};
__device__ __forceinline__ ShadeK make_shade(F3 n, F3 v, const Mat &m) {
#pragma clang fp contract(off)  // explicit fmaf only, as nee_eval
    ShadeK s;
    s.n = n;
    s.v = v;
    float om = 1.0f - m.metallic;
    s.f0 = F3{fmaf(m.color.x, m.metallic, 0.04f * om), fmaf(m.color.y, m.metallic, 0.04f * om),
              fmaf(m.color.z, m.metallic, 0.04f * om)};
    s.diffuse_over_pi = m.color * (om * 0.31830988618379067154f);
    float nv = fmaxf(dot(n, v), 1e-4f);
    float a2 = m.roughness * m.roughness;
    s.a2m1 = ggx_a2m1(a2);
    s.k = 0.5f * a2;
    s.four_nv = 4.0f * nv;
    s.spec_v = a2 * nv * rcp(3.14159265358979323846f * fmaf(nv, 1.0f - s.k, s.k));
    return s;
}

// renderer.rs:49-77: one visible light sample. Returns the contribution
// (already multiplied by intensity * area * falloff); `lit` tells whether the
// BRDF branch (:55) was reached. inv_dist2 = 1 / |lp - x|^2.
// The specular term D G / (4 n.v n.l + 1e-6) (ggx.rs:10-18, renderer.rs:65-70) is formed over ONE common
// denominator, q^2 (n.l (1 - k) + k) (4 n.v n.l + 1e-6), with the n.v-only factors hoisted into ShadeK:
// one v_rcp_f32 per sample instead of four (transcendentals issue at quarter rate).
// Contraction is spelt out (explicit fmaf, none implied): the lights loop evaluates a sample either in its own lane
// or, for the rays parked on the wave's list, later in another lane (Tracer::process_pairs); which of the two depends
// on what the wave has seen so far, and the two inlined copies must agree bit for bit (a roughness-0.01 highlight
// amplifies a last-bit difference 1e4 times; chunked and one-shot renders are compared to 2e-6).
__device__ __forceinline__ F3 nee_eval(const ShadeK &s, F3 l, float inv_dist2, F3 light_n, float light_area,
                                       F3 intensity, bool &lit) {
#pragma clang fp contract(off)
    // .max(0.0) of renderer.rs:49,53 is implied by the two `> 0` tests that follow it (:50,:55)
    float n_dot_l = dot(s.n, l);
    float cos_l = -dot(l, light_n);
    lit = (n_dot_l > 0.0f) & (cos_l > 0.0f);
    if (!lit) return F3{0.0f, 0.0f, 0.0f};
    float falloff = cos_l * inv_dist2;
    F3 h = normalize(s.v + l);
    float n_dot_h = fmaxf(dot(s.n, h), 0.0f);
    float v_dot_h = fmaxf(dot(s.v, h), 0.0f);
    F3 f = fresnel_schlick(v_dot_h, s.f0);
    float q = ggx_q(n_dot_h, s.a2m1);
    float den = (q * q) * fmaf(n_dot_l, 1.0f - s.k, s.k) * fmaf(s.four_nv, n_dot_l, 1e-6f);
    float spec = (s.spec_v * n_dot_l) * rcp(den);
    F3 kd = F3{1.0f - f.x, 1.0f - f.y, 1.0f - f.z};
    F3 brdf = F3{fmaf(s.diffuse_over_pi.x, kd.x, f.x * spec), fmaf(s.diffuse_over_pi.y, kd.y, f.y * spec),
                 fmaf(s.diffuse_over_pi.z, kd.z, f.z * spec)};
    float w = n_dot_l * (light_area * falloff);
    return brdf * intensity * w;
}

// renderer.rs:170-178 + algebra.rs:67-73
template <class Tbl>
__device__ __forceinline__ void camera_ray(const Tbl &tbl, const Params &P, uint32_t x, uint32_t y, float jx, float jy,
                                           float ds, float dt, F3 &ro, F3 &rd) {
    const float4 c0 = tbl.ld(P.cam_off), c1 = tbl.ld(P.cam_off + 1), c2 = tbl.ld(P.cam_off + 2);
    const float4 c3 = tbl.ld(P.cam_off + 3), c4 = tbl.ld(P.cam_off + 4);
    F3 pos = f3(c0), right = f3(c1), up = f3(c2), fwd = f3(c3);
    const float aspect = c0.w, scale = c1.w, aperture = c2.w, focus = c3.w;
    float u = (((float)x + jx) * c4.x - 0.5f) * 2.0f * aspect * scale;
    float v = -(((float)y + jy) * c4.y - 0.5f) * 2.0f * scale;
    F3 rd0 = normalize(right * u + up * v + fwd);
    float rad = aperture * fsqrt(dt);
    float dx = rad * cos2pi(ds), dy = rad * sin2pi(ds);
    F3 focal = madd(pos, rd0, focus);
    ro = pos + right * dx + up * dy;
    rd = normalize(focal - ro);
}

// tonemap.rs:10-19
__device__ __forceinline__ float aces1(float x) {
    const float a = 2.51f, b = 0.03f, c2 = 2.43f, d = 0.59f, e = 0.14f;
    return rclamp((x * (a * x + b)) / (x * (c2 * x + d) + e), 0.0f, 1.0f);
}
// renderer.rs:187-191 — correctly rounded division and libm powf here: the
// resolve runs once per pixel and its result is truncated to u8.
__device__ __forceinline__ uint8_t quantise(float tm) {
    float q = fminf(powf(tm, 1.0f / 2.2f) * 255.0f, 255.0f);
    if (!(q > 0.0f)) return 0;
    if (q >= 255.0f) return 255;
    return (uint8_t)q;
}

}  // namespace pt
