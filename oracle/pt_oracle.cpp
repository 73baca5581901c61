// pt_oracle.cpp — CPU restatement of the reference path tracer's hot path.
//
// TEST INFRASTRUCTURE ONLY. Nothing under pathtracer_amd/ may link, import or
// call this file; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it, and only as the checker / reported CPU baseline.
//
// What it restates (eliasstepanik/Pathtracer, paths under /root/reference):
//   src/algebra.rs, src/sphere.rs:20-39, src/plane.rs:19-56, src/object.rs:10-24,
//   src/ggx.rs, src/tonemap.rs:10-19, src/renderer.rs:19-85,156-369,
//   src/main.rs:41-51 (camera basis) and :126-144 (row-parallel loop).
// Each function cites the lines it follows. Arithmetic is IEEE f32, one
// rounding per operation in the reference's evaluation order; build with
// -ffp-contract=off (Rust never contracts a*b+c).
//
// PARITY STATUS: "parity unpinned" by reference tests — the reference holds no
// tests, golden vectors or images and its Rust toolchain is
// absent here, so the reference cannot be run. The restatement is pinned by
// closed-form known-answer tests derived from the cited source lines
// (tests/test_oracle_kat.py) and by the source text only.
//
// One deliberate difference: the reference draws from rand::thread_rng()
// (OS-seeded ChaCha, unreproducible). Every uniform here comes from an explicit
// argument or from the counter-seeded stream of DESIGN.md §RNG, drawn in the
// reference's source order, so that this restatement
// and the HIP kernels can be compared sample for sample.

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../include/hippt.h"

namespace {

constexpr float PI_F = 3.14159265358979323846f;  // std::f32::consts::PI
constexpr float INF_F = std::numeric_limits<float>::infinity();

// ---- src/algebra.rs:9-46 ---------------------------------------------------
struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(const float *p) { return V3{p[0], p[1], p[2]}; }
inline V3 scale(V3 a, float f) { return v3(a.x * f, a.y * f, a.z * f); }            // :13
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // :14
inline V3 cross(V3 a, V3 b) {                                                        // :15-17
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float norm(V3 a) { return std::sqrt(dot(a, a)); }                             // :18
inline V3 normalize(V3 a) { return scale(a, 1.0f / norm(a)); }                       // :19
inline V3 any_orthonormal(V3 a) {                                                    // :21-27
    if (std::fabs(a.z) < 0.9999999f) return v3(a.y, -a.x, 0.0f);
    return v3(0.0f, -a.z, a.y);
}
inline V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }           // :42
inline V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }           // :43
inline V3 mul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }           // :44
inline V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }                                 // :46

// Rust f32::max / f32::min: if one operand is NaN the other is returned.
inline float rmax(float a, float b) { return std::fmax(a, b); }
inline float rmin(float a, float b) { return std::fmin(a, b); }
// Rust f32::clamp: NaN stays NaN.
inline float rclamp(float x, float lo, float hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
inline float powi2(float x) { return x * x; }
inline float powi5(float x) {  // x.powi(5): square-and-multiply, x * (x^2)^2
    float x2 = x * x;
    float x4 = x2 * x2;
    return x * x4;
}

// ---- uniform stream (replaces rand::thread_rng) -------------
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
struct Rng {
    uint32_t s0, s1;
    Rng(uint32_t seed, uint32_t pixel, uint32_t sample) {
        uint32_t a = mix32(seed ^ 0x85ebca6bu);
        uint32_t b = mix32(pixel + a);
        // 63-bit stream key: start state and increment are two different hashes of (pixel, sample)
        uint32_t d = mix32(b ^ 0x5bd1e995u);
        s0 = mix32(sample ^ b);
        s1 = mix32(sample * 0x9e3779b1u + d) | 1u;
    }
    // rand's Standard f32: 24 random bits scaled by 2^-24, in [0,1).
    // (32-bit LCG with the stream's own odd increment; top 24 bits of the new state)
    float gen() {
        s0 = s0 * 2891336453u + s1;
        return (float)(s0 >> 8) * (1.0f / 16777216.0f);
    }
};

// ---- scene view -------------------------------------------------------------
struct Obj {  // src/object.rs:4-7, in scene order
    int kind;  // 0 sphere, 1 plane
    const hippt_sphere *s;
    const hippt_plane *p;
    hippt_material mat;
    int order;
    bool in_focus;
};

struct SceneView {
    const hippt_scene *sc;
    std::vector<Obj> objs;
    bool ok = true;
    explicit SceneView(const hippt_scene *scene) : sc(scene) {
        uint32_t n = scene->n_spheres + scene->n_planes;
        objs.resize(n);
        std::vector<char> seen(n, 0);
        auto place = [&](uint32_t order, Obj o) {
            if (order >= n || seen[order]) { ok = false; return; }
            seen[order] = 1;
            objs[order] = o;
        };
        for (uint32_t i = 0; i < scene->n_spheres; ++i) {
            const hippt_sphere *s = &scene->spheres[i];
            if (s->material >= scene->n_materials) { ok = false; continue; }
            place(s->order, Obj{0, s, nullptr, scene->materials[s->material], (int)s->order, s->in_focus != 0});
        }
        for (uint32_t i = 0; i < scene->n_planes; ++i) {
            const hippt_plane *p = &scene->planes[i];
            if (p->material >= scene->n_materials) { ok = false; continue; }
            place(p->order, Obj{1, nullptr, p, scene->materials[p->material], (int)p->order, p->in_focus != 0});
        }
    }
};

struct Hit {
    float t;
    V3 n;
    int idx;  // index into objs (scene order)
};

// src/sphere.rs:20-39
inline bool sphere_hit(const hippt_sphere *s, V3 ro, V3 rd, float *t_out, V3 *n_out) {
    V3 center = v3(s->center);
    V3 oc = sub(ro, center);
    float a = dot(rd, rd);
    float b = 2.0f * dot(oc, rd);
    float c = dot(oc, oc) - s->radius * s->radius;
    float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return false;
    float t = (-b - std::sqrt(disc)) / (2.0f * a);
    if (t <= 0.0f) return false;
    V3 hit = add(ro, scale(rd, t));
    *n_out = scale(sub(hit, center), 1.0f / s->radius);
    *t_out = t;
    return true;
}

// src/plane.rs:19-56
inline bool plane_hit(const hippt_plane *p, V3 ro, V3 rd, float *t_out, V3 *n_out) {
    V3 normal = v3(p->normal), point = v3(p->point), pu = v3(p->u), pv = v3(p->v);
    float denom = dot(normal, rd);
    if (std::fabs(denom) < 1e-6f) return false;
    float t = dot(sub(point, ro), normal) / denom;
    if (!std::isfinite(t) || t <= 1e-4f) return false;
    V3 hit_normal = (denom < 0.0f) ? normal : neg(normal);
    V3 hit = add(ro, scale(rd, t));
    V3 d = sub(hit, point);
    float du = dot(d, pu);
    float u2 = dot(pu, pu);
    if (std::fabs(du) > u2) return false;
    float dv = dot(d, pv);
    float v2 = dot(pv, pv);
    if (std::fabs(dv) > v2) return false;
    *t_out = t;
    *n_out = hit_normal;
    return true;
}

// src/object.rs:10-17
inline bool obj_hit(const Obj &o, V3 ro, V3 rd, float *t, V3 *n) {
    return o.kind == 0 ? sphere_hit(o.s, ro, rd, t, n) : plane_hit(o.p, ro, rd, t, n);
}

// src/renderer.rs:356-362 — Iterator::min_by keeps the FIRST minimum. (t is
// positive and finite for every hit the two tests return on finite input, so
// total_cmp reduces to '<'.)
inline bool intersect_closest(const SceneView &sv, V3 ro, V3 rd, Hit *out, bool focus_only = false) {
    bool found = false;
    Hit best{INF_F, v3(0, 0, 0), -1};
    for (size_t i = 0; i < sv.objs.size(); ++i) {
        if (focus_only && !sv.objs[i].in_focus) continue;
        float t;
        V3 n;
        if (obj_hit(sv.objs[i], ro, rd, &t, &n)) {
            if (!found || t < best.t) {
                best = Hit{t, n, (int)i};
                found = true;
            }
        }
    }
    *out = best;
    return found;
}

// ---- src/ggx.rs -------------------------------------------------------------
inline V3 reflect(V3 v, V3 n) { return sub(v, scale(scale(n, 2.0f), dot(v, n))); }  // :5  v - n*2.0*v.dot(n)
inline V3 fresnel_schlick(float cos_theta, V3 f0) {                                  // :7-9
    return add(f0, scale(sub(v3(1, 1, 1), f0), powi5(1.0f - cos_theta)));
}
inline float d_term(float nh, float a) {                                             // :10-12
    float a2 = a * a;
    return a2 / (PI_F * powi2(nh * nh * (a2 - 1.0f) + 1.0f));
}
inline float g_term(float nv, float nl, float a) {                                   // :13-18
    float k = a * a / 2.0f;
    float g1 = nv / (nv * (1.0f - k) + k);
    float g2 = nl / (nl * (1.0f - k) + k);
    return g1 * g2;
}
inline V3 sample_ggx_h(V3 n, float roughness, float r1, float r2) {                  // :20-45
    float a = roughness * roughness;
    float a2 = a * a;
    float phi = 2.0f * PI_F * r1;
    float cos_theta = std::sqrt((1.0f - r2) / (1.0f + (a2 - 1.0f) * r2));
    float sin_theta = std::sqrt(rmax(1.0f - cos_theta * cos_theta, 0.0f));
    V3 ht = v3(std::cos(phi) * sin_theta, std::sin(phi) * sin_theta, cos_theta);
    V3 w = n;
    V3 u = normalize(any_orthonormal(n));
    V3 v = cross(w, u);
    return add(add(scale(u, ht.x), scale(v, ht.y)), scale(w, ht.z));
}

// ---- src/renderer.rs:364-369 --------------------------------------------------
inline bool refract(V3 v, V3 n, float eta_ratio, V3 *out) {
    float cos_theta = rmin(dot(neg(v), n), 1.0f);
    V3 r_out_perp = scale(add(v, scale(n, cos_theta)), eta_ratio);
    V3 r_out_parallel = scale(n, -std::sqrt(std::fabs(1.0f - dot(r_out_perp, r_out_perp))));
    *out = add(r_out_perp, r_out_parallel);
    return dot(r_out_perp, r_out_perp) < 1.0f;
}

// src/renderer.rs:309-317: next direction at a glass vertex. rd = -v.
inline V3 glass_dir(V3 rd, V3 n, float ior, float u, int *branch) {
    V3 v = neg(rd);
    float cosi = rclamp(dot(v, n), -1.0f, 1.0f);
    float etai = cosi > 0.0f ? 1.0f : ior;
    float etat = cosi > 0.0f ? ior : 1.0f;
    V3 hit_normal = cosi > 0.0f ? n : neg(n);
    float r0 = powi2((etai - etat) / (etai + etat));
    float reflectance = r0 + (1.0f - r0) * powi5(1.0f - std::fabs(cosi));
    if (u < reflectance) {
        if (branch) *branch = 0;
        return reflect(neg(v), hit_normal);
    }
    V3 rdir;
    if (refract(neg(v), hit_normal, etai / etat, &rdir)) {
        if (branch) *branch = 1;
        return rdir;
    }
    if (branch) *branch = 2;
    return reflect(neg(v), hit_normal);
}

// src/renderer.rs:194-202 (one uniform in either branch)
inline float sample_phase_function(float g, float r) {
    if (std::fabs(g) < 1e-3f) return 1.0f - 2.0f * r;
    float g2 = g * g;
    float term = (1.0f - g2) / (1.0f - g + 2.0f * g * r);
    return (1.0f + g2 - term * term) / (2.0f * g);
}

// src/renderer.rs:276-282
inline V3 scatter_dir(V3 rd, float cos_theta, float u_phi) {
    V3 w = rd;
    V3 u = normalize(any_orthonormal(w));
    V3 v_cross = cross(w, u);
    float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
    float phi = 2.0f * PI_F * u_phi;
    return normalize(add(add(scale(scale(u, std::cos(phi)), sin_theta),
                             scale(scale(v_cross, std::sin(phi)), sin_theta)),
                         scale(w, cos_theta)));
}

// src/renderer.rs:330-335
inline V3 diffuse_dir(V3 n, float u_phi, float r2) {
    V3 w = n;
    V3 u = normalize(any_orthonormal(w));
    V3 v_cross = cross(w, u);
    float phi = 2.0f * PI_F * u_phi;
    return normalize(add(add(scale(scale(u, std::cos(phi)), std::sqrt(r2)),
                             scale(scale(v_cross, std::sin(phi)), std::sqrt(r2))),
                         scale(w, std::sqrt(1.0f - r2))));
}

// src/renderer.rs:49-77 — contribution of one visible light sample.
inline V3 nee_eval(V3 n, V3 v, V3 l, float dist2, const hippt_material &mat, const hippt_light &light) {
    V3 out = v3(0, 0, 0);
    float n_dot_l = rmax(dot(n, l), 0.0f);
    if (n_dot_l > 0.0f) {
        V3 lu = v3(light.u), lv = v3(light.v);
        float light_area = norm(cross(lu, lv));
        V3 light_normal = normalize(cross(lu, lv));
        float cos_theta_light = rmax(dot(neg(l), light_normal), 0.0f);
        if (cos_theta_light > 0.0f) {
            float falloff = cos_theta_light / dist2;
            V3 h = normalize(add(v, l));
            float n_dot_v = rmax(dot(n, v), 1e-4f);
            float n_dot_h = rmax(dot(n, h), 0.0f);
            float v_dot_h = rmax(dot(v, h), 0.0f);
            V3 color = v3(mat.rgb);
            V3 f0 = add(scale(v3(0.04f, 0.04f, 0.04f), 1.0f - mat.metallic), scale(color, mat.metallic));
            V3 f = fresnel_schlick(v_dot_h, f0);
            float d = d_term(n_dot_h, mat.roughness);
            float g = g_term(n_dot_v, n_dot_l, mat.roughness);
            V3 spec_numerator = scale(scale(f, d), g);
            float spec_denominator = 4.0f * n_dot_v * n_dot_l;
            V3 specular_brdf = scale(spec_numerator, 1.0f / (spec_denominator + 1e-6f));
            V3 diffuse_color = scale(color, 1.0f - mat.metallic);
            V3 k_d = sub(v3(1, 1, 1), f);
            V3 diffuse_brdf = scale(mul(diffuse_color, k_d), 1.0f / PI_F);
            V3 radiance = scale(add(diffuse_brdf, specular_brdf), n_dot_l);
            out = scale(mul(radiance, v3(light.intensity)), light_area * falloff);
        }
    }
    return out;
}

struct Counters {
    uint64_t c[HIPPT_N_COUNTERS];
    Counters() { std::memset(c, 0, sizeof(c)); }
};

// src/renderer.rs:46 — objs.iter().any(hit && t*t < dist2*0.999), scene order,
// stops at the first occluder.
inline bool occluded(const SceneView &sv, V3 ro, V3 rd, float dist2, Counters *cn) {
    for (const Obj &o : sv.objs) {
        float t;
        V3 n;
        if (cn) cn->c[o.kind == 0 ? HIPPT_C_SHADOW_SPHERE_TESTS : HIPPT_C_SHADOW_PLANE_TESTS]++;
        if (obj_hit(o, ro, rd, &t, &n) && t * t < dist2 * 0.999f) return true;
    }
    return false;
}

// src/renderer.rs:19-85
inline V3 direct_light_sample(const SceneView &sv, V3 hit, V3 n, V3 v, const hippt_material &mat, Rng &rng,
                              Counters *cn) {
    V3 total = v3(0, 0, 0);
    const uint32_t SHADOW_SAMPLES = 4;
    if (cn) cn->c[HIPPT_C_NEE_VERTICES]++;
    for (uint32_t li = 0; li < sv.sc->n_lights; ++li) {
        const hippt_light &light = sv.sc->lights[li];
        V3 contrib = v3(0, 0, 0);
        for (uint32_t s = 0; s < SHADOW_SAMPLES; ++s) {
            float e1 = rng.gen();
            float e2 = rng.gen();
            V3 lp = add(add(v3(light.pos), scale(v3(light.u), e1 - 0.5f)), scale(v3(light.v), e2 - 0.5f));
            V3 lvec = sub(lp, hit);
            float dist2 = dot(lvec, lvec);
            V3 l = normalize(lvec);
            V3 shadow_ro = add(hit, scale(l, 1e-4f));
            if (cn) cn->c[HIPPT_C_SHADOW_RAYS]++;
            if (occluded(sv, shadow_ro, l, dist2, cn)) continue;
            V3 r = nee_eval(n, v, l, dist2, mat, light);
            if (cn) {
                // count the evaluations that reach the BRDF (:55)
                float ndl = rmax(dot(n, l), 0.0f);
                if (ndl > 0.0f) {
                    V3 ln = normalize(cross(v3(light.u), v3(light.v)));
                    if (rmax(dot(neg(l), ln), 0.0f) > 0.0f) cn->c[HIPPT_C_LIT_EVALS]++;
                }
            }
            contrib = add(contrib, r);
        }
        total = add(total, scale(contrib, 1.0f / (float)SHADOW_SAMPLES));
    }
    return total;
}

constexpr uint32_t RUSSIAN_ROULETTE_DEPTH = 2;  // src/renderer.rs:16

struct Media {
    bool some;
    hippt_material m;
};

// src/renderer.rs:237-353 — literal recursive form.
V3 trace(const SceneView &sv, V3 ro, V3 rd, uint32_t depth, uint32_t max_depth, Rng &rng, Media current_media,
         Counters *cn) {
    if (depth >= max_depth) return v3(0, 0, 0);

    Hit sh;
    bool has_hit = intersect_closest(sv, ro, rd, &sh);
    if (cn) {
        cn->c[HIPPT_C_EXTEND_RAYS]++;
        cn->c[HIPPT_C_EXTEND_SPHERE_TESTS] += sv.sc->n_spheres;
        cn->c[HIPPT_C_EXTEND_PLANE_TESTS] += sv.sc->n_planes;
        if (has_hit) cn->c[HIPPT_C_EXTEND_HITS]++;
    }
    float t_surface = has_hit ? sh.t : INF_F;

    float t_media = INF_F;
    V3 absorption = v3(1, 1, 1);
    if (current_media.some) {
        const hippt_material &media = current_media.m;
        if (media.volume_density > 0.0f) {
            if (cn) cn->c[HIPPT_C_MEDIUM_VERTICES]++;
            float sample_dist = -std::log(rng.gen()) / media.volume_density;
            t_media = sample_dist;
            V3 coeff = v3(rmax(1.0f - media.rgb[0], 0.0f) * media.volume_density,
                          rmax(1.0f - media.rgb[1], 0.0f) * media.volume_density,
                          rmax(1.0f - media.rgb[2], 0.0f) * media.volume_density);
            V3 e = scale(neg(coeff), rmin(t_media, t_surface));
            absorption = v3(std::exp(e.x), std::exp(e.y), std::exp(e.z));
        }
    }

    if (t_media < t_surface) {  // :269
        if (cn) cn->c[HIPPT_C_SCATTER_EVENTS]++;
        V3 hit_point = add(ro, scale(rd, t_media));
        V3 direct = direct_light_sample(sv, hit_point, v3(0, 1, 0), neg(rd), current_media.m, rng, cn);
        float cos_theta = sample_phase_function(current_media.m.volume_anisotropy, rng.gen());
        float u_phi = rng.gen();
        V3 next_dir = scatter_dir(rd, cos_theta, u_phi);
        return mul(add(direct, trace(sv, hit_point, next_dir, depth + 1, max_depth, rng, current_media, cn)),
                   absorption);
    }

    if (!has_hit) return mul(v3(0, 0, 0), absorption);  // :291

    float t = sh.t;
    V3 n = sh.n;
    hippt_material mat = sv.objs[sh.idx].mat;
    V3 hit = add(ro, scale(rd, t));
    V3 v = neg(rd);

    mat.metallic = rclamp(mat.metallic, 0.0f, 1.0f);    // :297
    mat.roughness = rclamp(mat.roughness, 0.01f, 1.0f);  // :298

    Media next_media;  // :301-305
    if (mat.volume_density > 0.0f) {
        if (dot(v, n) > 0.0f) next_media = Media{true, mat};
        else next_media = Media{false, mat};
    } else {
        next_media = current_media;
    }

    if (mat.ior > 1.0f && mat.metallic < 0.1f) {  // :307
        if (cn) cn->c[HIPPT_C_GLASS_VERTICES]++;
        V3 next_dir = glass_dir(rd, n, mat.ior, rng.gen(), nullptr);
        return mul(trace(sv, add(hit, scale(next_dir, 1e-4f)), next_dir, depth + 1, max_depth, rng, next_media, cn),
                   absorption);
    }

    V3 direct = direct_light_sample(sv, hit, n, v, mat, rng, cn);  // :323
    V3 indirect = v3(0, 0, 0);

    float p = rmax(rmax(mat.rgb[0], mat.rgb[1]), mat.rgb[2]);  // :327
    if (depth < RUSSIAN_ROULETTE_DEPTH || rng.gen() < p) {
        V3 next_dir, brdf;
        if (rng.gen() < (1.0f - mat.metallic)) {  // :329
            if (cn) cn->c[HIPPT_C_DIFFUSE_LOBES]++;
            float u_phi = rng.gen();
            float r2 = rng.gen();
            next_dir = diffuse_dir(n, u_phi, r2);
            brdf = scale(v3(mat.rgb), 1.0f / PI_F);
        } else {
            if (cn) cn->c[HIPPT_C_SPEC_LOBES]++;
            float r1 = rng.gen();
            float r2 = rng.gen();
            V3 h = sample_ggx_h(n, mat.roughness, r1, r2);
            next_dir = reflect(neg(v), h);
            brdf = v3(1, 1, 1);
        }
        if (dot(next_dir, n) > 0.0f) {  // :343
            V3 incoming =
                trace(sv, add(hit, scale(next_dir, 1e-4f)), next_dir, depth + 1, max_depth, rng, next_media, cn);
            indirect = scale(mul(incoming, brdf), dot(next_dir, n));
            if (depth >= RUSSIAN_ROULETTE_DEPTH) indirect = scale(indirect, 1.0f / p);
        }
    }
    return mul(add(direct, indirect), absorption);  // :352
}

// The same integrator unrolled into the throughput-carrying loop the HIP
// kernels implement. Kept to prove the two forms agree.
V3 trace_iterative(const SceneView &sv, V3 ro, V3 rd, uint32_t max_depth, Rng &rng) {
    V3 T = v3(1, 1, 1), L = v3(0, 0, 0);
    Media media{false, hippt_material{}};
    for (uint32_t depth = 0; depth < max_depth; ++depth) {
        Hit sh;
        bool has_hit = intersect_closest(sv, ro, rd, &sh);
        float t_surface = has_hit ? sh.t : INF_F;
        float t_media = INF_F;
        V3 A = v3(1, 1, 1);
        if (media.some && media.m.volume_density > 0.0f) {
            t_media = -std::log(rng.gen()) / media.m.volume_density;
            V3 coeff = v3(rmax(1.0f - media.m.rgb[0], 0.0f) * media.m.volume_density,
                          rmax(1.0f - media.m.rgb[1], 0.0f) * media.m.volume_density,
                          rmax(1.0f - media.m.rgb[2], 0.0f) * media.m.volume_density);
            V3 e = scale(neg(coeff), rmin(t_media, t_surface));
            A = v3(std::exp(e.x), std::exp(e.y), std::exp(e.z));
        }
        if (t_media < t_surface) {
            V3 x = add(ro, scale(rd, t_media));
            V3 D = direct_light_sample(sv, x, v3(0, 1, 0), neg(rd), media.m, rng, nullptr);
            float cos_theta = sample_phase_function(media.m.volume_anisotropy, rng.gen());
            float u_phi = rng.gen();
            V3 nd = scatter_dir(rd, cos_theta, u_phi);
            T = mul(T, A);
            L = add(L, mul(T, D));
            ro = x;
            rd = nd;
            continue;
        }
        if (!has_hit) break;
        hippt_material mat = sv.objs[sh.idx].mat;
        V3 n = sh.n;
        V3 hit = add(ro, scale(rd, sh.t));
        V3 v = neg(rd);
        mat.metallic = rclamp(mat.metallic, 0.0f, 1.0f);
        mat.roughness = rclamp(mat.roughness, 0.01f, 1.0f);
        Media next_media = media;
        if (mat.volume_density > 0.0f) next_media = Media{dot(v, n) > 0.0f, mat};
        if (mat.ior > 1.0f && mat.metallic < 0.1f) {
            V3 nd = glass_dir(rd, n, mat.ior, rng.gen(), nullptr);
            T = mul(T, A);
            ro = add(hit, scale(nd, 1e-4f));
            rd = nd;
            media = next_media;
            continue;
        }
        V3 D = direct_light_sample(sv, hit, n, v, mat, rng, nullptr);
        float p = rmax(rmax(mat.rgb[0], mat.rgb[1]), mat.rgb[2]);
        T = mul(T, A);
        L = add(L, mul(T, D));
        if (!(depth < RUSSIAN_ROULETTE_DEPTH || rng.gen() < p)) break;
        V3 nd, f;
        if (rng.gen() < (1.0f - mat.metallic)) {
            float u_phi = rng.gen();
            float r2 = rng.gen();
            nd = diffuse_dir(n, u_phi, r2);
            f = scale(v3(mat.rgb), 1.0f / PI_F);
        } else {
            float r1 = rng.gen();
            float r2 = rng.gen();
            nd = reflect(neg(v), sample_ggx_h(n, mat.roughness, r1, r2));
            f = v3(1, 1, 1);
        }
        float c = dot(nd, n);
        if (!(c > 0.0f)) break;
        T = scale(mul(T, f), c);
        if (depth >= RUSSIAN_ROULETTE_DEPTH) T = scale(T, 1.0f / p);
        ro = add(hit, scale(nd, 1e-4f));
        rd = nd;
        media = next_media;
    }
    return L;
}

// src/renderer.rs:170-178 + src/algebra.rs:67-73
inline void camera_ray(const hippt_camera &cam, uint32_t w, uint32_t h, uint32_t x, uint32_t y, float jx, float jy,
                       float ds, float dt, V3 *ro, V3 *rd) {
    V3 right = v3(cam.right), up = v3(cam.up), forward = v3(cam.forward), pos = v3(cam.pos);
    float u = (((float)x + jx) / (float)w - 0.5f) * 2.0f * cam.aspect * cam.scale;
    float v = -(((float)y + jy) / (float)h - 0.5f) * 2.0f * cam.scale;
    V3 rd0 = normalize(add(add(scale(right, u), scale(up, v)), forward));
    float ang = 2.0f * PI_F * ds;  // sample_disk(aperture)
    float rad = cam.aperture * std::sqrt(dt);
    float dx = rad * std::cos(ang), dy = rad * std::sin(ang);
    V3 focal_pt = add(pos, scale(rd0, cam.focus));
    V3 origin = add(add(pos, scale(right, dx)), scale(up, dy));
    *ro = origin;
    *rd = normalize(sub(focal_pt, origin));
}

// src/tonemap.rs:10-19
inline float aces1(float x) {
    const float a = 2.51f, b = 0.03f, c2 = 2.43f, d = 0.59f, e = 0.14f;
    return rclamp((x * (a * x + b)) / (x * (c2 * x + d) + e), 0.0f, 1.0f);
}
// src/renderer.rs:187-191: (x.powf(1/2.2)*255).min(255) as u8 (saturating, NaN->0
// after the min, but min(NaN,255)=255).
inline uint8_t quantise(float tm) {
    float q = rmin(std::pow(tm, 1.0f / 2.2f) * 255.0f, 255.0f);
    if (!(q > 0.0f)) return 0;
    if (q >= 255.0f) return 255;
    return (uint8_t)q;
}

V3 path_radiance(const SceneView &sv, uint32_t x, uint32_t y, uint32_t sample, uint32_t seed, uint32_t max_depth,
                 bool iterative, Counters *cn) {
    const hippt_scene *sc = sv.sc;
    Rng rng(seed, y * sc->width + x, sample);
    float jx = rng.gen();
    float jy = rng.gen();
    float ds = rng.gen();
    float dt = rng.gen();
    V3 ro, rd;
    camera_ray(sc->camera, sc->width, sc->height, x, y, jx, jy, ds, dt, &ro, &rd);
    if (cn) cn->c[HIPPT_C_PATHS]++;
    if (iterative) return trace_iterative(sv, ro, rd, max_depth, rng);
    return trace(sv, ro, rd, 0, max_depth, rng, Media{false, hippt_material{}}, cn);
}

thread_local char g_err[256] = "";

int render_accum(const hippt_scene *scene, const hippt_options *opts, float *out, uint64_t *counters, bool iterative,
                 int n_threads) {
    SceneView sv(scene);
    if (!sv.ok) {
        std::snprintf(g_err, sizeof g_err, "inconsistent scene tables");
        return HIPPT_ERR_INVALID;
    }
    hippt_options o{};
    if (opts) o = *opts;
    uint32_t s0 = o.sample_begin;
    uint32_t cnt = o.sample_count ? o.sample_count : (scene->samples > s0 ? scene->samples - s0 : 0);
    uint32_t max_depth = o.max_depth ? o.max_depth : 5;  // MAX_DEPTH, src/renderer.rs:15
    uint32_t W = scene->width, H = scene->height;
    bool want_cn = counters && (o.flags & HIPPT_FLAG_COUNTERS);
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    n_threads = std::min<int>(n_threads, (int)std::max(1u, H));
    std::atomic<uint32_t> next_row{0};
    std::vector<Counters> tc(n_threads);
    // one task per image row, as src/main.rs:126
    auto worker = [&](int tid) {
        for (;;) {
            uint32_t y = next_row.fetch_add(1);
            if (y >= H) break;
            for (uint32_t x = 0; x < W; ++x) {
                V3 col = v3(0, 0, 0);
                for (uint32_t s = 0; s < cnt; ++s) {
                    V3 r = path_radiance(sv, x, y, s0 + s, o.seed, max_depth, iterative, want_cn ? &tc[tid] : nullptr);
                    col = add(col, r);  // src/renderer.rs:181
                }
                float *px = out + ((size_t)y * W + x) * 3;
                px[0] = col.x;
                px[1] = col.y;
                px[2] = col.z;
            }
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n_threads; ++i) th.emplace_back(worker, i);
    worker(0);
    for (auto &t : th) t.join();
    if (want_cn) {
        for (int k = 0; k < HIPPT_N_COUNTERS; ++k) {
            counters[k] = 0;
            for (auto &c : tc) counters[k] += c.c[k];
        }
    }
    return HIPPT_OK;
}

}  // namespace

extern "C" {

const char *orc_last_error(void) { return g_err; }

int orc_render_accum_f32(const hippt_scene *scene, const hippt_options *opts, float *out, uint64_t *counters) {
    return render_accum(scene, opts, out, counters, false, 0);
}
// extra knobs: iterative form, thread count (0 = all cores)
int orc_render_accum_ex(const hippt_scene *scene, const hippt_options *opts, float *out, uint64_t *counters,
                        int iterative, int n_threads) {
    return render_accum(scene, opts, out, counters, iterative != 0, n_threads);
}

// src/renderer.rs:184-191 on radiance sums
void orc_resolve(const float *accum, uint32_t n_pixels, uint32_t spp, uint8_t *rgba) {
    float inv = 1.0f / (float)spp;
    for (uint32_t i = 0; i < n_pixels; ++i) {
        for (int c = 0; c < 3; ++c) rgba[4 * i + c] = quantise(aces1(accum[3 * i + c] * inv));
        rgba[4 * i + 3] = 255;
    }
}

int orc_render_rgba8(const hippt_scene *scene, const hippt_options *opts, uint8_t *out_rgba) {
    std::vector<float> acc((size_t)scene->width * scene->height * 3);
    hippt_options o{};
    if (opts) o = *opts;
    o.sample_begin = 0;
    o.sample_count = scene->samples;
    int rc = render_accum(scene, &o, acc.data(), nullptr, false, 0);
    if (rc) return rc;
    orc_resolve(acc.data(), scene->width * scene->height, scene->samples, out_rgba);
    return HIPPT_OK;
}

// src/main.rs:41-51
void orc_camera_basis(const float pos[3], const float look_at[3], const float up[3], float fov_deg, uint32_t width,
                      uint32_t height, hippt_camera *out) {
    float fov_rad = fov_deg * (PI_F / 180.0f);  // f32::to_radians
    out->aspect = (float)width / (float)height;
    out->scale = std::tan(fov_rad * 0.5f);
    V3 p = v3(pos), la = v3(look_at), upv = v3(up);
    V3 forward = normalize(sub(la, p));
    V3 right = normalize(cross(upv, forward));
    V3 real_up = normalize(cross(forward, right));
    std::memcpy(out->pos, pos, 12);
    out->right[0] = right.x, out->right[1] = right.y, out->right[2] = right.z;
    out->up[0] = real_up.x, out->up[1] = real_up.y, out->up[2] = real_up.z;
    out->forward[0] = forward.x, out->forward[1] = forward.y, out->forward[2] = forward.z;
}

// src/renderer.rs:204-234
float orc_autofocus(const hippt_scene *scene) {
    SceneView sv(scene);
    bool any_focus = false;
    for (const Obj &o : sv.objs) any_focus |= o.in_focus;
    if (!any_focus) return 5.0f;
    const hippt_camera &cam = scene->camera;
    uint32_t w = scene->width, h = scene->height;
    float sum = 0.0f;
    uint32_t cnt = 0;
    for (int i = 0; i < 5; ++i) {
        for (int j = 0; j < 5; ++j) {
            float px = (float)(w / 2) + ((float)i - 2.0f);
            float py = (float)(h / 2) + ((float)j - 2.0f);
            float u = (px / (float)w - 0.5f) * 2.0f * cam.aspect * cam.scale;
            float v = -((py / (float)h - 0.5f) * 2.0f * cam.scale);
            V3 dir = normalize(add(add(scale(v3(cam.right), u), scale(v3(cam.up), v)), v3(cam.forward)));
            Hit hh;
            if (intersect_closest(sv, v3(cam.pos), dir, &hh, true)) {
                sum += hh.t;  // Iterator::sum in push order
                cnt++;
            }
        }
    }
    return cnt == 0 ? 10.0f : sum / (float)cnt;
}

// ---- unit exports: same signatures as hippt_test_* (device ignored) ---------
int orc_test_intersect(const hippt_scene *scene, uint32_t n, const float *ro, const float *rd, float *out_t,
                       float *out_normal, int32_t *out_index, int) {
    SceneView sv(scene);
    if (!sv.ok) return HIPPT_ERR_INVALID;
    for (uint32_t i = 0; i < n; ++i) {
        Hit h;
        bool ok = intersect_closest(sv, v3(ro + 3 * i), v3(rd + 3 * i), &h);
        out_t[i] = ok ? h.t : INF_F;
        out_normal[3 * i] = ok ? h.n.x : 0.0f;
        out_normal[3 * i + 1] = ok ? h.n.y : 0.0f;
        out_normal[3 * i + 2] = ok ? h.n.z : 0.0f;
        out_index[i] = ok ? h.idx : -1;
    }
    return HIPPT_OK;
}
int orc_test_occluded(const hippt_scene *scene, uint32_t n, const float *ro, const float *rd, const float *dist2,
                      uint8_t *out, int) {
    SceneView sv(scene);
    if (!sv.ok) return HIPPT_ERR_INVALID;
    for (uint32_t i = 0; i < n; ++i) out[i] = occluded(sv, v3(ro + 3 * i), v3(rd + 3 * i), dist2[i], nullptr) ? 1 : 0;
    return HIPPT_OK;
}
int orc_test_nee_eval(uint32_t n, const float *nrm, const float *v, const float *l, const float *dist2,
                      const float *mat, const hippt_light *light, float *out_rgb, int) {
    for (uint32_t i = 0; i < n; ++i) {
        hippt_material m;
        std::memcpy(&m, mat + 8 * i, sizeof m);
        V3 r = nee_eval(v3(nrm + 3 * i), v3(v + 3 * i), v3(l + 3 * i), dist2[i], m, *light);
        out_rgb[3 * i] = r.x, out_rgb[3 * i + 1] = r.y, out_rgb[3 * i + 2] = r.z;
    }
    return HIPPT_OK;
}
int orc_test_ggx_terms(uint32_t n, const float *nh, const float *nv, const float *nl, const float *vh,
                       const float *rough, const float *f0, float *out5, int) {
    for (uint32_t i = 0; i < n; ++i) {
        V3 f = fresnel_schlick(vh[i], v3(f0 + 3 * i));
        out5[5 * i] = f.x, out5[5 * i + 1] = f.y, out5[5 * i + 2] = f.z;
        out5[5 * i + 3] = d_term(nh[i], rough[i]);
        out5[5 * i + 4] = g_term(nv[i], nl[i], rough[i]);
    }
    return HIPPT_OK;
}
int orc_test_ggx_sample(uint32_t n, const float *nrm, const float *rough, const float *r1, const float *r2,
                        float *out_h, int) {
    for (uint32_t i = 0; i < n; ++i) {
        V3 h = sample_ggx_h(v3(nrm + 3 * i), rough[i], r1[i], r2[i]);
        out_h[3 * i] = h.x, out_h[3 * i + 1] = h.y, out_h[3 * i + 2] = h.z;
    }
    return HIPPT_OK;
}
int orc_test_diffuse_sample(uint32_t n, const float *nrm, const float *u_phi, const float *r2, float *out_dir, int) {
    for (uint32_t i = 0; i < n; ++i) {
        V3 d = diffuse_dir(v3(nrm + 3 * i), u_phi[i], r2[i]);
        out_dir[3 * i] = d.x, out_dir[3 * i + 1] = d.y, out_dir[3 * i + 2] = d.z;
    }
    return HIPPT_OK;
}
int orc_test_glass(uint32_t n, const float *rd, const float *nrm, const float *ior, const float *u, float *out_dir,
                   int) {
    for (uint32_t i = 0; i < n; ++i) {
        V3 d = glass_dir(v3(rd + 3 * i), v3(nrm + 3 * i), ior[i], u[i], nullptr);
        out_dir[3 * i] = d.x, out_dir[3 * i + 1] = d.y, out_dir[3 * i + 2] = d.z;
    }
    return HIPPT_OK;
}
// refract alone (src/renderer.rs:364-369): out4 = {valid, x, y, z}
int orc_test_refract(uint32_t n, const float *v, const float *nrm, const float *eta, float *out4) {
    for (uint32_t i = 0; i < n; ++i) {
        V3 r;
        bool ok = refract(v3(v + 3 * i), v3(nrm + 3 * i), eta[i], &r);
        out4[4 * i] = ok ? 1.0f : 0.0f, out4[4 * i + 1] = r.x, out4[4 * i + 2] = r.y, out4[4 * i + 3] = r.z;
    }
    return HIPPT_OK;
}
int orc_test_reflect(uint32_t n, const float *v, const float *nrm, float *out) {
    for (uint32_t i = 0; i < n; ++i) {
        V3 r = reflect(v3(v + 3 * i), v3(nrm + 3 * i));
        out[3 * i] = r.x, out[3 * i + 1] = r.y, out[3 * i + 2] = r.z;
    }
    return HIPPT_OK;
}
int orc_test_any_orthonormal(uint32_t n, const float *v, float *out) {
    for (uint32_t i = 0; i < n; ++i) {
        V3 r = any_orthonormal(v3(v + 3 * i));
        out[3 * i] = r.x, out[3 * i + 1] = r.y, out[3 * i + 2] = r.z;
    }
    return HIPPT_OK;
}
int orc_test_hg(uint32_t n, const float *rd, const float *g, const float *u_hg, const float *u_phi, float *out4,
                int) {
    for (uint32_t i = 0; i < n; ++i) {
        float c = sample_phase_function(g[i], u_hg[i]);
        V3 d = scatter_dir(v3(rd + 3 * i), c, u_phi[i]);
        out4[4 * i] = c, out4[4 * i + 1] = d.x, out4[4 * i + 2] = d.y, out4[4 * i + 3] = d.z;
    }
    return HIPPT_OK;
}
int orc_test_camera_ray(const hippt_scene *scene, uint32_t n, const uint32_t *xy, const float *u4, float *out_ro,
                        float *out_rd, int) {
    for (uint32_t i = 0; i < n; ++i) {
        V3 ro, rd;
        camera_ray(scene->camera, scene->width, scene->height, xy[2 * i], xy[2 * i + 1], u4[4 * i], u4[4 * i + 1],
                   u4[4 * i + 2], u4[4 * i + 3], &ro, &rd);
        out_ro[3 * i] = ro.x, out_ro[3 * i + 1] = ro.y, out_ro[3 * i + 2] = ro.z;
        out_rd[3 * i] = rd.x, out_rd[3 * i + 1] = rd.y, out_rd[3 * i + 2] = rd.z;
    }
    return HIPPT_OK;
}
int orc_test_tonemap(uint32_t n, const float *rgb, uint8_t *out_rgb8, int) {
    for (uint32_t i = 0; i < 3 * n; ++i) out_rgb8[i] = quantise(aces1(rgb[i]));
    return HIPPT_OK;
}
// aces_film alone (src/tonemap.rs:10-19)
int orc_test_aces(uint32_t n, const float *x, float *out) {
    for (uint32_t i = 0; i < n; ++i) out[i] = aces1(x[i]);
    return HIPPT_OK;
}
int orc_test_rng(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float *out, int) {
    Rng r(seed, pixel, sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = r.gen();
    return HIPPT_OK;
}
// radiance of single paths (x, y, sample) — for sample-exact comparisons
int orc_test_paths(const hippt_scene *scene, const hippt_options *opts, uint32_t n, const uint32_t *xys,
                   float *out_rgb, int iterative) {
    SceneView sv(scene);
    if (!sv.ok) return HIPPT_ERR_INVALID;
    uint32_t max_depth = (opts && opts->max_depth) ? opts->max_depth : 5;
    uint32_t seed = opts ? opts->seed : 0;
    for (uint32_t i = 0; i < n; ++i) {
        V3 r = path_radiance(sv, xys[3 * i], xys[3 * i + 1], xys[3 * i + 2], seed, max_depth, iterative != 0, nullptr);
        out_rgb[3 * i] = r.x, out_rgb[3 * i + 1] = r.y, out_rgb[3 * i + 2] = r.z;
    }
    return HIPPT_OK;
}

}  // extern "C"
