// pt_kernels.hip — gfx950 kernels of the path-trace hot path.
//
//   pt_megakernel  : renderer.rs:163-182 (sample loop) + :237-353 (trace) for
//                    every pixel; one lane = one pixel, paths regenerated in
//                    place so lanes never idle waiting for the wave's longest
//                    path; object / material / light tables staged in LDS.
//   pt_resolve     : renderer.rs:184-191 (average, ACES, gamma, u8).
//   pt_unit_*      : thin kernels around the same device functions, for the
//                    1e-5 unit-parity tests.
#include "pt_kernels.h"
#include "pt_trace.h"

namespace pt {

// Lane -> pixel: a 256-thread block covers a 16x16 pixel tile, each wave an
// 8x8 sub-tile (neighbouring pixels hit the same objects -> coherent branches).
__device__ __forceinline__ void lane_pixel(const Params &P, uint32_t &x, uint32_t &y) {
    uint32_t tile = blockIdx.x;
    uint32_t ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    x = tx * 16u + (wave & 1u) * 8u + (lane & 7u);
    y = ty * 16u + (wave >> 1) * 8u + (lane >> 3);
}

// 0: every kernel runs the plain lights loop (Tracer::direct_light), for A/B builds (tools/build_variant.sh)
#ifndef PT_SHARED_NEE
#define PT_SHARED_NEE 1
#endif
// Per-wave scratch in LDS: the 64 pixels' radiance sums and their RNG pixel keys ...
struct WaveAcc {
    float acc[64 * 3];
    uint32_t mix[64];
};
// ... and, for the kernels that run the wave-shared lights loop, its state (the LBVH megakernel does not: its LDS
// goes to the traversal stacks, up to BVH_STACK x 1 KiB per block)
struct WaveLds : WaveAcc {
    NeeShare nee;  // Tracer::direct_light_shared
};
static_assert(4 * sizeof(WaveAcc) + (size_t)BVH_STACK * 256 * 4 <= 64 * 1024, "LBVH megakernel: stacks + pixel sums exceed a block's 64 KiB of LDS");

// One wave renders the sample_count samples of its 8x8 pixels as a POOL of 64 * sample_count work items
// (item i = sample i / 64 of pixel i % 64): a lane whose path has ended takes the next item, whichever
// pixel it belongs to, so no lane idles while a neighbour with longer paths (a pixel on a sphere's rim, in
// fog, next to the sky) finishes its samples. Every path is a function of (pixel, global sample index)
// only, and a finished path's radiance is added to its pixel's sum in LDS, so the image is the same as
// with one lane per pixel up to the order of the f32 additions (which is fixed: the item order and the LDS
// atomics of a wave are deterministic).
template <class Tbl, bool COUNT, bool BVH = false, class WL = WaveLds, bool EXEC = false>
__device__ __forceinline__ void trace_pixel(const Params &P, Tbl tbl, uint32_t x_own, uint32_t y_own, bool in_image,
                                            float *__restrict__ accum, unsigned long long *__restrict__ counters,
                                            WL *wl, Bvh bvh = Bvh{}, uint32_t *stack = nullptr, bool overwrite = false) {
    constexpr bool kSharedNee = PT_SHARED_NEE && (!COUNT || EXEC) && !BVH;
    Tracer<Tbl, COUNT, BVH, EXEC> tr(P, tbl, bvh, stack);
#ifdef PT_NEE_DIAG
    tr.diag = counters;
#endif
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t pixel = y_own * P.width + x_own;
    const uint32_t seed_mixed = mix32(P.seed ^ 0x85ebca6bu);
    const uint32_t wx0 = x_own - (lane & 7u), wy0 = y_own - (lane >> 3);  // this wave's 8x8 tile
    wl->mix[lane] = mix32(pixel + seed_mixed);
    wl->acc[lane * 3 + 0] = 0.0f;
    wl->acc[lane * 3 + 1] = 0.0f;
    wl->acc[lane * 3 + 2] = 0.0f;
    if constexpr (kSharedNee) {
        if (lane < kHintLights) wl->nee.hint[lane] = kNoHint;
        if (lane == 0u) wl->nee.pending = 0u;
    }
    __builtin_amdgcn_wave_barrier();

    uint32_t cur_lane = 0, cur_sample = 0;  // wave-uniform cursor into the item sequence
    uint32_t my_pl = lane;                  // pixel (0..63 within the tile) of this lane's current path
    F3 psum = F3{0.0f, 0.0f, 0.0f};         // radiance of the current path
    // path state
    Rng rng{0u, 1u};
    F3 ro = F3{0, 0, 0}, rd = F3{0, 0, 1}, T = F3{1, 1, 1};
    uint32_t depth = 0;
    int media = -1;  // material id of the current medium, -1 = none
    bool alive = false;

    if (__any(in_image))
    for (;;) {
        const unsigned long long need = ballot64(!alive);
        if (need != 0ull) {
            uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            uint32_t pos = cur_lane + rank;
            uint32_t tot = cur_lane + (uint32_t)__popcll(need);
            uint32_t my_sample = cur_sample + (pos >> 6);
            cur_sample += tot >> 6;
            cur_lane = tot & 63u;
            if (!alive) {
                if (my_sample >= P.sample_count) break;  // pool exhausted: this lane is finished
                my_pl = pos & 63u;
                uint32_t px = wx0 + (my_pl & 7u), py = wy0 + (my_pl >> 3);
                if ((px < P.width) & (py < P.height)) {
                    // renderer.rs:170-178: new camera path
                    rng.seed(seed_mixed, wl->mix[my_pl], P.sample_begin + my_sample);
                    float jx = rng.gen(), jy = rng.gen(), ds = rng.gen(), dt = rng.gen();
                    camera_ray(tbl, P, px, py, jx, jy, ds, dt, ro, rd);
                    T = F3{1, 1, 1};
                    psum = F3{0.0f, 0.0f, 0.0f};
                    depth = 0;
                    media = -1;
                    alive = true;
                    tr.bump(HIPPT_C_PATHS);
                }
            }
        }
        if (!alive) continue;  // the item's pixel lies outside the image (ragged edge tiles only)

        // ---- one vertex of renderer.rs:237-353 ----
        RayK ray = make_ray(ro, rd);
        HitRec h = tr.closest(ray);
        tr.bump(HIPPT_C_EXTEND_RAYS);
        if (!BVH) tr.bump(HIPPT_C_EXTEND_SPHERE_TESTS, P.n_sph);
        tr.bump(HIPPT_C_EXTEND_PLANE_TESTS, P.n_pl);
        if (h.id >= 0) tr.bump(HIPPT_C_EXTEND_HITS);

        bool done = false;  // path ends after this vertex
        F3 A = F3{1, 1, 1};
        float t_media = __builtin_inff();
        Mat m;  // the medium's material, then the material shaded at this vertex
        if (media >= 0) {  // only media with density > 0 are ever entered (:301)
            m = load_mat(tbl, P.mat_off, (uint32_t)media);
            tr.bump(HIPPT_C_MEDIUM_VERTICES);
            t_media = -fln(rng.gen()) * rcp(m.density);  // :257
            float d = fminf(t_media, h.t);
            float k = -m.density * d;
            A = F3{fexp(fmaxf(1.0f - m.color.x, 0.0f) * k), fexp(fmaxf(1.0f - m.color.y, 0.0f) * k),
                   fexp(fmaxf(1.0f - m.color.z, 0.0f) * k)};  // :261-262
        }

        // Classify the vertex; scatter points and opaque surfaces then share ONE next-event-estimation
        // block, so a wave whose lanes are split between fog and surfaces runs it once, not twice.
        int kind = 0;  // 0: miss or glass (no NEE), 1: opaque surface, 2: scatter point inside the medium
        F3 x = F3{0, 0, 0}, n = F3{0.0f, 1.0f, 0.0f};  // :273: scatter points shade with a fixed "up" normal (Q11)
        int next_media = media;
        uint32_t nee_mat = (uint32_t)media;  // material id of m at a vertex that is lit
        if (t_media < h.t) {  // :269
            tr.bump(HIPPT_C_SCATTER_EVENTS);
            x = point_at_ref(ro, rd, t_media);
            T = T * A;
            kind = 2;
        } else if (h.id < 0) {  // :291 sky is black
            done = true;
        } else {
            uint32_t mid, order;
            tr.finish(ray, h, x, n, mid, order);
            m = load_mat(tbl, P.mat_off, mid);
            nee_mat = mid;
            if (m.density > 0.0f) next_media = (dot(-rd, n) > 0.0f) ? (int)mid : -1;  // :301-305
            T = T * A;
            if (m.ior > 1.0f && m.metallic < 0.1f) {  // :307 glass
                tr.bump(HIPPT_C_GLASS_VERTICES);
                F3 nd = glass_dir(rd, n, m.ior, rng.gen());
                ro = madd(x, nd, 1e-4f);
                rd = nd;
                media = next_media;
            } else {
                kind = 1;
            }
        }

        F3 D;
        if constexpr (kSharedNee) tr.direct_light_shared(kind != 0, x, n, -rd, nee_mat, rng, T, psum, my_pl, &wl->nee, wl->acc);
        if (kind != 0) {
            // the shared lights loop reads the material from the tables itself; so does the lobe sampling below, and
            // no material lives in registers across the loop
            if constexpr (kSharedNee) m = load_mat(tbl, P.mat_off, nee_mat);
            if constexpr (!kSharedNee) {
                D = tr.direct_light(x, make_shade(n, -rd, m), rng);
                psum = psum + T * D;  // :285 / :325
            }
            if (kind == 2) {
                float ct = sample_phase_function(m.aniso, rng.gen());
                float up = rng.gen();
                rd = scatter_dir(rd, ct, up);
                ro = x;  // no epsilon (:285)
            } else {  // :323 opaque
                float p = fmaxf(fmaxf(m.color.x, m.color.y), m.color.z);  // :327
                bool cont = depth < 2u;
                if (!cont) cont = rng.gen() < p;  // drawn only when depth >= 2 (:328)
                if (cont) {
                    const bool diffuse = rng.gen() < (1.0f - m.metallic);  // :329
                    tr.bump(diffuse ? HIPPT_C_DIFFUSE_LOBES : HIPPT_C_SPEC_LOBES);
                    const float ra = rng.gen(), rb = rng.gen();  // (u_phi, r2) of :331-332 or (r1, r2) of ggx.rs:21-22
                    const F3 nd = lobe_dir(diffuse, n, rd, m.roughness, ra, rb);
                    const F3 f = diffuse ? m.color * 0.31830988618379067154f : F3{1.0f, 1.0f, 1.0f};
                    float c = dot(nd, n);
                    if (c > 0.0f) {  // :343
                        float w = depth >= 2u ? c * rcp(p) : c;
                        T = T * f * w;
                        ro = madd(x, nd, 1e-4f);
                        rd = nd;
                        media = next_media;
                    } else {
                        done = true;
                    }
                } else {
                    done = true;
                }
            }
        }
        ++depth;
        if (done || depth >= P.max_depth) {  // :246; the path's radiance goes to its pixel's sum (:181)
            alive = false;
            __hip_atomic_fetch_add(&wl->acc[my_pl * 3 + 0], psum.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&wl->acc[my_pl * 3 + 1], psum.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&wl->acc[my_pl * 3 + 2], psum.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }

    if constexpr (kSharedNee) tr.flush_pairs(&wl->nee, wl->acc);
    __builtin_amdgcn_wave_barrier();
    if (in_image) {
        float *px = accum + (size_t)pixel * 3;
        if (overwrite) {  // a slice of the launch's sample range: partial sums, added up by pt_add_partials
            px[0] = wl->acc[lane * 3 + 0];
            px[1] = wl->acc[lane * 3 + 1];
            px[2] = wl->acc[lane * 3 + 2];
        } else {
            px[0] += wl->acc[lane * 3 + 0];
            px[1] += wl->acc[lane * 3 + 1];
            px[2] += wl->acc[lane * 3 + 2];
        }
    }
    if (COUNT) {
        // wave-reduce then one atomic per wave and counter
        for (int k = 0; k < HIPPT_N_COUNTERS; ++k) {
            uint32_t v = tr.cnt[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&counters[k], (unsigned long long)v);
        }
    }
}

// A launch may cut its sample range into gridDim.y slices (launch_megakernel): block (tile, j) renders slice j of the
// tile's samples and writes partial sums to slice j of `accum` (then a partial-sum buffer). With ~8 000 tiles at
// 1080p and ~1 300 resident blocks a whole-range block lasts a sixth of the launch and the last ones run on a
// half-empty chip; slices shorten that tail.
__device__ __forceinline__ float *sample_slice(Params &P, float *accum) {
    const uint32_t J = gridDim.y, j = blockIdx.y;
    if (J == 1u) return accum;
    const uint32_t base = P.sample_count / J, rem = P.sample_count % J;
    P.sample_begin += j * base + min(j, rem);
    P.sample_count = base + (j < rem ? 1u : 0u);
    return accum + (size_t)j * P.width * P.height * 3;
}

// Scene tables in LDS (north-star layout).
#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 5  // waves per SIMD the register allocator must allow (launch_bounds 2nd argument)
#endif
template <bool COUNT, bool EXEC = false>
__global__ __launch_bounds__(256, PT_MIN_WAVES) void pt_megakernel_lds(Params P, const float4 *__restrict__ blob,
                                                         float *__restrict__ accum,
                                                         unsigned long long *__restrict__ counters) {
    extern __shared__ float4 s_tbl[];
    for (uint32_t i = threadIdx.x; i < P.blob_f4; i += 256u) s_tbl[i] = blob[i];
    __syncthreads();
    uint32_t x, y;
    lane_pixel(P, x, y);
    bool in_image = (x < P.width) & (y < P.height);
    __shared__ WaveLds s_wl[4];
    float *out = sample_slice(P, accum);
    trace_pixel<TblLds, COUNT, false, WaveLds, EXEC>(P, TblLds{s_tbl}, x, y, in_image, out, counters, &s_wl[threadIdx.x >> 6], Bvh{}, nullptr, gridDim.y > 1u);
}

// Scene tables read from global memory (scenes too large for LDS).
template <bool COUNT, bool EXEC = false>
__global__ __launch_bounds__(256) void pt_megakernel_global(Params P, const float4 *__restrict__ blob,
                                                            float *__restrict__ accum,
                                                            unsigned long long *__restrict__ counters) {
    uint32_t x, y;
    lane_pixel(P, x, y);
    bool in_image = (x < P.width) & (y < P.height);
    __shared__ WaveLds s_wl[4];
    float *out = sample_slice(P, accum);
    trace_pixel<TblGlobal, COUNT, false, WaveLds, EXEC>(P, TblGlobal{blob}, x, y, in_image, out, counters, &s_wl[threadIdx.x >> 6], Bvh{}, nullptr, gridDim.y > 1u);
}

// Many-sphere scenes: spheres through the LBVH, everything else from global tables.
template <bool COUNT>
__global__ __launch_bounds__(256) void pt_megakernel_bvh(Params P, const float4 *__restrict__ blob, Bvh bvh,
                                                         float *__restrict__ accum,
                                                         unsigned long long *__restrict__ counters) {
    extern __shared__ uint32_t s_stack[];  // stack_entries x 256 (launch_megakernel_bvh sizes it from the tree depth)
    uint32_t x, y;
    lane_pixel(P, x, y);
    bool in_image = (x < P.width) & (y < P.height);
    __shared__ WaveAcc s_wl[4];
    trace_pixel<TblGlobal, COUNT, true, WaveAcc>(P, TblGlobal{blob}, x, y, in_image, accum, counters, &s_wl[threadIdx.x >> 6], bvh,
                                                 s_stack + threadIdx.x);
}

// renderer.rs:184-191
__global__ __launch_bounds__(256) void pt_resolve(const float *__restrict__ accum, uint32_t n_pixels, float inv_spp,
                                                  uint8_t *__restrict__ rgba) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_pixels) return;
    uchar4 o;
    o.x = quantise(aces1(accum[3 * i + 0] * inv_spp));
    o.y = quantise(aces1(accum[3 * i + 1] * inv_spp));
    o.z = quantise(aces1(accum[3 * i + 2] * inv_spp));
    o.w = 255;
    reinterpret_cast<uchar4 *>(rgba)[i] = o;
}

// ---- unit kernels ---------------------------------------------------------------
__device__ __forceinline__ F3 ld3(const float *p, uint32_t i) { return F3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ void st3(float *p, uint32_t i, F3 v) {
    p[3 * i] = v.x;
    p[3 * i + 1] = v.y;
    p[3 * i + 2] = v.z;
}

template <class Tbl>
__device__ __forceinline__ void unit_intersect_body(const Params &P, Tbl tbl, uint32_t n, const float *ro, const float *rd,
                                                    float *out_t, float *out_n, int32_t *out_idx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool act = i < n;
    uint32_t j = act ? i : 0;
    RayK r = make_ray(ld3(ro, j), ld3(rd, j));
    HitRec h = closest_hit(tbl, P, r);
    F3 hit, nn = F3{0, 0, 0};
    uint32_t mid, order = 0;
    if (h.id >= 0) finish_hit(tbl, P, r, h, hit, nn, mid, order);
    if (act) {
        out_t[i] = h.t;
        st3(out_n, i, nn);
        out_idx[i] = h.id >= 0 ? (int32_t)order : -1;
    }
}
__global__ void pt_unit_intersect(Params P, const float4 *__restrict__ blob, uint32_t n, const float *ro,
                                  const float *rd, float *out_t, float *out_n, int32_t *out_idx) {
    extern __shared__ float4 s_tbl[];
    for (uint32_t i = threadIdx.x; i < P.blob_f4; i += blockDim.x) s_tbl[i] = blob[i];
    __syncthreads();
    unit_intersect_body(P, TblLds{s_tbl}, n, ro, rd, out_t, out_n, out_idx);
}
__global__ void pt_unit_intersect_global(Params P, const float4 *__restrict__ blob, uint32_t n, const float *ro,
                                         const float *rd, float *out_t, float *out_n, int32_t *out_idx) {
    unit_intersect_body(P, TblGlobal{blob}, n, ro, rd, out_t, out_n, out_idx);
}
__global__ __launch_bounds__(256) void pt_unit_intersect_bvh(Params P, const float4 *__restrict__ blob, Bvh bvh, uint32_t n,
                                                             const float *ro, const float *rd, float *out_t, float *out_n,
                                                             int32_t *out_idx) {
    __shared__ uint32_t s_stack[BVH_STACK * 256];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool act = i < n;
    uint32_t j = act ? i : 0;
    Tracer<TblGlobal, false, true> tr(P, TblGlobal{blob}, bvh, s_stack + threadIdx.x);
    RayK r = make_ray(ld3(ro, j), ld3(rd, j));
    HitRec h = tr.closest(r);
    F3 hit, nn = F3{0, 0, 0};
    uint32_t mid, order = 0;
    if (h.id >= 0) tr.finish(r, h, hit, nn, mid, order);
    if (act) {
        out_t[i] = h.t;
        st3(out_n, i, nn);
        out_idx[i] = h.id >= 0 ? (int32_t)order : -1;
    }
}

__global__ void pt_unit_occluded(Params P, const float4 *__restrict__ blob, uint32_t n, const float *ro,
                                 const float *rd, const float *dist2, uint8_t *out) {
    extern __shared__ float4 s_tbl[];
    for (uint32_t i = threadIdx.x; i < P.blob_f4; i += blockDim.x) s_tbl[i] = blob[i];
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    TblLds tbl{s_tbl};
    // production shadow routine with the origin offset folded out (eps = 0); rd must be unit length
    F3 l[1] = {ld3(rd, i)};
    float lim[1] = {fsqrt(dist2[i] * 0.999f)};
    bool occ[1];
    uint32_t a = 0, b = 0;
    occluded_multi<TblLds, 1, false>(tbl, P, ld3(ro, i), 0.0f, l, lim, occ, a, b);
    out[i] = occ[0] ? 1 : 0;
}

// The shadow test AS NEXT-EVENT ESTIMATION USES IT (renderer.rs:44-46): 4 rays per lane from the shading point x,
// the origin offset eps = 1e-4 folded into the arithmetic (pt_device.h: sphere_shadow_row / plane_shadow_row), tmax
// formed from dist2 exactly as Tracer::direct_light forms it. The lane's ray sits in all 4 slots; bit 1 of the output
// is set if the slots disagree.
__global__ void pt_unit_shadow_nee(Params P, const float4 *__restrict__ blob, uint32_t n, const float *x,
                                   const float *l, const float *dist2, uint8_t *out) {
    extern __shared__ float4 s_tbl[];
    for (uint32_t i = threadIdx.x; i < P.blob_f4; i += blockDim.x) s_tbl[i] = blob[i];
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    TblLds tbl{s_tbl};
    const F3 li = ld3(l, i);
    const float d2 = dist2[i];
    const float tm = (d2 * rsq(d2)) * 0.99949987493746f;  // |lvec| sqrt(0.999)
    F3 ls[4] = {li, li, li, li};
    float lim[4] = {tm, tm, tm, tm};
    bool occ[4];
    uint32_t a = 0, b = 0;
    occluded_multi<TblLds, 4, false>(tbl, P, ld3(x, i), 1e-4f, ls, lim, occ, a, b);
    const bool same = (occ[0] == occ[1]) & (occ[1] == occ[2]) & (occ[2] == occ[3]);
    out[i] = (occ[0] ? 1 : 0) | (same ? 0 : 2);
}

__global__ void pt_unit_nee_eval(uint32_t n, const float *nrm, const float *v, const float *l, const float *dist2,
                                 const float *mat, hippt_light light, float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Mat m;
    m.color = F3{mat[8 * i], mat[8 * i + 1], mat[8 * i + 2]};
    m.metallic = mat[8 * i + 3];
    m.roughness = mat[8 * i + 4];
    m.ior = mat[8 * i + 5];
    m.density = mat[8 * i + 6];
    m.aniso = mat[8 * i + 7];
    ShadeK sk = make_shade(ld3(nrm, i), ld3(v, i), m);
    F3 lu = F3{light.u[0], light.u[1], light.u[2]}, lv = F3{light.v[0], light.v[1], light.v[2]};
    F3 cr = cross(lu, lv);
    float area = fsqrt(dot(cr, cr));
    F3 ln = normalize(cr);
    bool lit;
    F3 c = nee_eval(sk, ld3(l, i), rcp(dist2[i]), ln, area, F3{light.intensity[0], light.intensity[1], light.intensity[2]},
                    lit);
    st3(out, i, c);
}

__global__ void pt_unit_ggx_terms(uint32_t n, const float *nh, const float *nv, const float *nl, const float *vh,
                                  const float *rough, const float *f0, float *out5) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 f = fresnel_schlick(vh[i], ld3(f0, i));
    out5[5 * i] = f.x;
    out5[5 * i + 1] = f.y;
    out5[5 * i + 2] = f.z;
    out5[5 * i + 3] = d_term(nh[i], rough[i]);
    out5[5 * i + 4] = g_term(nv[i], nl[i], rough[i]);
}

__global__ void pt_unit_ggx_sample(uint32_t n, const float *nrm, const float *rough, const float *r1,
                                   const float *r2, float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st3(out, i, sample_ggx_h(ld3(nrm, i), rough[i], r1[i], r2[i]));
}

__global__ void pt_unit_diffuse(uint32_t n, const float *nrm, const float *u_phi, const float *r2, float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st3(out, i, diffuse_dir(ld3(nrm, i), u_phi[i], r2[i]));
}

__global__ void pt_unit_glass(uint32_t n, const float *rd, const float *nrm, const float *ior, const float *u,
                              float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st3(out, i, glass_dir(ld3(rd, i), ld3(nrm, i), ior[i], u[i]));
}

__global__ void pt_unit_hg(uint32_t n, const float *rd, const float *g, const float *u_hg, const float *u_phi,
                           float *out4) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float c = sample_phase_function(g[i], u_hg[i]);
    F3 d = scatter_dir(ld3(rd, i), c, u_phi[i]);
    out4[4 * i] = c;
    out4[4 * i + 1] = d.x;
    out4[4 * i + 2] = d.y;
    out4[4 * i + 3] = d.z;
}

__global__ void pt_unit_camera(Params P, const float4 *__restrict__ blob, uint32_t n, const uint32_t *xy, const float *u4,
                               float *out_ro, float *out_rd) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 ro, rd;
    camera_ray(TblGlobal{blob}, P, xy[2 * i], xy[2 * i + 1], u4[4 * i], u4[4 * i + 1], u4[4 * i + 2], u4[4 * i + 3], ro, rd);
    st3(out_ro, i, ro);
    st3(out_rd, i, rd);
}

__global__ void pt_unit_tonemap(uint32_t n3, const float *rgb, uint8_t *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    out[i] = quantise(aces1(rgb[i]));
}

__global__ void pt_unit_rng(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float *out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t sm = mix32(seed ^ 0x85ebca6bu);
    Rng r;
    r.seed(sm, mix32(pixel + sm), sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = r.gen();
}

// ---- launchers (host) -------------------------------------------------------------
static inline unsigned grid_for(uint32_t n, unsigned bs = 256) { return (n + bs - 1) / bs; }

__global__ __launch_bounds__(256) void pt_add_partials(float *__restrict__ accum, const float *__restrict__ partial,
                                                       uint32_t slices, size_t n) {
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float v = partial[i];
    for (uint32_t j = 1; j < slices; ++j) v += partial[(size_t)j * n + i];  // fixed order: deterministic
    accum[i] += v;
}

// slices > 1: the sample range is cut into `slices` parts rendered by separate blocks into `partial`
// (slices x width*height*3 floats), which are then added to accum in a fixed order.
hipError_t launch_megakernel(const Params &P, const float4 *blob, float *accum, unsigned long long *counters,
                             bool count, bool lds, hipStream_t st, float *partial, uint32_t slices, bool count_executed) {
    uint32_t tiles_y = (P.height + 15u) / 16u;
    if (slices < 1u || !partial) slices = 1u;
    dim3 grid(P.tiles_x * tiles_y, slices), block(256);
    float *out = slices > 1u ? partial : accum;
    size_t shmem = (size_t)P.blob_f4 * sizeof(float4);
    if (count && count_executed) {  // the TIMED kernels' own code (wave-shared lights loop), counting the tests it executes
        if (lds)
            hipLaunchKernelGGL((pt_megakernel_lds<true, true>), grid, block, shmem, st, P, blob, out, counters);
        else
            hipLaunchKernelGGL((pt_megakernel_global<true, true>), grid, block, 0, st, P, blob, out, counters);
    } else if (lds) {
        if (count)
            hipLaunchKernelGGL(pt_megakernel_lds<true>, grid, block, shmem, st, P, blob, out, counters);
        else
            hipLaunchKernelGGL(pt_megakernel_lds<false>, grid, block, shmem, st, P, blob, out, counters);
    } else {
        if (count)
            hipLaunchKernelGGL(pt_megakernel_global<true>, grid, block, 0, st, P, blob, out, counters);
        else
            hipLaunchKernelGGL(pt_megakernel_global<false>, grid, block, 0, st, P, blob, out, counters);
    }
    if (slices > 1u) {
        size_t n = (size_t)P.width * P.height * 3;
        hipLaunchKernelGGL(pt_add_partials, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, accum, (const float *)partial, slices, n);
    }
    return hipGetLastError();
}

// stack_entries: per-lane traversal stack in LDS, >= tree depth (the smaller, the more blocks per CU)
hipError_t launch_megakernel_bvh(const Params &P, const float4 *blob, const Bvh &bvh, uint32_t stack_entries, float *accum,
                                 unsigned long long *counters, bool count, hipStream_t st) {
    uint32_t tiles_y = (P.height + 15u) / 16u;
    dim3 grid(P.tiles_x * tiles_y), block(256);
    if (4 * sizeof(WaveAcc) + (size_t)stack_entries * 256 * 4 > 64 * 1024) return hipErrorInvalidValue;  // a block's LDS without opt-in
    if (count)
        hipLaunchKernelGGL(pt_megakernel_bvh<true>, grid, block, (size_t)stack_entries * 256 * 4, st, P, blob, bvh, accum, counters);
    else
        hipLaunchKernelGGL(pt_megakernel_bvh<false>, grid, block, (size_t)stack_entries * 256 * 4, st, P, blob, bvh, accum, counters);
    return hipGetLastError();
}

hipError_t launch_add_slices(float *accum, const float *partial, uint32_t slices, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(pt_add_partials, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, accum, partial, slices, n);
    return hipGetLastError();
}
hipError_t launch_add(float *dst, const float *src, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(pt_add_partials, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, src, 1u, n);
    return hipGetLastError();
}

hipError_t launch_resolve(const float *accum, uint32_t n_pixels, uint32_t spp, uint8_t *rgba, hipStream_t st) {
    hipLaunchKernelGGL(pt_resolve, dim3(grid_for(n_pixels)), dim3(256), 0, st, accum, n_pixels, 1.0f / (float)spp,
                       rgba);
    return hipGetLastError();
}

hipError_t launch_unit_intersect(const Params &P, const float4 *blob, uint32_t n, const float *ro, const float *rd,
                                 float *t, float *nn, int32_t *idx, hipStream_t st) {
    hipLaunchKernelGGL(pt_unit_intersect, dim3(grid_for(n)), dim3(256), (size_t)P.blob_f4 * 16, st, P, blob, n, ro, rd,
                       t, nn, idx);
    return hipGetLastError();
}
hipError_t launch_unit_intersect_global(const Params &P, const float4 *blob, uint32_t n, const float *ro, const float *rd,
                                        float *t, float *nn, int32_t *idx, hipStream_t st) {
    hipLaunchKernelGGL(pt_unit_intersect_global, dim3(grid_for(n)), dim3(256), 0, st, P, blob, n, ro, rd, t, nn, idx);
    return hipGetLastError();
}
hipError_t launch_unit_intersect_bvh(const Params &P, const float4 *blob, const Bvh &bvh, uint32_t n, const float *ro,
                                     const float *rd, float *t, float *nn, int32_t *idx, hipStream_t st) {
    hipLaunchKernelGGL(pt_unit_intersect_bvh, dim3(grid_for(n)), dim3(256), 0, st, P, blob, bvh, n, ro, rd, t, nn, idx);
    return hipGetLastError();
}
hipError_t launch_unit_occluded(const Params &P, const float4 *blob, uint32_t n, const float *ro, const float *rd,
                                const float *dist2, uint8_t *out, hipStream_t st) {
    hipLaunchKernelGGL(pt_unit_occluded, dim3(grid_for(n)), dim3(256), (size_t)P.blob_f4 * 16, st, P, blob, n, ro, rd,
                       dist2, out);
    return hipGetLastError();
}
hipError_t launch_unit_shadow_nee(const Params &P, const float4 *blob, uint32_t n, const float *x, const float *l,
                                  const float *dist2, uint8_t *out, hipStream_t st) {
    hipLaunchKernelGGL(pt_unit_shadow_nee, dim3(grid_for(n)), dim3(256), (size_t)P.blob_f4 * 16, st, P, blob, n, x, l, dist2, out);
    return hipGetLastError();
}
hipError_t launch_unit_nee_eval(uint32_t n, const float *nrm, const float *v, const float *l, const float *dist2,
                                const float *mat, const hippt_light &light, float *out, hipStream_t st) {
    hipLaunchKernelGGL(pt_unit_nee_eval, dim3(grid_for(n)), dim3(256), 0, st, n, nrm, v, l, dist2, mat, light, out);
    return hipGetLastError();
}
hipError_t launch_unit_ggx_terms(uint32_t n, const float *nh, const float *nv, const float *nl, const float *vh,
                                 const float *rough, const float *f0, float *out5, hipStream_t st) {
    hipLaunchKernelGGL(pt_unit_ggx_terms, dim3(grid_for(n)), dim3(256), 0, st, n, nh, nv, nl, vh, rough, f0, out5);
    return hipGetLastError();
}
hipError_t launch_unit_ggx_sample(uint32_t n, const float *nrm, const float *rough, const float *r1, const float *r2,
                                  float *out, hipStream_t st) {
    hipLaunchKernelGGL(pt_unit_ggx_sample, dim3(grid_for(n)), dim3(256), 0, st, n, nrm, rough, r1, r2, out);
    return hipGetLastError();
}
hipError_t launch_unit_diffuse(uint32_t n, const float *nrm, const float *u_phi, const float *r2, float *out,
                               hipStream_t st) {
    hipLaunchKernelGGL(pt_unit_diffuse, dim3(grid_for(n)), dim3(256), 0, st, n, nrm, u_phi, r2, out);
    return hipGetLastError();
}
hipError_t launch_unit_glass(uint32_t n, const float *rd, const float *nrm, const float *ior, const float *u,
                             float *out, hipStream_t st) {
    hipLaunchKernelGGL(pt_unit_glass, dim3(grid_for(n)), dim3(256), 0, st, n, rd, nrm, ior, u, out);
    return hipGetLastError();
}
hipError_t launch_unit_hg(uint32_t n, const float *rd, const float *g, const float *u_hg, const float *u_phi,
                          float *out4, hipStream_t st) {
    hipLaunchKernelGGL(pt_unit_hg, dim3(grid_for(n)), dim3(256), 0, st, n, rd, g, u_hg, u_phi, out4);
    return hipGetLastError();
}
size_t megakernel_static_lds() { return 4 * sizeof(WaveLds); }
hipError_t launch_unit_camera(const Params &P, const float4 *blob, uint32_t n, const uint32_t *xy, const float *u4, float *out_ro,
                              float *out_rd, hipStream_t st) {
    hipLaunchKernelGGL(pt_unit_camera, dim3(grid_for(n)), dim3(256), 0, st, P, blob, n, xy, u4, out_ro, out_rd);
    return hipGetLastError();
}
hipError_t launch_unit_tonemap(uint32_t n3, const float *rgb, uint8_t *out, hipStream_t st) {
    hipLaunchKernelGGL(pt_unit_tonemap, dim3(grid_for(n3)), dim3(256), 0, st, n3, rgb, out);
    return hipGetLastError();
}
hipError_t launch_unit_rng(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float *out, hipStream_t st) {
    hipLaunchKernelGGL(pt_unit_rng, dim3(1), dim3(64), 0, st, seed, pixel, sample, n, out);
    return hipGetLastError();
}

}  // namespace pt
