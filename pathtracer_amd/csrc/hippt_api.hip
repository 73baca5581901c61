// hippt_api.hip — C ABI (include/hippt.h) over the gfx950 kernels.
//
// Replaces the host half of the reference's wgpu backend
// (src/gpu_renderer.rs:76-200): device setup, table packing, dispatch sizing,
// accumulation and the final average/tonemap, for the HIP kernels in
// pt_kernels.hip. There is no CPU fallback: every entry point needs a HIP
// device and fails with HIPPT_ERR_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hippt.h"
#include "hippt_internal.h"
#include "pt_kernels.h"

namespace {

thread_local char g_err[512] = "";

}  // namespace

int hippt_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return hippt_set_error(HIPPT_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                   __FILE__, __LINE__);                                            \
    } while (0)

namespace {

// Paths per launch: bounds one launch to a fraction of a second so that long
// renders stay observable (analogue of the 600 M budget, gpu_renderer.rs:94).
constexpr uint64_t kPathsPerLaunch = 4ull << 30;
// Samples per pixel and launch: a launch keeps a pixel's running sum of a slice (an eighth of its samples) in f32, and
// a running f32 sum of n terms loses ~6e-8 sqrt(n) of its value. 4 096 samples per launch = 512 per slice keep that at
// ~1e-6 for any sample count (the showcase job has 131 072, README.md:265-268; SURVEY §7.3 "hierarchical accumulation");
// the launches' sums are then added to the accumulator one by one.
constexpr uint64_t kSamplesPerLaunch = 4096;
// LDS budget for the staged tables: a block may use 64 KiB without opting in to more, and the megakernel
// keeps 4 x WaveLds (pixel sums, pending shadow rays) of static LDS beside the tables.
const size_t kMaxLdsBlob = 64 * 1024 - pt::megakernel_static_lds();
// Sphere count from which hippt_set_scene builds the LBVH (brute force wins below).
constexpr uint32_t kBvhMinSpheres = 64;

// ---- hippt_render_accum_device_noalloc: device scratch that exists from load time -------------------------------
// The megakernel's only scratch buffer is the partial-sum buffer of its sample slices (launch_megakernel). The
// ordinary entry point allocates it on demand; the no-alloc entry point uses this array instead: 96 MiB of device
// memory per GPU the library runs on, held until the process ends.
constexpr size_t kScratchFloats = (96ull << 20) / sizeof(float);
constexpr uint64_t kNoAllocMaxPixels = kScratchFloats / (3 * 2);  // at least two slices must fit: 4 194 304 pixels
constexpr int kMaxDevices = 64;
static __device__ __align__(256) float scratch[kScratchFloats];
// `scratch` above is a host-side stand-in, not a device address: host code uses the pointer hipGetSymbolAddress
// returns, and every GPU has its own copy of the array at its own address.
float *g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mutex;

// Called by hippt_create with `device` current.
int scratch_init(int device) {
    if (device < 0 || device >= kMaxDevices) return HIPPT_OK;  // such a context has no no-alloc entry point
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return HIPPT_OK;
    float *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr)
        return hippt_set_error(HIPPT_ERR_DEVICE, "hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(e));
    g_scratch_ptr[device] = ptr;
    return HIPPT_OK;
}

// Sample slices of one megakernel launch of `c` samples per pixel: enough blocks for ~50 rounds of the chip's ~1 300
// resident blocks (measured: 3 slices +3.4 %, 8 slices +4.6 %, 16 no more), at most `cap_floats` of partial sums.
uint32_t megakernel_slices(const pt::Params &P, uint32_t c, uint32_t flags, size_t cap_floats) {
    const uint64_t tiles = (uint64_t)P.tiles_x * ((P.height + 15u) / 16u);
    const uint64_t pixels = (uint64_t)P.width * P.height;
    // at most 8 slices, each of at least 32 samples: a block relearns its occluder hints and restages the tables, so
    // short slices cost more than the tail they remove (1080p: 64 spp 2 slices +4.4 % over 8, 128 spp 4 slices +1.4 %,
    // 256 spp 4 or 8 alike, 1024 spp 8)
    uint64_t slices = std::min<uint64_t>(std::min<uint64_t>(8, std::max<uint32_t>(1, c / 32)), (65280 + tiles - 1) / tiles);
    slices = std::min<uint64_t>(slices, cap_floats / (pixels * 3));
    if ((flags & HIPPT_FLAG_NO_SLICES) || slices < 1) slices = 1;
    return (uint32_t)slices;
}

struct PackedScene {
    pt::Params P{};
    std::vector<float4> blob;
};

float4 mk4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
float as_f(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// Table packing: the counterpart of get_object_data (gpu_renderer.rs:203-223),
// without its 32+32 object cap and with volumes and every light kept.
int pack_scene(const hippt_scene *sc, PackedScene &out) {
    if (!sc) return hippt_set_error(HIPPT_ERR_INVALID, "scene is NULL");
    if (sc->width == 0 || sc->height == 0) return hippt_set_error(HIPPT_ERR_INVALID, "empty image %ux%u", sc->width, sc->height);
    if ((uint64_t)sc->width * sc->height > (1ull << 31))
        return hippt_set_error(HIPPT_ERR_LIMIT, "image %ux%u exceeds 2^31 pixels", sc->width, sc->height);
    uint32_t n_obj = sc->n_spheres + sc->n_planes;
    if (sc->n_materials > (1u << 23)) return hippt_set_error(HIPPT_ERR_LIMIT, "%u materials exceed 2^23", sc->n_materials);
    if ((sc->n_spheres && !sc->spheres) || (sc->n_planes && !sc->planes) || (sc->n_lights && !sc->lights) ||
        (sc->n_materials && !sc->materials))
        return hippt_set_error(HIPPT_ERR_INVALID, "NULL table with non-zero count");
    pt::Params &P = out.P;
    const hippt_camera &c = sc->camera;
    P.width = sc->width;
    P.height = sc->height;
    P.n_sph = sc->n_spheres;
    P.n_pl = sc->n_planes;
    P.n_mat = sc->n_materials;
    P.n_light = sc->n_lights;
    P.sph_off = 0;
    P.pl_off = P.sph_off + P.n_sph;
    P.aux_off = P.pl_off + P.n_pl * 3u;
    P.mat_off = P.aux_off + P.n_sph + P.n_pl;
    P.light_off = P.mat_off + sc->n_materials * pt::MAT_F4;
    P.cam_off = P.light_off + sc->n_lights * pt::LIGHT_F4;
    P.cull_off = P.cam_off + pt::CAM_F4;
    // The light-side table is n_lights x n_planes words: it pays on small scenes (it is staged in LDS with the other
    // tables). Beyond kMaxCullEntries all lights share ONE all-zero row ("straddles": never skip), so the table can
    // neither grow quadratically nor push a scene out of the LDS budget. Sizes are formed in 64 bits (the
    // 32-bit product wrapped for ~66k lights x ~66k planes and the fill loop then wrote past the blob).
    constexpr uint64_t kMaxCullEntries = 4096;
    const uint64_t cull_entries = (uint64_t)sc->n_lights * sc->n_planes;
    const bool cull_table = cull_entries <= kMaxCullEntries;
    P.cull_stride = cull_table ? sc->n_planes : 0u;
    const uint64_t blob_f4 = (uint64_t)P.cull_off + (cull_table ? cull_entries : sc->n_planes);
    const uint64_t off_check = (uint64_t)sc->n_spheres * 2 + (uint64_t)sc->n_planes * 4 + (uint64_t)sc->n_materials * pt::MAT_F4 +
                               (uint64_t)sc->n_lights * pt::LIGHT_F4 + pt::CAM_F4;  // the 32-bit offsets above, in 64 bits
    if (off_check + (cull_table ? cull_entries : sc->n_planes) != blob_f4 || blob_f4 > (1ull << 28))
        return hippt_set_error(HIPPT_ERR_LIMIT, "scene tables of %llu x 16 bytes exceed 4 GiB", (unsigned long long)(off_check + cull_entries));
    P.blob_f4 = (uint32_t)blob_f4;
    P.tiles_x = (sc->width + 15u) / 16u;
    out.blob.assign(P.blob_f4, mk4(0, 0, 0, 0));
    {
        float4 *q = &out.blob[P.cam_off];
        q[0] = mk4(c.pos[0], c.pos[1], c.pos[2], c.aspect);
        q[1] = mk4(c.right[0], c.right[1], c.right[2], c.scale);
        q[2] = mk4(c.up[0], c.up[1], c.up[2], c.aperture);
        q[3] = mk4(c.forward[0], c.forward[1], c.forward[2], c.focus);
        q[4] = mk4(1.0f / (float)sc->width, 1.0f / (float)sc->height, 0, 0);
    }

    // Each table is written in ascending scene order, so that strict '<' inside a table keeps
    // the first minimum (renderer.rs:359-361).
    std::vector<char> seen(n_obj, 0);
    std::vector<uint32_t> si(sc->n_spheres), pi(sc->n_planes);
    for (uint32_t i = 0; i < sc->n_spheres; ++i) si[i] = i;
    for (uint32_t i = 0; i < sc->n_planes; ++i) pi[i] = i;
    std::sort(si.begin(), si.end(), [&](uint32_t a, uint32_t b) { return sc->spheres[a].order < sc->spheres[b].order; });
    std::sort(pi.begin(), pi.end(), [&](uint32_t a, uint32_t b) { return sc->planes[a].order < sc->planes[b].order; });
    for (uint32_t k = 0; k < sc->n_spheres; ++k) {
        const hippt_sphere &s = sc->spheres[si[k]];
        if (s.order >= n_obj || seen[s.order]) return hippt_set_error(HIPPT_ERR_INVALID, "sphere %u: bad order %u", si[k], s.order);
        if (s.material >= sc->n_materials) return hippt_set_error(HIPPT_ERR_INVALID, "sphere %u: bad material %u", si[k], s.material);
        seen[s.order] = 1;
        out.blob[P.sph_off + k] = mk4(s.center[0], s.center[1], s.center[2], s.radius * s.radius);  // sphere.rs:29
        out.blob[P.aux_off + k] = mk4(1.0f / s.radius, as_f(s.material), as_f(s.order), s.radius);  // sphere.rs:37
    }
    for (uint32_t k = 0; k < sc->n_planes; ++k) {
        const hippt_plane &p = sc->planes[pi[k]];
        if (p.order >= n_obj || seen[p.order]) return hippt_set_error(HIPPT_ERR_INVALID, "plane %u: bad order %u", pi[k], p.order);
        if (p.material >= sc->n_materials) return hippt_set_error(HIPPT_ERR_INVALID, "plane %u: bad material %u", pi[k], p.material);
        seen[p.order] = 1;
        float uu = p.u[0] * p.u[0] + p.u[1] * p.u[1] + p.u[2] * p.u[2];  // plane.rs:45
        float vv = p.v[0] * p.v[0] + p.v[1] * p.v[1] + p.v[2] * p.v[2];  // plane.rs:50
        float iu = 1.0f / uu, iv = 1.0f / vv;  // |d.u| <= u.u  <=>  |d.(u/u.u)| <= 1
        float4 *q = &out.blob[P.pl_off + (size_t)k * 3];
        q[0] = mk4(p.point[0], p.point[1], p.point[2], p.normal[0]);
        q[1] = mk4(p.normal[1], p.normal[2], p.u[0] * iu, p.u[1] * iu);
        q[2] = mk4(p.u[2] * iu, p.v[0] * iv, p.v[1] * iv, p.v[2] * iv);
        out.blob[P.aux_off + P.n_sph + k] = mk4(0, as_f(p.material), as_f(p.order), 0);
    }
    for (uint32_t i = 0; i < sc->n_materials; ++i) {
        const hippt_material &m = sc->materials[i];
        float4 *q = &out.blob[P.mat_off + (size_t)i * pt::MAT_F4];
        q[0] = mk4(m.rgb[0], m.rgb[1], m.rgb[2], m.metallic);
        q[1] = mk4(m.roughness, m.ior, m.volume_density, m.volume_anisotropy);
    }
    for (uint32_t i = 0; i < sc->n_lights; ++i) {
        const hippt_light &l = sc->lights[i];
        // renderer.rs:51-52, hoisted: area = |u x v|, normal = normalize(u x v)
        float cx = l.u[1] * l.v[2] - l.u[2] * l.v[1];
        float cy = l.u[2] * l.v[0] - l.u[0] * l.v[2];
        float cz = l.u[0] * l.v[1] - l.u[1] * l.v[0];
        float area = std::sqrt(cx * cx + cy * cy + cz * cz);
        float inv = 1.0f / area;
        float4 *q = &out.blob[P.light_off + (size_t)i * pt::LIGHT_F4];
        q[0] = mk4(l.pos[0], l.pos[1], l.pos[2], area);
        q[1] = mk4(l.u[0], l.u[1], l.u[2], 0);
        q[2] = mk4(l.v[0], l.v[1], l.v[2], 0);
        q[3] = mk4(l.intensity[0], l.intensity[1], l.intensity[2], 0);
        q[4] = mk4(cx * inv, cy * inv, cz * inv, 0);
        // Which side of each plane the light's rectangle lies on (pt_device.h: occluded_multi, CULL). Light points are
        // pos + u e1 + v e2 with e in [-1/2, 1/2) (renderer.rs:37-39): the four corners bound (y - p).n. The margin
        // keeps the f32 arithmetic of the kernel (and of the light points it forms) well inside the decision.
        for (uint32_t k = 0; cull_table && k < sc->n_planes; ++k) {
            const hippt_plane &p = sc->planes[pi[k]];
            double lo = INFINITY, hi = -INFINITY, scale = 0.0;
            for (int c = 0; c < 4; ++c) {
                const double e1 = (c & 1) ? 0.5 : -0.5, e2 = (c & 2) ? 0.5 : -0.5;
                double g = 0.0, ext = 0.0;
                for (int a = 0; a < 3; ++a) {
                    const double y = (double)l.pos[a] + (double)l.u[a] * e1 + (double)l.v[a] * e2;
                    g += (y - (double)p.point[a]) * (double)p.normal[a];
                    ext = std::max(ext, std::max(std::fabs(y), std::fabs((double)p.point[a])));
                }
                lo = std::min(lo, g);
                hi = std::max(hi, g);
                scale = std::max(scale, ext);
            }
            const double margin = 1e-4 * (1.0 + scale);
            const float side = lo > margin ? 1.0f : (hi < -margin ? -1.0f : 0.0f);
            out.blob[P.cull_off + (size_t)i * sc->n_planes + k] = mk4(std::isfinite(lo) && std::isfinite(hi) ? side : 0.0f, 0, 0, 0);
        }
    }
    return HIPPT_OK;
}

}  // namespace

namespace {
int build_bvh(const float4 *d_blob, const pt::Params &P, pt::BvhBuffers &out, hipStream_t st, int *depth) {
    if (P.n_sph < 2) return hippt_set_error(HIPPT_ERR_INVALID, "LBVH needs at least 2 spheres");
    hipError_t e = pt::bvh_build(d_blob + P.sph_off, d_blob + P.aux_off, P.n_sph, out, st, depth);
    if (e != hipSuccess) {
        pt::bvh_free(out);
        return hippt_set_error(HIPPT_ERR_DEVICE, "LBVH build failed: %s", hipGetErrorString(e));
    }
    if (std::getenv("HIPPT_DEBUG")) std::fprintf(stderr, "[hippt] LBVH over %u spheres: depth %d\n", P.n_sph, *depth);
    if (*depth > pt::BVH_STACK) {
        pt::bvh_free(out);
        return hippt_set_error(HIPPT_ERR_LIMIT, "LBVH depth %d exceeds the traversal stack (%d)", *depth, pt::BVH_STACK);
    }
    return HIPPT_OK;
}
pt::Bvh bvh_view(const pt::BvhBuffers &b) {
    return pt::Bvh{b.nodes, b.top, b.sph_sorted, b.aux_sorted, b.n, {b.grid_lo[0], b.grid_lo[1], b.grid_lo[2]},
                   {b.grid_cell[0], b.grid_cell[1], b.grid_cell[2]}, b.wide, b.n_wide};
}
}  // namespace

struct hippt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    PackedScene scene;
    bool has_scene = false;
    float4 *d_blob = nullptr;
    size_t d_blob_cap = 0;
    unsigned long long *d_counters = nullptr;  // HIPPT_N_COUNTERS event counters + 10 traversal statistics
    uint32_t samples = 0;
    void *wf_ws = nullptr;  // wavefront path-state workspace (allocated on first use)
    size_t wf_ws_bytes = 0;
    uint32_t wf_streams_cap = 31;  // log2 of the sample streams per pixel the queue allocations of this context have been found to allow
    pt::BvhBuffers bvh;  // LBVH over the spheres (many-sphere scenes only)
    int bvh_depth = 0;
    std::string bvh_error;  // why the last hippt_set_scene could not build the LBVH (the scene then renders only with HIPPT_FLAG_NO_BVH)
    int bvh_error_code = HIPPT_OK;
    float *slice_buf = nullptr;  // partial sums of sample slices (launch_megakernel)
    size_t slice_bytes = 0;
    hipStream_t stream2 = nullptr;  // second pipeline of the LBVH queue variant (forked from / joined to the launch stream)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void *chunk_buf = nullptr;  // device staging of the progressive entry points
    size_t chunk_bytes = 0;
    // kernel timing (events on the launch stream)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
};

extern "C" {

const char *hippt_last_error(void) { return g_err; }

int hippt_create(int device, hippt_context **out_ctx) {
    if (!out_ctx) return hippt_set_error(HIPPT_ERR_INVALID, "out_ctx is NULL");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return hippt_set_error(HIPPT_ERR_DEVICE, "HIP device %d not present (%d devices)", device, n);
    HIP_TRY(hipSetDevice(device));
    hippt_context *ctx = new hippt_context();
    ctx->device = device;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_counters, (HIPPT_N_COUNTERS + 10) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(ctx->d_counters, 0, (HIPPT_N_COUNTERS + 10) * sizeof(unsigned long long));
    if (e != hipSuccess) {
        hippt_destroy(ctx);
        return hippt_set_error(HIPPT_ERR_DEVICE, "context setup failed: %s", hipGetErrorString(e));
    }
    if (int rc = scratch_init(device)) {  // the device is current: this GPU's address of the no-alloc scratch array
        hippt_destroy(ctx);
        return rc;
    }
    *out_ctx = ctx;
    return HIPPT_OK;
}

int hippt_destroy(hippt_context *ctx) {
    if (!ctx) return HIPPT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->ev_pool) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    pt::bvh_free(ctx->bvh);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->wf_ws) (void)hipFree(ctx->wf_ws);
    if (ctx->chunk_buf) (void)hipFree(ctx->chunk_buf);
    if (ctx->slice_buf) (void)hipFree(ctx->slice_buf);
    if (ctx->d_blob) (void)hipFree(ctx->d_blob);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return HIPPT_OK;
}

int hippt_set_scene(hippt_context *ctx, const hippt_scene *scene) {
    if (!ctx) return hippt_set_error(HIPPT_ERR_INVALID, "ctx is NULL");
    PackedScene ps;
    int rc = pack_scene(scene, ps);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    size_t bytes = ps.blob.size() * sizeof(float4);
    if (bytes > ctx->d_blob_cap) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->d_blob) HIP_TRY(hipFree(ctx->d_blob));
        ctx->d_blob = nullptr;
        ctx->d_blob_cap = 0;
        HIP_TRY(hipMalloc(&ctx->d_blob, bytes));
        ctx->d_blob_cap = bytes;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->has_scene = false;  // until the blob and its Params are both in place
    HIP_TRY(hipMemcpy(ctx->d_blob, ps.blob.data(), bytes, hipMemcpyHostToDevice));
    pt::bvh_free(ctx->bvh);
    ctx->bvh_depth = 0;
    ctx->bvh_error.clear();
    ctx->bvh_error_code = HIPPT_OK;
    ctx->scene = std::move(ps);
    ctx->samples = scene->samples;
    ctx->has_scene = true;
    if (scene->n_spheres >= kBvhMinSpheres) {
        // A failed build leaves a consistent scene WITHOUT an LBVH: the call reports the failure, later renders
        // repeat it unless they ask for the brute-force loops (HIPPT_FLAG_NO_BVH), which need no tree.
        int rc2 = build_bvh(ctx->d_blob, ctx->scene.P, ctx->bvh, ctx->stream, &ctx->bvh_depth);
        if (rc2) {
            ctx->bvh_error = hippt_last_error();
            ctx->bvh_error_code = rc2;
            return rc2;
        }
    }
    return HIPPT_OK;
}

int hippt_render_accum_device(hippt_context *ctx, const hippt_options *opts, float *d_accum, void *stream) {
    if (!ctx || !ctx->has_scene) return hippt_set_error(HIPPT_ERR_INVALID, "no scene set");
    if (!d_accum) return hippt_set_error(HIPPT_ERR_INVALID, "d_accum is NULL");
    hippt_options o{};
    if (opts) o = *opts;
    uint32_t s0 = o.sample_begin;
    uint32_t cnt = o.sample_count ? o.sample_count : (ctx->samples > s0 ? ctx->samples - s0 : 0);
    if ((uint64_t)s0 + cnt > 0xffffffffull) return hippt_set_error(HIPPT_ERR_LIMIT, "sample range overflows 32 bits");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    pt::Params P = ctx->scene.P;
    P.max_depth = o.max_depth ? o.max_depth : 5u;  // MAX_DEPTH, renderer.rs:15
    P.seed = o.seed;
    bool count = (o.flags & HIPPT_FLAG_COUNTERS) != 0;
    bool lds = (size_t)P.blob_f4 * sizeof(float4) <= kMaxLdsBlob;
    if (o.flags & HIPPT_FLAG_TABLES_GLOBAL) lds = false;
    bool use_bvh = ctx->bvh.n > 0 && !(o.flags & HIPPT_FLAG_NO_BVH);
    const uint64_t pixels = (uint64_t)P.width * P.height;
    if (ctx->bvh_error_code && !(o.flags & HIPPT_FLAG_NO_BVH))
        return hippt_set_error(ctx->bvh_error_code, "%s (render with HIPPT_FLAG_NO_BVH for the brute-force loops)", ctx->bvh_error.c_str());
    if ((o.flags & HIPPT_FLAG_BVH) && !use_bvh) {
        if (o.flags & HIPPT_FLAG_NO_BVH) return hippt_set_error(HIPPT_ERR_INVALID, "HIPPT_FLAG_BVH and HIPPT_FLAG_NO_BVH are exclusive");
        if (ctx->bvh.n == 0) {  // forced on a scene below the automatic threshold
            int rc2 = build_bvh(ctx->d_blob, ctx->scene.P, ctx->bvh, st, &ctx->bvh_depth);
            if (rc2) return rc2;
        }
        use_bvh = true;
    }
    bool wavefront = (o.flags & HIPPT_FLAG_WAVEFRONT) != 0;
    if ((o.flags & HIPPT_FLAG_WAVEFRONT) && (o.flags & HIPPT_FLAG_MEGAKERNEL))
        return hippt_set_error(HIPPT_ERR_INVALID, "HIPPT_FLAG_WAVEFRONT and HIPPT_FLAG_MEGAKERNEL are exclusive");
    // LBVH scenes default to the queue pipeline with persistent traversal kernels (2.4x the BVH megakernel on the
    // 100 000-sphere scene); the counting variant exists only as a megakernel
    // (also kept off the queues: trees too large or deep for its item keys and stack, and images whose 4 L shadow samples
    // per pixel would overflow the 32-bit ray ids even with one sample stream per pixel)
    const bool queue_ok = use_bvh && pt::wfb_supports(ctx->bvh) && pixels * 3u * 4u * std::max(1u, P.n_light) < pt::wfb_max_ray_ids();  // (a queue may hold up to 3 positions per entry)
    if (use_bvh && !(o.flags & HIPPT_FLAG_MEGAKERNEL) && !count && P.n_mat <= 0xffffu && P.max_depth <= 31u && P.n_light <= 96u && queue_ok) wavefront = true;
    if (wavefront) {
        if ((!use_bvh && !lds) || count || P.n_mat > 0xffffu || P.max_depth > 31u || (use_bvh && (P.n_light > 96u || !queue_ok)))
            return hippt_set_error(HIPPT_ERR_INVALID, "the wavefront variants support LDS-sized scenes or LBVH scenes, no counters, "
                                                      "<= 65535 materials, max_depth <= 31 and (LBVH scenes) <= 96 lights, < 2^24 spheres + nodes, "
                                                      "pixels x 4 x lights < 2^32");
        // LBVH pipeline: several sample streams per pixel (pt_wavefront_bvh.hip), sized for the largest chunk of this call
        const uint64_t per_chunk = std::max<uint64_t>(1, std::min(kPathsPerLaunch / pixels, kSamplesPerLaunch));
        if (use_bvh && !(o.flags & HIPPT_FLAG_NO_SLICES))
            P.wf_streams_log2 = pt::wfb_streams_log2((uint32_t)pixels, (uint32_t)std::min<uint64_t>(cnt, per_chunk), P.n_light);
        if (use_bvh) P.wf_streams_log2 = std::min(P.wf_streams_log2, ctx->wf_streams_cap);  // what an earlier call found to fit
        // The LBVH queues are sized for throughput (tens of GiB of an otherwise empty 288 GB GPU), and the streams' partial
        // sums grow with them. On a GPU that cannot spare that much, halve the number of sample streams until BOTH fit:
        // slower, same image. The stream count that fitted is remembered, so later calls do not repeat the failing
        // allocations.
        for (;;) {
            const size_t need = use_bvh ? pt::wfb_workspace_bytes((uint32_t)pixels << P.wf_streams_log2, P.n_light) : pt::wf_workspace_bytes((uint32_t)pixels);
            const size_t need_slices = use_bvh ? ((size_t)pixels * 3 * sizeof(float)) << P.wf_streams_log2 : 0;  // the streams' running sums
            hipError_t me = hipSuccess;
            if (need > ctx->wf_ws_bytes) {
                HIP_TRY(hipStreamSynchronize(st));
                if (ctx->wf_ws) HIP_TRY(hipFree(ctx->wf_ws));
                ctx->wf_ws = nullptr;
                ctx->wf_ws_bytes = 0;
                me = hipMalloc(&ctx->wf_ws, need);
                if (me == hipSuccess) {
                    ctx->wf_ws_bytes = need;
                    if (std::getenv("HIPPT_DEBUG"))
                        std::fprintf(stderr, "[hippt] queue workspace %.2f GiB, %u sample streams per pixel\n", (double)need / (1ull << 30), 1u << P.wf_streams_log2);
                }
            }
            if (me == hipSuccess && need_slices > ctx->slice_bytes) {
                HIP_TRY(hipStreamSynchronize(st));
                if (ctx->slice_buf) HIP_TRY(hipFree(ctx->slice_buf));
                ctx->slice_buf = nullptr;
                ctx->slice_bytes = 0;
                me = hipMalloc((void **)&ctx->slice_buf, need_slices);
                if (me == hipSuccess) ctx->slice_bytes = need_slices;
            }
            if (me == hipSuccess) break;
            if (me != hipErrorOutOfMemory || !use_bvh || P.wf_streams_log2 == 0) HIP_TRY(me);
            (void)hipGetLastError();
            --P.wf_streams_log2;
            ctx->wf_streams_cap = P.wf_streams_log2;
        }
        if (use_bvh && !ctx->stream2) {
            HIP_TRY(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        }
    }
    uint64_t per = std::min(kPathsPerLaunch / pixels, kSamplesPerLaunch);
    if (per < 1) per = 1;
    while (cnt > 0) {
        uint32_t c = (uint32_t)std::min<uint64_t>(cnt, per);
        P.sample_begin = s0;
        P.sample_count = c;
        // time each launch with events on the stream it runs on
        if (ctx->ev_used == ctx->ev_pool.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            ctx->ev_pool.emplace_back(a, b);
        }
        auto &ev = ctx->ev_pool[ctx->ev_used++];
        HIP_TRY(hipEventRecord(ev.first, st));
#ifdef HIPPT_STACK_OVERRIDE  // experiment only: more resident traversal waves (wrong images if a ray's stack overflows)
        const uint32_t stack_entries = HIPPT_STACK_OVERRIDE;
#else
        const uint32_t stack_entries = (uint32_t)std::min(pt::BVH_STACK, std::max(8, ctx->bvh_depth + 2));
#endif
        if (wavefront && use_bvh) {
            pt::Params Pc = P;  // a short last chunk may have fewer samples than streams
            Pc.wf_streams_log2 = std::min(P.wf_streams_log2, pt::wfb_streams_log2((uint32_t)pixels, c, P.n_light));
            HIP_TRY(pt::wavefront_bvh_render(Pc, ctx->d_blob, bvh_view(ctx->bvh), pt::wfb_stack_entries(ctx->bvh), d_accum, ctx->wf_ws, st, nullptr,
                                             (o.flags & HIPPT_FLAG_TRAVERSAL_STATS) ? ctx->d_counters + HIPPT_N_COUNTERS : nullptr,
                                             ctx->slice_buf, ctx->stream2, ctx->ev_fork, ctx->ev_join));
        }
        else if (wavefront)
            HIP_TRY(pt::wavefront_render(P, ctx->d_blob, d_accum, ctx->wf_ws, st, nullptr));
        else if (use_bvh)
            HIP_TRY(pt::launch_megakernel_bvh(P, ctx->d_blob, bvh_view(ctx->bvh), stack_entries, d_accum, ctx->d_counters, count, st));
        else {
            uint32_t slices = megakernel_slices(P, c, o.flags, SIZE_MAX);
            if (slices > 1) {
                size_t need = (size_t)slices * pixels * 3 * sizeof(float);
                if (need > ctx->slice_bytes) {
                    HIP_TRY(hipStreamSynchronize(st));
                    if (ctx->slice_buf) HIP_TRY(hipFree(ctx->slice_buf));
                    ctx->slice_buf = nullptr;
                    ctx->slice_bytes = 0;
                    HIP_TRY(hipMalloc((void **)&ctx->slice_buf, need));
                    ctx->slice_bytes = need;
                }
            }
            HIP_TRY(pt::launch_megakernel(P, ctx->d_blob, d_accum, ctx->d_counters, count, lds, st, ctx->slice_buf, slices,
                                          (o.flags & HIPPT_FLAG_COUNT_EXECUTED) != 0));
        }
        HIP_TRY(hipEventRecord(ev.second, st));
        s0 += c;
        cnt -= c;
    }
    return HIPPT_OK;
}

// The same launches as hippt_render_accum_device for table scenes (megakernel), with the partial-sum slices in this
// GPU's copy of `scratch` instead of a buffer allocated on demand: no hipMalloc / hipFree, no event creation (these
// launches are not timed by hippt_kernel_time_ms), nothing that allocates.
int hippt_render_accum_device_noalloc(hippt_context *ctx, const hippt_options *opts, float *d_accum, void *stream) {
    if (!ctx || !ctx->has_scene) return hippt_set_error(HIPPT_ERR_INVALID, "no scene set");
    if (!d_accum) return hippt_set_error(HIPPT_ERR_INVALID, "d_accum is NULL");
    hippt_options o{};
    if (opts) o = *opts;
    const uint32_t allowed = HIPPT_FLAG_COUNTERS | HIPPT_FLAG_NO_BVH | HIPPT_FLAG_TABLES_GLOBAL | HIPPT_FLAG_NO_SLICES | HIPPT_FLAG_MEGAKERNEL;
    if (o.flags & ~allowed)
        return hippt_set_error(HIPPT_ERR_INVALID, "hippt_render_accum_device_noalloc runs the table-scene megakernel only (flags 0x%x not supported)", o.flags & ~allowed);
    if (ctx->bvh_error_code && !(o.flags & HIPPT_FLAG_NO_BVH))  // as hippt_render_accum_device: a failed LBVH build is reported, not papered over
        return hippt_set_error(ctx->bvh_error_code, "%s (render with HIPPT_FLAG_NO_BVH for the brute-force loops)", ctx->bvh_error.c_str());
    if (ctx->bvh.n > 0 && !(o.flags & HIPPT_FLAG_NO_BVH))
        return hippt_set_error(HIPPT_ERR_INVALID, "hippt_render_accum_device_noalloc: LBVH scenes need the queue workspace of hippt_render_accum_device");
    if (ctx->device < 0 || ctx->device >= kMaxDevices || !g_scratch_ptr[ctx->device])
        return hippt_set_error(HIPPT_ERR_DEVICE, "no scratch address for device %d", ctx->device);
    float *const scratch_dev = g_scratch_ptr[ctx->device];  // the DEVICE address (hipGetSymbolAddress), never the array's name
    pt::Params P = ctx->scene.P;
    const uint64_t pixels = (uint64_t)P.width * P.height;
    if (pixels > kNoAllocMaxPixels)
        return hippt_set_error(HIPPT_ERR_LIMIT, "hippt_render_accum_device_noalloc: %ux%u = %llu pixels exceed the %llu the %zu MiB scratch array holds two sample slices of",
                               P.width, P.height, (unsigned long long)pixels, (unsigned long long)kNoAllocMaxPixels, (kScratchFloats * sizeof(float)) >> 20);
    uint32_t s0 = o.sample_begin;
    uint32_t cnt = o.sample_count ? o.sample_count : (ctx->samples > s0 ? ctx->samples - s0 : 0);
    if ((uint64_t)s0 + cnt > 0xffffffffull) return hippt_set_error(HIPPT_ERR_LIMIT, "sample range overflows 32 bits");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    P.max_depth = o.max_depth ? o.max_depth : 5u;  // MAX_DEPTH, renderer.rs:15
    P.seed = o.seed;
    const bool count = (o.flags & HIPPT_FLAG_COUNTERS) != 0;
    bool lds = (size_t)P.blob_f4 * sizeof(float4) <= kMaxLdsBlob;
    if (o.flags & HIPPT_FLAG_TABLES_GLOBAL) lds = false;
    uint64_t per = std::min(kPathsPerLaunch / pixels, kSamplesPerLaunch);
    if (per < 1) per = 1;
    while (cnt > 0) {
        const uint32_t c = (uint32_t)std::min<uint64_t>(cnt, per);
        P.sample_begin = s0;
        P.sample_count = c;
        // every slice block overwrites all of its tile's in-image pixels, and pt_add_partials reads exactly the
        // slices written here: nothing of an earlier call's contents is read, so nothing needs clearing
        const uint32_t slices = megakernel_slices(P, c, o.flags, kScratchFloats);
        HIP_TRY(pt::launch_megakernel(P, ctx->d_blob, d_accum, ctx->d_counters, count, lds, st, scratch_dev, slices));
        s0 += c;
        cnt -= c;
    }
    return HIPPT_OK;
}

int hippt_resolve_device(hippt_context *ctx, const float *d_accum, uint32_t spp, uint8_t *d_rgba, void *stream) {
    if (!ctx || !ctx->has_scene) return hippt_set_error(HIPPT_ERR_INVALID, "no scene set");
    if (!d_accum || !d_rgba || spp == 0) return hippt_set_error(HIPPT_ERR_INVALID, "bad resolve arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(pt::launch_resolve(d_accum, ctx->scene.P.width * ctx->scene.P.height, spp, d_rgba, st));
    return HIPPT_OK;
}

static int chunk_reserve(hippt_context *ctx, size_t bytes) {
    if (bytes <= ctx->chunk_bytes) return HIPPT_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->chunk_buf) HIP_TRY(hipFree(ctx->chunk_buf));
    ctx->chunk_buf = nullptr;
    ctx->chunk_bytes = 0;
    HIP_TRY(hipMalloc(&ctx->chunk_buf, bytes));
    ctx->chunk_bytes = bytes;
    return HIPPT_OK;
}

int hippt_render_accum_add_f32(hippt_context *ctx, const hippt_options *opts, float *host_accum) {
    if (!ctx || !ctx->has_scene) return hippt_set_error(HIPPT_ERR_INVALID, "no scene set");
    if (!host_accum) return hippt_set_error(HIPPT_ERR_INVALID, "host_accum is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    size_t n = (size_t)ctx->scene.P.width * ctx->scene.P.height * 3;
    int rc = chunk_reserve(ctx, n * sizeof(float));
    if (rc) return rc;
    float *d = (float *)ctx->chunk_buf;
    HIP_TRY(hipMemsetAsync(d, 0, n * sizeof(float), ctx->stream));
    rc = hippt_render_accum_device(ctx, opts, d, nullptr);
    if (rc) return rc;
    std::vector<float> part(n);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(part.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) host_accum[i] += part[i];  // gpu_renderer.rs:175-179
    return HIPPT_OK;
}

int hippt_resolve_f32(hippt_context *ctx, const float *host_accum, uint32_t n_pixels, uint32_t spp, uint8_t *out_rgba) {
    if (!ctx) return hippt_set_error(HIPPT_ERR_INVALID, "ctx is NULL");
    if (!host_accum || !out_rgba || !n_pixels || !spp) return hippt_set_error(HIPPT_ERR_INVALID, "bad resolve arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    size_t fb = (size_t)n_pixels * 12, ob = (size_t)n_pixels * 4;
    int rc = chunk_reserve(ctx, fb + ob);
    if (rc) return rc;
    float *d = (float *)ctx->chunk_buf;
    uint8_t *o = (uint8_t *)ctx->chunk_buf + fb;
    HIP_TRY(hipMemcpyAsync(d, host_accum, fb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(pt::launch_resolve(d, n_pixels, spp, o, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out_rgba, o, ob, hipMemcpyDeviceToHost));
    return HIPPT_OK;
}

int hippt_read_counters(hippt_context *ctx, uint64_t *counters) {
    if (!ctx || !counters) return hippt_set_error(HIPPT_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(counters, ctx->d_counters, HIPPT_N_COUNTERS * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(ctx->d_counters, 0, HIPPT_N_COUNTERS * sizeof(uint64_t)));
    return HIPPT_OK;
}

int hippt_read_traversal_stats(hippt_context *ctx, uint64_t *stats10) {
    if (!ctx || !stats10) return hippt_set_error(HIPPT_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(stats10, ctx->d_counters + HIPPT_N_COUNTERS, 10 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(ctx->d_counters + HIPPT_N_COUNTERS, 0, 10 * sizeof(uint64_t)));
    return HIPPT_OK;
}

int hippt_kernel_time_ms(hippt_context *ctx, double *total_ms, uint32_t *launches) {
    if (!ctx) return hippt_set_error(HIPPT_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    double tot = 0;
    for (size_t i = 0; i < ctx->ev_used; ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_pool[i].first, ctx->ev_pool[i].second));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (uint32_t)ctx->ev_used;
    ctx->ev_used = 0;
    return HIPPT_OK;
}

int hippt_render_accum_f32(const hippt_scene *scene, const hippt_options *opts, float *out_accum, uint64_t *counters) {
    if (!scene || !out_accum) return hippt_set_error(HIPPT_ERR_INVALID, "NULL argument");
    hippt_context *ctx = nullptr;
    int rc = hippt_create(opts ? opts->device : 0, &ctx);
    if (rc) return rc;
    float *d_accum = nullptr;
    size_t bytes = (size_t)scene->width * scene->height * 3 * sizeof(float);
    auto fail = [&](int code) {
        if (d_accum) (void)hipFree(d_accum);
        hippt_destroy(ctx);
        return code;
    };
    rc = hippt_set_scene(ctx, scene);
    if (rc) return fail(rc);
    if (hipMalloc(&d_accum, bytes) != hipSuccess) return fail(hippt_set_error(HIPPT_ERR_DEVICE, "hipMalloc(%zu) failed", bytes));
    if (hipMemsetAsync(d_accum, 0, bytes, ctx->stream) != hipSuccess) return fail(hippt_set_error(HIPPT_ERR_DEVICE, "memset failed"));
    rc = hippt_render_accum_device(ctx, opts, d_accum, nullptr);
    if (rc) return fail(rc);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(out_accum, d_accum, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(hippt_set_error(HIPPT_ERR_DEVICE, "render failed: %s", hipGetErrorString(e)));
    if (counters) {
        rc = hippt_read_counters(ctx, counters);
        if (rc) return fail(rc);
    }
    return fail(HIPPT_OK);
}

int hippt_render_rgba8(const hippt_scene *scene, const hippt_options *opts, uint8_t *out_rgba) {
    if (!scene || !out_rgba) return hippt_set_error(HIPPT_ERR_INVALID, "NULL argument");
    if (scene->samples == 0) return hippt_set_error(HIPPT_ERR_INVALID, "render.samples is 0");
    hippt_context *ctx = nullptr;
    int rc = hippt_create(opts ? opts->device : 0, &ctx);
    if (rc) return rc;
    float *d_accum = nullptr;
    uint8_t *d_rgba = nullptr;
    size_t npx = (size_t)scene->width * scene->height;
    auto fail = [&](int code) {
        if (d_accum) (void)hipFree(d_accum);
        if (d_rgba) (void)hipFree(d_rgba);
        hippt_destroy(ctx);
        return code;
    };
    rc = hippt_set_scene(ctx, scene);
    if (rc) return fail(rc);
    if (hipMalloc(&d_accum, npx * 12) != hipSuccess || hipMalloc(&d_rgba, npx * 4) != hipSuccess)
        return fail(hippt_set_error(HIPPT_ERR_DEVICE, "hipMalloc failed for %zu pixels", npx));
    if (hipMemsetAsync(d_accum, 0, npx * 12, ctx->stream) != hipSuccess) return fail(hippt_set_error(HIPPT_ERR_DEVICE, "memset failed"));
    hippt_options o{};
    if (opts) o = *opts;
    o.sample_begin = 0;
    o.sample_count = scene->samples;
    rc = hippt_render_accum_device(ctx, &o, d_accum, nullptr);
    if (rc) return fail(rc);
    rc = hippt_resolve_device(ctx, d_accum, scene->samples, d_rgba, nullptr);
    if (rc) return fail(rc);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(out_rgba, d_rgba, npx * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(hippt_set_error(HIPPT_ERR_DEVICE, "render failed: %s", hipGetErrorString(e)));
    return fail(HIPPT_OK);
}

}  // extern "C"

// ---- unit-kernel test exports ------------------------------------------------------
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 4); }
    template <class T>
    T *as() { return (T *)p; }
};
hipError_t up(DevBuf &b, const void *src, size_t bytes) {
    hipError_t e = b.alloc(bytes);
    if (e != hipSuccess) return e;
    return bytes ? hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
}
int use_device(int device) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return hippt_set_error(HIPPT_ERR_DEVICE, "HIP device %d not present (%d devices)", device, n);
    HIP_TRY(hipSetDevice(device));
    return HIPPT_OK;
}
}  // namespace

extern "C" {

#define UP(buf, src, bytes) HIP_TRY(up(buf, src, bytes))
#define DOWN(dst, buf, bytes)                                        \
    do {                                                             \
        HIP_TRY(hipDeviceSynchronize());                             \
        HIP_TRY(hipMemcpy(dst, (buf).p, bytes, hipMemcpyDeviceToHost)); \
    } while (0)

int hippt_test_intersect(const hippt_scene *scene, uint32_t n, const float *ro, const float *rd, float *out_t,
                         float *out_normal, int32_t *out_index, int device) {
    int rc = use_device(device);
    if (rc) return rc;
    PackedScene ps;
    rc = pack_scene(scene, ps);
    if (rc) return rc;
    bool lds = ps.blob.size() * 16 <= kMaxLdsBlob;
    DevBuf blob, dro, drd, dt, dn, di;
    UP(blob, ps.blob.data(), ps.blob.size() * 16);
    UP(dro, ro, (size_t)n * 12);
    UP(drd, rd, (size_t)n * 12);
    HIP_TRY(dt.alloc((size_t)n * 4));
    HIP_TRY(dn.alloc((size_t)n * 12));
    HIP_TRY(di.alloc((size_t)n * 4));
    if (n && lds) HIP_TRY(pt::launch_unit_intersect(ps.P, blob.as<float4>(), n, dro.as<float>(), drd.as<float>(), dt.as<float>(), dn.as<float>(), di.as<int32_t>(), nullptr));
    if (n && !lds) HIP_TRY(pt::launch_unit_intersect_global(ps.P, blob.as<float4>(), n, dro.as<float>(), drd.as<float>(), dt.as<float>(), dn.as<float>(), di.as<int32_t>(), nullptr));
    DOWN(out_t, dt, (size_t)n * 4);
    DOWN(out_normal, dn, (size_t)n * 12);
    DOWN(out_index, di, (size_t)n * 4);
    return HIPPT_OK;
}

int hippt_test_occluded(const hippt_scene *scene, uint32_t n, const float *ro, const float *rd, const float *dist2,
                        uint8_t *out, int device) {
    int rc = use_device(device);
    if (rc) return rc;
    PackedScene ps;
    rc = pack_scene(scene, ps);
    if (rc) return rc;
    if (ps.blob.size() * 16 > kMaxLdsBlob) return hippt_set_error(HIPPT_ERR_LIMIT, "scene tables exceed the LDS budget of the unit kernel");
    DevBuf blob, dro, drd, dd, dout;
    UP(blob, ps.blob.data(), ps.blob.size() * 16);
    UP(dro, ro, (size_t)n * 12);
    UP(drd, rd, (size_t)n * 12);
    UP(dd, dist2, (size_t)n * 4);
    HIP_TRY(dout.alloc(n));
    if (n) HIP_TRY(pt::launch_unit_occluded(ps.P, blob.as<float4>(), n, dro.as<float>(), drd.as<float>(), dd.as<float>(), dout.as<uint8_t>(), nullptr));
    DOWN(out, dout, n);
    return HIPPT_OK;
}

int hippt_test_shadow_nee(const hippt_scene *scene, uint32_t n, const float *x, const float *l, const float *dist2,
                          uint8_t *out, int device) {
    int rc = use_device(device);
    if (rc) return rc;
    PackedScene ps;
    rc = pack_scene(scene, ps);
    if (rc) return rc;
    if (ps.blob.size() * 16 > kMaxLdsBlob) return hippt_set_error(HIPPT_ERR_LIMIT, "scene tables exceed the LDS budget of the unit kernel");
    DevBuf blob, dx, dl, dd, dout;
    UP(blob, ps.blob.data(), ps.blob.size() * 16);
    UP(dx, x, (size_t)n * 12);
    UP(dl, l, (size_t)n * 12);
    UP(dd, dist2, (size_t)n * 4);
    HIP_TRY(dout.alloc(n));
    if (n) HIP_TRY(pt::launch_unit_shadow_nee(ps.P, blob.as<float4>(), n, dx.as<float>(), dl.as<float>(), dd.as<float>(), dout.as<uint8_t>(), nullptr));
    DOWN(out, dout, n);
    return HIPPT_OK;
}

int hippt_test_nee_eval(uint32_t n, const float *nrm, const float *v, const float *l, const float *dist2,
                        const float *mat, const hippt_light *light, float *out_rgb, int device) {
    int rc = use_device(device);
    if (rc) return rc;
    if (!light) return hippt_set_error(HIPPT_ERR_INVALID, "light is NULL");
    DevBuf dn, dv, dl, dd, dm, dout;
    UP(dn, nrm, (size_t)n * 12);
    UP(dv, v, (size_t)n * 12);
    UP(dl, l, (size_t)n * 12);
    UP(dd, dist2, (size_t)n * 4);
    UP(dm, mat, (size_t)n * 32);
    HIP_TRY(dout.alloc((size_t)n * 12));
    if (n) HIP_TRY(pt::launch_unit_nee_eval(n, dn.as<float>(), dv.as<float>(), dl.as<float>(), dd.as<float>(), dm.as<float>(), *light, dout.as<float>(), nullptr));
    DOWN(out_rgb, dout, (size_t)n * 12);
    return HIPPT_OK;
}

int hippt_test_ggx_terms(uint32_t n, const float *nh, const float *nv, const float *nl, const float *vh,
                         const float *rough, const float *f0, float *out5, int device) {
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf a, b, c, d, e, f, o;
    UP(a, nh, (size_t)n * 4);
    UP(b, nv, (size_t)n * 4);
    UP(c, nl, (size_t)n * 4);
    UP(d, vh, (size_t)n * 4);
    UP(e, rough, (size_t)n * 4);
    UP(f, f0, (size_t)n * 12);
    HIP_TRY(o.alloc((size_t)n * 20));
    if (n) HIP_TRY(pt::launch_unit_ggx_terms(n, a.as<float>(), b.as<float>(), c.as<float>(), d.as<float>(), e.as<float>(), f.as<float>(), o.as<float>(), nullptr));
    DOWN(out5, o, (size_t)n * 20);
    return HIPPT_OK;
}

int hippt_test_ggx_sample(uint32_t n, const float *nrm, const float *rough, const float *r1, const float *r2,
                          float *out_h, int device) {
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf a, b, c, d, o;
    UP(a, nrm, (size_t)n * 12);
    UP(b, rough, (size_t)n * 4);
    UP(c, r1, (size_t)n * 4);
    UP(d, r2, (size_t)n * 4);
    HIP_TRY(o.alloc((size_t)n * 12));
    if (n) HIP_TRY(pt::launch_unit_ggx_sample(n, a.as<float>(), b.as<float>(), c.as<float>(), d.as<float>(), o.as<float>(), nullptr));
    DOWN(out_h, o, (size_t)n * 12);
    return HIPPT_OK;
}

int hippt_test_diffuse_sample(uint32_t n, const float *nrm, const float *u_phi, const float *r2, float *out_dir,
                              int device) {
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf a, b, c, o;
    UP(a, nrm, (size_t)n * 12);
    UP(b, u_phi, (size_t)n * 4);
    UP(c, r2, (size_t)n * 4);
    HIP_TRY(o.alloc((size_t)n * 12));
    if (n) HIP_TRY(pt::launch_unit_diffuse(n, a.as<float>(), b.as<float>(), c.as<float>(), o.as<float>(), nullptr));
    DOWN(out_dir, o, (size_t)n * 12);
    return HIPPT_OK;
}

int hippt_test_glass(uint32_t n, const float *rd, const float *nrm, const float *ior, const float *u, float *out_dir,
                     int device) {
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf a, b, c, d, o;
    UP(a, rd, (size_t)n * 12);
    UP(b, nrm, (size_t)n * 12);
    UP(c, ior, (size_t)n * 4);
    UP(d, u, (size_t)n * 4);
    HIP_TRY(o.alloc((size_t)n * 12));
    if (n) HIP_TRY(pt::launch_unit_glass(n, a.as<float>(), b.as<float>(), c.as<float>(), d.as<float>(), o.as<float>(), nullptr));
    DOWN(out_dir, o, (size_t)n * 12);
    return HIPPT_OK;
}

int hippt_test_hg(uint32_t n, const float *rd, const float *g, const float *u_hg, const float *u_phi, float *out4,
                  int device) {
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf a, b, c, d, o;
    UP(a, rd, (size_t)n * 12);
    UP(b, g, (size_t)n * 4);
    UP(c, u_hg, (size_t)n * 4);
    UP(d, u_phi, (size_t)n * 4);
    HIP_TRY(o.alloc((size_t)n * 16));
    if (n) HIP_TRY(pt::launch_unit_hg(n, a.as<float>(), b.as<float>(), c.as<float>(), d.as<float>(), o.as<float>(), nullptr));
    DOWN(out4, o, (size_t)n * 16);
    return HIPPT_OK;
}

int hippt_test_camera_ray(const hippt_scene *scene, uint32_t n, const uint32_t *xy, const float *u4, float *out_ro,
                          float *out_rd, int device) {
    int rc = use_device(device);
    if (rc) return rc;
    PackedScene ps;
    rc = pack_scene(scene, ps);
    if (rc) return rc;
    DevBuf blob, a, b, o1, o2;
    UP(blob, ps.blob.data(), ps.blob.size() * 16);
    UP(a, xy, (size_t)n * 8);
    UP(b, u4, (size_t)n * 16);
    HIP_TRY(o1.alloc((size_t)n * 12));
    HIP_TRY(o2.alloc((size_t)n * 12));
    if (n) HIP_TRY(pt::launch_unit_camera(ps.P, blob.as<float4>(), n, a.as<uint32_t>(), b.as<float>(), o1.as<float>(), o2.as<float>(), nullptr));
    DOWN(out_ro, o1, (size_t)n * 12);
    DOWN(out_rd, o2, (size_t)n * 12);
    return HIPPT_OK;
}

int hippt_test_tonemap(uint32_t n, const float *rgb, uint8_t *out_rgb8, int device) {
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf a, o;
    UP(a, rgb, (size_t)n * 12);
    HIP_TRY(o.alloc((size_t)n * 3));
    if (n) HIP_TRY(pt::launch_unit_tonemap(n * 3, a.as<float>(), o.as<uint8_t>(), nullptr));
    DOWN(out_rgb8, o, (size_t)n * 3);
    return HIPPT_OK;
}

int hippt_test_rng(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float *out, int device) {
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf o;
    HIP_TRY(o.alloc((size_t)n * 4));
    if (n) HIP_TRY(pt::launch_unit_rng(seed, pixel, sample, n, o.as<float>(), nullptr));
    DOWN(out, o, (size_t)n * 4);
    return HIPPT_OK;
}

int hippt_test_intersect_bvh(const hippt_scene *scene, uint32_t n, const float *ro, const float *rd, float *out_t,
                             float *out_normal, int32_t *out_index, int device) {
    int rc = use_device(device);
    if (rc) return rc;
    PackedScene ps;
    rc = pack_scene(scene, ps);
    if (rc) return rc;
    DevBuf blob, dro, drd, dt, dn, di;
    UP(blob, ps.blob.data(), ps.blob.size() * 16);
    pt::BvhBuffers bb;
    int depth = 0;
    rc = build_bvh(blob.as<float4>(), ps.P, bb, nullptr, &depth);
    if (rc) return rc;
    struct Guard {
        pt::BvhBuffers &b;
        ~Guard() { pt::bvh_free(b); }
    } guard{bb};
    UP(dro, ro, (size_t)n * 12);
    UP(drd, rd, (size_t)n * 12);
    HIP_TRY(dt.alloc((size_t)n * 4));
    HIP_TRY(dn.alloc((size_t)n * 12));
    HIP_TRY(di.alloc((size_t)n * 4));
    if (n) HIP_TRY(pt::launch_unit_intersect_bvh(ps.P, blob.as<float4>(), bvh_view(bb), n, dro.as<float>(), drd.as<float>(), dt.as<float>(), dn.as<float>(), di.as<int32_t>(), nullptr));
    DOWN(out_t, dt, (size_t)n * 4);
    DOWN(out_normal, dn, (size_t)n * 12);
    DOWN(out_index, di, (size_t)n * 4);
    return HIPPT_OK;
}

}  // extern "C"
