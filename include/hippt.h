/* hippt.h — C ABI of the MI355X (gfx950) path-trace backend.
 *
 * This is the drop-in boundary for the reference's `--gpu` backend:
 *     pub fn render(scene: &Scene) -> RgbaImage        (src/gpu_renderer.rs:11-13,
 *                                                        sole caller src/main.rs:91)
 * Everything the reference passes across that call (the loaded `Scene`,
 * src/scene.rs:87-92) is marshalled into the POD tables below; everything it
 * gets back (an 8-bit RGBA image, row 0 = top, alpha 255,
 * src/gpu_renderer.rs:186-198) comes back through a caller-owned buffer.
 *
 * Conventions
 *   - plain pointers and sizes, no C++/torch types; all structs are POD with
 *     natural alignment (only 4-byte scalars, then pointers);
 *   - every function returns 0 on success, non-zero on error; the message is
 *     available from hippt_last_error() (thread-local). No exceptions cross
 *     the ABI. The reference panics via `expect` (gpu_renderer.rs:78,86,171);
 *     a host wrapper turns non-zero into the same abort-with-message;
 *   - the caller owns every pointer passed in or out; the library copies
 *     inputs and never frees caller memory;
 *   - the arithmetic specification is the reference CPU renderer
 *     (src/renderer.rs), not the WGSL shader.
 */
#ifndef HIPPT_H
#define HIPPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPPT_OK 0
#define HIPPT_ERR_INVALID 1   /* bad argument / inconsistent tables        */
#define HIPPT_ERR_DEVICE 2    /* HIP runtime error (message has the code)   */
#define HIPPT_ERR_IO 3        /* file / JSON / PNG error                    */
#define HIPPT_ERR_LIMIT 4     /* size beyond a documented limit             */

#define HIPPT_N_COUNTERS 16

/* ---- scene tables ------------------------------------------------------ */

/* src/material.rs:4-11 */
typedef struct hippt_material {
    float rgb[3];
    float metallic;
    float roughness;
    float ior;
    float volume_density;
    float volume_anisotropy;
} hippt_material;

/* src/sphere.rs:9-15. `order` = index of the object in the scene's object
 * list (src/scene.rs:124-151); it carries the first-minimum tie rule of
 * `intersect_closest` (src/renderer.rs:359-361) across the two tables. */
typedef struct hippt_sphere {
    float center[3];
    float radius;
    uint32_t material; /* index into materials[] */
    uint32_t order;
    uint32_t in_focus; /* src/object.rs:19-24 */
} hippt_sphere;

/* src/plane.rs:8-16. `normal` = normalize(u x v) (src/scene.rs:139). */
typedef struct hippt_plane {
    float point[3];
    float u[3];
    float v[3];
    float normal[3];
    uint32_t material;
    uint32_t order;
    uint32_t in_focus;
} hippt_plane;

/* src/light.rs:4-9 */
typedef struct hippt_light {
    float pos[3];
    float u[3];
    float v[3];
    float intensity[3];
} hippt_light;

/* Camera basis exactly as src/main.rs:44-51 derives it:
 *   forward = normalize(look_at - pos); right = normalize(up x forward);
 *   up = normalize(forward x right); aspect = w/h; scale = tan(fov/2).
 * `focus` is the autofocus distance (src/renderer.rs:204-234), `aperture`
 * the lens radius (src/algebra.rs:67-73). */
typedef struct hippt_camera {
    float pos[3];
    float right[3];
    float up[3];
    float forward[3];
    float aspect;
    float scale;
    float aperture;
    float focus;
} hippt_camera;

typedef struct hippt_scene {
    hippt_camera camera;
    uint32_t width;   /* scene.render.width   (src/scene.rs:18) */
    uint32_t height;  /* scene.render.height                      */
    uint32_t samples; /* scene.render.samples = divisor of the final average */
    uint32_t n_materials;
    uint32_t n_spheres;
    uint32_t n_planes;
    uint32_t n_lights;
    uint32_t reserved;
    const hippt_material *materials;
    const hippt_sphere *spheres;
    const hippt_plane *planes;
    const hippt_light *lights;
} hippt_scene;

/* flags */
#define HIPPT_FLAG_COUNTERS 1u   /* run the event-counting variant          */
#define HIPPT_FLAG_WAVEFRONT 2u  /* queue-based extend/shade/shadow pipeline (the default for LBVH scenes) */
#define HIPPT_FLAG_BVH 4u        /* LBVH over the sphere table               */
#define HIPPT_FLAG_NO_BVH 8u     /* force the brute-force object loop        */
#define HIPPT_FLAG_TABLES_GLOBAL 16u /* read scene tables from global memory
                                        (scalar loads) instead of staging in LDS */
#define HIPPT_FLAG_NO_SLICES 64u  /* one block renders a tile's whole sample range (no partial-sum slices) */
#define HIPPT_FLAG_MEGAKERNEL 32u /* force the one-kernel variant (the default for scenes without LBVH) */
#define HIPPT_FLAG_LOCAL_REDUCE 128u /* hippt_render_*_multi only: add the ranks' accumulators up on devices[0] with
                                        peer copies + a device add kernel instead of RCCL; a device may then be
                                        listed more than once (several ranks share it: a one-GPU box can rehearse the
                                        split, the per-rank contexts / threads / streams and the sum) */

#define HIPPT_FLAG_COUNT_EXECUTED 512u /* with HIPPT_FLAG_COUNTERS, table-scene megakernel: the timed kernel's own code runs
                                        * and shadow_*_tests / lit_evals report the tests and evaluations IT executes
                                        * (behind remembered occluders, without culled plane rows), not the reference's */
#define HIPPT_FLAG_TRAVERSAL_STATS 256u /* LBVH queue pipeline: count what its rays fetch (hippt_read_traversal_stats) */

typedef struct hippt_options {
    uint32_t sample_begin; /* first global sample index rendered by this call */
    uint32_t sample_count; /* 0 = scene.samples - sample_begin                */
    uint32_t max_depth;    /* 0 = 5 = MAX_DEPTH (src/renderer.rs:15)          */
    uint32_t seed;         /* global RNG seed; stream = f(seed, pixel, sample)*/
    uint32_t flags;
    int32_t device;        /* HIP device ordinal (one-shot entry points)      */
} hippt_options;

/* Event counters (index into the uint64[HIPPT_N_COUNTERS] array); the
 * algorithmic FLOP model of SURVEY.md §8(d) is a dot product with these. */
enum {
    HIPPT_C_PATHS = 0,
    HIPPT_C_EXTEND_RAYS = 1,
    HIPPT_C_EXTEND_HITS = 2,
    HIPPT_C_SHADOW_RAYS = 3,
    HIPPT_C_SHADOW_SPHERE_TESTS = 4, /* as the reference executes them: scene order, up to a ray's first occluder (renderer.rs:46) */
    HIPPT_C_SHADOW_PLANE_TESTS = 5,
    HIPPT_C_NEE_VERTICES = 6,        /* opaque + scatter vertices */
    HIPPT_C_LIT_EVALS = 7,           /* unoccluded, front-facing shadow samples */
    HIPPT_C_DIFFUSE_LOBES = 8,
    HIPPT_C_SPEC_LOBES = 9,
    HIPPT_C_GLASS_VERTICES = 10,
    HIPPT_C_MEDIUM_VERTICES = 11,
    HIPPT_C_SCATTER_EVENTS = 12,
    HIPPT_C_EXTEND_SPHERE_TESTS = 13,
    HIPPT_C_EXTEND_PLANE_TESTS = 14,
    HIPPT_C_RESERVED = 15
};

const char *hippt_last_error(void);

/* ---- one-shot entry points (what the Rust `hip_renderer::render` binds) -- */

/* Replaces gpu_renderer::render (src/gpu_renderer.rs:11-13). Renders
 * scene.samples samples per pixel on opts->device, averages, applies ACES +
 * gamma 1/2.2 + truncating u8 quantisation (src/renderer.rs:184-191) and
 * writes width*height*4 bytes RGBA (alpha 255, row 0 = top). opts may be NULL.
 * Zero lights is legal and renders black (CPU path, src/renderer.rs:31),
 * unlike the wgpu backend's `expect` at gpu_renderer.rs:116. */
int hippt_render_rgba8(const hippt_scene *scene, const hippt_options *opts,
                       uint8_t *out_rgba);

/* Same render, but returns the raw radiance SUMS over the rendered sample
 * range as width*height*3 floats (RGB interleaved, row 0 = top): the
 * analogue of `accumulated_color` (src/gpu_renderer.rs:104,175-179).
 * counters may be NULL; filled only with HIPPT_FLAG_COUNTERS. */
int hippt_render_accum_f32(const hippt_scene *scene, const hippt_options *opts,
                           float *out_accum, uint64_t *counters);

/* ---- one process, several GPUs ---------------------------------------------
 * The reference has one device and adds its dispatches up on the host
 * (src/gpu_renderer.rs:94-102,175-179). Here the image's sample range is split
 * over `devices` (hippt_sample_range), every device renders all pixels for its share of
 * the global sample indices, and the f32 sums are added up on devices[0] with one
 * RCCL reduce (librccl.so is opened at the first call; HIPPT_RCCL_LIB overrides the
 * name) before the tonemap. The image does not depend on the number of devices
 * beyond the order of the f32 additions. opts->device is ignored. */
int hippt_render_rgba8_multi(const hippt_scene *scene, const hippt_options *opts,
                             const int *devices, int n_devices, uint8_t *out_rgba);
/* Radiance sums of opts' sample range; sample_count 0 = from sample_begin to scene->samples, as the one-device
 * entry points read it. */
int hippt_render_accum_f32_multi(const hippt_scene *scene, const hippt_options *opts,
                                 const int *devices, int n_devices, float *out_accum);
/* Rank's share of `total` samples over `world` ranks: contiguous, counts differ by at
 * most one (the reference over-renders an uneven last dispatch instead,
 * gpu_renderer.rs:98-102). */
void hippt_sample_range(uint32_t total, uint32_t rank, uint32_t world,
                        uint32_t *begin, uint32_t *count);
/* Number of HIP devices this process sees (the reference asks wgpu for one adapter,
 * gpu_renderer.rs:77-86); 0 with HIPPT_ERR_DEVICE if there is none. */
int hippt_device_count(int *count);

/* ---- resident-context entry points (bench, multi-GPU, repeated renders) -- */

typedef struct hippt_context hippt_context;

/* Creates a context on HIP device `device` (stream, counter buffer). */
int hippt_create(int device, hippt_context **out_ctx);
int hippt_destroy(hippt_context *ctx);

/* Copies the scene tables to device memory (replaces the per-dispatch buffer
 * creation of src/gpu_renderer.rs:243-275; done once, not per dispatch) and builds the LBVH for scenes with
 * >= 64 spheres. If only the LBVH build fails (tree deeper than the traversal stack, out of memory) the call
 * returns that error but the scene IS set: renders with HIPPT_FLAG_NO_BVH (brute-force loops) work, renders
 * without it repeat the error. Any other failure leaves the context without a scene. */
int hippt_set_scene(hippt_context *ctx, const hippt_scene *scene);

/* Launches the trace kernels for samples [sample_begin, sample_begin+count)
 * of every pixel and ADDS the radiance sums into d_accum (device pointer,
 * width*height*3 floats, caller-zeroed). Asynchronous on `stream` (a
 * hipStream_t, NULL = the context's own stream) for the megakernel variants; the queue
 * pipelines (HIPPT_FLAG_WAVEFRONT, and LBVH scenes by default) poll their queue
 * counters and therefore synchronise the stream before returning. The LBVH queue pipeline keeps its
 * path queues in device memory that the context allocates on first use and holds until hippt_destroy:
 * up to 96 GiB for large images and sample counts (2^28 paths in flight), less when the GPU cannot
 * spare it (it then runs fewer concurrent sample streams: slower, same image; the stream count that fitted is
 * remembered by the context); it runs two pipeline groups (WFB_GROUPS = 2, the default since the end of round 3) and
 * therefore also launches on a second internal stream forked from / joined to `stream`. One call may cover any number of
 * samples: launches are cut at 4 096 samples per pixel and their sums added to d_accum one by one. This is the hot path
 * (src/renderer.rs:163-182 + 237-353 for every pixel). */
int hippt_render_accum_device(hippt_context *ctx, const hippt_options *opts,
                              float *d_accum, void *stream);

/* hippt_render_accum_device for callers that CANNOT ALLOCATE: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc. No counterpart in the reference (its backend creates seven buffers per dispatch,
 * src/gpu_renderer.rs:243-275). Same arguments, same kernels and the same sums as hippt_render_accum_device for
 * scenes that render with the table megakernel (no LBVH, no HIPPT_FLAG_WAVEFRONT); the megakernel's scratch buffer
 * (partial sums of its sample slices) is a 96 MiB array that lives in the library's device image instead of a buffer
 * allocated on demand, so the call performs no hipMalloc / hipFree / event creation of its own. Limits: images of at
 * most 4 194 304 pixels (two sample slices must fit; larger ones fail with HIPPT_ERR_LIMIT); at 1920x1080 the array
 * holds 4 slices where the allocating entry point uses 8, so the sums agree with it to f32 summation order there and
 * bit for bit on images of up to 1 048 576 pixels. Every call on a GPU shares that GPU's array: calls that overlap on
 * one GPU (two streams, two contexts, two threads) race. Its launches are not included in hippt_kernel_time_ms. */
int hippt_render_accum_device_noalloc(hippt_context *ctx, const hippt_options *opts,
                                      float *d_accum, void *stream);

/* Average / ACES / gamma / quantise on the device (src/renderer.rs:184-191):
 * d_accum (sums) / spp -> d_rgba (width*height*4 bytes). */
int hippt_resolve_device(hippt_context *ctx, const float *d_accum, uint32_t spp,
                         uint8_t *d_rgba, void *stream);

/* Waits for the context's stream and copies out the event counters
 * accumulated since the last call (then zeroes them). */
int hippt_read_counters(hippt_context *ctx, uint64_t *counters);

/* LBVH queue pipeline launched with HIPPT_FLAG_TRAVERSAL_STATS: what its traversal kernels fetched since the last
 * call (then zeroed): stats10 = {rays, 4-wide nodes visited (64 B each), spheres fetched (16 B each), aux rows read on
 * exact ties (16 B each)} of the closest-hit traversals, then the same four of the shadow-ray traversals, then how many of the
 * node visits of either kind were served from the LDS copy of the tree's first levels. No reference analogue
 * (the reference's closest hit is a loop over all objects, src/renderer.rs:356-362): measurement only. */
int hippt_read_traversal_stats(hippt_context *ctx, uint64_t *stats10);

/* Average duration in ms of the trace-kernel launches issued through this
 * context since the last call (HIP events on the launch stream), and their
 * number. Synchronises the stream. */
int hippt_kernel_time_ms(hippt_context *ctx, double *total_ms, uint32_t *launches);

/* ---- progressive / checkpointed accumulation ------------------------------
 * The analogue of the reference's dispatch loop with host-side accumulation
 * (src/gpu_renderer.rs:94-183): render a sample range, add it to an accumulator the
 * caller keeps on the HOST, and go on; any sample count and any chunk size (the
 * reference over-renders its last dispatch, gpu_renderer.rs:98-102 — not reproduced).
 * Used for long jobs such as the 131072-spp showcase render: the host accumulator plus
 * the number of samples done is a complete checkpoint. */

/* Renders samples [opts->sample_begin, +sample_count) of every pixel of the context's
 * scene into a context-owned device buffer and ADDS the per-pixel sums to host_accum
 * (width*height*3 floats, RGB interleaved, row 0 = top). Blocking. */
int hippt_render_accum_add_f32(hippt_context *ctx, const hippt_options *opts, float *host_accum);

/* Average / ACES / gamma / quantise (src/renderer.rs:184-191) of a HOST accumulator of
 * n_pixels*3 floats through the same device kernel as hippt_resolve_device. */
int hippt_resolve_f32(hippt_context *ctx, const float *host_accum, uint32_t n_pixels, uint32_t spp,
                      uint8_t *out_rgba);

/* Checkpoint file: 32-byte header {"HIPPTAC1", width, height, samples_done, seed, 0, 0}
 * (little-endian u32) followed by width*height*3 f32 radiance sums. save writes to
 * path + ".tmp" and renames, so a killed job never leaves a truncated checkpoint. load
 * fails with HIPPT_ERR_INVALID if the file's size does not match width x height. */
int hippt_checkpoint_save(const char *path, const float *accum, uint32_t width, uint32_t height,
                          uint32_t samples_done, uint32_t seed);
int hippt_checkpoint_load(const char *path, float *accum, uint32_t width, uint32_t height,
                          uint32_t *samples_done, uint32_t *seed);

/* ---- host-side scene plumbing (src/scene.rs, src/main.rs) ---------------- */

typedef struct hippt_scene_file hippt_scene_file; /* owns the tables */

/* Parses a scene.json (schema: src/scene.rs:6-84), resolves materials by name
 * with the magenta default (src/scene.rs:112-128), precomputes plane normals
 * (:139), derives the camera basis (src/main.rs:44-51) and the autofocus
 * distance (src/renderer.rs:204-234). */
int hippt_scene_load(const char *path, hippt_scene_file **out);
int hippt_scene_parse(const char *json_text, size_t len, hippt_scene_file **out);
const hippt_scene *hippt_scene_get(const hippt_scene_file *f);
/* Overrides render.width/height/samples and re-derives aspect and autofocus. */
int hippt_scene_set_render(hippt_scene_file *f, uint32_t width, uint32_t height,
                           uint32_t samples);
void hippt_scene_free(hippt_scene_file *f);
/* The camera / object / light dump that main() prints before it renders (src/main.rs:60-87), byte for byte:
 * position / look_at / up as Rust `{:?}` of Vec3 (shortest round-trip digits, "1.0" for integral values),
 * fov `{:.2}`, aspect / aperture / autofocus `{:.4}`, objects with their names, lights. Writes at most cap-1
 * bytes + NUL to out (may be NULL) and returns the length of the whole text, -1 if f is NULL. */
int hippt_scene_dump(const hippt_scene_file *f, char *out, size_t cap);

/* Camera basis from pos/look_at/up/fov(deg) as src/main.rs:41-51. */
void hippt_camera_basis(const float pos[3], const float look_at[3], const float up[3],
                        float fov_deg, uint32_t width, uint32_t height,
                        hippt_camera *out);
/* src/renderer.rs:204-234 (5x5 probe rays over in_focus objects). */
float hippt_autofocus(const hippt_scene *scene);

/* `renders/render_{w}x{h}_s{s}_ap{ap:.2}_f{f:.1}_{6 alnum}.png`
 * (src/renderer.rs:156-161). Returns the length written (excluding NUL). */
int hippt_render_image_name(uint32_t w, uint32_t h, uint32_t s, float ap, float f,
                            char *out, size_t cap);
/* 8-bit RGBA PNG (what `RgbaImage::save` produces, src/main.rs:104). */
int hippt_write_png_rgba8(const char *path, const uint8_t *rgba, uint32_t w, uint32_t h);

/* ---- unit-kernel test exports (1e-5 parity clause; no reference analogue) --
 * Each runs a HIP kernel over n independent inputs (host pointers in/out,
 * arrays of n items, vec3 = 3 consecutive floats). */

/* intersect_closest (src/renderer.rs:356-362) over the scene's objects.
 * out_index = object `order` of the hit, -1 on miss; t = +inf on miss. */
int hippt_test_intersect(const hippt_scene *scene, uint32_t n, const float *ro,
                         const float *rd, float *out_t, float *out_normal,
                         int32_t *out_index, int device);
/* LBVH closest hit (same outputs; requires a prior build over the scene). */
int hippt_test_intersect_bvh(const hippt_scene *scene, uint32_t n, const float *ro,
                             const float *rd, float *out_t, float *out_normal,
                             int32_t *out_index, int device);
/* shadow test of src/renderer.rs:46: any hit with t > 0 and t*t < dist2*0.999, for rays
 * ro + t*rd with UNIT-length rd (the routine the NEE kernel uses, origin offset folded out). */
int hippt_test_occluded(const hippt_scene *scene, uint32_t n, const float *ro,
                        const float *rd, const float *dist2, uint8_t *out, int device);
/* The same test as next-event estimation runs it (src/renderer.rs:44-46): shading point x, UNIT direction l towards
 * the light sample, dist2 = |light point - x|^2; the shadow ray starts at x + l*1e-4 and is occluded iff some object
 * has t > 0 (plane: t > 1e-4) and t*t < dist2*0.999. Runs the production 4-rays-per-lane routine with the origin
 * offset folded into its arithmetic. out: bit 0 = occluded, bit 1 = the four slots of the lane disagreed (a bug). */
int hippt_test_shadow_nee(const hippt_scene *scene, uint32_t n, const float *x, const float *l,
                          const float *dist2, uint8_t *out, int device);
/* one unoccluded light sample, src/renderer.rs:49-77. mat = n x 8 floats in
 * hippt_material order; light = one hippt_light. out = rgb contribution
 * (before the 1/SHADOW_SAMPLES average). */
int hippt_test_nee_eval(uint32_t n, const float *nrm, const float *v, const float *l,
                        const float *dist2, const float *mat, const hippt_light *light,
                        float *out_rgb, int device);
/* src/ggx.rs:7-18: out = {F.r, F.g, F.b, D, G} per item. */
int hippt_test_ggx_terms(uint32_t n, const float *nh, const float *nv, const float *nl,
                         const float *vh, const float *rough, const float *f0,
                         float *out5, int device);
/* src/ggx.rs:20-45 with explicit uniforms r1,r2. */
int hippt_test_ggx_sample(uint32_t n, const float *nrm, const float *rough,
                          const float *r1, const float *r2, float *out_h, int device);
/* src/renderer.rs:330-335 with explicit uniforms (u_phi, r2). */
int hippt_test_diffuse_sample(uint32_t n, const float *nrm, const float *u_phi,
                              const float *r2, float *out_dir, int device);
/* src/renderer.rs:309-317 + refract :364-369; rd = incoming direction (-v),
 * nrm = hit normal, u = the uniform of :315. out_dir = next direction. */
int hippt_test_glass(uint32_t n, const float *rd, const float *nrm, const float *ior,
                     const float *u, float *out_dir, int device);
/* src/renderer.rs:194-202 and the scatter frame :276-282.
 * out4 = {cos_theta, dir.x, dir.y, dir.z}. */
int hippt_test_hg(uint32_t n, const float *rd, const float *g, const float *u_hg,
                  const float *u_phi, float *out4, int device);
/* src/renderer.rs:170-178 + src/algebra.rs:67-73; u4 = {jx, jy, s, t}. */
int hippt_test_camera_ray(const hippt_scene *scene, uint32_t n, const uint32_t *xy,
                          const float *u4, float *out_ro, float *out_rd, int device);
/* src/tonemap.rs:10-19 + src/renderer.rs:187-191 on mean radiance. */
int hippt_test_tonemap(uint32_t n, const float *rgb, uint8_t *out_rgb8, int device);
/* first `n` uniforms of the stream (seed, pixel, sample): DESIGN.md §RNG. */
int hippt_test_rng(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t n,
                   float *out, int device);

#ifdef __cplusplus
}
#endif
#endif /* HIPPT_H */
