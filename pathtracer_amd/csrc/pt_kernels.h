// pt_kernels.h — host-callable launchers of the kernels in pt_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hippt.h"
#include "pt_bvh.h"
#include "pt_bvh_build.h"
#include "pt_device.h"

namespace pt {

hipError_t launch_megakernel(const Params &P, const float4 *blob, float *accum, unsigned long long *counters,
                             bool count, bool lds, hipStream_t st, float *partial = nullptr, uint32_t slices = 1,
                             bool count_executed = false);
hipError_t launch_megakernel_bvh(const Params &P, const float4 *blob, const Bvh &bvh, uint32_t stack_entries, float *accum,
                                 unsigned long long *counters, bool count, hipStream_t st);
hipError_t launch_unit_intersect_global(const Params &P, const float4 *blob, uint32_t n, const float *ro, const float *rd,
                                        float *t, float *nn, int32_t *idx, hipStream_t st);
hipError_t launch_unit_intersect_bvh(const Params &P, const float4 *blob, const Bvh &bvh, uint32_t n, const float *ro,
                                     const float *rd, float *t, float *nn, int32_t *idx, hipStream_t st);
// pt_wavefront.hip
size_t wf_workspace_bytes(uint32_t n_pixels);
// queue pipeline with persistent LBVH traversal kernels for many-sphere scenes (pt_wavefront_bvh.hip)
size_t wfb_workspace_bytes(uint32_t n_pixels, uint32_t n_light);
// stats (optional, device, 8 x u64): {rays, nodes, leaf fetches, aux rows} of the closest-hit and of the any-hit traversals
hipError_t wavefront_bvh_render(const Params &P, const float4 *blob, const Bvh &bvh, uint32_t stack_entries, float *accum,
                                void *workspace, hipStream_t st, uint32_t *iterations_out, unsigned long long *stats = nullptr,
                                float *partial = nullptr, hipStream_t st2 = nullptr, hipEvent_t ev_fork = nullptr,
                                hipEvent_t ev_join = nullptr);
// whether the queue pipeline takes this tree (index width, depth), and the stack entries its rays may need
bool wfb_supports(const BvhBuffers &b);
uint32_t wfb_stack_entries(const BvhBuffers &b);
uint64_t wfb_max_ray_ids();  // pixels x streams x 4 x lights must stay below this
// sample streams per pixel (log2) for a launch of sample_count samples over n_pixels pixels: up to 2^28 paths in flight, queues of at most 96 GiB
uint32_t wfb_streams_log2(uint32_t n_pixels, uint32_t sample_count, uint32_t n_light);
// accum[i] += slice 0 + slice 1 + ... (fixed order) of `partial` (slices x n floats)
hipError_t launch_add_slices(float *accum, const float *partial, uint32_t slices, size_t n, hipStream_t st);
hipError_t wavefront_render(const Params &P, const float4 *blob, float *accum, void *workspace, hipStream_t st,
                            uint32_t *iterations_out);
// dst[i] += src[i] (sum of per-rank accumulators on one device)
hipError_t launch_add(float *dst, const float *src, size_t n, hipStream_t st);
hipError_t launch_resolve(const float *accum, uint32_t n_pixels, uint32_t spp, uint8_t *rgba, hipStream_t st);

hipError_t launch_unit_intersect(const Params &P, const float4 *blob, uint32_t n, const float *ro, const float *rd,
                                 float *t, float *nn, int32_t *idx, hipStream_t st);
hipError_t launch_unit_occluded(const Params &P, const float4 *blob, uint32_t n, const float *ro, const float *rd,
                                const float *dist2, uint8_t *out, hipStream_t st);
hipError_t launch_unit_shadow_nee(const Params &P, const float4 *blob, uint32_t n, const float *x, const float *l,
                                  const float *dist2, uint8_t *out, hipStream_t st);
hipError_t launch_unit_nee_eval(uint32_t n, const float *nrm, const float *v, const float *l, const float *dist2,
                                const float *mat, const hippt_light &light, float *out, hipStream_t st);
hipError_t launch_unit_ggx_terms(uint32_t n, const float *nh, const float *nv, const float *nl, const float *vh,
                                 const float *rough, const float *f0, float *out5, hipStream_t st);
hipError_t launch_unit_ggx_sample(uint32_t n, const float *nrm, const float *rough, const float *r1, const float *r2,
                                  float *out, hipStream_t st);
hipError_t launch_unit_diffuse(uint32_t n, const float *nrm, const float *u_phi, const float *r2, float *out,
                               hipStream_t st);
hipError_t launch_unit_glass(uint32_t n, const float *rd, const float *nrm, const float *ior, const float *u,
                             float *out, hipStream_t st);
hipError_t launch_unit_hg(uint32_t n, const float *rd, const float *g, const float *u_hg, const float *u_phi,
                          float *out4, hipStream_t st);
// static LDS per block of the megakernels (4 x WaveLds)
size_t megakernel_static_lds();
hipError_t launch_unit_camera(const Params &P, const float4 *blob, uint32_t n, const uint32_t *xy, const float *u4, float *out_ro,
                              float *out_rd, hipStream_t st);
hipError_t launch_unit_tonemap(uint32_t n3, const float *rgb, uint8_t *out, hipStream_t st);
hipError_t launch_unit_rng(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float *out, hipStream_t st);

}  // namespace pt
