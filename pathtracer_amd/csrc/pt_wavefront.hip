// pt_wavefront.hip — wavefront (queue-based) variant of the trace loop, built to decide
// "megakernel vs wavefront" from measurements (north star).
//
// One path slot per pixel, path state in SoA buffers in HBM (coalesced: lane i of a wave reads
// element queue[i] of each array). Per iteration, two kernels over compacted index queues:
//
//   wf_extend : regenerate finished paths (renderer.rs:170-178), closest hit (:356-362),
//               medium free flight (:253-264), classify; glass vertices (:307-320) are finished
//               here; opaque / scatter vertices are appended to the shade queue, everything
//               that still has a ray to trace to the next extend queue.
//   wf_shade  : next-event estimation (:19-85) + Russian roulette + lobe sampling (:323-349)
//               or phase-function scatter (:269-286) for the shade queue only — every lane of
//               every wave has NEE work, which is what compaction buys.
//
// Queue appends use a wave ballot + popcount prefix and one atomicAdd per wave.
// The arithmetic is the same device code as the megakernel (pt_trace.h), with the same RNG
// stream per (pixel, sample), so both variants render the same image.
#include "pt_kernels.h"
#include "pt_trace.h"

namespace pt {

struct WfState {  // SoA over n slots (slot = pixel index)
    float *ro, *rd, *T, *sum;  // 3 planes of n floats each: v[c * n + slot]
    float *x, *nrm;            // hit / scatter point and normal of the pending shade vertex
    uint32_t *rng0, *rng1;
    uint32_t *meta;            // bit 0 alive, bits 1-5 depth, bit 6 pending vertex is a scatter, bits 8.. media+1
    uint32_t *next_sample;     // next global sample index this slot will start
    uint32_t *mat;             // material id of the pending shade vertex | next_media+1 << 16
    uint32_t n;
};

__device__ __forceinline__ F3 ld3s(const float *p, uint32_t n, uint32_t i) { return F3{p[i], p[n + i], p[2 * n + i]}; }
__device__ __forceinline__ void st3s(float *p, uint32_t n, uint32_t i, F3 v) {
    p[i] = v.x;
    p[n + i] = v.y;
    p[2 * n + i] = v.z;
}

// Appends `slot` to queue q for the lanes with `want` set: one atomic per wave.
__device__ __forceinline__ void queue_push(bool want, uint32_t slot, uint32_t *q, uint32_t *q_count) {
    unsigned long long m = __ballot(want);
    if (m == 0) return;
    uint32_t lane = threadIdx.x & 63u;
    uint32_t rank = __popcll(m & ((1ull << lane) - 1ull));
    uint32_t base = 0;
    if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(q_count, (uint32_t)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (want) q[base + rank] = slot;
}

__global__ __launch_bounds__(256) void wf_init(WfState S, uint32_t sample_begin, uint32_t *q, uint32_t *q_count) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= S.n) return;
    S.meta[i] = 0;  // dead: wf_extend starts the first sample
    S.next_sample[i] = sample_begin;
    st3s(S.sum, S.n, i, F3{0, 0, 0});
    q[i] = i;
    if (i == 0) *q_count = S.n;
}

__global__ __launch_bounds__(256) void wf_extend(Params P, const float4 *__restrict__ blob, WfState S,
                                                 const uint32_t *__restrict__ q_in, const uint32_t *__restrict__ q_in_count,
                                                 uint32_t *q_next, uint32_t *q_next_count, uint32_t *q_shade,
                                                 uint32_t *q_shade_count) {
    extern __shared__ float4 s_tbl[];
    for (uint32_t i = threadIdx.x; i < P.blob_f4; i += 256u) s_tbl[i] = blob[i];
    __syncthreads();
    TblLds tbl{s_tbl};
    Tracer<TblLds, false, false> tr(P, tbl, Bvh{}, nullptr);
    const uint32_t n = S.n, count = *q_in_count;
    const uint32_t s_end = P.sample_begin + P.sample_count;
    const uint32_t seed_mixed = mix32(P.seed ^ 0x85ebca6bu);
    for (uint32_t base = blockIdx.x * 256u; base < count; base += gridDim.x * 256u) {
        uint32_t qi = base + threadIdx.x;
        bool act = qi < count;
        uint32_t slot = act ? q_in[qi] : 0u;
        bool to_next = false, to_shade = false;
        if (act) {
            uint32_t meta = S.meta[slot];
            bool alive = meta & 1u;
            uint32_t depth = (meta >> 1) & 31u;
            int media = (int)(meta >> 8) - 1;
            Rng rng;
            F3 ro, rd, T;
            if (!alive) {
                uint32_t s = S.next_sample[slot];  // queued only while s < s_end
                uint32_t px = slot % P.width, py = slot / P.width;
                rng.seed(seed_mixed, mix32(slot + seed_mixed), s);
                float jx = rng.gen(), jy = rng.gen(), ds = rng.gen(), dt = rng.gen();
                camera_ray(P, px, py, jx, jy, ds, dt, ro, rd);
                T = F3{1, 1, 1};
                depth = 0;
                media = -1;
                alive = true;
                S.next_sample[slot] = s + 1;
            } else {
                ro = ld3s(S.ro, n, slot);
                rd = ld3s(S.rd, n, slot);
                T = ld3s(S.T, n, slot);
                rng.s0 = S.rng0[slot];
                rng.s1 = S.rng1[slot];
            }
            RayK ray = make_ray(ro, rd);
            HitRec h = tr.closest(ray);
            F3 A = F3{1, 1, 1};
            float t_media = __builtin_inff();
            Mat mm;
            if (media >= 0) {
                mm = load_mat(tbl, P.mat_off, (uint32_t)media);
                t_media = -fln(rng.gen()) * rcp(mm.density);
                float k = -mm.density * fminf(t_media, h.t);
                A = F3{fexp(fmaxf(1.0f - mm.color.x, 0.0f) * k), fexp(fmaxf(1.0f - mm.color.y, 0.0f) * k),
                       fexp(fmaxf(1.0f - mm.color.z, 0.0f) * k)};
            }
            bool scatter = false;
            if (t_media < h.t) {
                T = T * A;
                st3s(S.x, n, slot, madd(ro, rd, t_media));
                st3s(S.nrm, n, slot, F3{0.0f, 1.0f, 0.0f});
                S.mat[slot] = (uint32_t)media | ((uint32_t)(media + 1) << 16);
                scatter = true;
                to_shade = true;
            } else if (h.id < 0) {
                alive = false;
            } else {
                F3 hit, nn;
                uint32_t mid, order;
                tr.finish(ray, h, hit, nn, mid, order);
                Mat m = load_mat(tbl, P.mat_off, mid);
                int next_media = media;
                if (m.density > 0.0f) next_media = (dot(-rd, nn) > 0.0f) ? (int)mid : -1;
                T = T * A;
                if (m.ior > 1.0f && m.metallic < 0.1f) {
                    F3 nd = glass_dir(rd, nn, m.ior, rng.gen());
                    ro = madd(hit, nd, 1e-4f);
                    rd = nd;
                    media = next_media;
                    ++depth;
                    if (depth >= P.max_depth) alive = false;
                } else {
                    st3s(S.x, n, slot, hit);
                    st3s(S.nrm, n, slot, nn);
                    S.mat[slot] = mid | ((uint32_t)(next_media + 1) << 16);
                    to_shade = true;
                }
            }
            if (!to_shade) {
                // finished paths come back through this kernel to start their next sample
                to_next = alive || S.next_sample[slot] < s_end;
                if (alive) {
                    st3s(S.ro, n, slot, ro);
                    st3s(S.rd, n, slot, rd);
                }
            } else {
                st3s(S.rd, n, slot, rd);  // wf_shade needs v = -rd (and rd for the scatter frame)
            }
            if (alive) {
                st3s(S.T, n, slot, T);
                S.rng0[slot] = rng.s0;
                S.rng1[slot] = rng.s1;
            }
            S.meta[slot] = (alive ? 1u : 0u) | (depth << 1) | (scatter ? 64u : 0u) | ((uint32_t)(media + 1) << 8);
        }
        queue_push(to_next, slot, q_next, q_next_count);
        queue_push(to_shade, slot, q_shade, q_shade_count);
    }
}

__global__ __launch_bounds__(256) void wf_shade(Params P, const float4 *__restrict__ blob, WfState S,
                                                const uint32_t *__restrict__ q_shade, const uint32_t *__restrict__ q_shade_count,
                                                uint32_t *q_next, uint32_t *q_next_count) {
    extern __shared__ float4 s_tbl[];
    for (uint32_t i = threadIdx.x; i < P.blob_f4; i += 256u) s_tbl[i] = blob[i];
    __syncthreads();
    TblLds tbl{s_tbl};
    Tracer<TblLds, false, false> tr(P, tbl, Bvh{}, nullptr);
    const uint32_t n = S.n, count = *q_shade_count;
    const uint32_t s_end = P.sample_begin + P.sample_count;
    for (uint32_t base = blockIdx.x * 256u; base < count; base += gridDim.x * 256u) {
        uint32_t qi = base + threadIdx.x;
        bool act = qi < count;
        uint32_t slot = act ? q_shade[qi] : 0u;
        bool to_next = false;
        if (act) {
            uint32_t meta = S.meta[slot];
            uint32_t depth = (meta >> 1) & 31u;
            bool scatter = meta & 64u;
            int media = (int)(meta >> 8) - 1;
            uint32_t mw = S.mat[slot];
            uint32_t mid = mw & 0xffffu;
            int next_media = (int)(mw >> 16) - 1;
            F3 x = ld3s(S.x, n, slot), nn = ld3s(S.nrm, n, slot), rd = ld3s(S.rd, n, slot), T = ld3s(S.T, n, slot);
            F3 sum = ld3s(S.sum, n, slot);
            Rng rng{S.rng0[slot], S.rng1[slot]};
            Mat m = load_mat(tbl, P.mat_off, mid);
            ShadeK sk = make_shade(nn, -rd, m);
            F3 D = tr.direct_light(x, sk, rng);
            sum = sum + T * D;
            bool alive = true;
            F3 ro;
            if (scatter) {
                float ct = sample_phase_function(m.aniso, rng.gen());
                float up = rng.gen();
                ro = x;
                rd = scatter_dir(rd, ct, up);
            } else {
                float p = fmaxf(fmaxf(m.color.x, m.color.y), m.color.z);
                bool cont = depth < 2u;
                if (!cont) cont = rng.gen() < p;
                if (cont) {
                    F3 nd, f;
                    if (rng.gen() < (1.0f - m.metallic)) {
                        float up = rng.gen(), r2 = rng.gen();
                        nd = diffuse_dir(nn, up, r2);
                        f = m.color * 0.31830988618379067154f;
                    } else {
                        float r1 = rng.gen(), r2 = rng.gen();
                        nd = reflect(rd, sample_ggx_h(nn, m.roughness, r1, r2));
                        f = F3{1.0f, 1.0f, 1.0f};
                    }
                    float c = dot(nd, nn);
                    if (c > 0.0f) {
                        float w = depth >= 2u ? c * rcp(p) : c;
                        T = T * f * w;
                        ro = madd(x, nd, 1e-4f);
                        rd = nd;
                        media = next_media;
                    } else {
                        alive = false;
                    }
                } else {
                    alive = false;
                }
            }
            ++depth;
            if (depth >= P.max_depth) alive = false;
            st3s(S.sum, n, slot, sum);
            if (alive) {
                st3s(S.ro, n, slot, ro);
                st3s(S.rd, n, slot, rd);
                st3s(S.T, n, slot, T);
                S.rng0[slot] = rng.s0;
                S.rng1[slot] = rng.s1;
            }
            S.meta[slot] = (alive ? 1u : 0u) | (depth << 1) | ((uint32_t)(media + 1) << 8);
            to_next = alive || S.next_sample[slot] < s_end;
        }
        queue_push(to_next, slot, q_next, q_next_count);
    }
}

__global__ __launch_bounds__(256) void wf_finalize(WfState S, float *__restrict__ accum) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= S.n) return;
    accum[3 * i + 0] += S.sum[i];
    accum[3 * i + 1] += S.sum[S.n + i];
    accum[3 * i + 2] += S.sum[2 * S.n + i];
}

// ---- host driver -----------------------------------------------------------------------
size_t wf_workspace_bytes(uint32_t n_pixels) {
    // 18 float planes + 5 uint planes + 3 queues + counters
    return (size_t)n_pixels * 4 * (18 + 5 + 3) + 4096 + 32 * 256;
}

// Renders P.sample_count samples per pixel into accum (adds). Workspace from wf_workspace_bytes().
// iterations_out / launches_out: bookkeeping for the bench.
hipError_t wavefront_render(const Params &P, const float4 *blob, float *accum, void *workspace, hipStream_t st,
                            uint32_t *iterations_out) {
    const uint32_t n = P.width * P.height;
    uint8_t *w = (uint8_t *)workspace;
    auto take = [&](size_t bytes) {
        void *p = w;
        w += (bytes + 255) & ~(size_t)255;
        return p;
    };
    uint32_t *counters = (uint32_t *)take(4096);  // [0],[1]: extend queue counts (ping-pong), [2]: shade queue count
    WfState S;
    S.n = n;
    S.ro = (float *)take((size_t)n * 12);
    S.rd = (float *)take((size_t)n * 12);
    S.T = (float *)take((size_t)n * 12);
    S.sum = (float *)take((size_t)n * 12);
    S.x = (float *)take((size_t)n * 12);
    S.nrm = (float *)take((size_t)n * 12);
    S.rng0 = (uint32_t *)take((size_t)n * 4);
    S.rng1 = (uint32_t *)take((size_t)n * 4);
    S.meta = (uint32_t *)take((size_t)n * 4);
    S.next_sample = (uint32_t *)take((size_t)n * 4);
    S.mat = (uint32_t *)take((size_t)n * 4);
    uint32_t *q[2] = {(uint32_t *)take((size_t)n * 4), (uint32_t *)take((size_t)n * 4)};
    uint32_t *q_shade = (uint32_t *)take((size_t)n * 4);

    const unsigned blocks_all = (n + 255) / 256;
    const unsigned grid = std::min(blocks_all, 256u * 8u);  // grid-stride over the queues
    const size_t shmem = (size_t)P.blob_f4 * sizeof(float4);
    hipError_t e;
    if ((e = hipMemsetAsync(counters, 0, 4096, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(wf_init, dim3(blocks_all), dim3(256), 0, st, S, P.sample_begin, q[0], &counters[0]);
    // every sample needs at most max_depth extend iterations; finished slots leave the queues, so
    // the queue is empty after at most sample_count * max_depth iterations. Poll the count now and then.
    const uint32_t max_iter = P.sample_count * P.max_depth + 1;
    uint32_t it = 0;
    int cur = 0;
    for (; it < max_iter; ++it) {
        if ((e = hipMemsetAsync(&counters[1 - cur], 0, 4, st)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(&counters[2], 0, 4, st)) != hipSuccess) return e;
        hipLaunchKernelGGL(wf_extend, dim3(grid), dim3(256), shmem, st, P, blob, S, q[cur], &counters[cur], q[1 - cur],
                           &counters[1 - cur], q_shade, &counters[2]);
        hipLaunchKernelGGL(wf_shade, dim3(grid), dim3(256), shmem, st, P, blob, S, q_shade, &counters[2], q[1 - cur],
                           &counters[1 - cur]);
        cur = 1 - cur;
        if ((it & 63u) == 63u) {
            uint32_t left = 0;
            if ((e = hipMemcpyAsync(&left, &counters[cur], 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
            if (left == 0) {
                ++it;
                break;
            }
        }
    }
    hipLaunchKernelGGL(wf_finalize, dim3(blocks_all), dim3(256), 0, st, S, accum);
    if (iterations_out) *iterations_out = it;
    return hipGetLastError();
}

}  // namespace pt
