// pt_wavefront.hip — wavefront (queue-based) variant of the trace loop, built to decide
// "megakernel vs wavefront" from measurements (north star).
//
// One path per pixel in flight; the path state travels WITH its queue entry, in SoA planes over
// queue positions, so every read (entry i by lane i) and every append (base + rank in wave) is
// coalesced. (A first version kept the state in per-pixel arrays and queued indices: its gathers
// of 4-byte elements at compacted, i.e. scattered, indices ran 16x slower than the megakernel.)
// Per round, two kernels over the compacted queues:
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
#include "pt_wfqueue.h"

namespace pt {

__global__ __launch_bounds__(256) void wf_init(WfQueue Q, uint32_t n, uint32_t sample_begin) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    Q.stu(U_PIX, i, i);
    Q.stu(U_META, i, 0);  // dead: wf_extend starts the first sample
    Q.stu(U_NEXT, i, sample_begin);
    Q.st3(F_SUM, i, F3{0, 0, 0});
    if (i == 0) *Q.count = n;
}

__global__ __launch_bounds__(256) void wf_extend(Params P, const float4 *__restrict__ blob, WfQueue Qin, WfQueue Qnext,
                                                 WfQueue Qshade, float *__restrict__ accum) {
    extern __shared__ float4 s_tbl[];
    __shared__ uint32_t s_cnt[2][4], s_base[2];
    for (uint32_t i = threadIdx.x; i < P.blob_f4; i += 256u) s_tbl[i] = blob[i];
    __syncthreads();
    TblLds tbl{s_tbl};
    Tracer<TblLds, false, false> tr(P, tbl, Bvh{}, nullptr);
    const uint32_t count = *Qin.count;
    const uint32_t s_end = P.sample_begin + P.sample_count;
    const uint32_t seed_mixed = mix32(P.seed ^ 0x85ebca6bu);
    for (uint32_t base = blockIdx.x * 256u; base < count; base += gridDim.x * 256u) {
        uint32_t qi = base + threadIdx.x;
        bool act = qi < count;
        bool to_next = false, to_shade = false;
        uint32_t pix = 0, next = 0, depth = 0, mat_word = 0;
        int media = -1;
        bool alive = false, scatter = false;
        Rng rng{0u, 1u};
        F3 ro = F3{0, 0, 0}, rd = F3{0, 0, 1}, T = F3{1, 1, 1}, sum = F3{0, 0, 0}, x = F3{0, 0, 0}, nn = F3{0, 1, 0};
        if (act) {
            pix = Qin.ldu(U_PIX, qi);
            uint32_t meta = Qin.ldu(U_META, qi);
            next = Qin.ldu(U_NEXT, qi);
            sum = Qin.ld3(F_SUM, qi);
            alive = meta & 1u;
            depth = (meta >> 1) & 31u;
            media = (int)(meta >> 8) - 1;
            if (!alive) {  // queued only while next < s_end
                uint32_t px = pix % P.width, py = pix / P.width;
                rng.seed(seed_mixed, mix32(pix + seed_mixed), next);
                float jx = rng.gen(), jy = rng.gen(), ds = rng.gen(), dt = rng.gen();
                camera_ray(tbl, P, px, py, jx, jy, ds, dt, ro, rd);
                depth = 0;
                media = -1;
                alive = true;
                ++next;
            } else {
                ro = Qin.ld3(F_RO, qi);
                rd = Qin.ld3(F_RD, qi);
                T = Qin.ld3(F_T, qi);
                rng.s0 = Qin.ldu(U_RNG0, qi);
                rng.s1 = Qin.ldu(U_RNG1, qi);
            }
            RayK ray = make_ray(ro, rd);
            HitRec h = tr.closest(ray);
            F3 A = F3{1, 1, 1};
            float t_media = __builtin_inff();
            if (media >= 0) {
                Mat mm = load_mat(tbl, P.mat_off, (uint32_t)media);
                t_media = -fln(rng.gen()) * rcp(mm.density);
                float k = -mm.density * fminf(t_media, h.t);
                A = F3{fexp(fmaxf(1.0f - mm.color.x, 0.0f) * k), fexp(fmaxf(1.0f - mm.color.y, 0.0f) * k),
                       fexp(fmaxf(1.0f - mm.color.z, 0.0f) * k)};
            }
            if (t_media < h.t) {
                T = T * A;
                x = point_at_ref(ro, rd, t_media);
                nn = F3{0.0f, 1.0f, 0.0f};
                mat_word = (uint32_t)media | ((uint32_t)(media + 1) << 16);
                scatter = true;
                to_shade = true;
            } else if (h.id < 0) {
                alive = false;
            } else {
                uint32_t mid, order;
                tr.finish(ray, h, x, nn, mid, order);
                Mat m = load_mat(tbl, P.mat_off, mid);
                int next_media = media;
                if (m.density > 0.0f) next_media = (dot(-rd, nn) > 0.0f) ? (int)mid : -1;
                T = T * A;
                if (m.ior > 1.0f && m.metallic < 0.1f) {
                    F3 nd = glass_dir(rd, nn, m.ior, rng.gen());
                    ro = madd(x, nd, 1e-4f);
                    rd = nd;
                    media = next_media;
                    ++depth;
                    if (depth >= P.max_depth) alive = false;
                } else {
                    mat_word = mid | ((uint32_t)(next_media + 1) << 16);
                    to_shade = true;
                }
            }
            if (!to_shade) {
                to_next = alive || next < s_end;  // a finished path comes back to start the pixel's next sample
                if (!to_next) {                   // pixel done: its sum goes to the accumulator
                    accum[3 * (size_t)pix + 0] += sum.x;
                    accum[3 * (size_t)pix + 1] += sum.y;
                    accum[3 * (size_t)pix + 2] += sum.z;
                }
            }
        }
        uint32_t meta_out = (alive ? 1u : 0u) | (depth << 1) | (scatter ? 64u : 0u) | ((uint32_t)(media + 1) << 8);
        uint32_t pn, ps;
        queue_reserve2(to_next, Qnext.count, pn, to_shade, Qshade.count, ps, s_cnt, s_base);
        if (to_next) {
            Qnext.stu(U_PIX, pn, pix);
            Qnext.stu(U_META, pn, meta_out);
            Qnext.stu(U_NEXT, pn, next);
            Qnext.st3(F_SUM, pn, sum);
            if (alive) {
                Qnext.st3(F_RO, pn, ro);
                Qnext.st3(F_RD, pn, rd);
                Qnext.st3(F_T, pn, T);
                Qnext.stu(U_RNG0, pn, rng.s0);
                Qnext.stu(U_RNG1, pn, rng.s1);
            }
        }
        if (to_shade) {
            Qshade.stu(U_PIX, ps, pix);
            Qshade.stu(U_META, ps, meta_out);
            Qshade.stu(U_NEXT, ps, next);
            Qshade.stu(U_MAT, ps, mat_word);
            Qshade.stu(U_RNG0, ps, rng.s0);
            Qshade.stu(U_RNG1, ps, rng.s1);
            Qshade.st3(F_SUM, ps, sum);
            Qshade.st3(F_RD, ps, rd);
            Qshade.st3(F_T, ps, T);
            Qshade.st3(F_X, ps, x);
            Qshade.st3(F_N, ps, nn);
        }
    }
}

__global__ __launch_bounds__(256) void wf_shade(Params P, const float4 *__restrict__ blob, WfQueue Qshade, WfQueue Qnext,
                                                float *__restrict__ accum) {
    extern __shared__ float4 s_tbl[];
    __shared__ uint32_t s_cnt[2][4], s_base[2];
    for (uint32_t i = threadIdx.x; i < P.blob_f4; i += 256u) s_tbl[i] = blob[i];
    __syncthreads();
    TblLds tbl{s_tbl};
    Tracer<TblLds, false, false> tr(P, tbl, Bvh{}, nullptr);
    const uint32_t count = *Qshade.count;
    const uint32_t s_end = P.sample_begin + P.sample_count;
    for (uint32_t base = blockIdx.x * 256u; base < count; base += gridDim.x * 256u) {
        uint32_t qi = base + threadIdx.x;
        bool act = qi < count;
        bool to_next = false, alive = false;
        uint32_t pix = 0, next = 0, depth = 0;
        int media = -1;
        Rng rng{0u, 1u};
        F3 ro = F3{0, 0, 0}, rd = F3{0, 0, 1}, T = F3{1, 1, 1}, sum = F3{0, 0, 0};
        if (act) {
            pix = Qshade.ldu(U_PIX, qi);
            uint32_t meta = Qshade.ldu(U_META, qi);
            next = Qshade.ldu(U_NEXT, qi);
            uint32_t mw = Qshade.ldu(U_MAT, qi);
            rng.s0 = Qshade.ldu(U_RNG0, qi);
            rng.s1 = Qshade.ldu(U_RNG1, qi);
            depth = (meta >> 1) & 31u;
            bool scatter = meta & 64u;
            media = (int)(meta >> 8) - 1;
            uint32_t mid = mw & 0xffffu;
            int next_media = (int)(mw >> 16) - 1;
            F3 x = Qshade.ld3(F_X, qi), nn = Qshade.ld3(F_N, qi);
            rd = Qshade.ld3(F_RD, qi);
            T = Qshade.ld3(F_T, qi);
            sum = Qshade.ld3(F_SUM, qi);
            Mat m = load_mat(tbl, P.mat_off, mid);
            ShadeK sk = make_shade(nn, -rd, m);
            F3 D = tr.direct_light(x, sk, rng);
            sum = sum + T * D;
            alive = true;
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
            to_next = alive || next < s_end;
            if (!to_next) {
                accum[3 * (size_t)pix + 0] += sum.x;
                accum[3 * (size_t)pix + 1] += sum.y;
                accum[3 * (size_t)pix + 2] += sum.z;
            }
        }
        uint32_t pn, unused;
        queue_reserve2(to_next, Qnext.count, pn, false, nullptr, unused, s_cnt, s_base);
        if (to_next) {
            Qnext.stu(U_PIX, pn, pix);
            Qnext.stu(U_META, pn, (alive ? 1u : 0u) | (depth << 1) | ((uint32_t)(media + 1) << 8));
            Qnext.stu(U_NEXT, pn, next);
            Qnext.st3(F_SUM, pn, sum);
            if (alive) {
                Qnext.st3(F_RO, pn, ro);
                Qnext.st3(F_RD, pn, rd);
                Qnext.st3(F_T, pn, T);
                Qnext.stu(U_RNG0, pn, rng.s0);
                Qnext.stu(U_RNG1, pn, rng.s1);
            }
        }
    }
}

// ---- host driver -----------------------------------------------------------------------
size_t wf_workspace_bytes(uint32_t n_pixels) {
    // three queues x (NF float + NU uint planes) + counters
    size_t per_queue = (queue_bytes(n_pixels) + 511) & ~(size_t)255;
    return 3 * per_queue + 4096;
}

// Renders P.sample_count samples per pixel and adds each pixel's radiance sum into accum.
// Workspace from wf_workspace_bytes(). iterations_out: number of extend/shade rounds run.
hipError_t wavefront_render(const Params &P, const float4 *blob, float *accum, void *workspace, hipStream_t st,
                            uint32_t *iterations_out) {
    const uint32_t n = P.width * P.height;
    uint8_t *w = (uint8_t *)workspace;
    uint32_t *counters = (uint32_t *)w;  // [0],[1]: ray queue counts (ping-pong), [2]: shade queue count
    w += 4096;
    size_t per_queue = (queue_bytes(n) + 511) & ~(size_t)255;
    WfQueue Q[3];
    for (int k = 0; k < 3; ++k) {
        Q[k].f = (float *)(w + per_queue * k);
        Q[k].count = &counters[k];
        Q[k].cap = n;
    }
    const unsigned blocks_all = (n + 255) / 256;
    const unsigned grid = std::min(blocks_all, 256u * 8u);  // grid-stride over the queues
    const size_t shmem = (size_t)P.blob_f4 * sizeof(float4);
    hipError_t e;
    if ((e = hipMemsetAsync(counters, 0, 4096, st)) != hipSuccess) return e;
    if (P.sample_count == 0) return hipSuccess;
    hipLaunchKernelGGL(wf_init, dim3(blocks_all), dim3(256), 0, st, Q[0], n, P.sample_begin);
    // every sample needs at most max_depth rounds, so the ray queue is empty after at most
    // sample_count * max_depth rounds; the count is polled every 64 rounds to stop early.
    const uint32_t max_iter = P.sample_count * P.max_depth + 1;
    uint32_t it = 0;
    int cur = 0;
    for (; it < max_iter; ++it) {
        if ((e = hipMemsetAsync(&counters[1 - cur], 0, 4, st)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(&counters[2], 0, 4, st)) != hipSuccess) return e;
        hipLaunchKernelGGL(wf_extend, dim3(grid), dim3(256), shmem, st, P, blob, Q[cur], Q[1 - cur], Q[2], accum);
        hipLaunchKernelGGL(wf_shade, dim3(grid), dim3(256), shmem, st, P, blob, Q[2], Q[1 - cur], accum);
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
    if (iterations_out) *iterations_out = it;
    return hipGetLastError();
}

}  // namespace pt
