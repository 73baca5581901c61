// hippt_multi.hip — one process, several GPUs: the sample range of an image is split over the devices, every
// device renders all pixels for its share of the GLOBAL sample indices (the RNG stream is keyed by (seed, pixel,
// global sample), so the image does not depend on the number of devices), and the f32 radiance sums are added up on
// the first device with one RCCL reduce over xGMI before the tonemap (north star "Multi-GPU
// partitioning"). The reference has one device and adds its dispatches up on the host (gpu_renderer.rs:175-179).
//
// A host with one process per GPU (bench.py under torch.distributed) does the same with its own communicator and
// hippt_render_accum_device; this file is for the single-process host, i.e. the reference's own `--gpu` binary.
// librccl is opened at the first call (dlopen), so that libhippt has no load-time dependency on it and never brings
// a second RCCL into a process that already holds one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hippt_internal.h"
#include "pt_kernels.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

const Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {std::getenv("HIPPT_RCCL_LIB"), "librccl.so.1", "librccl.so"};
        // an RCCL the process already holds comes first
        for (const char *n : names)
            if (n && !r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        for (const char *n : names)
            if (n && !r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) return;
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.handle, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.Reduce = (decltype(r.Reduce))dlsym(r.handle, "ncclReduce");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.handle, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.handle, "ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
        r.ok = r.CommInitAll && r.CommDestroy && r.Reduce && r.GroupStart && r.GroupEnd && r.GetErrorString;
    });
    return r;
}

struct Rank {
    hippt_context *ctx = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    float *d_accum = nullptr;
    ncclComm_t comm = nullptr;
    hipEvent_t done = nullptr;  // HIPPT_FLAG_LOCAL_REDUCE: this rank's render has finished
};

void release(std::vector<Rank> &ranks, float *staging = nullptr) {
    const Rccl &R = rccl();
    if (staging && !ranks.empty()) {
        (void)hipSetDevice(ranks[0].device);
        (void)hipFree(staging);
    }
    for (Rank &k : ranks) {
        (void)hipSetDevice(k.device);
        if (k.stream) (void)hipStreamSynchronize(k.stream);
        if (k.comm && R.ok) (void)R.CommDestroy(k.comm);
        if (k.done) (void)hipEventDestroy(k.done);
        if (k.d_accum) (void)hipFree(k.d_accum);
        if (k.stream) (void)hipStreamDestroy(k.stream);
        if (k.ctx) hippt_destroy(k.ctx);
    }
    ranks.clear();
}

// Sum of the ranks' accumulators on ranks[0] without RCCL (HIPPT_FLAG_LOCAL_REDUCE): rank 0's stream waits for each
// rank's render, fetches the rank's sums (in place if the rank shares rank 0's device, by peer copy into a staging
// buffer otherwise) and adds them with a small kernel, ranks in ascending order (deterministic).
int local_reduce(std::vector<Rank> &ranks, size_t n_float, float **staging) {
    Rank &r0 = ranks[0];
    for (size_t i = 1; i < ranks.size(); ++i) {
        Rank &k = ranks[i];
        hipError_t e = hipSetDevice(k.device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&k.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(k.done, k.stream);
        if (e == hipSuccess) e = hipSetDevice(r0.device);
        if (e == hipSuccess) e = hipStreamWaitEvent(r0.stream, k.done, 0);
        const float *src = k.d_accum;
        if (e == hipSuccess && k.device != r0.device) {
            if (!*staging) e = hipMalloc((void **)staging, n_float * sizeof(float));
            if (e == hipSuccess) e = hipMemcpyPeerAsync(*staging, r0.device, k.d_accum, k.device, n_float * sizeof(float), r0.stream);
            src = *staging;
        }
        if (e == hipSuccess) e = pt::launch_add(r0.d_accum, src, n_float, r0.stream);
        if (e != hipSuccess) return hippt_set_error(HIPPT_ERR_DEVICE, "local reduce of device %d: %s", k.device, hipGetErrorString(e));
    }
    return HIPPT_OK;
}

// Renders the sample range of `opts` (from sample_begin to the image's last sample if its count is 0, as the
// one-device entry points read it) split over the devices and leaves the reduced sums in ranks[0].d_accum; the caller
// resolves or copies them and then calls release().
int render_reduced(const hippt_scene *scene, const hippt_options *opts, const int *devices, int n_devices,
                   std::vector<Rank> &ranks, uint32_t *spp_out, float **staging) {
    if (!scene || !devices || n_devices < 1) return hippt_set_error(HIPPT_ERR_INVALID, "bad multi-GPU arguments");
    hippt_options base{};
    if (opts) base = *opts;
    const bool local = (base.flags & HIPPT_FLAG_LOCAL_REDUCE) != 0;
    base.flags &= ~HIPPT_FLAG_LOCAL_REDUCE;
    if (!local)  // RCCL wants one rank per device; the local reduce lets a one-GPU box rehearse several ranks
        for (int i = 0; i < n_devices; ++i)
            for (int j = 0; j < i; ++j)
                if (devices[i] == devices[j]) return hippt_set_error(HIPPT_ERR_INVALID, "device %d is listed twice", devices[i]);
    const Rccl &R = rccl();
    if (!local && !R.ok) {
        const char *why = dlerror();
        return hippt_set_error(HIPPT_ERR_DEVICE, "librccl.so could not be opened (set HIPPT_RCCL_LIB): %s", why ? why : "symbols missing");
    }
    const uint32_t s0 = base.sample_begin;
    const uint32_t total = base.sample_count ? base.sample_count : (scene->samples > s0 ? scene->samples - s0 : 0u);
    if (spp_out) *spp_out = total;
    const size_t n_float = (size_t)scene->width * scene->height * 3;

    ranks.assign((size_t)n_devices, Rank{});
    for (int i = 0; i < n_devices; ++i) {
        Rank &k = ranks[(size_t)i];
        k.device = devices[i];
        int rc = hippt_create(k.device, &k.ctx);
        if (rc) return rc;
        rc = hippt_set_scene(k.ctx, scene);
        if (rc) return rc;
        if (hipSetDevice(k.device) != hipSuccess || hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc((void **)&k.d_accum, n_float * sizeof(float)) != hipSuccess ||
            hipMemsetAsync(k.d_accum, 0, n_float * sizeof(float), k.stream) != hipSuccess)
            return hippt_set_error(HIPPT_ERR_DEVICE, "device %d: stream or accumulator allocation failed (%zu bytes)", k.device,
                                   n_float * sizeof(float));
    }
    // One communicator per device, BEFORE any render is queued: communicator setup allocates and synchronises on
    // every device, and a failure here must not cost a full render.
    if (!local) {
        std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
        ncclResult_t nr = R.CommInitAll(comms.data(), n_devices, devices);
        for (int i = 0; i < n_devices; ++i) ranks[(size_t)i].comm = comms[(size_t)i];
        if (nr != ncclSuccess) return hippt_set_error(HIPPT_ERR_DEVICE, "ncclCommInitAll over %d devices: %s", n_devices, R.GetErrorString(nr));
    }
    // one host thread per device: the queue pipelines (LBVH scenes) poll their counters and block their caller
    std::vector<int> codes((size_t)n_devices, HIPPT_OK);
    std::vector<std::string> messages((size_t)n_devices);
    std::vector<std::thread> workers;
    for (int i = 0; i < n_devices; ++i) {
        workers.emplace_back([&, i] {
            Rank &k = ranks[(size_t)i];
            hippt_options o = base;
            hippt_sample_range(total, (uint32_t)i, (uint32_t)n_devices, &o.sample_begin, &o.sample_count);
            o.sample_begin += s0;
            o.device = k.device;
            if (o.sample_count == 0) return;  // more devices than samples
            codes[(size_t)i] = hippt_render_accum_device(k.ctx, &o, k.d_accum, (void *)k.stream);
            if (codes[(size_t)i]) messages[(size_t)i] = hippt_last_error();  // the message is per thread
        });
    }
    for (std::thread &t : workers) t.join();
    for (int i = 0; i < n_devices; ++i)
        if (codes[(size_t)i]) return hippt_set_error(codes[(size_t)i], "device %d: %s", devices[i], messages[(size_t)i].c_str());
    if (local) {
        int rc = local_reduce(ranks, n_float, staging);
        if (rc) return rc;
    } else {
        // one grouped reduce onto rank 0 (in place there), each rank's part on its device's stream, behind its render
        ncclResult_t nr = R.GroupStart();
        for (int i = 0; i < n_devices && nr == ncclSuccess; ++i) {
            Rank &k = ranks[(size_t)i];
            nr = R.Reduce(k.d_accum, k.d_accum, n_float, ncclFloat32, ncclSum, 0, k.comm, k.stream);
        }
        ncclResult_t ne = R.GroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) return hippt_set_error(HIPPT_ERR_DEVICE, "ncclReduce: %s", R.GetErrorString(nr));
    }
    for (Rank &k : ranks) {
        hipError_t e = hipSetDevice(k.device);
        if (e == hipSuccess) e = hipStreamSynchronize(k.stream);
        if (e != hipSuccess) return hippt_set_error(HIPPT_ERR_DEVICE, "device %d: %s", k.device, hipGetErrorString(e));
    }
    return HIPPT_OK;
}

}  // namespace

extern "C" {

int hippt_device_count(int *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = e == hipSuccess ? n : 0;
    if (e != hipSuccess || n == 0) return hippt_set_error(HIPPT_ERR_DEVICE, "no HIP device: %s", hipGetErrorString(e));
    return HIPPT_OK;
}

void hippt_sample_range(uint32_t total, uint32_t rank, uint32_t world, uint32_t *begin, uint32_t *count) {
    if (world == 0) world = 1;
    const uint32_t base = total / world, rem = total % world;
    if (begin) *begin = rank * base + (rank < rem ? rank : rem);
    if (count) *count = base + (rank < rem ? 1u : 0u);
}

int hippt_render_accum_f32_multi(const hippt_scene *scene, const hippt_options *opts, const int *devices, int n_devices,
                                 float *out_accum) {
    if (!out_accum) return hippt_set_error(HIPPT_ERR_INVALID, "out_accum is NULL");
    std::vector<Rank> ranks;
    float *staging = nullptr;
    int rc = render_reduced(scene, opts, devices, n_devices, ranks, nullptr, &staging);
    if (rc == HIPPT_OK) {
        hipError_t e = hipSetDevice(ranks[0].device);
        if (e == hipSuccess)
            e = hipMemcpy(out_accum, ranks[0].d_accum, (size_t)scene->width * scene->height * 3 * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hippt_set_error(HIPPT_ERR_DEVICE, "copy out: %s", hipGetErrorString(e));
    }
    release(ranks, staging);
    return rc;
}

int hippt_render_rgba8_multi(const hippt_scene *scene, const hippt_options *opts, const int *devices, int n_devices,
                             uint8_t *out_rgba) {
    if (!out_rgba) return hippt_set_error(HIPPT_ERR_INVALID, "out_rgba is NULL");
    if (scene && scene->samples == 0) return hippt_set_error(HIPPT_ERR_INVALID, "render.samples is 0");
    hippt_options whole{};
    if (opts) whole = *opts;
    whole.sample_begin = 0;
    whole.sample_count = 0;  // the image's own sample count
    std::vector<Rank> ranks;
    uint32_t spp = 0;
    float *staging = nullptr;
    int rc = render_reduced(scene, &whole, devices, n_devices, ranks, &spp, &staging);
    uint8_t *d_rgba = nullptr;
    if (rc == HIPPT_OK) {
        const size_t npx = (size_t)scene->width * scene->height;
        hipError_t e = hipSetDevice(ranks[0].device);
        if (e == hipSuccess) e = hipMalloc((void **)&d_rgba, npx * 4);
        if (e != hipSuccess) rc = hippt_set_error(HIPPT_ERR_DEVICE, "hipMalloc(%zu): %s", npx * 4, hipGetErrorString(e));
        if (rc == HIPPT_OK) rc = hippt_resolve_device(ranks[0].ctx, ranks[0].d_accum, spp, d_rgba, (void *)ranks[0].stream);
        if (rc == HIPPT_OK) {
            e = hipStreamSynchronize(ranks[0].stream);
            if (e == hipSuccess) e = hipMemcpy(out_rgba, d_rgba, npx * 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = hippt_set_error(HIPPT_ERR_DEVICE, "resolve: %s", hipGetErrorString(e));
        }
    }
    if (d_rgba) (void)hipFree(d_rgba);
    release(ranks, staging);
    return rc;
}

}  // extern "C"
