// hippt_cli.cpp — host driver with the behaviour of the reference's main()
// on its `--gpu` branch (src/main.rs:29-108): load ./scene.json, print the
// camera / object / light dump, render through the HIP backend, save
// renders/render_{w}x{h}_s{spp}_ap{ap:.2}_f{focus:.1}_{6 alnum}.png.
//
// Flags of the reference: --gpu (accepted; this binary has only the GPU path),
// --quiet / -q. Extra flags default to reference behaviour:
//   --scene PATH  --width W --height H --samples S  --seed N  --device D  --out PATH
// Progressive / checkpointed rendering (analogue of the dispatch loop of src/gpu_renderer.rs:94-183):
//   --chunk N          render N samples per pixel at a time, accumulating on the host
//   --checkpoint PATH  after every chunk, save the accumulator + samples done to PATH
//   --resume           start from PATH if it exists (same size and seed), render the remaining samples
//   --preview          after every chunk, also write the image of the samples so far to the output path
// Several GPUs in this process (sample-range split + one RCCL reduce, hippt_render_rgba8_multi):
//   --devices 0,1,...  the HIP devices to use, or `all` (not combined with --chunk / --checkpoint)
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hippt.h"

static void die(const char *what) {
    std::fprintf(stderr, "%s: %s\n", what, hippt_last_error());
    std::exit(101);  // Rust panic exit code
}

int main(int argc, char **argv) {
    std::string scene_path = "scene.json", out_path;
    uint32_t w = 0, h = 0, spp = 0, seed = 0;
    int device = 0;
    bool have_spp = false, resume = false, preview = false;
    uint32_t chunk = 0;
    std::string ckpt_path;
    std::vector<int> devices;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "missing value for %s\n", a.c_str());
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "--gpu" || a == "--quiet" || a == "-q") continue;
        else if (a == "--scene") scene_path = next();
        else if (a == "--width") w = (uint32_t)std::strtoul(next(), nullptr, 10);
        else if (a == "--height") h = (uint32_t)std::strtoul(next(), nullptr, 10);
        else if (a == "--samples") spp = (uint32_t)std::strtoul(next(), nullptr, 10), have_spp = true;
        else if (a == "--seed") seed = (uint32_t)std::strtoul(next(), nullptr, 10);
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--out") out_path = next();
        else if (a == "--chunk") chunk = (uint32_t)std::strtoul(next(), nullptr, 10);
        else if (a == "--checkpoint") ckpt_path = next();
        else if (a == "--resume") resume = true;
        else if (a == "--preview") preview = true;
        else if (a == "--devices") {
            const char *list = next();
            if (std::strcmp(list, "all") == 0) {
                int n = 0;
                if (hippt_device_count(&n)) die("devices");
                for (int d = 0; d < n; ++d) devices.push_back(d);
                list = "";
            }
            for (const char *p = list; *p;) {
                char *end = nullptr;
                devices.push_back((int)std::strtol(p, &end, 10));
                if (end == p) {
                    std::fprintf(stderr, "--devices wants a comma-separated list of device numbers\n");
                    return 2;
                }
                p = *end == ',' ? end + 1 : end;
            }
        } else {
            std::fprintf(stderr, "unknown flag %s\n", a.c_str());
            return 2;
        }
    }

    hippt_scene_file *sf = nullptr;
    if (hippt_scene_load(scene_path.c_str(), &sf)) die("scene file");
    const hippt_scene *sc = hippt_scene_get(sf);
    if (w || h || have_spp) {
        if (hippt_scene_set_render(sf, w ? w : sc->width, h ? h : sc->height, have_spp ? spp : sc->samples))
            die("render override");
        sc = hippt_scene_get(sf);
    }
    const hippt_camera &c = sc->camera;

    // src/main.rs:60-87
    {
        std::vector<char> dump((size_t)hippt_scene_dump(sf, nullptr, 0) + 1);
        hippt_scene_dump(sf, dump.data(), dump.size());
        std::fputs(dump.data(), stdout);
    }

    char name[512];
    if (out_path.empty()) {
        hippt_render_image_name(sc->width, sc->height, sc->samples, c.aperture, c.focus, name, sizeof name);
        mkdir("renders", 0755);  // fs::create_dir_all (src/main.rs:99-101)
    } else {
        std::snprintf(name, sizeof name, "%s", out_path.c_str());
    }

    std::printf("Running GPU renderer...\n");  // src/main.rs:90
    std::vector<uint8_t> rgba((size_t)sc->width * sc->height * 4);
    hippt_options opt{};
    opt.seed = seed;
    opt.device = device;
    auto t0 = std::chrono::steady_clock::now();
    uint32_t rendered = sc->samples;
    if (!devices.empty()) {
        if (chunk != 0 || !ckpt_path.empty()) {
            std::fprintf(stderr, "--devices is not combined with --chunk / --checkpoint\n");
            return 2;
        }
        if (hippt_render_rgba8_multi(sc, &opt, devices.data(), (int)devices.size(), rgba.data())) die("render");
    } else if (chunk == 0 && ckpt_path.empty()) {
        if (hippt_render_rgba8(sc, &opt, rgba.data())) die("render");
    } else {
        if (sc->samples == 0) {
            std::fprintf(stderr, "render.samples is 0\n");
            return 101;
        }
        if (chunk == 0) chunk = sc->samples;
        size_t npx = (size_t)sc->width * sc->height;
        std::vector<float> accum(npx * 3, 0.0f);
        uint32_t done = 0;
        if (resume && !ckpt_path.empty()) {
            FILE *probe = std::fopen(ckpt_path.c_str(), "rb");
            if (probe) {
                std::fclose(probe);
                uint32_t ck_seed = 0;
                if (hippt_checkpoint_load(ckpt_path.c_str(), accum.data(), sc->width, sc->height, &done, &ck_seed)) die("checkpoint");
                if (ck_seed != seed || done > sc->samples) {
                    std::fprintf(stderr, "checkpoint %s: seed %u / %u samples do not fit this render (seed %u, %u samples)\n",
                                 ckpt_path.c_str(), ck_seed, done, seed, sc->samples);
                    return 101;
                }
                std::printf("Resuming at sample %u of %u\n", done, sc->samples);
            }
        }
        rendered = sc->samples - done;
        hippt_context *ctx = nullptr;
        if (hippt_create(device, &ctx)) die("device");
        if (hippt_set_scene(ctx, sc)) die("scene");
        while (done < sc->samples) {
            uint32_t n = sc->samples - done < chunk ? sc->samples - done : chunk;  // the last chunk is short, not over-rendered
            opt.sample_begin = done;
            opt.sample_count = n;
            if (hippt_render_accum_add_f32(ctx, &opt, accum.data())) die("render");
            done += n;
            if (!ckpt_path.empty() && hippt_checkpoint_save(ckpt_path.c_str(), accum.data(), sc->width, sc->height, done, seed))
                die("checkpoint");
            if (preview && done < sc->samples) {
                if (hippt_resolve_f32(ctx, accum.data(), (uint32_t)npx, done, rgba.data())) die("resolve");
                if (hippt_write_png_rgba8(name, rgba.data(), sc->width, sc->height)) die("save");
            }
            std::printf("  %u / %u samples\n", done, sc->samples);
        }
        if (hippt_resolve_f32(ctx, accum.data(), (uint32_t)npx, sc->samples, rgba.data())) die("resolve");
        hippt_destroy(ctx);
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double paths = (double)sc->width * sc->height * rendered;
    std::printf("Rendered %ux%u at %u spp in %.3f s (%.1f Mpaths/s incl. device setup)\n", sc->width, sc->height,
                rendered, secs, paths / secs / 1e6);

    if (hippt_write_png_rgba8(name, rgba.data(), sc->width, sc->height)) die("save");
    std::printf("Saved \xE2\x86\x92 %s\n", name);  // src/main.rs:106
    hippt_scene_free(sf);
    return 0;
}
