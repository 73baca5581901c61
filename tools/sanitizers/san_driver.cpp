// san_driver.cpp — host code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU builds only: GPU sanitizers are
// not available on this pool). Links pathtracer_amd/csrc/host_scene.cpp (the scene.json loader, dump, PNG and
// checkpoint writers of the product) and oracle/pt_oracle.cpp (the CPU restatement used as the checker) directly:
//   1. N random mutations of scenes/scene.example.json through hippt_scene_parse (the mutator of
//      tests/test_abi_host.py::test_loader_survives_mutated_scene_files, restated in C++), every scene that parses is
//      dumped (main.rs:60-87 format) and freed;
//   2. the example scene parsed, given a small size, rendered by the oracle on several threads (accumulator, counters,
//      RGBA8), written as PNG and as a checkpoint, read back.
// usage: san_driver <scene.example.json> <N> <tmp dir>; exit status 0 = no finding (a finding aborts the process).
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hippt.h"

// host_scene.cpp reports errors through libhippt's hippt_set_error (hippt_api.hip, GPU translation unit): same contract here
static char g_err[512];
int hippt_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
extern "C" {
int orc_render_accum_ex(const hippt_scene *scene, const hippt_options *opts, float *out, uint64_t *counters, int iterative, int n_threads);
int orc_render_rgba8(const hippt_scene *scene, const hippt_options *opts, uint8_t *out_rgba);
}

static uint64_t g_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd(uint32_t n) {  // splitmix64
    uint64_t z = (g_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return n ? (uint32_t)((z ^ (z >> 31)) % n) : 0u;
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::string base;
    char buf[4096];
    for (size_t n; (n = fread(buf, 1, sizeof buf, f)) > 0;) base.append(buf, n);
    fclose(f);
    const long N = strtol(argv[2], nullptr, 10);
    const std::string tmp = argv[3];
    const char *tokens[] = {"{", "}", "[", "]", ",", ":", "\"", "null", "true", "-", "1e999", "1e-999", "NaN", "\\u00", "\\",
                            "0.", "-0", "99999999999999999999", "\"Sphere\"", "\"Plane\"", "\"material\"", "\"objects\""};
    long ok = 0, bad = 0;
    std::vector<char> dump(1 << 16);
    for (long it = 0; it < N; ++it) {
        std::string s = base;
        for (uint32_t m = 1 + rnd(4); m > 0; --m) {
            const size_t pos = rnd((uint32_t)s.size() + 1);
            switch (rnd(5)) {
                case 0: if (!s.empty()) s.erase(pos % s.size(), 1 + rnd(8)); break;
                case 1: s.insert(pos, tokens[rnd(sizeof tokens / sizeof *tokens)]); break;
                case 2: if (!s.empty()) s[pos % s.size()] = (char)rnd(256); break;
                case 3: s.erase(pos); break;
                default: if (!s.empty()) { const size_t a = rnd((uint32_t)s.size()); s.insert(pos, s.substr(a, rnd(64))); }
            }
        }
        hippt_scene_file *sf = nullptr;
        if (hippt_scene_parse(s.data(), s.size(), &sf) == 0 && sf) {
            hippt_scene_dump(sf, dump.data(), dump.size());
            hippt_scene_free(sf);
            ++ok;
        } else {
            ++bad;
        }
    }
    for (const char *opener : {"[", "{\"a\":", "{\"camera\":["}) {  // million-deep nesting
        std::string deep;
        for (int i = 0; i < 1000000; ++i) deep += opener;
        hippt_scene_file *sf = nullptr;
        if (hippt_scene_parse(deep.data(), deep.size(), &sf) == 0) return 3;
    }
    hippt_scene_file *sf = nullptr;
    if (hippt_scene_parse(base.data(), base.size(), &sf) != 0) return 4;
    if (hippt_scene_set_render(sf, 40, 30, 3) != 0) return 5;
    const hippt_scene *scp = hippt_scene_get(sf);
    if (!scp) return 6;
    const hippt_scene sc = *scp;
    std::vector<float> acc((size_t)sc.width * sc.height * 3);
    std::vector<uint64_t> cn(HIPPT_N_COUNTERS);
    hippt_options o{};
    o.flags = HIPPT_FLAG_COUNTERS;
    for (int iterative = 0; iterative < 2; ++iterative)
        if (orc_render_accum_ex(&sc, &o, acc.data(), cn.data(), iterative, 3) != 0) return 7;
    std::vector<uint8_t> rgba((size_t)sc.width * sc.height * 4);
    if (orc_render_rgba8(&sc, &o, rgba.data()) != 0) return 8;
    const std::string png = tmp + "/san.png", ck = tmp + "/san.acc";
    if (hippt_write_png_rgba8(png.c_str(), rgba.data(), sc.width, sc.height) != 0) return 9;
    if (hippt_checkpoint_save(ck.c_str(), acc.data(), sc.width, sc.height, 3, 0) != 0) return 10;
    uint32_t done = 0, seed = 1;
    std::vector<float> back(acc.size());
    if (hippt_checkpoint_load(ck.c_str(), back.data(), sc.width, sc.height, &done, &seed) != 0 || done != 3 || back != acc) return 11;
    if (hippt_checkpoint_load(ck.c_str(), back.data(), sc.width + 1, sc.height, &done, &seed) == 0) return 12;  // wrong size: rejected
    char name[256];
    hippt_render_image_name(sc.width, sc.height, 3, 0.02f, 5.0f, name, sizeof name);
    hippt_scene_free(sf);
    printf("sanitizer driver: %ld mutants (%ld parsed, %ld rejected), oracle render %ux%u x 3 spp x 2 forms, png + checkpoint round trip: clean\n",
           N, ok, bad, sc.width, sc.height);
    return 0;
}
