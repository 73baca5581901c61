// host_scene.cpp — host-side plumbing either side of the hot path:
//   scene.json loader          (reference src/scene.rs:6-158)
//   camera basis               (src/main.rs:41-51)
//   autofocus                  (src/renderer.rs:204-234)
//   output file name           (src/renderer.rs:156-161)
//   RGBA8 PNG writer           (what image::RgbaImage::save does, src/main.rs:104)
// Plain C++ (no HIP). Built with -ffp-contract=off so the camera basis and the
// autofocus distance round exactly as the reference's f32 arithmetic does.
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <random>
#include <string>
#include <utility>
#include <vector>

#include "../../include/hippt.h"
#include "hippt_internal.h"

namespace {

// ---- minimal JSON (RFC 8259) reader ------------------------------------------------
struct JValue;
using JObject = std::vector<std::pair<std::string, JValue>>;  // keeps file order
struct JValue {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0.0;
    bool num_is_integer = false;  // no fraction / exponent in the token
    std::string str;
    std::vector<JValue> arr;
    JObject obj;
    const JValue *get(const char *key) const {
        // serde keeps the LAST duplicate of a struct field as an error, of a map key as the winner;
        // scanning backwards gives the map behaviour, which is the lenient one.
        for (size_t i = obj.size(); i-- > 0;)
            if (obj[i].first == key) return &obj[i].second;
        return nullptr;
    }
};

struct JParser {
    const char *p, *end;
    std::string err;
    bool fail(const std::string &m) {
        if (err.empty()) err = m + " at byte " + std::to_string((long)(p - start));
        return false;
    }
    const char *start;
    JParser(const char *s, size_t n) : p(s), end(s + n), start(s) {}
    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool parse_value(JValue &v, int depth) {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end of input");
        char c = *p;
        if (c == '{') return parse_object(v, depth);
        if (c == '[') return parse_array(v, depth);
        if (c == '"') {
            v.kind = JValue::Str;
            return parse_string(v.str);
        }
        if (c == 't' && end - p >= 4 && !std::strncmp(p, "true", 4)) {
            v.kind = JValue::Bool, v.b = true, p += 4;
            return true;
        }
        if (c == 'f' && end - p >= 5 && !std::strncmp(p, "false", 5)) {
            v.kind = JValue::Bool, v.b = false, p += 5;
            return true;
        }
        if (c == 'n' && end - p >= 4 && !std::strncmp(p, "null", 4)) {
            v.kind = JValue::Null, p += 4;
            return true;
        }
        if (c == '-' || (c >= '0' && c <= '9')) return parse_number(v);
        return fail("unexpected character");
    }
    bool parse_number(JValue &v) {
        const char *s = p;
        if (p < end && *p == '-') ++p;
        if (p >= end || !std::isdigit((unsigned char)*p)) return fail("bad number");
        if (*p == '0') ++p;
        else
            while (p < end && std::isdigit((unsigned char)*p)) ++p;
        bool integer = true;
        if (p < end && *p == '.') {
            integer = false;
            ++p;
            if (p >= end || !std::isdigit((unsigned char)*p)) return fail("bad fraction");
            while (p < end && std::isdigit((unsigned char)*p)) ++p;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            integer = false;
            ++p;
            if (p < end && (*p == '+' || *p == '-')) ++p;
            if (p >= end || !std::isdigit((unsigned char)*p)) return fail("bad exponent");
            while (p < end && std::isdigit((unsigned char)*p)) ++p;
        }
        std::string tok(s, p);
        v.kind = JValue::Num;
        v.num = std::strtod(tok.c_str(), nullptr);  // decimal -> f64, as serde_json
        v.num_is_integer = integer;
        return true;
    }
    static void put_utf8(std::string &out, unsigned cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) out += (char)(0xC0 | (cp >> 6)), out += (char)(0x80 | (cp & 0x3F));
        else if (cp < 0x10000)
            out += (char)(0xE0 | (cp >> 12)), out += (char)(0x80 | ((cp >> 6) & 0x3F)), out += (char)(0x80 | (cp & 0x3F));
        else
            out += (char)(0xF0 | (cp >> 18)), out += (char)(0x80 | ((cp >> 12) & 0x3F)),
                out += (char)(0x80 | ((cp >> 6) & 0x3F)), out += (char)(0x80 | (cp & 0x3F));
    }
    bool hex4(unsigned &v) {
        if (end - p < 4) return fail("bad \\u escape");
        v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0';
            else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else return fail("bad \\u escape");
        }
        return true;
    }
    bool parse_string(std::string &out) {
        ++p;  // opening quote
        out.clear();
        while (p < end) {
            unsigned char c = (unsigned char)*p++;
            if (c == '"') return true;
            if (c < 0x20) return fail("control character in string");
            if (c != '\\') {
                out += (char)c;
                continue;
            }
            if (p >= end) break;
            char e = *p++;
            switch (e) {
                case '"': out += '"'; break;
                case '\\': out += '\\'; break;
                case '/': out += '/'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'n': out += '\n'; break;
                case 'r': out += '\r'; break;
                case 't': out += '\t'; break;
                case 'u': {
                    unsigned cp;
                    if (!hex4(cp)) return false;
                    if (cp >= 0xD800 && cp <= 0xDBFF && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                        p += 2;
                        unsigned lo;
                        if (!hex4(lo)) return false;
                        if (lo >= 0xDC00 && lo <= 0xDFFF) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        else return fail("bad surrogate pair");
                    }
                    put_utf8(out, cp);
                    break;
                }
                default: return fail("bad escape");
            }
        }
        return fail("unterminated string");
    }
    bool parse_array(JValue &v, int depth) {
        v.kind = JValue::Arr;
        ++p;
        ws();
        if (p < end && *p == ']') {
            ++p;
            return true;
        }
        for (;;) {
            v.arr.emplace_back();
            if (!parse_value(v.arr.back(), depth + 1)) return false;
            ws();
            if (p >= end) return fail("unterminated array");
            if (*p == ',') {
                ++p;
                continue;
            }
            if (*p == ']') {
                ++p;
                return true;
            }
            return fail("expected ',' or ']'");
        }
    }
    bool parse_object(JValue &v, int depth) {
        v.kind = JValue::Obj;
        ++p;
        ws();
        if (p < end && *p == '}') {
            ++p;
            return true;
        }
        for (;;) {
            ws();
            if (p >= end || *p != '"') return fail("expected object key");
            std::string key;
            if (!parse_string(key)) return false;
            ws();
            if (p >= end || *p != ':') return fail("expected ':'");
            ++p;
            v.obj.emplace_back(std::move(key), JValue());
            if (!parse_value(v.obj.back().second, depth + 1)) return false;
            ws();
            if (p >= end) return fail("unterminated object");
            if (*p == ',') {
                ++p;
                continue;
            }
            if (*p == '}') {
                ++p;
                return true;
            }
            return fail("expected ',' or '}'");
        }
    }
};

// ---- f32 vector helpers in the reference's evaluation order (algebra.rs:12-46) ---------
struct V3 {
    float x, y, z;
};
V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
V3 add(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
V3 scale(V3 a, float f) { return V3{a.x * f, a.y * f, a.z * f}; }
float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
V3 normalize(V3 a) { return scale(a, 1.0f / std::sqrt(dot(a, a))); }
V3 ld(const float *p) { return V3{p[0], p[1], p[2]}; }
void st(float *p, V3 v) { p[0] = v.x, p[1] = v.y, p[2] = v.z; }

// ---- typed extraction with serde-like errors ---------------------------------------------
struct Extract {
    std::string err;
    bool fail(const std::string &m) {
        if (err.empty()) err = m;
        return false;
    }
    const JValue *field(const JValue &o, const char *key, const char *ctx, bool required = true) {
        if (o.kind != JValue::Obj) {
            fail(std::string(ctx) + ": expected an object");
            return nullptr;
        }
        const JValue *v = o.get(key);
        if (!v && required) fail(std::string(ctx) + ": missing field `" + key + "`");
        return v;
    }
    bool f32(const JValue &o, const char *key, const char *ctx, float &out, bool required = true) {
        const JValue *v = field(o, key, ctx, required);
        if (!v) return !required && err.empty();
        if (v->kind != JValue::Num) return fail(std::string(ctx) + "." + key + ": expected a number");
        out = (float)v->num;  // f64 -> f32 narrowing, as serde
        return true;
    }
    bool u32(const JValue &o, const char *key, const char *ctx, uint32_t &out) {
        const JValue *v = field(o, key, ctx);
        if (!v) return false;
        if (v->kind != JValue::Num || !v->num_is_integer || v->num < 0 || v->num > 4294967295.0)
            return fail(std::string(ctx) + "." + key + ": expected an unsigned 32-bit integer");
        out = (uint32_t)v->num;
        return true;
    }
    bool vec3(const JValue &o, const char *key, const char *ctx, float out[3]) {
        const JValue *v = field(o, key, ctx);
        if (!v) return false;
        if (v->kind != JValue::Arr || v->arr.size() != 3) return fail(std::string(ctx) + "." + key + ": expected an array of 3 numbers");
        for (int i = 0; i < 3; ++i) {
            if (v->arr[i].kind != JValue::Num) return fail(std::string(ctx) + "." + key + ": expected an array of 3 numbers");
            out[i] = (float)v->arr[i].num;
        }
        return true;
    }
    bool str(const JValue &o, const char *key, const char *ctx, std::string &out) {
        const JValue *v = field(o, key, ctx);
        if (!v) return false;
        if (v->kind != JValue::Str) return fail(std::string(ctx) + "." + key + ": expected a string");
        out = v->str;
        return true;
    }
    bool boolean(const JValue &o, const char *key, const char *ctx, bool &out) {  // #[serde(default)]
        const JValue *v = field(o, key, ctx, false);
        if (!v) return err.empty();
        if (v->kind != JValue::Bool) return fail(std::string(ctx) + "." + key + ": expected a boolean");
        out = v->b;
        return true;
    }
};

}  // namespace

// Owns every table a hippt_scene points to.
struct hippt_scene_file {
    hippt_scene scene{};
    std::vector<hippt_material> materials;
    std::vector<hippt_sphere> spheres;
    std::vector<hippt_plane> planes;
    std::vector<hippt_light> lights;
    std::vector<std::string> names;  // object names in scene order (sphere.rs:10, plane.rs:9; shown by main.rs:73,77)
    // raw camera inputs, kept so that a render-size override can re-derive the basis (and for main.rs:61-64's dump)
    float cam_pos[3], cam_look_at[3], cam_up[3], cam_fov, cam_aperture;
    void rebind() {
        scene.n_materials = (uint32_t)materials.size();
        scene.n_spheres = (uint32_t)spheres.size();
        scene.n_planes = (uint32_t)planes.size();
        scene.n_lights = (uint32_t)lights.size();
        scene.materials = materials.data();
        scene.spheres = spheres.data();
        scene.planes = planes.data();
        scene.lights = lights.data();
    }
    void derive_camera() {
        hippt_camera_basis(cam_pos, cam_look_at, cam_up, cam_fov, scene.width, scene.height, &scene.camera);
        scene.camera.aperture = cam_aperture;
        scene.camera.focus = hippt_autofocus(&scene);
    }
};

extern "C" {

// src/main.rs:41-51
void hippt_camera_basis(const float pos[3], const float look_at[3], const float up[3], float fov_deg, uint32_t width,
                        uint32_t height, hippt_camera *out) {
    const float pi = 3.14159265358979323846f;
    float fov_rad = fov_deg * (pi / 180.0f);  // f32::to_radians
    out->aspect = (float)width / (float)height;
    out->scale = std::tan(fov_rad * 0.5f);
    V3 forward = normalize(sub(ld(look_at), ld(pos)));
    V3 right = normalize(cross(ld(up), forward));
    V3 real_up = normalize(cross(forward, right));
    std::memcpy(out->pos, pos, 12);
    st(out->right, right);
    st(out->up, real_up);
    st(out->forward, forward);
}

// src/renderer.rs:204-234. Probe rays use the reference's hit tests restricted
// to the in_focus objects (sphere.rs:26-34, plane.rs:25-52), closest first-min.
float hippt_autofocus(const hippt_scene *sc) {
    bool any_focus = false;
    for (uint32_t i = 0; i < sc->n_spheres; ++i) any_focus |= sc->spheres[i].in_focus != 0;
    for (uint32_t i = 0; i < sc->n_planes; ++i) any_focus |= sc->planes[i].in_focus != 0;
    if (!any_focus) return 5.0f;
    const hippt_camera &cam = sc->camera;
    uint32_t w = sc->width, h = sc->height;
    V3 ro = ld(cam.pos);
    float sum = 0.0f;
    uint32_t cnt = 0;
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) {
            float px = (float)(w / 2) + ((float)i - 2.0f);
            float py = (float)(h / 2) + ((float)j - 2.0f);
            float u = (px / (float)w - 0.5f) * 2.0f * cam.aspect * cam.scale;
            float v = -((py / (float)h - 0.5f) * 2.0f * cam.scale);
            V3 rd = normalize(add(add(scale(ld(cam.right), u), scale(ld(cam.up), v)), ld(cam.forward)));
            float best = INFINITY;
            uint32_t best_order = 0xffffffffu;
            auto consider = [&](float t, uint32_t order) {
                if (t < best || (t == best && order < best_order)) best = t, best_order = order;
            };
            for (uint32_t k = 0; k < sc->n_spheres; ++k) {
                const hippt_sphere &s = sc->spheres[k];
                if (!s.in_focus) continue;
                V3 oc = sub(ro, ld(s.center));
                float a = dot(rd, rd), b = 2.0f * dot(oc, rd), c = dot(oc, oc) - s.radius * s.radius;
                float disc = b * b - 4.0f * a * c;
                if (disc < 0.0f) continue;
                float t = (-b - std::sqrt(disc)) / (2.0f * a);
                if (t <= 0.0f) continue;
                consider(t, s.order);
            }
            for (uint32_t k = 0; k < sc->n_planes; ++k) {
                const hippt_plane &p = sc->planes[k];
                if (!p.in_focus) continue;
                float denom = dot(ld(p.normal), rd);
                if (std::fabs(denom) < 1e-6f) continue;
                float t = dot(sub(ld(p.point), ro), ld(p.normal)) / denom;
                if (!std::isfinite(t) || t <= 1e-4f) continue;
                V3 d = sub(add(ro, scale(rd, t)), ld(p.point));
                if (std::fabs(dot(d, ld(p.u))) > dot(ld(p.u), ld(p.u))) continue;
                if (std::fabs(dot(d, ld(p.v))) > dot(ld(p.v), ld(p.v))) continue;
                consider(t, p.order);
            }
            if (best_order != 0xffffffffu) sum += best, ++cnt;
        }
    return cnt == 0 ? 10.0f : sum / (float)cnt;
}

int hippt_scene_parse(const char *json_text, size_t len, hippt_scene_file **out) {
    if (!json_text || !out) return hippt_set_error(HIPPT_ERR_INVALID, "NULL argument");
    JParser jp(json_text, len);
    JValue root;
    if (!jp.parse_value(root, 0)) return hippt_set_error(HIPPT_ERR_IO, "json parse: %s", jp.err.c_str());
    jp.ws();
    if (jp.p != jp.end) return hippt_set_error(HIPPT_ERR_IO, "json parse: trailing characters at byte %ld", (long)(jp.p - jp.start));

    Extract ex;
    auto f = std::make_unique<hippt_scene_file>();
    auto bad = [&]() { return hippt_set_error(HIPPT_ERR_IO, "json parse: %s", ex.err.c_str()); };

    // scene.rs:78-84: all five sections are required
    const JValue *cam = ex.field(root, "camera", "scene");
    const JValue *ren = ex.field(root, "render", "scene");
    const JValue *mats = ex.field(root, "materials", "scene");
    const JValue *objs = ex.field(root, "objects", "scene");
    const JValue *lights = ex.field(root, "lights", "scene");
    if (!cam || !ren || !mats || !objs || !lights) return bad();

    // scene.rs:7-16
    if (!ex.vec3(*cam, "pos", "camera", f->cam_pos) || !ex.vec3(*cam, "look_at", "camera", f->cam_look_at) ||
        !ex.vec3(*cam, "up", "camera", f->cam_up) || !ex.f32(*cam, "fov", "camera", f->cam_fov) ||
        !ex.f32(*cam, "aperture", "camera", f->cam_aperture))
        return bad();
    // scene.rs:18
    if (!ex.u32(*ren, "width", "render", f->scene.width) || !ex.u32(*ren, "height", "render", f->scene.height) ||
        !ex.u32(*ren, "samples", "render", f->scene.samples))
        return bad();

    // scene.rs:20-29, 99-110
    if (mats->kind != JValue::Obj) {
        ex.fail("materials: expected a map");
        return bad();
    }
    std::map<std::string, uint32_t> mat_index;
    for (const auto &kv : mats->obj) {
        hippt_material m{};
        const JValue &mv = kv.second;
        std::string ctx = "materials." + kv.first;
        float rgb[3];
        if (!ex.vec3(mv, "rgb", ctx.c_str(), rgb) || !ex.f32(mv, "metallic", ctx.c_str(), m.metallic) ||
            !ex.f32(mv, "roughness", ctx.c_str(), m.roughness) || !ex.f32(mv, "ior", ctx.c_str(), m.ior) ||
            !ex.f32(mv, "volume_density", ctx.c_str(), m.volume_density, false) ||
            !ex.f32(mv, "volume_anisotropy", ctx.c_str(), m.volume_anisotropy, false))
            return bad();
        std::memcpy(m.rgb, rgb, 12);
        auto it = mat_index.find(kv.first);
        if (it != mat_index.end()) f->materials[it->second] = m;  // later duplicate key wins (HashMap insert)
        else {
            mat_index[kv.first] = (uint32_t)f->materials.size();
            f->materials.push_back(m);
        }
    }
    // scene.rs:112-120: magenta default for unknown material names
    const uint32_t default_mat = (uint32_t)f->materials.size();
    f->materials.push_back(hippt_material{{1.0f, 0.0f, 1.0f}, 0.0f, 1.0f, 1.0f, 0.0f, 0.0f});
    auto lookup = [&](const std::string &name) {
        auto it = mat_index.find(name);
        return it == mat_index.end() ? default_mat : it->second;
    };

    // scene.rs:32-61, 124-151: untagged enum, Sphere variant tried first
    if (objs->kind != JValue::Arr) {
        ex.fail("objects: expected an array");
        return bad();
    }
    uint32_t order = 0;
    for (const JValue &ov : objs->arr) {
        std::string ctx = "objects[" + std::to_string(order) + "]";
        const JValue *sv = ov.kind == JValue::Obj ? ov.get("sphere") : nullptr;
        const JValue *pv = ov.kind == JValue::Obj ? ov.get("plane") : nullptr;
        std::string name, mat;
        bool in_focus = false;
        bool done = false;
        if (sv) {
            Extract es;
            hippt_sphere s{};
            std::string c2 = ctx + ".sphere";
            if (es.str(*sv, "name", c2.c_str(), name) && es.vec3(*sv, "center", c2.c_str(), s.center) &&
                es.f32(*sv, "radius", c2.c_str(), s.radius) && es.str(*sv, "mat", c2.c_str(), mat) &&
                es.boolean(*sv, "in_focus", c2.c_str(), in_focus)) {
                s.material = lookup(mat);
                s.order = order;
                s.in_focus = in_focus ? 1u : 0u;
                f->spheres.push_back(s);
                f->names.push_back(name);
                done = true;
            } else if (!pv) {
                ex.err = es.err;
                return bad();
            }
        }
        if (!done && pv) {
            hippt_plane p{};
            std::string c2 = ctx + ".plane";
            if (!ex.str(*pv, "name", c2.c_str(), name) || !ex.vec3(*pv, "point", c2.c_str(), p.point) ||
                !ex.vec3(*pv, "u", c2.c_str(), p.u) || !ex.vec3(*pv, "v", c2.c_str(), p.v) ||
                !ex.str(*pv, "mat", c2.c_str(), mat) || !ex.boolean(*pv, "in_focus", c2.c_str(), in_focus))
                return bad();
            st(p.normal, normalize(cross(ld(p.u), ld(p.v))));  // scene.rs:139
            p.material = lookup(mat);
            p.order = order;
            p.in_focus = in_focus ? 1u : 0u;
            f->planes.push_back(p);
            f->names.push_back(name);
            done = true;
        }
        if (!done) {
            ex.fail(ctx + ": data did not match any variant of untagged enum ObjectJson");
            return bad();
        }
        ++order;
    }

    // scene.rs:64-74, 153-155
    if (lights->kind != JValue::Arr) {
        ex.fail("lights: expected an array");
        return bad();
    }
    for (size_t i = 0; i < lights->arr.size(); ++i) {
        hippt_light l{};
        std::string ctx = "lights[" + std::to_string(i) + "]";
        const JValue &lv = lights->arr[i];
        if (!ex.vec3(lv, "pos", ctx.c_str(), l.pos) || !ex.vec3(lv, "u", ctx.c_str(), l.u) ||
            !ex.vec3(lv, "v", ctx.c_str(), l.v) || !ex.vec3(lv, "intensity", ctx.c_str(), l.intensity))
            return bad();
        f->lights.push_back(l);
    }

    f->rebind();
    f->derive_camera();
    *out = f.release();
    return HIPPT_OK;
}

int hippt_scene_load(const char *path, hippt_scene_file **out) {
    if (!path || !out) return hippt_set_error(HIPPT_ERR_INVALID, "NULL argument");
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return hippt_set_error(HIPPT_ERR_IO, "scene file: %s: %s", path, std::strerror(errno));
    std::string data;
    char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, fp)) > 0) data.append(buf, n);
    std::fclose(fp);
    return hippt_scene_parse(data.data(), data.size(), out);
}

const hippt_scene *hippt_scene_get(const hippt_scene_file *f) { return f ? &f->scene : nullptr; }

int hippt_scene_set_render(hippt_scene_file *f, uint32_t width, uint32_t height, uint32_t samples) {
    if (!f || width == 0 || height == 0) return hippt_set_error(HIPPT_ERR_INVALID, "bad render override");
    f->scene.width = width;
    f->scene.height = height;
    f->scene.samples = samples;
    f->derive_camera();
    return HIPPT_OK;
}

void hippt_scene_free(hippt_scene_file *f) { delete f; }

}  // extern "C"

// ---- src/main.rs:60-87: the debug dump ------------------------------------------------
namespace {
// Rust's `{:?}` of an f32 (core::fmt::float::float_to_general_debug): the shortest digits that read back as the
// same f32; decimal notation with at least one fractional digit ("1.0") for 1e-4 <= |x| < 1e16 and for zero,
// otherwise `d.ddde-7` style exponent notation.
std::string rust_debug_f32(float x) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x < 0 ? "-inf" : "inf";
    char buf[64];
    int prec = 0;
    for (; prec < 9; ++prec) {
        std::snprintf(buf, sizeof buf, "%.*e", prec, (double)x);
        if (std::strtof(buf, nullptr) == x) break;
    }
    // buf = [-]d[.ddd]e[+-]XX
    std::string t = buf;
    const bool neg = t[0] == '-';
    if (neg) t.erase(0, 1);
    const size_t epos = t.find('e');
    int e10 = std::atoi(t.c_str() + epos + 1);
    std::string digits;
    for (char c : t.substr(0, epos))
        if (c != '.') digits.push_back(c);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    const float ax = std::fabs(x);
    std::string out = neg ? "-" : "";
    if (ax != 0.0f && (ax >= 1e16f || ax < 1e-4f)) {
        out += digits.substr(0, 1);
        if (digits.size() > 1) out += "." + digits.substr(1);
        out += "e" + std::to_string(e10);
        return out;
    }
    if (ax == 0.0f) return out + "0.0";
    if (e10 >= 0) {
        std::string ip = digits.substr(0, std::min(digits.size(), (size_t)e10 + 1));
        ip.append((size_t)e10 + 1 - ip.size(), '0');
        std::string fp = digits.size() > (size_t)e10 + 1 ? digits.substr((size_t)e10 + 1) : "0";
        return out + ip + "." + fp;
    }
    return out + "0." + std::string((size_t)(-e10 - 1), '0') + digits;
}
std::string rust_debug_vec3(const float v[3]) {
    return "Vec3(" + rust_debug_f32(v[0]) + ", " + rust_debug_f32(v[1]) + ", " + rust_debug_f32(v[2]) + ")";
}
std::string fixed(float x, int prec) {
    char b[64];
    std::snprintf(b, sizeof b, "%.*f", prec, (double)x);
    return b;
}
}  // namespace

extern "C" {

int hippt_scene_dump(const hippt_scene_file *f, char *out, size_t cap) {
    if (!f) return -1;
    const hippt_scene &sc = f->scene;
    std::string s;
    s += "=== CAMERA INFO ===\n";
    s += " position : " + rust_debug_vec3(f->cam_pos) + "\n";
    s += " look_at  : " + rust_debug_vec3(f->cam_look_at) + "\n";
    s += " up       : " + rust_debug_vec3(f->cam_up) + "\n";
    s += " fov (\xC2\xB0)  : " + fixed(f->cam_fov, 2) + "\n";
    s += " aspect   : " + fixed(sc.camera.aspect, 4) + "\n";
    s += " aperture : " + fixed(f->cam_aperture, 4) + "\n";
    s += " autofocus: " + fixed(sc.camera.focus, 4) + "\n";
    const uint32_t n_obj = sc.n_spheres + sc.n_planes;
    s += "\n=== OBJECTS (" + std::to_string(n_obj) + ") ===\n";
    for (uint32_t o = 0; o < n_obj; ++o) {
        const std::string name = o < f->names.size() ? f->names[o] : std::string();
        for (uint32_t i = 0; i < sc.n_spheres; ++i)
            if (sc.spheres[i].order == o) {
                const hippt_sphere &sp = sc.spheres[i];
                s += " [" + std::to_string(o) + "] Sphere '" + name + "' { center: " + rust_debug_vec3(sp.center) +
                     ", radius: " + fixed(sp.radius, 4) + ", mat_color: " + rust_debug_vec3(sc.materials[sp.material].rgb) + " }\n";
            }
        for (uint32_t i = 0; i < sc.n_planes; ++i)
            if (sc.planes[i].order == o) {
                const hippt_plane &pl = sc.planes[i];
                s += " [" + std::to_string(o) + "] Plane '" + name + "' { point: " + rust_debug_vec3(pl.point) +
                     ", normal: " + rust_debug_vec3(pl.normal) + ", mat_color: " + rust_debug_vec3(sc.materials[pl.material].rgb) + " }\n";
            }
    }
    s += "\n=== LIGHTS (" + std::to_string(sc.n_lights) + ") ===\n";
    for (uint32_t i = 0; i < sc.n_lights; ++i) {
        const hippt_light &l = sc.lights[i];
        s += " [" + std::to_string(i) + "] Light { pos: " + rust_debug_vec3(l.pos) + ", u: " + rust_debug_vec3(l.u) +
             ", v: " + rust_debug_vec3(l.v) + ", intensity: " + rust_debug_vec3(l.intensity) + " }\n";
    }
    if (out && cap) {
        const size_t n = std::min(cap - 1, s.size());
        std::memcpy(out, s.data(), n);
        out[n] = 0;
    }
    return (int)s.size();
}

// src/renderer.rs:156-161
int hippt_render_image_name(uint32_t w, uint32_t h, uint32_t s, float ap, float f, char *out, size_t cap) {
    static const char alnum[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789";
    std::random_device rd;
    char suf[7];
    for (int i = 0; i < 6; ++i) suf[i] = alnum[rd() % 62u];
    suf[6] = 0;
    return std::snprintf(out, cap, "renders/render_%ux%u_s%u_ap%.2f_f%.1f_%s.png", w, h, s, (double)ap, (double)f, suf);
}

// Checkpoint of a progressive render (include/hippt.h): header + raw f32 sums.
namespace {
struct CkptHeader {
    char magic[8];
    uint32_t width, height, samples_done, seed, zero0, zero1;
};
static_assert(sizeof(CkptHeader) == 32, "checkpoint header is 32 bytes");
}  // namespace

int hippt_checkpoint_save(const char *path, const float *accum, uint32_t width, uint32_t height, uint32_t samples_done,
                          uint32_t seed) {
    if (!path || !accum || !width || !height) return hippt_set_error(HIPPT_ERR_INVALID, "bad checkpoint arguments");
    std::string tmp = std::string(path) + ".tmp";
    FILE *fp = std::fopen(tmp.c_str(), "wb");
    if (!fp) return hippt_set_error(HIPPT_ERR_IO, "%s: %s", tmp.c_str(), std::strerror(errno));
    CkptHeader h{{'H', 'I', 'P', 'P', 'T', 'A', 'C', '1'}, width, height, samples_done, seed, 0, 0};
    size_t n = (size_t)width * height * 3;
    bool ok = std::fwrite(&h, sizeof h, 1, fp) == 1 && std::fwrite(accum, sizeof(float), n, fp) == n;
    ok = (std::fclose(fp) == 0) && ok;
    if (!ok) {
        std::remove(tmp.c_str());
        return hippt_set_error(HIPPT_ERR_IO, "%s: short write", tmp.c_str());
    }
    if (std::rename(tmp.c_str(), path) != 0) return hippt_set_error(HIPPT_ERR_IO, "%s: %s", path, std::strerror(errno));
    return HIPPT_OK;
}

int hippt_checkpoint_load(const char *path, float *accum, uint32_t width, uint32_t height, uint32_t *samples_done,
                          uint32_t *seed) {
    if (!path || !accum || !width || !height) return hippt_set_error(HIPPT_ERR_INVALID, "bad checkpoint arguments");
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return hippt_set_error(HIPPT_ERR_IO, "%s: %s", path, std::strerror(errno));
    CkptHeader h{};
    size_t n = (size_t)width * height * 3;
    int rc = HIPPT_OK;
    if (std::fread(&h, sizeof h, 1, fp) != 1 || std::memcmp(h.magic, "HIPPTAC1", 8) != 0)
        rc = hippt_set_error(HIPPT_ERR_INVALID, "%s: not a hippt checkpoint", path);
    else if (h.width != width || h.height != height)
        rc = hippt_set_error(HIPPT_ERR_INVALID, "%s: checkpoint is %ux%u, render is %ux%u", path, h.width, h.height, width, height);
    else if (std::fread(accum, sizeof(float), n, fp) != n || std::fgetc(fp) != EOF)
        rc = hippt_set_error(HIPPT_ERR_INVALID, "%s: size does not match %ux%u", path, width, height);
    std::fclose(fp);
    if (rc) return rc;
    if (samples_done) *samples_done = h.samples_done;
    if (seed) *seed = h.seed;
    return HIPPT_OK;
}

// 8-bit RGBA, non-interlaced PNG: signature, IHDR, one IDAT (zlib), IEND.
int hippt_write_png_rgba8(const char *path, const uint8_t *rgba, uint32_t w, uint32_t h) {
    if (!path || !rgba || !w || !h) return hippt_set_error(HIPPT_ERR_INVALID, "bad PNG arguments");
    size_t stride = (size_t)w * 4;
    std::vector<uint8_t> raw((stride + 1) * h);
    for (uint32_t y = 0; y < h; ++y) {
        raw[(stride + 1) * y] = 0;  // filter: None
        std::memcpy(&raw[(stride + 1) * y + 1], rgba + stride * y, stride);
    }
    uLongf zcap = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zcap);
    if (compress2(z.data(), &zcap, raw.data(), (uLong)raw.size(), 6) != Z_OK) return hippt_set_error(HIPPT_ERR_IO, "zlib compress failed");
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return hippt_set_error(HIPPT_ERR_IO, "%s: %s", path, std::strerror(errno));
    auto be32 = [](uint8_t *p, uint32_t v) { p[0] = v >> 24, p[1] = v >> 16, p[2] = v >> 8, p[3] = v; };
    bool ok = true;
    auto chunk = [&](const char *type, const uint8_t *data, uint32_t n) {
        uint8_t hdr[8];
        be32(hdr, n);
        std::memcpy(hdr + 4, type, 4);
        uLong crc = crc32(0L, hdr + 4, 4);
        if (n) crc = crc32(crc, data, n);
        uint8_t tail[4];
        be32(tail, (uint32_t)crc);
        ok &= std::fwrite(hdr, 1, 8, fp) == 8;
        if (n) ok &= std::fwrite(data, 1, n, fp) == n;
        ok &= std::fwrite(tail, 1, 4, fp) == 4;
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    ok &= std::fwrite(sig, 1, 8, fp) == 8;
    uint8_t ihdr[13];
    be32(ihdr, w);
    be32(ihdr + 4, h);
    ihdr[8] = 8, ihdr[9] = 6, ihdr[10] = 0, ihdr[11] = 0, ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t)zcap);
    chunk("IEND", nullptr, 0);
    ok &= std::fclose(fp) == 0;
    if (!ok) return hippt_set_error(HIPPT_ERR_IO, "%s: write failed", path);
    return HIPPT_OK;
}

}  // extern "C"
