"""C-ABI surface and host-side plumbing (no GPU compute): exports, scene.json loader
(src/scene.rs), file name (renderer.rs:156-161), PNG writer, error behaviour."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLE = os.path.join(ROOT, "scenes", "scene.example.json")


def _has_gpu():
    return os.path.exists("/dev/kfd")


def test_library_exports_every_declared_symbol():
    from pathtracer_amd import _abi
    lib = _abi.load_library()
    header = open(os.path.join(ROOT, "include", "hippt.h")).read()
    declared = set(re.findall(r"\b(hippt_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    for sym in declared:
        assert hasattr(lib, sym), f"libhippt.so lacks {sym}"
    assert declared == set(_abi.EXPORTED_SYMBOLS)


def test_struct_layouts_match_header():
    from pathtracer_amd import _abi
    assert C.sizeof(_abi.Material) == 32 and C.sizeof(_abi.Sphere) == 28 and C.sizeof(_abi.Plane) == 60
    assert C.sizeof(_abi.Light) == 48 and C.sizeof(_abi.Camera) == 64 and C.sizeof(_abi.Options) == 24
    assert C.sizeof(_abi.SceneC) == 64 + 8 * 4 + 4 * 8


def test_example_scene_loads():
    from pathtracer_amd import Scene
    s = Scene.load(EXAMPLE)
    assert (s.width, s.height, s.samples) == (800, 600, 256)
    assert len(s.spheres) == 3 and len(s.planes) == 4 and len(s.lights) == 2
    assert len(s.materials) == 5 + 1                       # + magenta default (scene.rs:112-120)
    assert list(s.planes["order"]) == [0, 4, 5, 6] and list(s.spheres["order"]) == [1, 2, 3]
    assert s.camera.focus == 5.0 and np.isclose(s.camera.aperture, 0.02)
    assert np.isclose(s.camera.aspect, 800 / 600) and np.isclose(s.camera.scale, np.tan(np.radians(30)), rtol=1e-6)
    j = json.load(open(EXAMPLE))
    glass = j["materials"]["Glass.006"]
    sph2 = s.spheres[2]
    m = s.materials[sph2["material"]]
    assert np.isclose(m["ior"], glass["ior"]) and np.allclose(m["rgb"], glass["rgb"])
    # plane normal = normalize(u x v), scene.rs:139
    p = s.planes[3]
    n = np.cross(p["u"].astype(np.float64), p["v"].astype(np.float64))
    assert np.allclose(p["normal"], n / np.linalg.norm(n), atol=1e-6)


MINI = {
    "camera": {"pos": [0, 0, 0], "look_at": [0, 0, 1], "up": [0, 1, 0], "fov": 60, "aperture": 0.0},
    "render": {"width": 16, "height": 8, "samples": 2},
    "materials": {"a": {"rgb": [0.1, 0.2, 0.3], "metallic": 0.5, "roughness": 0.25, "ior": 1.0}},
    "objects": [
        {"sphere": {"name": "s", "center": [0, 0, 5], "radius": 1, "mat": "a", "in_focus": True}},
        {"plane": {"name": "p", "point": [0, -1, 0], "u": [1, 0, 0], "v": [0, 0, 1], "mat": "nope", "junk": 7}},
    ],
    "lights": [{"pos": [0, 5, 0], "u": [1, 0, 0], "v": [0, 0, 1], "intensity": [1, 2, 3]}],
    "ignored_top_level_key": [1, 2, {"x": None}],
}


def test_loader_defaults_and_untagged_objects(oracle):
    from pathtracer_amd import Scene
    s = Scene.parse(json.dumps(MINI))
    assert s.materials[0]["volume_density"] == 0 and s.materials[0]["volume_anisotropy"] == 0   # serde default
    magenta = s.materials[s.planes[0]["material"]]                                              # unknown name
    assert list(magenta["rgb"]) == [1, 0, 1] and magenta["roughness"] == 1 and magenta["ior"] == 1
    assert s.spheres[0]["in_focus"] == 1 and s.planes[0]["in_focus"] == 0
    # autofocus over the in_focus sphere (5x5 probes around the centre of a 16x8 image): host == oracle, bit for bit
    assert 4.0 < s.camera.focus < 4.5 and s.camera.focus == np.float32(oracle.autofocus(s))
    assert list(s.planes[0]["normal"]) == [0, -1, 0]
    s2 = Scene.parse(json.dumps(MINI), width=32, height=32, samples=9)
    assert (s2.width, s2.height, s2.samples) == (32, 32, 9) and s2.camera.aspect == 1.0


@pytest.mark.parametrize("mutate,needle", [
    (lambda d: d.pop("lights"), "lights"),
    (lambda d: d["camera"].pop("fov"), "fov"),
    (lambda d: d["render"].__setitem__("width", -3), "width"),
    (lambda d: d["render"].__setitem__("samples", 1.5), "samples"),
    (lambda d: d["materials"]["a"].pop("ior"), "ior"),
    (lambda d: d["objects"].append({"cube": {}}), "untagged"),
    (lambda d: d["objects"][0]["sphere"].pop("name"), "name"),
    (lambda d: d["lights"][0].__setitem__("pos", [1, 2]), "pos"),
])
def test_loader_rejects_bad_scenes(mutate, needle):
    """serde would fail `expect("json parse")` (scene.rs:96); here: HIPPT_ERR_IO with a message."""
    import copy
    from pathtracer_amd import HipptError, Scene
    d = copy.deepcopy(MINI)
    mutate(d)
    with pytest.raises(HipptError) as e:
        Scene.parse(json.dumps(d))
    assert "json parse" in str(e.value) and needle in str(e.value)


def test_loader_rejects_malformed_json_and_missing_file():
    from pathtracer_amd import HipptError, Scene
    for bad in ("{", '{"camera": }', "[1,2", '{"a": 1} x', ""):
        with pytest.raises(HipptError):
            Scene.parse(bad)
    with pytest.raises(HipptError) as e:
        Scene.load("/nonexistent/scene.json")
    assert "scene file" in str(e.value)                     # scene.rs:95 expect("scene file")


def test_loader_survives_mutated_scene_files():
    """4 000 random mutations of scene.example.json (deleted spans, inserted tokens, random bytes, truncation, duplicated
    spans, extreme numbers) and million-deep nesting: the loader either returns a scene or an error, never crashes or
    hangs. (The same mutator ran 30 000 cases under ASan + UBSan against host_scene.cpp alone: clean.)"""
    import random
    from pathtracer_amd import HipptError, Scene
    rnd = random.Random(12345)
    base = open(os.path.join(ROOT, "scenes", "scene.example.json"), "rb").read()
    tokens = [b"{", b"}", b"[", b"]", b",", b":", b'"', b"null", b"true", b"-", b"1e999", b"1e-999", b"NaN", b"\\u00", b"\\",
              b"0.", b"-0", b"99999999999999999999", b'"Sphere"', b'"Plane"', b'"material"', b'"objects"']
    ok = bad = 0
    for _ in range(4000):
        s = bytearray(base)
        for _m in range(rnd.randint(1, 4)):
            pos = rnd.randrange(len(s) + 1)
            kind = rnd.randrange(5)
            if kind == 0 and s:
                del s[pos % len(s):pos % len(s) + rnd.randint(1, 8)]
            elif kind == 1:
                s[pos:pos] = rnd.choice(tokens)
            elif kind == 2 and s:
                s[pos % len(s)] = rnd.randrange(256)
            elif kind == 3:
                del s[pos:]
            elif s:
                a = rnd.randrange(len(s))
                s[pos:pos] = s[a:a + rnd.randrange(64)]
        try:
            sc = Scene.parse(bytes(s))
            assert sc.width > 0 and sc.height > 0
            ok += 1
        except HipptError:
            bad += 1
    assert ok > 50 and bad > 2000, (ok, bad)
    for opener in (b"[", b'{"a":', b'{"camera":['):
        with pytest.raises(HipptError, match="nesting too deep"):
            Scene.parse(opener * 1000000)


def test_json_string_escapes_and_numbers():
    from pathtracer_amd import Scene
    d = json.loads(json.dumps(MINI))
    d["materials"] = {"métal \"x\"\\": {"rgb": [1e-1, 2.5E-1, 0.5], "metallic": 1, "roughness": 1e0, "ior": 1}}
    d["objects"][0]["sphere"]["mat"] = "métal \"x\"\\"
    for ensure_ascii in (True, False):
        s = Scene.parse(json.dumps(d, ensure_ascii=ensure_ascii))
        assert s.spheres[0]["material"] == 0 and np.allclose(s.materials[0]["rgb"], [0.1, 0.25, 0.5])


def test_render_image_name_pattern():
    from pathtracer_amd import hip_renderer
    a = hip_renderer.render_image_name(1920, 1080, 1024, 0.02, 5.0)
    b = hip_renderer.render_image_name(1920, 1080, 1024, 0.02, 5.0)
    assert re.fullmatch(r"renders/render_1920x1080_s1024_ap0\.02_f5\.0_[A-Za-z0-9]{6}\.png", a), a
    assert a != b


def test_png_writer_roundtrip(tmp_path):
    from PIL import Image
    from pathtracer_amd import hip_renderer
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    path = tmp_path / "x.png"
    hip_renderer.save_png(path, img)
    back = np.asarray(Image.open(path))
    assert back.shape == (37, 53, 4) and np.array_equal(back, img)


def test_checkpoint_roundtrip_and_rejections(tmp_path):
    """Progressive accumulation: the host accumulator + samples done is the checkpoint."""
    from pathtracer_amd import hip_renderer, _abi
    rng = np.random.default_rng(0)
    acc = rng.random((5, 7, 3), np.float32) * 1000
    p = tmp_path / "job.acc"
    hip_renderer.checkpoint_save(p, acc, 12345, seed=77)
    assert p.stat().st_size == 32 + 5 * 7 * 3 * 4 and not (tmp_path / "job.acc.tmp").exists()
    back, done, seed = hip_renderer.checkpoint_load(p, 7, 5)
    assert np.array_equal(back, acc) and (done, seed) == (12345, 77)
    raw = p.read_bytes()
    assert raw[:8] == b"HIPPTAC1" and np.frombuffer(raw[8:24], "<u4").tolist() == [7, 5, 12345, 77]
    with pytest.raises(_abi.HipptError, match="checkpoint is 7x5"):
        hip_renderer.checkpoint_load(p, 5, 7)
    (tmp_path / "short.acc").write_bytes(raw[:-4])
    with pytest.raises(_abi.HipptError, match="size does not match"):
        hip_renderer.checkpoint_load(tmp_path / "short.acc", 7, 5)
    (tmp_path / "long.acc").write_bytes(raw + b"x")
    with pytest.raises(_abi.HipptError, match="size does not match"):
        hip_renderer.checkpoint_load(tmp_path / "long.acc", 7, 5)
    (tmp_path / "bad.acc").write_bytes(b"NOTACKPT" + raw[8:])
    with pytest.raises(_abi.HipptError, match="not a hippt checkpoint"):
        hip_renderer.checkpoint_load(tmp_path / "bad.acc", 7, 5)
    with pytest.raises(_abi.HipptError):
        hip_renderer.checkpoint_load(tmp_path / "missing.acc", 7, 5)


def test_invalid_tables_are_rejected_before_any_launch():
    from pathtracer_amd import HipptError, Scene, hip_renderer
    s = Scene.parse(json.dumps(MINI))
    s.spheres["material"][0] = 99
    with pytest.raises(HipptError):
        hip_renderer.render_accum(s)


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly_without_fallback():
    """No CPU fallback in the product path: without a HIP device every render entry point errors."""
    from pathtracer_amd import HipptError, Scene, hip_renderer
    s = Scene.parse(json.dumps(MINI))
    with pytest.raises(HipptError):
        hip_renderer.render(s)
    with pytest.raises(HipptError):
        hip_renderer.unit_ops().rng(0, 0, 0, 4)


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "pathtracer_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in txt and "pt_oracle" not in txt and "oracle_lib" not in txt, f


def test_sample_range_matches_partition():
    """hippt_sample_range (the split hippt_render_*_multi uses) == pathtracer_amd.partition.sample_range."""
    from pathtracer_amd import hip_renderer
    from pathtracer_amd.partition import sample_range
    for total in (0, 1, 7, 8, 1024, 4096, 131072, 1000003):
        for world in (1, 2, 3, 4, 8, 13):
            got = [hip_renderer.sample_range(total, r, world) for r in range(world)]
            assert got == [sample_range(total, r, world) for r in range(world)]
            assert sum(c for _, c in got) == total and got[0][0] == 0
            assert all(got[r][0] + got[r][1] == got[r + 1][0] for r in range(world - 1))


def test_debug_dump_is_main_rs_60_87():
    """hippt_scene_dump == the text of src/main.rs:60-87 for scene.example.json: position / look_at / up / fov lines,
    object names, and floats as Rust's `{:?}` prints them (shortest round-trip digits, "1.0", "-0.0", "1.509958e-7").
    The expected block is written out literally; the f32 digits are cross-checked against numpy's shortest repr."""
    from pathtracer_amd import scene
    expected = """=== CAMERA INFO ===
 position : Vec3(0.0, 0.4057229, -1.1699998)
 look_at  : Vec3(1.509958e-7, 0.40572315, -0.16999984)
 up       : Vec3(-2.719304e-14, 1.0, -2.8212997e-7)
 fov (°)  : 60.00
 aspect   : 1.3333
 aperture : 0.0200
 autofocus: 5.0000

=== OBJECTS (7) ===
 [0] Plane 'Plane' { point: Vec3(0.0, -0.81, 8.54), normal: Vec3(-3.727507e-6, -1.0, 3.7591824e-6), mat_color: Vec3(0.27521268, 0.68945605, 0.80049694) }
 [1] Sphere 'Sphere' { center: Vec3(1.1354003, -0.3700193, 4.418177), radius: 1.0000, mat_color: Vec3(0.8, 0.8, 0.8) }
 [2] Sphere 'Sphere.001' { center: Vec3(-1.1011227, -0.5954898, 4.7919927), radius: 1.0000, mat_color: Vec3(0.50519305, 0.7576715, 0.80034703) }
 [3] Sphere 'Sphere.002' { center: Vec3(0.013796618, -0.7068826, 2.941975), radius: 0.5200, mat_color: Vec3(0.8, 0.8, 0.8) }
 [4] Plane 'Plane.001' { point: Vec3(-2.5, 0.0, 0.0), normal: Vec3(1.0, -4.5144347e-8, 0.0), mat_color: Vec3(0.7061524, 0.33920133, 0.8003553) }
 [5] Plane 'Plane.002' { point: Vec3(2.5, 0.0, 0.0), normal: Vec3(-1.0, -4.5144347e-8, -0.0), mat_color: Vec3(0.7061524, 0.33920133, 0.8003553) }
 [6] Plane 'Plane.003' { point: Vec3(0.0, 0.0, 7.2), normal: Vec3(0.0, 0.7986356, -0.601815), mat_color: Vec3(0.8, 0.8, 0.8) }

=== LIGHTS (2) ===
 [0] Light { pos: Vec3(0.0, 2.95, 4.0), u: Vec3(-1.0, -0.0, -0.0), v: Vec3(0.0, 2.3841858e-7, -1.0), intensity: Vec3(25.0, 25.0, 25.0) }
 [1] Light { pos: Vec3(0.0, 0.32999998, 11.379999), u: Vec3(-1.0, -0.0, -0.0), v: Vec3(0.0, 0.9993908, 0.03489971), intensity: Vec3(50.0, 50.0, 50.0) }
"""
    got = scene.dump(os.path.join(ROOT, "scenes", "scene.example.json"))
    assert got == expected
    # independent witness for the digit strings: numpy prints the shortest round-trip form of an f32 too
    import json
    cam = json.load(open(os.path.join(ROOT, "scenes", "scene.example.json")))["camera"]
    for v in cam["pos"] + cam["look_at"] + cam["up"]:
        f = np.float32(v)
        s = np.format_float_scientific(f, unique=True, trim="-", exp_digits=1).replace("e+", "e") if (f != 0 and (abs(f) < 1e-4 or abs(f) >= 1e16)) \
            else np.format_float_positional(f, unique=True, trim="0")
        assert s in got, s
    # render overrides change aspect only
    assert " aspect   : 1.7778\n" in scene.dump(os.path.join(ROOT, "scenes", "scene.example.json"), 1920, 1080, 4)


def test_one_hip_runtime_whatever_the_import_order():
    """libhippt loaded BEFORE torch used to leave two HIP runtimes in the process (the system ROCm's for libhippt,
    the wheel's own for torch): a torch stream passed to the C ABI then belonged to the other runtime. _abi loads the
    wheel's copy first when torch is installed."""
    import subprocess
    import sys
    code = ("from pathtracer_amd import _abi; _abi.load_library(); import torch; "
            "m = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}; print(len(m)); print(sorted(m))")
    for order in (code, "import torch; " + code):
        r = subprocess.run([sys.executable, "-c", order], cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert r.stdout.split()[0] == "1", r.stdout
