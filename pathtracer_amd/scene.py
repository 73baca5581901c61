"""Loaded scene (reference `Scene`, src/scene.rs:87-92) as numpy tables.

`Scene.load(path)` goes through the C loader in libhippt (host_scene.cpp),
so there is one implementation of the scene.json schema. Scenes can also be
built programmatically (synthetic benchmarks, tests); `to_c()` yields the
`hippt_scene` the C ABI takes.
"""
import ctypes as C

import numpy as np

from . import _abi

MAT_DTYPE = np.dtype([("rgb", "<f4", 3), ("metallic", "<f4"), ("roughness", "<f4"), ("ior", "<f4"),
                      ("volume_density", "<f4"), ("volume_anisotropy", "<f4")])
SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("material", "<u4"), ("order", "<u4"),
                         ("in_focus", "<u4")])
PLANE_DTYPE = np.dtype([("point", "<f4", 3), ("u", "<f4", 3), ("v", "<f4", 3), ("normal", "<f4", 3),
                        ("material", "<u4"), ("order", "<u4"), ("in_focus", "<u4")])
LIGHT_DTYPE = np.dtype([("pos", "<f4", 3), ("u", "<f4", 3), ("v", "<f4", 3), ("intensity", "<f4", 3)])

assert MAT_DTYPE.itemsize == C.sizeof(_abi.Material)
assert SPHERE_DTYPE.itemsize == C.sizeof(_abi.Sphere)
assert PLANE_DTYPE.itemsize == C.sizeof(_abi.Plane)
assert LIGHT_DTYPE.itemsize == C.sizeof(_abi.Light)


def _f3(a):
    return (C.c_float * 3)(*[float(x) for x in a])


class Scene:
    """camera + render settings + material / sphere / plane / light tables."""

    def __init__(self):
        self.width = 0
        self.height = 0
        self.samples = 0
        self.camera = _abi.Camera()
        # raw camera inputs (scene.rs:7-16), kept to re-derive the basis on resize
        self.cam_pos = (0.0, 0.0, 0.0)
        self.cam_look_at = (0.0, 0.0, 1.0)
        self.cam_up = (0.0, 1.0, 0.0)
        self.cam_fov = 60.0
        self.cam_aperture = 0.0
        self.materials = np.zeros(0, MAT_DTYPE)
        self.spheres = np.zeros(0, SPHERE_DTYPE)
        self.planes = np.zeros(0, PLANE_DTYPE)
        self.lights = np.zeros(0, LIGHT_DTYPE)

    # -- construction -----------------------------------------------------------
    @classmethod
    def load(cls, path, width=None, height=None, samples=None):
        """scene::load (src/scene.rs:94-158) + camera basis + autofocus (src/main.rs:44-56)."""
        lib = _abi.load_library()
        h = C.c_void_p()
        _abi.check(lib.hippt_scene_load(str(path).encode(), C.byref(h)), lib)
        try:
            return cls._from_handle(lib, h, width, height, samples)
        finally:
            lib.hippt_scene_free(h)

    @classmethod
    def parse(cls, text, width=None, height=None, samples=None):
        lib = _abi.load_library()
        data = text.encode() if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        _abi.check(lib.hippt_scene_parse(data, len(data), C.byref(h)), lib)
        try:
            return cls._from_handle(lib, h, width, height, samples)
        finally:
            lib.hippt_scene_free(h)

    @classmethod
    def _from_handle(cls, lib, h, width, height, samples):
        sc = lib.hippt_scene_get(h).contents
        if width or height or samples:
            _abi.check(lib.hippt_scene_set_render(h, width or sc.width, height or sc.height,
                                                  samples if samples is not None else sc.samples), lib)
            sc = lib.hippt_scene_get(h).contents
        s = cls()
        s.width, s.height, s.samples = sc.width, sc.height, sc.samples
        s.camera = _abi.Camera.from_buffer_copy(sc.camera)

        def table(ptr, n, dtype):
            if n == 0:
                return np.zeros(0, dtype)
            buf = C.string_at(ptr, n * dtype.itemsize)
            return np.frombuffer(buf, dtype=dtype).copy()

        s.materials = table(sc.materials, sc.n_materials, MAT_DTYPE)
        s.spheres = table(sc.spheres, sc.n_spheres, SPHERE_DTYPE)
        s.planes = table(sc.planes, sc.n_planes, PLANE_DTYPE)
        s.lights = table(sc.lights, sc.n_lights, LIGHT_DTYPE)
        s.cam_pos = tuple(sc.camera.pos)
        s.cam_aperture = sc.camera.aperture
        s._basis_from_file = True
        return s

    def set_camera(self, pos, look_at, up, fov_deg, aperture, focus=None):
        """Camera basis as src/main.rs:44-51; focus=None runs autofocus (renderer.rs:204-234)."""
        lib = _abi.load_library()
        self.cam_pos, self.cam_look_at, self.cam_up = tuple(pos), tuple(look_at), tuple(up)
        self.cam_fov, self.cam_aperture = float(fov_deg), float(aperture)
        cam = _abi.Camera()
        lib.hippt_camera_basis(_f3(pos), _f3(look_at), _f3(up), C.c_float(fov_deg), self.width, self.height,
                               C.byref(cam))
        cam.aperture = aperture
        self.camera = cam
        if focus is None:
            sc, _keep = self.to_c()
            focus = lib.hippt_autofocus(C.byref(sc))
        self.camera.focus = focus
        return self

    def add_material(self, rgb, metallic=0.0, roughness=1.0, ior=1.0, volume_density=0.0, volume_anisotropy=0.0):
        m = np.zeros(1, MAT_DTYPE)
        m["rgb"], m["metallic"], m["roughness"], m["ior"] = rgb, metallic, roughness, ior
        m["volume_density"], m["volume_anisotropy"] = volume_density, volume_anisotropy
        self.materials = np.concatenate([self.materials, m])
        return len(self.materials) - 1

    def n_objects(self):
        return len(self.spheres) + len(self.planes)

    def add_sphere(self, center, radius, material, in_focus=False):
        s = np.zeros(1, SPHERE_DTYPE)
        s["center"], s["radius"], s["material"] = center, radius, material
        s["order"], s["in_focus"] = self.n_objects(), int(in_focus)
        self.spheres = np.concatenate([self.spheres, s])

    def add_plane(self, point, u, v, material, in_focus=False):
        p = np.zeros(1, PLANE_DTYPE)
        u32, v32 = np.asarray(u, np.float32), np.asarray(v, np.float32)
        # normal = normalize(u x v) in f32, one rounding per op (scene.rs:139)
        cx = np.float32(u32[1] * v32[2]) - np.float32(u32[2] * v32[1])
        cy = np.float32(u32[2] * v32[0]) - np.float32(u32[0] * v32[2])
        cz = np.float32(u32[0] * v32[1]) - np.float32(u32[1] * v32[0])
        n2 = np.float32(np.float32(np.float32(cx * cx) + np.float32(cy * cy)) + np.float32(cz * cz))
        inv = np.float32(1.0) / np.sqrt(n2, dtype=np.float32)
        p["point"], p["u"], p["v"], p["material"] = point, u32, v32, material
        p["normal"] = [cx * inv, cy * inv, cz * inv]
        p["order"], p["in_focus"] = self.n_objects(), int(in_focus)
        self.planes = np.concatenate([self.planes, p])

    def add_light(self, pos, u, v, intensity):
        l = np.zeros(1, LIGHT_DTYPE)
        l["pos"], l["u"], l["v"], l["intensity"] = pos, u, v, intensity
        self.lights = np.concatenate([self.lights, l])

    def with_render(self, width, height, samples):
        """Copy with render.{width,height,samples} overridden; aspect re-derived."""
        import copy
        s = copy.copy(self)
        s.camera = _abi.Camera.from_buffer_copy(self.camera)
        s.width, s.height, s.samples = width, height, samples
        s.camera.aspect = np.float32(width) / np.float32(height)
        return s

    # -- C view -----------------------------------------------------------------
    def to_c(self):
        """Returns (hippt_scene, keepalive)."""
        keep = [np.ascontiguousarray(self.materials), np.ascontiguousarray(self.spheres),
                np.ascontiguousarray(self.planes), np.ascontiguousarray(self.lights)]
        sc = _abi.SceneC()
        sc.camera = self.camera
        sc.width, sc.height, sc.samples = int(self.width), int(self.height), int(self.samples)
        sc.n_materials, sc.n_spheres, sc.n_planes, sc.n_lights = (len(k) for k in keep)
        sc.materials = keep[0].ctypes.data_as(C.POINTER(_abi.Material))
        sc.spheres = keep[1].ctypes.data_as(C.POINTER(_abi.Sphere))
        sc.planes = keep[2].ctypes.data_as(C.POINTER(_abi.Plane))
        sc.lights = keep[3].ctypes.data_as(C.POINTER(_abi.Light))
        return sc, keep


def dump(path, width=None, height=None, samples=None):
    """The text main() prints before rendering (src/main.rs:60-87) for the scene file at `path`."""
    lib = _abi.load_library()
    h = C.c_void_p()
    _abi.check(lib.hippt_scene_load(str(path).encode(), C.byref(h)), lib)
    try:
        if width or height or samples:
            sc = lib.hippt_scene_get(h).contents
            _abi.check(lib.hippt_scene_set_render(h, width or sc.width, height or sc.height,
                                                  samples if samples is not None else sc.samples), lib)
        n = lib.hippt_scene_dump(h, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib.hippt_scene_dump(h, buf, n + 1)
        return buf.value.decode()
    finally:
        lib.hippt_scene_free(h)
