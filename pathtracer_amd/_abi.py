"""ctypes view of include/hippt.h: POD structs and the loader of libhippt.so.

The structs mirror the reference's loaded scene (src/scene.rs:87-92,
src/material.rs:4-11, src/sphere.rs:9-15, src/plane.rs:8-16, src/light.rs:4-9).
There is no Python or CPU fallback: if the HIP library is missing, loading
fails loudly.
"""
import ctypes as C
import os

N_COUNTERS = 16

FLAG_COUNTERS = 1
FLAG_WAVEFRONT = 2
FLAG_BVH = 4
FLAG_NO_BVH = 8
FLAG_TABLES_GLOBAL = 16
FLAG_MEGAKERNEL = 32
FLAG_NO_SLICES = 64
FLAG_LOCAL_REDUCE = 128
FLAG_TRAVERSAL_STATS = 256
FLAG_COUNT_EXECUTED = 512

COUNTER_NAMES = (
    "paths", "extend_rays", "extend_hits", "shadow_rays", "shadow_sphere_tests",
    "shadow_plane_tests", "nee_vertices", "lit_evals", "diffuse_lobes", "spec_lobes",
    "glass_vertices", "medium_vertices", "scatter_events", "extend_sphere_tests",
    "extend_plane_tests", "reserved",
)


class Material(C.Structure):
    _fields_ = [("rgb", C.c_float * 3), ("metallic", C.c_float), ("roughness", C.c_float),
                ("ior", C.c_float), ("volume_density", C.c_float), ("volume_anisotropy", C.c_float)]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("material", C.c_uint32),
                ("order", C.c_uint32), ("in_focus", C.c_uint32)]


class Plane(C.Structure):
    _fields_ = [("point", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3),
                ("normal", C.c_float * 3), ("material", C.c_uint32), ("order", C.c_uint32),
                ("in_focus", C.c_uint32)]


class Light(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3),
                ("intensity", C.c_float * 3)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("forward", C.c_float * 3), ("aspect", C.c_float), ("scale", C.c_float),
                ("aperture", C.c_float), ("focus", C.c_float)]


class SceneC(C.Structure):
    _fields_ = [("camera", Camera), ("width", C.c_uint32), ("height", C.c_uint32),
                ("samples", C.c_uint32), ("n_materials", C.c_uint32), ("n_spheres", C.c_uint32),
                ("n_planes", C.c_uint32), ("n_lights", C.c_uint32), ("reserved", C.c_uint32),
                ("materials", C.POINTER(Material)), ("spheres", C.POINTER(Sphere)),
                ("planes", C.POINTER(Plane)), ("lights", C.POINTER(Light))]


class Options(C.Structure):
    _fields_ = [("sample_begin", C.c_uint32), ("sample_count", C.c_uint32), ("max_depth", C.c_uint32),
                ("seed", C.c_uint32), ("flags", C.c_uint32), ("device", C.c_int32)]


_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# HIPPT_LIB: another build of the same library (A/B timing of kernel variants with tools/ab_bench.sh)
LIB_PATH = os.environ.get("HIPPT_LIB") or os.path.join(_PKG_DIR, "lib", "libhippt.so")

# every symbol include/hippt.h declares
EXPORTED_SYMBOLS = (
    "hippt_last_error", "hippt_render_rgba8", "hippt_render_accum_f32", "hippt_create", "hippt_destroy",
    "hippt_set_scene", "hippt_render_accum_device", "hippt_resolve_device", "hippt_read_counters",
    "hippt_kernel_time_ms", "hippt_scene_load", "hippt_scene_parse", "hippt_scene_get",
    "hippt_scene_set_render", "hippt_scene_free", "hippt_camera_basis", "hippt_autofocus",
    "hippt_render_image_name", "hippt_write_png_rgba8", "hippt_test_intersect", "hippt_test_intersect_bvh",
    "hippt_test_occluded", "hippt_test_nee_eval", "hippt_test_ggx_terms", "hippt_test_ggx_sample",
    "hippt_test_diffuse_sample", "hippt_test_glass", "hippt_test_hg", "hippt_test_camera_ray",
    "hippt_test_tonemap", "hippt_test_rng",
    "hippt_render_accum_add_f32", "hippt_resolve_f32", "hippt_checkpoint_save", "hippt_checkpoint_load",
    "hippt_render_rgba8_multi", "hippt_render_accum_f32_multi", "hippt_sample_range", "hippt_device_count",
    "hippt_scene_dump", "hippt_test_shadow_nee", "hippt_render_accum_device_noalloc", "hippt_read_traversal_stats",
)

_lib = None


class HipptError(RuntimeError):
    """Non-zero status from the C ABI (the reference panics via `expect`)."""


def _one_hip_runtime():
    """One HIP runtime per process. libhippt.so needs `libamdhip64.so.7` and finds the system ROCm's; a PyTorch-ROCm
    wheel ships its OWN copy (same SONAME) and asks for it as `libamdhip64.so`. If PyTorch is imported first, libhippt
    binds to PyTorch's copy by SONAME and all is well; the other way round the loader takes PyTorch's request for a
    different name and the process ends up with two runtimes, in which a torch stream or device pointer handed to
    libhippt belongs to the wrong one (seen as an abort inside hippt_render_accum_device). So when PyTorch is
    installed but not imported yet, its copy of the runtime is loaded first, and both libraries share it."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library():
    """Loads pathtracer_amd/lib/libhippt.so (built by __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    _one_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise HipptError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C pathtracer_amd/csrc). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.hippt_last_error.restype = C.c_char_p
    lib.hippt_autofocus.restype = C.c_float
    lib.hippt_autofocus.argtypes = [C.POINTER(SceneC)]
    lib.hippt_camera_basis.restype = None
    lib.hippt_camera_basis.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                       C.c_float, C.c_uint32, C.c_uint32, C.POINTER(Camera)]
    lib.hippt_scene_get.restype = C.POINTER(SceneC)
    lib.hippt_scene_get.argtypes = [C.c_void_p]
    lib.hippt_scene_free.restype = None
    lib.hippt_scene_free.argtypes = [C.c_void_p]
    lib.hippt_scene_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    lib.hippt_scene_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.hippt_scene_set_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.hippt_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.hippt_destroy.argtypes = [C.c_void_p]
    lib.hippt_set_scene.argtypes = [C.c_void_p, C.POINTER(SceneC)]
    lib.hippt_render_accum_device.argtypes = [C.c_void_p, C.POINTER(Options), C.c_void_p, C.c_void_p]
    lib.hippt_render_accum_device_noalloc.argtypes = [C.c_void_p, C.POINTER(Options), C.c_void_p, C.c_void_p]
    lib.hippt_resolve_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.hippt_read_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.hippt_read_traversal_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.hippt_kernel_time_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    lib.hippt_render_rgba8.argtypes = [C.POINTER(SceneC), C.POINTER(Options), C.c_void_p]
    lib.hippt_render_accum_f32.argtypes = [C.POINTER(SceneC), C.POINTER(Options), C.c_void_p, C.c_void_p]
    lib.hippt_render_image_name.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float,
                                            C.c_char_p, C.c_size_t]
    lib.hippt_write_png_rgba8.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
    lib.hippt_render_rgba8_multi.argtypes = [C.POINTER(SceneC), C.POINTER(Options), C.POINTER(C.c_int), C.c_int, C.c_void_p]
    lib.hippt_render_accum_f32_multi.argtypes = [C.POINTER(SceneC), C.POINTER(Options), C.POINTER(C.c_int), C.c_int, C.c_void_p]
    lib.hippt_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.hippt_scene_dump.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.hippt_render_accum_add_f32.argtypes = [C.c_void_p, C.POINTER(Options), C.c_void_p]
    lib.hippt_resolve_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.hippt_checkpoint_save.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.hippt_checkpoint_load.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_uint32)]
    lib.hippt_sample_range.restype = None
    lib.hippt_sample_range.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    _lib = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load_library()
        msg = lib.hippt_last_error()
        raise HipptError(f"hippt error {rc}: {msg.decode(errors='replace') if msg else ''}")
