"""hip_renderer — the drop-in for the reference's `gpu_renderer` module.

    reference:  pub fn render(scene: &Scene) -> RgbaImage     (src/gpu_renderer.rs:11-13)
    here:       render(scene) -> uint8 ndarray [height, width, 4]

Same inputs (the loaded Scene), same output (8-bit RGBA, alpha 255, row 0 = top,
ACES + gamma 2.2, truncating quantisation), same blocking single-call shape.
Everything runs in the HIP kernels of libhippt.so through the C ABI of
include/hippt.h; a missing library or GPU raises HipptError (the reference
panics via `expect`, gpu_renderer.rs:78,86).
"""
import ctypes as C

import numpy as np

from . import _abi
from .unit_ops import UnitOps


def _opts(sample_begin=0, sample_count=0, max_depth=0, seed=0, flags=0, device=0):
    return _abi.Options(sample_begin, sample_count, max_depth, seed, flags, device)


def render(scene, seed=0, device=0, max_depth=0, flags=0):
    """gpu_renderer::render: tonemapped RGBA8 image of `scene` (scene.samples spp)."""
    lib = _abi.load_library()
    sc, _keep = scene.to_c()
    out = np.empty((scene.height, scene.width, 4), np.uint8)
    o = _opts(0, scene.samples, max_depth, seed, flags, device)
    _abi.check(lib.hippt_render_rgba8(C.byref(sc), C.byref(o), out.ctypes.data_as(C.c_void_p)), lib)
    return out


def render_accum(scene, sample_begin=0, sample_count=0, seed=0, device=0, max_depth=0, flags=0, counters=False):
    """Radiance SUMS over a sample range: float32 [height, width, 3] (+ counter dict)."""
    lib = _abi.load_library()
    sc, _keep = scene.to_c()
    out = np.empty((scene.height, scene.width, 3), np.float32)
    cn = (C.c_uint64 * _abi.N_COUNTERS)()
    if counters:
        flags |= _abi.FLAG_COUNTERS
    o = _opts(sample_begin, sample_count, max_depth, seed, flags, device)
    _abi.check(lib.hippt_render_accum_f32(C.byref(sc), C.byref(o), out.ctypes.data_as(C.c_void_p),
                                          C.cast(cn, C.c_void_p) if counters else None), lib)
    if counters:
        return out, dict(zip(_abi.COUNTER_NAMES, [int(x) for x in cn]))
    return out


def render_multi(scene, devices, seed=0, max_depth=0, flags=0):
    """gpu_renderer::render on several GPUs of this process: sample-range split, one RCCL reduce, tonemap on devices[0]."""
    lib = _abi.load_library()
    sc, _keep = scene.to_c()
    out = np.empty((scene.height, scene.width, 4), np.uint8)
    o = _opts(0, 0, max_depth, seed, flags, 0)
    dv = (C.c_int * len(devices))(*devices)
    _abi.check(lib.hippt_render_rgba8_multi(C.byref(sc), C.byref(o), dv, len(devices), out.ctypes.data_as(C.c_void_p)), lib)
    return out


def render_accum_multi(scene, devices, sample_begin=0, sample_count=0, seed=0, max_depth=0, flags=0):
    """Radiance sums of a sample range rendered by several GPUs of this process and reduced with RCCL."""
    lib = _abi.load_library()
    sc, _keep = scene.to_c()
    out = np.empty((scene.height, scene.width, 3), np.float32)
    o = _opts(sample_begin, sample_count, max_depth, seed, flags, 0)
    dv = (C.c_int * len(devices))(*devices)
    _abi.check(lib.hippt_render_accum_f32_multi(C.byref(sc), C.byref(o), dv, len(devices), out.ctypes.data_as(C.c_void_p)), lib)
    return out


def sample_range(total, rank, world):
    """hippt_sample_range: the C ABI's copy of pathtracer_amd.partition.sample_range."""
    lib = _abi.load_library()
    b, c = C.c_uint32(), C.c_uint32()
    lib.hippt_sample_range(total, rank, world, C.byref(b), C.byref(c))
    return b.value, c.value


def unit_ops(device=0):
    """Unit-kernel exports (hippt_test_*) of the HIP library."""
    lib = _abi.load_library()
    return UnitOps(lib, "hippt_test_", device, lib.hippt_last_error)


class Context:
    """Resident context: scene tables stay on the GPU; launches are asynchronous
    on a caller-provided HIP stream and add into a caller-owned device buffer."""

    def __init__(self, device=0):
        self.lib = _abi.load_library()
        self.h = C.c_void_p()
        _abi.check(self.lib.hippt_create(device, C.byref(self.h)), self.lib)
        self.scene = None

    def close(self):
        if self.h:
            self.lib.hippt_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, scene):
        sc, _keep = scene.to_c()
        _abi.check(self.lib.hippt_set_scene(self.h, C.byref(sc)), self.lib)
        self.scene = scene

    def render_accum_device(self, d_accum_ptr, sample_begin, sample_count, seed=0, max_depth=0, flags=0, stream=0):
        o = _opts(sample_begin, sample_count, max_depth, seed, flags, 0)
        _abi.check(self.lib.hippt_render_accum_device(self.h, C.byref(o), C.c_void_p(d_accum_ptr),
                                                      C.c_void_p(stream)), self.lib)

    def render_accum_device_noalloc(self, d_accum_ptr, sample_begin, sample_count, seed=0, max_depth=0, flags=0, stream=0):
        """hippt_render_accum_device_noalloc: the same launches with the library's static device scratch (no allocation)."""
        o = _opts(sample_begin, sample_count, max_depth, seed, flags, 0)
        _abi.check(self.lib.hippt_render_accum_device_noalloc(self.h, C.byref(o), C.c_void_p(d_accum_ptr),
                                                              C.c_void_p(stream)), self.lib)

    def resolve_device(self, d_accum_ptr, spp, d_rgba_ptr, stream=0):
        _abi.check(self.lib.hippt_resolve_device(self.h, C.c_void_p(d_accum_ptr), spp, C.c_void_p(d_rgba_ptr),
                                                 C.c_void_p(stream)), self.lib)

    def render_accum_add(self, host_accum, sample_begin, sample_count, seed=0, max_depth=0, flags=0):
        """Progressive step (gpu_renderer.rs:94-183): renders a sample range and adds it to a HOST accumulator."""
        assert host_accum.dtype == np.float32 and host_accum.flags["C_CONTIGUOUS"]
        o = _opts(sample_begin, sample_count, max_depth, seed, flags, 0)
        _abi.check(self.lib.hippt_render_accum_add_f32(self.h, C.byref(o), host_accum.ctypes.data_as(C.c_void_p)), self.lib)

    def resolve(self, host_accum, spp):
        """Average / ACES / gamma / u8 of a host accumulator [h, w, 3] -> uint8 [h, w, 4]."""
        acc = np.ascontiguousarray(host_accum, np.float32)
        n = acc.size // 3
        out = np.empty(acc.shape[:-1] + (4,), np.uint8)
        _abi.check(self.lib.hippt_resolve_f32(self.h, acc.ctypes.data_as(C.c_void_p), n, int(spp),
                                              out.ctypes.data_as(C.c_void_p)), self.lib)
        return out

    def read_counters(self):
        cn = (C.c_uint64 * _abi.N_COUNTERS)()
        _abi.check(self.lib.hippt_read_counters(self.h, cn), self.lib)
        return dict(zip(_abi.COUNTER_NAMES, [int(x) for x in cn]))

    def read_traversal_stats(self):
        """{closest, any} x {rays, nodes, leaf_fetches, aux_rows} of LBVH queue launches made with FLAG_TRAVERSAL_STATS."""
        st = (C.c_uint64 * 10)()
        _abi.check(self.lib.hippt_read_traversal_stats(self.h, st), self.lib)
        names = ("rays", "nodes", "leaf_fetches", "aux_rows")
        out = {"closest": dict(zip(names, [int(x) for x in st[:4]])), "any": dict(zip(names, [int(x) for x in st[4:8]]))}
        out["closest"]["nodes_from_lds"], out["any"]["nodes_from_lds"] = int(st[8]), int(st[9])
        return out

    def kernel_time_ms(self):
        tot, n = C.c_double(), C.c_uint32()
        _abi.check(self.lib.hippt_kernel_time_ms(self.h, C.byref(tot), C.byref(n)), self.lib)
        return tot.value, n.value


def render_image_name(w, h, s, ap, f):
    """renderer::render_image_name (src/renderer.rs:156-161)."""
    lib = _abi.load_library()
    buf = C.create_string_buffer(256)
    lib.hippt_render_image_name(w, h, s, C.c_float(ap), C.c_float(f), buf, 256)
    return buf.value.decode()


def checkpoint_save(path, accum, samples_done, seed=0):
    """Accumulator [h, w, 3] + samples done -> checkpoint file (include/hippt.h)."""
    lib = _abi.load_library()
    acc = np.ascontiguousarray(accum, np.float32)
    h, w = acc.shape[:2]
    _abi.check(lib.hippt_checkpoint_save(str(path).encode(), acc.ctypes.data_as(C.c_void_p), w, h, int(samples_done), int(seed)), lib)


def checkpoint_load(path, width, height):
    """-> (accum [h, w, 3] float32, samples_done, seed)."""
    lib = _abi.load_library()
    acc = np.empty((height, width, 3), np.float32)
    done, seed = C.c_uint32(), C.c_uint32()
    _abi.check(lib.hippt_checkpoint_load(str(path).encode(), acc.ctypes.data_as(C.c_void_p), width, height,
                                         C.byref(done), C.byref(seed)), lib)
    return acc, done.value, seed.value


def save_png(path, rgba):
    lib = _abi.load_library()
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = rgba.shape[:2]
    _abi.check(lib.hippt_write_png_rgba8(str(path).encode(), rgba.ctypes.data_as(C.c_void_p), w, h), lib)
