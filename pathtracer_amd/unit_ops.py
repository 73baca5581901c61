"""numpy front-end of the unit-kernel exports of include/hippt.h.

`UnitOps(lib, prefix)` binds `<prefix>intersect`, `<prefix>ggx_sample`, ... of a
loaded C library. The product binding is `hip_renderer.unit_ops()` (libhippt,
prefix `hippt_test_`, HIP kernels). The tests bind the same class to the CPU
oracle's identically-shaped exports to compare the two.
"""
import ctypes as C

import numpy as np

from . import _abi


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class UnitOps:
    def __init__(self, lib, prefix, device=0, error_fn=None):
        self.lib, self.prefix, self.device = lib, prefix, device
        self._err = error_fn

    def _call(self, name, *args):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype = C.c_int
        rc = fn(*args)
        if rc != 0:
            msg = self._err() if self._err else b""
            raise _abi.HipptError(f"{self.prefix}{name} failed ({rc}): {msg.decode(errors='replace') if msg else ''}")

    def intersect(self, scene, ro, rd):
        ro, rd = _f32(ro, (-1, 3)), _f32(rd, (-1, 3))
        n = len(ro)
        t = np.empty(n, np.float32)
        nrm = np.empty((n, 3), np.float32)
        idx = np.empty(n, np.int32)
        sc, _keep = scene.to_c()
        self._call("intersect", C.byref(sc), C.c_uint32(n), _p(ro), _p(rd), _p(t), _p(nrm), _p(idx),
                   C.c_int(self.device))
        return t, nrm, idx

    def intersect_bvh(self, scene, ro, rd):
        ro, rd = _f32(ro, (-1, 3)), _f32(rd, (-1, 3))
        n = len(ro)
        t = np.empty(n, np.float32)
        nrm = np.empty((n, 3), np.float32)
        idx = np.empty(n, np.int32)
        sc, _keep = scene.to_c()
        self._call("intersect_bvh", C.byref(sc), C.c_uint32(n), _p(ro), _p(rd), _p(t), _p(nrm), _p(idx),
                   C.c_int(self.device))
        return t, nrm, idx

    def occluded(self, scene, ro, rd, dist2):
        ro, rd, dist2 = _f32(ro, (-1, 3)), _f32(rd, (-1, 3)), _f32(dist2)
        n = len(ro)
        out = np.empty(n, np.uint8)
        sc, _keep = scene.to_c()
        self._call("occluded", C.byref(sc), C.c_uint32(n), _p(ro), _p(rd), _p(dist2), _p(out), C.c_int(self.device))
        return out

    def shadow_nee(self, scene, x, l, dist2):
        """renderer.rs:44-46 as NEE runs it: shading point x, unit direction l, dist2 -> bit 0 occluded, bit 1 slot mismatch."""
        x, l, dist2 = _f32(x, (-1, 3)), _f32(l, (-1, 3)), _f32(dist2)
        n = len(x)
        out = np.empty(n, np.uint8)
        sc, _keep = scene.to_c()
        self._call("shadow_nee", C.byref(sc), C.c_uint32(n), _p(x), _p(l), _p(dist2), _p(out), C.c_int(self.device))
        return out

    def nee_eval(self, nrm, v, l, dist2, mat, light):
        nrm, v, l, dist2, mat = _f32(nrm, (-1, 3)), _f32(v, (-1, 3)), _f32(l, (-1, 3)), _f32(dist2), _f32(mat, (-1, 8))
        n = len(nrm)
        out = np.empty((n, 3), np.float32)
        lt = _abi.Light()
        lt.pos[:], lt.u[:], lt.v[:], lt.intensity[:] = [[float(x) for x in light[k]] for k in ("pos", "u", "v", "intensity")]
        self._call("nee_eval", C.c_uint32(n), _p(nrm), _p(v), _p(l), _p(dist2), _p(mat), C.byref(lt), _p(out),
                   C.c_int(self.device))
        return out

    def ggx_terms(self, nh, nv, nl, vh, rough, f0):
        nh, nv, nl, vh, rough, f0 = _f32(nh), _f32(nv), _f32(nl), _f32(vh), _f32(rough), _f32(f0, (-1, 3))
        n = len(nh)
        out = np.empty((n, 5), np.float32)
        self._call("ggx_terms", C.c_uint32(n), _p(nh), _p(nv), _p(nl), _p(vh), _p(rough), _p(f0), _p(out),
                   C.c_int(self.device))
        return out

    def ggx_sample(self, nrm, rough, r1, r2):
        nrm, rough, r1, r2 = _f32(nrm, (-1, 3)), _f32(rough), _f32(r1), _f32(r2)
        n = len(nrm)
        out = np.empty((n, 3), np.float32)
        self._call("ggx_sample", C.c_uint32(n), _p(nrm), _p(rough), _p(r1), _p(r2), _p(out), C.c_int(self.device))
        return out

    def diffuse_sample(self, nrm, u_phi, r2):
        nrm, u_phi, r2 = _f32(nrm, (-1, 3)), _f32(u_phi), _f32(r2)
        n = len(nrm)
        out = np.empty((n, 3), np.float32)
        self._call("diffuse_sample", C.c_uint32(n), _p(nrm), _p(u_phi), _p(r2), _p(out), C.c_int(self.device))
        return out

    def glass(self, rd, nrm, ior, u):
        rd, nrm, ior, u = _f32(rd, (-1, 3)), _f32(nrm, (-1, 3)), _f32(ior), _f32(u)
        n = len(rd)
        out = np.empty((n, 3), np.float32)
        self._call("glass", C.c_uint32(n), _p(rd), _p(nrm), _p(ior), _p(u), _p(out), C.c_int(self.device))
        return out

    def hg(self, rd, g, u_hg, u_phi):
        rd, g, u_hg, u_phi = _f32(rd, (-1, 3)), _f32(g), _f32(u_hg), _f32(u_phi)
        n = len(rd)
        out = np.empty((n, 4), np.float32)
        self._call("hg", C.c_uint32(n), _p(rd), _p(g), _p(u_hg), _p(u_phi), _p(out), C.c_int(self.device))
        return out[:, 0], out[:, 1:]

    def camera_ray(self, scene, xy, u4):
        xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
        u4 = _f32(u4, (-1, 4))
        n = len(xy)
        ro = np.empty((n, 3), np.float32)
        rd = np.empty((n, 3), np.float32)
        sc, _keep = scene.to_c()
        self._call("camera_ray", C.byref(sc), C.c_uint32(n), _p(xy), _p(u4), _p(ro), _p(rd), C.c_int(self.device))
        return ro, rd

    def tonemap(self, rgb):
        rgb = _f32(rgb, (-1, 3))
        n = len(rgb)
        out = np.empty((n, 3), np.uint8)
        self._call("tonemap", C.c_uint32(n), _p(rgb), _p(out), C.c_int(self.device))
        return out

    def rng(self, seed, pixel, sample, n):
        out = np.empty(n, np.float32)
        self._call("rng", C.c_uint32(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(n), _p(out),
                   C.c_int(self.device))
        return out
