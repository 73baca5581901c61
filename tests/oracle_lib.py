"""Test-side binding of the CPU oracle (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this. The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from pathtracer_amd import _abi
from pathtracer_amd.unit_ops import UnitOps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")


def build_oracle():
    src = os.path.join(ORACLE_DIR, "pt_oracle.cpp")
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return ORACLE_SO


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def build_fast_oracle(out_dir):
    """The CPU-baseline build of the same source (bench.py, SURVEY §8d: -O3 -march=native), compiled ON the machine that
    times it — a -march=native library built elsewhere may not run here — into out_dir. Returns (path, flags)."""
    flags = "-O3 -march=native -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread"
    out = os.path.join(out_dir, "liboracle_fast.so")
    subprocess.check_call(["g++"] + flags.split() + ["-shared", "-o", out, os.path.join(ORACLE_DIR, "pt_oracle.cpp")],
                          stdout=subprocess.DEVNULL)
    return out, flags


class Oracle:
    def __init__(self, path=None):
        self.lib = C.CDLL(path or build_oracle())
        self.lib.orc_last_error.restype = C.c_char_p
        self.lib.orc_autofocus.restype = C.c_float
        self.unit = UnitOps(self.lib, "orc_test_", 0, self.lib.orc_last_error)

    def render_accum(self, scene, sample_begin=0, sample_count=0, seed=0, max_depth=0, counters=False,
                     iterative=False, threads=0):
        sc, _keep = scene.to_c()
        out = np.empty((scene.height, scene.width, 3), np.float32)
        cn = (C.c_uint64 * _abi.N_COUNTERS)()
        o = _abi.Options(sample_begin, sample_count, max_depth, seed, _abi.FLAG_COUNTERS if counters else 0, 0)
        rc = self.lib.orc_render_accum_ex(C.byref(sc), C.byref(o), _p(out), cn, int(iterative), int(threads))
        assert rc == 0, self.lib.orc_last_error()
        if counters:
            return out, dict(zip(_abi.COUNTER_NAMES, [int(x) for x in cn]))
        return out

    def render_rgba8(self, scene, seed=0):
        sc, _keep = scene.to_c()
        out = np.empty((scene.height, scene.width, 4), np.uint8)
        o = _abi.Options(0, 0, 0, seed, 0, 0)
        rc = self.lib.orc_render_rgba8(C.byref(sc), C.byref(o), _p(out))
        assert rc == 0
        return out

    def resolve(self, accum, spp):
        accum = np.ascontiguousarray(accum, np.float32)
        n = accum.size // 3
        out = np.empty((n, 4), np.uint8)
        self.lib.orc_resolve(_p(accum), C.c_uint32(n), C.c_uint32(spp), _p(out))
        return out.reshape(accum.shape[:-1] + (4,))

    def paths(self, scene, xys, seed=0, max_depth=0, iterative=False):
        xys = np.ascontiguousarray(xys, np.uint32).reshape(-1, 3)
        sc, _keep = scene.to_c()
        out = np.empty((len(xys), 3), np.float32)
        o = _abi.Options(0, 0, max_depth, seed, 0, 0)
        rc = self.lib.orc_test_paths(C.byref(sc), C.byref(o), C.c_uint32(len(xys)), _p(xys), _p(out), int(iterative))
        assert rc == 0
        return out

    def camera_basis(self, pos, look_at, up, fov, w, h):
        cam = _abi.Camera()
        f3 = lambda a: (C.c_float * 3)(*[float(x) for x in a])
        self.lib.orc_camera_basis(f3(pos), f3(look_at), f3(up), C.c_float(fov), C.c_uint32(w), C.c_uint32(h),
                                  C.byref(cam))
        return cam

    def autofocus(self, scene):
        sc, _keep = scene.to_c()
        return float(self.lib.orc_autofocus(C.byref(sc)))

    def reflect(self, v, n):
        v, n = np.ascontiguousarray(v, np.float32).reshape(-1, 3), np.ascontiguousarray(n, np.float32).reshape(-1, 3)
        out = np.empty_like(v)
        self.lib.orc_test_reflect(C.c_uint32(len(v)), _p(v), _p(n), _p(out))
        return out

    def refract(self, v, n, eta):
        v, n = np.ascontiguousarray(v, np.float32).reshape(-1, 3), np.ascontiguousarray(n, np.float32).reshape(-1, 3)
        eta = np.ascontiguousarray(eta, np.float32)
        out = np.empty((len(v), 4), np.float32)
        self.lib.orc_test_refract(C.c_uint32(len(v)), _p(v), _p(n), _p(eta), _p(out))
        return out[:, 0] > 0.5, out[:, 1:]

    def any_orthonormal(self, v):
        v = np.ascontiguousarray(v, np.float32).reshape(-1, 3)
        out = np.empty_like(v)
        self.lib.orc_test_any_orthonormal(C.c_uint32(len(v)), _p(v), _p(out))
        return out

    def aces(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        self.lib.orc_test_aces(C.c_uint32(x.size), _p(x), _p(out))
        return out
