"""BASELINE config C4 check: LBVH vs brute-force closest hit on the first 2^20 camera rays of the
100k-sphere scene (run on the GPU box). Prints agreement and the two kernels' wall times."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import hip_renderer  # noqa: E402
from pathtracer_amd.synthetic import synthetic_spheres  # noqa: E402


def main():
    n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    s = synthetic_spheres(100000, 0x5EED, 1920, 1080, 256)
    hip = hip_renderer.unit_ops(0)
    i = np.arange(n_rays)
    xy = np.stack([i % s.width, i // s.width], 1)
    u4 = np.tile(np.array([0.5, 0.5, 0.0, 0.0], np.float32), (n_rays, 1))
    ro, rd = hip.camera_ray(s, xy, u4)
    t0 = time.perf_counter()
    tv, nv, iv = hip.intersect_bvh(s, ro, rd)
    t1 = time.perf_counter()
    tb, nb, ib = hip.intersect(s, ro, rd)
    t2 = time.perf_counter()
    print(f"rays {n_rays}: index equal {np.array_equal(ib, iv)}, t equal {np.array_equal(tb, tv)}, normal equal "
          f"{np.array_equal(nb, nv)}; sphere hits {(ib >= 1).mean():.3f}, plane hits {(ib == 0).mean():.3f}; "
          f"LBVH call {t1 - t0:.2f} s (incl. build + copies), brute force call {t2 - t1:.2f} s")
    assert np.array_equal(ib, iv) and np.array_equal(tb, tv) and np.array_equal(nb, nv)


if __name__ == "__main__":
    main()
