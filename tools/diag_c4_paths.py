"""Diagnostic: distribution of |LBVH kernel - oracle| over complete C4 paths (used to set the bounds in
tests/test_gpu_full_size.py::test_c4_100k_spheres_lbvh_1080p)."""
import sys, time, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_lib
from pathtracer_amd import hip_renderer
from pathtracer_amd.synthetic import synthetic_spheres
o = oracle_lib.Oracle()
s = synthetic_spheres(100000, 0x5EED, 1920, 1080, 256)
rng = np.random.default_rng(4)
for cnt in (2, 8):
    acc = hip_renderer.render_accum(s, 256 - cnt, cnt, seed=0)
    n = 4000
    xs, ys = rng.integers(0, 1920, n), rng.integers(0, 1080, n)
    xys = np.stack([np.repeat(xs, cnt), np.repeat(ys, cnt), np.tile(np.arange(256 - cnt, 256), n)], 1)
    t = time.time()
    ref = o.paths(s, xys, seed=0, iterative=True).astype(np.float64).reshape(n, cnt, 3).sum(axis=(1, 2))
    print('oracle s', time.time() - t)
    got = acc[ys, xs].astype(np.float64).sum(axis=1)
    d = got - ref
    for tol in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2):
        print(cnt, tol, (np.abs(d) <= tol * np.maximum(np.abs(ref), 1e-3 * cnt)).mean())
    print(cnt, 'mean rel', d.sum() / ref.sum(), 'se', d.std() / np.sqrt(n) / ref.mean(), 'ref mean', ref.mean())
