"""Diagnostic: |LBVH pipeline - oracle| per pixel on C4 with paths cut after TWO vertices (max_depth = 2)."""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_lib
from pathtracer_amd import hip_renderer
from pathtracer_amd.synthetic import synthetic_spheres
o = oracle_lib.Oracle()
s = synthetic_spheres(100000, 0x5EED, 1920, 1080, 256)
rng = np.random.default_rng(14)
for depth in (1, 2, 3):
    acc = hip_renderer.render_accum(s, 254, 2, seed=0, max_depth=depth)
    n = 3000
    xs, ys = rng.integers(0, 1920, n), rng.integers(0, 1080, n)
    xys = np.stack([np.repeat(xs, 2), np.repeat(ys, 2), np.tile(np.arange(254, 256), n)], 1)
    ref = o.paths(s, xys, seed=0, max_depth=depth, iterative=True).astype(np.float64).reshape(n, 2, 3).sum(axis=1)
    got = acc[ys, xs].astype(np.float64)
    r = np.abs(got - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 2e-3)
    print('depth', depth, 'quantiles 50/90/95/99', [float(np.quantile(r, q)) for q in (0.5, 0.9, 0.95, 0.99)], 'within 1e-4', (r < 1e-4).mean(), 'sum rel', (got.sum() - ref.sum()) / ref.sum())
