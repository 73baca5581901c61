"""Parity at BASELINE.json's FULL configuration sizes (C2..C5).

A whole-image oracle render at these sizes would take hours, but every path is a pure function of
(scene, width, seed, x, y, global sample index) — the RNG stream is keyed by exactly those — so the
oracle can recompute any individual path of the full-size job (`orc_test_paths`). Each test renders
the full-size image on the GPU through the C ABI at the configuration's resolution and in the
configuration's range of global sample indices (few samples, so it runs in seconds), then checks a
random subset of pixels against the oracle path by path.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_pixels(oracle, scene, acc, s0, cnt, n_pix, seed, rng, q=(2e-6, 3e-5, 5e-3), max_depth=0, sum_tol=2e-4):
    h, w = scene.height, scene.width
    xs, ys = rng.integers(0, w, n_pix), rng.integers(0, h, n_pix)
    xys = np.stack([np.repeat(xs, cnt), np.repeat(ys, cnt), np.tile(np.arange(s0, s0 + cnt), n_pix)], 1)
    rad = oracle.paths(scene, xys, seed=seed, max_depth=max_depth, iterative=True).astype(np.float64).reshape(n_pix, cnt, 3)
    ref = rad.sum(axis=1)
    got = acc[ys, xs].astype(np.float64)
    assert np.isfinite(got).all()
    err = np.abs(got - ref).max(axis=1)
    scale = np.maximum(np.abs(ref).max(axis=1), 1e-3 * cnt)
    r = err / scale
    # same bounds as tests/test_gpu_render_parity.py::_compare: rounding-level agreement except where a
    # last-bit difference flips a branch or lands on a roughness-0.01 highlight
    qs = [float(np.quantile(r, p)) for p in (0.5, 0.9, 0.99)]
    assert qs[0] < q[0] and qs[1] < q[1] and qs[2] < q[2], qs
    tot_ref, tot_got = ref.sum(), got.sum()
    keep = r <= np.quantile(r, 0.995)
    assert abs(got[keep].sum() - ref[keep].sum()) <= sum_tol * ref[keep].sum() + 1e-6, (tot_got, tot_ref)
    return (ref > 0).any(axis=1).mean()


def test_c2_1080p_paths_match_oracle(oracle):
    """C2: scene.example.json at 1920x1080, samples at the far end of the 1024-spp range."""
    from pathtracer_amd import Scene, hip_renderer
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 1920, 1080, 1024)
    acc = hip_renderer.render_accum(s, 1016, 8, seed=0)
    lit = _check_pixels(oracle, s, acc, 1016, 8, 3000, 0, np.random.default_rng(2))
    assert lit > 0.9
    # launches are deterministic
    assert np.array_equal(acc, hip_renderer.render_accum(s, 1016, 8, seed=0))


def test_c3_glass_fog_1080p_rank7_range(oracle):
    """C3: glass + fog at 1920x1080, 4096 spp split over 8 GPUs: the last samples of rank 7's range."""
    from pathtracer_amd import Scene, hip_renderer
    from pathtracer_amd.partition import sample_range
    s = Scene.load(os.path.join(ROOT, "scenes", "glass_fog.json"), 1920, 1080, 4096)
    b, c = sample_range(4096, 7, 8)
    assert (b, c) == (3584, 512)
    acc = hip_renderer.render_accum(s, b + c - 6, 6, seed=0)
    _check_pixels(oracle, s, acc, b + c - 6, 6, 2500, 0, np.random.default_rng(3))


def test_c4_100k_spheres_lbvh_1080p(oracle):
    """C4: the 100 000-sphere scene at 1920x1080 through the LBVH kernel vs the oracle's brute-force loop.

    Hit records are bit-identical to the reference's (tests/test_gpu_unit_parity.py), but a path that
    bounces among spheres of radius 0.05-0.35 at distances of 5-50 amplifies a last-bit difference in a
    sampled direction by ~distance/radius per bounce, so only the first vertex (camera hit + next-event
    estimation through the LBVH) is compared pixel by pixel with the oracle; complete paths are compared with the
    oracle through the paired mean over the sampled pixels, and pixel by pixel with the same kernel arithmetic
    run over the brute-force loops (part c), which separates the tree from the amplification."""
    from pathtracer_amd.synthetic import synthetic_spheres
    from pathtracer_amd import hip_renderer
    s = synthetic_spheres(100000, 0x5EED, 1920, 1080, 256)
    rng = np.random.default_rng(4)
    # (a) one vertex: camera hit + NEE (shadow rays through the LBVH)
    acc, cn = hip_renderer.render_accum(s, 254, 2, seed=0, max_depth=1, counters=True)
    assert cn["paths"] == 2 * 1920 * 1080
    assert cn["extend_sphere_tests"] < 1e-3 * cn["extend_rays"] * 100000  # the LBVH prunes
    # 99th percentile: a shadow ray through 100 000 spheres grazes one often enough that ~1 % of the pixels
    # have one of their 8 shadow samples decided on a threshold
    lit = _check_pixels(oracle, s, acc, 254, 2, 1200, 0, rng, q=(2e-6, 3e-5, 5e-2), max_depth=1, sum_tol=2e-3)
    assert lit > 0.1
    # (a2) TWO vertices (max_depth = 2: the first bounce and its next-event estimation), pixel by pixel: the bounce
    # direction differs from the oracle's in its last bits (v_sin / v_cos / v_rsq against libm), which a second hit on a
    # sphere of radius 0.05-0.35 amplifies by ~distance / radius; half the pixels still agree exactly and five in six to
    # 1e-4 (measured, tools/diag_c4_depth2.py: median 0, 83.8 % within 1e-4, 90th percentile 1.3e-3; depth 3: 77 %)
    acc2 = hip_renderer.render_accum(s, 254, 2, seed=0, max_depth=2)
    n2 = 2000
    xs, ys = rng.integers(0, 1920, n2), rng.integers(0, 1080, n2)
    xys = np.stack([np.repeat(xs, 2), np.repeat(ys, 2), np.tile(np.arange(254, 256), n2)], 1)
    ref2 = oracle.paths(s, xys, seed=0, max_depth=2, iterative=True).astype(np.float64).reshape(n2, 2, 3).sum(axis=1)
    got2 = acc2[ys, xs].astype(np.float64)
    r2 = np.abs(got2 - ref2).max(axis=1) / np.maximum(np.abs(ref2).max(axis=1), 2e-3)
    assert np.median(r2) < 1e-6 and (r2 < 1e-4).mean() > 0.78 and np.quantile(r2, 0.9) < 1e-2, \
        (np.median(r2), (r2 < 1e-4).mean(), np.quantile(r2, 0.9))
    # (b) complete paths (max depth 5): most pixels still agree to rounding, and the differences of the
    # rest are unbiased: |mean(got - ref)| within 4 standard errors of the paired difference
    acc = hip_renderer.render_accum(s, 254, 2, seed=0)
    n = 1500
    xs, ys = rng.integers(0, 1920, n), rng.integers(0, 1080, n)
    xys = np.stack([np.repeat(xs, 2), np.repeat(ys, 2), np.tile(np.arange(254, 256), n)], 1)
    ref = oracle.paths(s, xys, seed=0, iterative=True).astype(np.float64).reshape(n, 2, 3).sum(axis=(1, 2))
    got = acc[ys, xs].astype(np.float64).sum(axis=1)
    d = got - ref
    close = np.abs(d) <= 1e-4 * np.maximum(np.abs(ref), 2e-3)
    assert close.mean() > 0.6, close.mean()       # measured 0.706 (0.64 within 1e-6): tools/diag_c4_paths.py
    assert abs(d.mean()) <= 4.0 * d.std() / np.sqrt(n) + 1e-9, (d.mean(), d.std() / np.sqrt(n))
    # (c) that the remaining differences come from the sampled directions' last bits and not from the tree: the
    # same kernel arithmetic with the brute-force loops (no LBVH, 100 000 sphere tests per ray) renders the
    # same complete paths at the full size, pixel by pixel
    from pathtracer_amd import _abi
    one = hip_renderer.render_accum(s, 255, 1, seed=0)
    brute = hip_renderer.render_accum(s, 255, 1, seed=0, flags=_abi.FLAG_NO_BVH)
    e = np.abs(one.astype(np.float64) - brute).max(axis=2) / np.maximum(np.abs(brute).max(axis=2), 1e-3)
    assert np.quantile(e, 0.99) < 1e-5 and (e < 1e-4).mean() > 0.995, (np.quantile(e, [0.5, 0.99, 0.999]), (e < 1e-4).mean())
    tot = brute.sum(dtype=np.float64)
    assert tot > 0 and abs(one.sum(dtype=np.float64) - tot) <= 1e-3 * tot, (one.sum(dtype=np.float64), tot)


def test_c5_8k_showcase_sample_indices(oracle):
    """C5: 7680x4320 at aperture 0.02, global sample indices at the end of the 131072-spp job."""
    from pathtracer_amd import Scene, hip_renderer
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 7680, 4320, 131072)
    assert abs(s.cam_aperture - 0.02) < 1e-6
    acc = hip_renderer.render_accum(s, 131070, 2, seed=0)
    assert acc.shape == (4320, 7680, 3)
    _check_pixels(oracle, s, acc, 131070, 2, 4000, 0, np.random.default_rng(5))
    # the split used by 8 ranks composes at this size too: [131070,131071) + [131071,131072)
    parts = hip_renderer.render_accum(s, 131070, 1, seed=0) + hip_renderer.render_accum(s, 131071, 1, seed=0)
    assert np.allclose(acc, parts, rtol=2e-6, atol=1e-6)
