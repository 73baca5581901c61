"""Differential stress of the LBVH paths against the brute-force kernel (same arithmetic, no tree) on scenes chosen to
strain the tree's half-float box format and the queue pipeline's bookkeeping: far-away, tiny and huge scenes, flat and
elongated clouds, a clump of coincident spheres, radii over three decades, several lights (vertex-major shadow-ray
list), planes in between, odd image sizes and sample counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

def cloud(seed, n, centre, extent, r_lo, r_hi, lights=1, planes=0, w=97, h=61, spp=7, flat_axis=None, cam=1.3):
    from pathtracer_amd import Scene
    from pathtracer_amd.scene import SPHERE_DTYPE
    rng = np.random.default_rng(seed)
    s = Scene()
    s.width, s.height, s.samples = w, h, spp
    mats = [s.add_material(tuple(rng.uniform(0.2, 0.9, 3)), roughness=float(rng.uniform(0.2, 1.0)),
                           metallic=float(rng.choice([0.0, 1.0], p=[0.7, 0.3]))) for _ in range(5)]
    c = np.asarray(centre, np.float64)
    e = np.asarray(extent, np.float64)
    size = float(e.max())
    for k in range(planes):                                   # scene order 0 .. planes-1, the spheres follow
        s.add_plane(tuple(c + [0, -0.6 * e[1] - 0.1 * size * k, 0]), (size, 0, 0), (0, 0, -size), mats[k % 5])
    sp = np.zeros(n, SPHERE_DTYPE)
    pos = rng.uniform(-0.5, 0.5, (n, 3)) * e
    if flat_axis is not None:
        pos[:, flat_axis] = 0.0
    pos[: n // 16] = pos[0]                                   # a clump of coincident spheres
    sp["center"] = (pos + c).astype(np.float32)
    sp["radius"] = rng.uniform(r_lo, r_hi, n).astype(np.float32)
    sp["material"] = rng.integers(0, len(mats), n)
    sp["order"] = rng.permutation(n).astype(np.uint32) + planes
    s.spheres = sp
    for k in range(lights):
        s.add_light(tuple(c + [0.3 * size * (k - 1), 0.9 * size, 0.2 * size * k]), (-0.3 * size, 0, 0), (0, 0, -0.3 * size),
                    tuple(rng.uniform(20, 60, 3) * size * size / 100.0))
    s.set_camera(tuple(c + [0.1 * size * cam, 0.4 * size * cam, -cam * size]), tuple(c), (0, 1, 0), 55.0, 0.0)
    return s


CASES = {
    "unit cloud": dict(seed=1, n=5000, centre=(0, 0, 0), extent=(10, 10, 10), r_lo=0.05, r_hi=0.4),
    "far from the origin": dict(seed=2, n=4000, centre=(9000, -7000, 12000), extent=(10, 10, 10), r_lo=0.05, r_hi=0.4),
    "tiny scene": dict(seed=3, n=3000, centre=(0, 0, 0), extent=(1e-2, 1e-2, 1e-2), r_lo=5e-5, r_hi=4e-4),
    "huge scene": dict(seed=4, n=3000, centre=(0, 0, 0), extent=(1e5, 1e5, 1e5), r_lo=500, r_hi=4000),
    "flat slab": dict(seed=5, n=6000, centre=(0, 0, 0), extent=(60, 1, 60), r_lo=0.1, r_hi=0.5, flat_axis=1),
    "needle": dict(seed=6, n=4000, centre=(0, 0, 0), extent=(0.5, 0.5, 200), r_lo=0.05, r_hi=0.3, cam=0.05),
    "mixed radii": dict(seed=7, n=5000, centre=(0, 0, 0), extent=(20, 20, 20), r_lo=0.001, r_hi=1.5, cam=0.9),
    "three lights, two planes": dict(seed=8, n=5000, centre=(0, 0, 0), extent=(10, 4, 10), r_lo=0.05, r_hi=0.4, lights=3, planes=2),
    "five lights, odd sizes": dict(seed=9, n=2000, centre=(3, 1, 9), extent=(8, 3, 8), r_lo=0.05, r_hi=0.5, lights=5, planes=1, w=131, h=37, spp=13),
    "64 spheres": dict(seed=10, n=64, centre=(0, 0, 0), extent=(4, 4, 4), r_lo=0.1, r_hi=0.6, lights=2),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_lbvh_paths_equal_brute_force(name):
    from pathtracer_amd import _abi, hip_renderer
    s = cloud(**CASES[name])
    a = hip_renderer.render_accum(s, seed=3)                                   # LBVH queue pipeline
    b = hip_renderer.render_accum(s, seed=3, flags=_abi.FLAG_NO_BVH)            # brute-force loops
    m = hip_renderer.render_accum(s, seed=3, flags=_abi.FLAG_MEGAKERNEL)        # LBVH megakernel
    assert np.isfinite(a).all() and np.isfinite(m).all() and b.mean() > 0
    scale = np.maximum(np.abs(b).max(axis=2), 1e-3 * b.max())
    for got in (a, m):
        # the three kernels inline the same device functions into different loops: a pixel here and there differs by a
        # flipped branch; everything else agrees to rounding
        e = np.abs(got.astype(np.float64) - b).max(axis=2) / scale
        assert np.quantile(e, 0.98) < 1e-5, (name, np.quantile(e, [0.5, 0.98, 1.0]))
        tot = b.sum(dtype=np.float64)
        assert abs(got.sum(dtype=np.float64) - tot) <= 1e-3 * tot


@pytest.mark.parametrize("name", ["far from the origin", "flat slab", "needle", "mixed radii", "huge scene", "64 spheres"])
def test_ploc_tree_paths_equal_brute_force(name, monkeypatch):
    """The same comparison with the opt-in PLOC tree (HIPPT_BVH_BUILDER=ploc): other boxes, other depth, same image."""
    from pathtracer_amd import _abi, hip_renderer
    s = cloud(**CASES[name])
    b = hip_renderer.render_accum(s, seed=3, flags=_abi.FLAG_NO_BVH)
    monkeypatch.setenv("HIPPT_BVH_BUILDER", "ploc")
    scale = np.maximum(np.abs(b).max(axis=2), 1e-3 * b.max())
    for flags in (0, _abi.FLAG_MEGAKERNEL):
        got = hip_renderer.render_accum(s, seed=3, flags=flags)
        e = np.abs(got.astype(np.float64) - b).max(axis=2) / scale
        assert np.isfinite(got).all() and np.quantile(e, 0.98) < 1e-5, (name, flags, np.quantile(e, [0.5, 0.98, 1.0]))
        tot = b.sum(dtype=np.float64)
        assert abs(got.sum(dtype=np.float64) - tot) <= 1e-3 * tot


def test_more_lights_than_the_queue_pipeline_takes():
    """100 lights: the jump-ahead table of the queue pipeline holds 96, so the library renders the scene with the
    one-kernel LBVH variant by itself (and refuses an explicit HIPPT_FLAG_WAVEFRONT); image == brute force."""
    from pathtracer_amd import _abi, hip_renderer
    s = cloud(seed=11, n=500, centre=(0, 0, 0), extent=(6, 3, 6), r_lo=0.1, r_hi=0.5, lights=100, w=40, h=24, spp=2)
    a = hip_renderer.render_accum(s, seed=3)
    b = hip_renderer.render_accum(s, seed=3, flags=_abi.FLAG_NO_BVH)
    e = np.abs(a.astype(np.float64) - b).max(axis=2) / np.maximum(np.abs(b).max(axis=2), 1e-3 * b.max())
    assert b.mean() > 0 and np.quantile(e, 0.98) < 1e-5
    with pytest.raises(_abi.HipptError, match="96 lights"):
        hip_renderer.render_accum(s, seed=3, flags=_abi.FLAG_WAVEFRONT)
