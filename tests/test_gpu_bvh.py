"""GPU-built LBVH (many-sphere scenes, BASELINE C4) vs the brute-force object loop.

The requirement (north star): the LBVH returns the same closest hit as the reference's
brute-force loop (src/renderer.rs:356-362) — same t, same object, first-in-scene-order on ties.
Both GPU paths call the same sphere_t(), so the comparison is exact; the oracle pins the
brute-force path on a subset.
"""
import numpy as np
import pytest

from helpers import unit_vectors

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene20k():
    from pathtracer_amd.synthetic import synthetic_spheres
    return synthetic_spheres(n=20000, seed=0x5EED, width=256, height=144, samples=4)


def _camera_rays(hip, s, n, rng):
    xy = np.stack([rng.integers(0, s.width, n), rng.integers(0, s.height, n)], 1)
    return hip.camera_ray(s, xy, rng.random((n, 4), np.float32))


def test_bvh_matches_brute_force_exactly(hip, oracle, scene20k):
    rng = np.random.default_rng(0)
    n = 1 << 15
    ro, rd = _camera_rays(hip, scene20k, n, rng)
    # secondary-like rays too: origins inside the sphere cloud, random directions
    ro2 = np.stack([rng.uniform(-40, 40, n), rng.uniform(0.1, 12, n), rng.uniform(4, 84, n)], 1).astype(np.float32)
    ro, rd = np.concatenate([ro, ro2]), np.concatenate([rd, unit_vectors(rng, n)])
    t_b, n_b, i_b = hip.intersect(scene20k, ro, rd)          # brute force, global tables (blob > LDS budget)
    t_v, n_v, i_v = hip.intersect_bvh(scene20k, ro, rd)
    assert (i_b >= 1).mean() > 0.3                            # plenty of sphere hits (order 0 is the ground plane)
    assert np.array_equal(i_b, i_v)
    assert np.array_equal(t_b, t_v)
    assert np.array_equal(n_b, n_v)
    # and the brute-force kernel agrees with the CPU oracle on a subset
    k = 2048
    t_o, n_o, i_o = oracle.unit.intersect(scene20k, ro[:k], rd[:k])
    same = i_o == i_b[:k]
    assert same.mean() > 0.999
    hit = same & (i_o >= 0)
    # small far spheres: disc = h^2 - a c with h^2 ~ 600 and disc ~ r^2 ~ 0.04 is a cancellation on BOTH sides
    # (SURVEY §7.3), so a few roots only agree to ~1e-4
    e = np.abs(t_b[:k][hit] - t_o[hit]) / t_o[hit]
    assert np.median(e) < 1e-6 and np.quantile(e, 0.95) < 1e-4 and e.max() < 1e-3


def test_ploc_builder_gives_the_same_hits_and_the_same_image(hip, scene20k, monkeypatch):
    """The opt-in PLOC tree (HIPPT_BVH_BUILDER=ploc, pt_bvh_build.hip) only changes which boxes are visited: closest
    hits equal the brute-force loop's, and the queue pipeline's image equals the radix tree's bit for bit."""
    from pathtracer_amd import _abi, hip_renderer
    rng = np.random.default_rng(11)
    n = 1 << 14
    ro, rd = _camera_rays(hip, scene20k, n, rng)
    ro2 = np.stack([rng.uniform(-40, 40, n), rng.uniform(0.1, 12, n), rng.uniform(4, 84, n)], 1).astype(np.float32)
    ro, rd = np.concatenate([ro, ro2]), np.concatenate([rd, unit_vectors(rng, n)])
    t_b, n_b, i_b = hip.intersect(scene20k, ro, rd)
    s = scene20k.with_render(128, 72, 5)
    img_radix = hip_renderer.render_accum(s, seed=9, flags=_abi.FLAG_WAVEFRONT)
    for radius in ("2", "8", "40"):
        monkeypatch.setenv("HIPPT_BVH_BUILDER", "ploc")
        monkeypatch.setenv("HIPPT_BVH_PLOC_RADIUS", radius)
        t_v, n_v, i_v = hip.intersect_bvh(scene20k, ro, rd)
        assert np.array_equal(i_b, i_v) and np.array_equal(t_b, t_v) and np.array_equal(n_b, n_v), radius
        assert np.array_equal(hip_renderer.render_accum(s, seed=9, flags=_abi.FLAG_WAVEFRONT), img_radix), radius


def _coincident_scene():
    from pathtracer_amd import Scene
    from pathtracer_amd.scene import SPHERE_DTYPE
    s = Scene()
    s.width, s.height, s.samples = 16, 16, 1
    m = s.add_material((0.5, 0.5, 0.5))
    rng = np.random.default_rng(3)
    n = 300
    sp = np.zeros(n, SPHERE_DTYPE)
    sp["center"] = rng.uniform(-5, 5, (n, 3)) + [0, 0, 20]
    sp["radius"] = rng.uniform(0.2, 0.6, n)
    sp["center"][100:140] = [0.5, 0.25, 12.0]                 # 40 coincident spheres
    sp["radius"][100:140] = 1.0
    sp["material"] = m
    sp["order"] = rng.permutation(n).astype(np.uint32)        # scene order unrelated to table order
    s.spheres = sp
    s.add_light((0, 30, 0), (-1, 0, 0), (0, 0, -1), (1, 1, 1))
    s.set_camera((0, 0, 0), (0, 0, 1), (0, 1, 0), 60, 0.0)
    ro = np.zeros((64, 3), np.float32)
    rd = np.tile(np.array([0.5, 0.25, 12.0], np.float32) / np.linalg.norm([0.5, 0.25, 12.0]), (64, 1))
    rd[1:] += rng.normal(0, 0.02, (63, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    return s, sp, ro, rd


@pytest.mark.parametrize("builder", ["lbvh", "ploc"])
def test_bvh_tie_rule_and_duplicates(hip, monkeypatch, builder):
    """Coincident spheres (identical centre and radius, many of them, so identical Morton codes):
    the lowest scene order must win, as Iterator::min_by keeps the first minimum. Both tree builders."""
    monkeypatch.setenv("HIPPT_BVH_BUILDER", builder)
    s, sp, ro, rd = _coincident_scene()
    t_b, _, i_b = hip.intersect(s, ro, rd)
    t_v, _, i_v = hip.intersect_bvh(s, ro, rd)
    assert np.array_equal(i_b, i_v) and np.array_equal(t_b, t_v)
    assert i_b[0] == sp["order"][100:140].min()


def test_bvh_render_matches_brute_force_render(scene20k):
    """Whole integrator with the LBVH vs with the brute-force loop on the same samples."""
    from pathtracer_amd import _abi, hip_renderer
    s = scene20k.with_render(96, 54, 4)
    a, ca = hip_renderer.render_accum(s, seed=2, flags=_abi.FLAG_BVH, counters=True)
    b, cb = hip_renderer.render_accum(s, seed=2, flags=_abi.FLAG_NO_BVH, counters=True)
    assert np.isfinite(a).all() and a.mean() > 0
    # closest hits are identical; shadow rays use two formulations of the same test (origin folded out
    # vs explicit), which can disagree only on a threshold
    for k in ("paths", "extend_rays", "extend_hits", "nee_vertices", "diffuse_lobes", "spec_lobes"):
        assert abs(ca[k] - cb[k]) <= 1e-3 * cb[k] + 2, (k, ca[k], cb[k])
    err = np.abs(a - b).max(axis=2) / np.maximum(np.abs(b).max(axis=2), 1e-3)
    assert np.quantile(err, 0.99) < 1e-4
    assert ca["extend_sphere_tests"] < 0.02 * cb["extend_sphere_tests"]     # the point of the BVH


def test_small_scene_can_force_bvh(example_scene):
    from pathtracer_amd import _abi, hip_renderer
    s = example_scene.with_render(48, 36, 8)
    a = hip_renderer.render_accum(s, seed=1, flags=_abi.FLAG_BVH)
    b = hip_renderer.render_accum(s, seed=1)
    err = np.abs(a - b).max(axis=2) / np.maximum(np.abs(b).max(axis=2), 1e-3)
    assert np.quantile(err, 0.99) < 1e-4


def test_wavefront_bvh_pipeline_matches_bvh_megakernel(scene20k, oracle):
    """Queue pipeline with persistent LBVH traversal kernels (pt_wavefront_bvh.hip) vs the BVH megakernel:
    same device functions, same RNG streams and draw order -> the same image."""
    from pathtracer_amd import _abi, hip_renderer
    s = scene20k.with_render(160, 90, 6)
    a, ca = hip_renderer.render_accum(s, seed=5, flags=_abi.FLAG_BVH | _abi.FLAG_MEGAKERNEL, counters=True)
    b = hip_renderer.render_accum(s, seed=5, flags=_abi.FLAG_BVH | _abi.FLAG_WAVEFRONT)
    assert np.isfinite(b).all() and b.mean() > 0
    err = np.abs(a - b).max(axis=2) / np.maximum(np.abs(a).max(axis=2), 1e-3 * 6)
    assert np.quantile(err, 0.99) < 1e-4, np.quantile(err, [0.5, 0.9, 0.99, 1.0])
    assert abs(a.sum(dtype=np.float64) - b.sum(dtype=np.float64)) < 1e-4 * a.sum(dtype=np.float64)
    # sample ranges compose, uneven image size, more lights
    s2 = scene20k.with_render(75, 41, 5)
    s2.add_light((10, 25, 30), (6, 0, 0), (0, 0, 6), (30, 20, 10))
    full = hip_renderer.render_accum(s2, 0, 5, seed=2, flags=_abi.FLAG_WAVEFRONT)
    parts = (hip_renderer.render_accum(s2, 0, 2, seed=2, flags=_abi.FLAG_WAVEFRONT)
             + hip_renderer.render_accum(s2, 2, 3, seed=2, flags=_abi.FLAG_WAVEFRONT))
    assert np.allclose(full, parts, rtol=2e-6, atol=1e-6)
    mk = hip_renderer.render_accum(s2, 0, 5, seed=2, flags=_abi.FLAG_MEGAKERNEL)
    err = np.abs(full - mk).max(axis=2) / np.maximum(np.abs(mk).max(axis=2), 1e-3 * 5)
    assert np.quantile(err, 0.99) < 1e-4


def test_queue_pipeline_edge_cases(scene20k):
    """LBVH queue pipeline: no lights (black, no shadow rays), one sample, max_depth 1, and the automatic
    fallback to the megakernel when the path depth does not fit the queue entry's 5-bit depth field."""
    from pathtracer_amd import _abi, hip_renderer
    s = scene20k.with_render(40, 24, 2)
    dark = scene20k.with_render(40, 24, 2)
    dark.lights = dark.lights[:0]
    assert (hip_renderer.render_accum(dark) == 0).all()
    assert (hip_renderer.render_accum(dark, flags=_abi.FLAG_MEGAKERNEL) == 0).all()
    for kw in (dict(sample_count=1), dict(max_depth=1), dict(max_depth=40)):
        a = hip_renderer.render_accum(s, seed=3, **kw)
        b = hip_renderer.render_accum(s, seed=3, flags=_abi.FLAG_MEGAKERNEL, **kw)
        err = np.abs(a - b).max(axis=2) / np.maximum(np.abs(b).max(axis=2), 2e-3)
        assert np.isfinite(a).all() and np.quantile(err, 0.99) < 1e-4, kw
    with pytest.raises(_abi.HipptError):
        hip_renderer.render_accum(s, flags=_abi.FLAG_WAVEFRONT | _abi.FLAG_MEGAKERNEL)
    with pytest.raises(_abi.HipptError):
        hip_renderer.render_accum(s, flags=_abi.FLAG_WAVEFRONT, max_depth=40)


def test_c4_lbvh_equals_brute_force_on_the_first_2p20_camera_rays(hip):
    """BASELINE config 4 / SURVEY §8d: LBVH vs brute-force closest hit on the first 2^20 camera rays (pixels in
    row-major order, pixel centres, no lens offset) of the 100 000-sphere scene at 1920x1080: index, t and normal
    must be EQUAL (the tree only prunes). Also the traversal statistics of the queue pipeline on that scene."""
    from pathtracer_amd.synthetic import synthetic_spheres
    s = synthetic_spheres(100000, 0x5EED, 1920, 1080, 256)
    n_rays = 1 << 20
    i = np.arange(n_rays)
    xy = np.stack([i % s.width, i // s.width], 1)
    u4 = np.tile(np.array([0.5, 0.5, 0.0, 0.0], np.float32), (n_rays, 1))
    ro, rd = hip.camera_ray(s, xy, u4)
    t_v, n_v, i_v = hip.intersect_bvh(s, ro, rd)
    t_b, n_b, i_b = hip.intersect(s, ro, rd)
    assert (i_b >= 1).mean() > 0.2 and (i_b < 0).mean() > 0.05        # the upper half of the image: spheres and sky
    assert np.array_equal(i_b, i_v)
    assert np.array_equal(t_b, t_v)
    assert np.array_equal(n_b, n_v)


def test_traversal_stats_of_the_queue_pipeline(scene20k):
    import torch
    from pathtracer_amd import _abi, hip_renderer
    s = scene20k.with_render(128, 72, 4)
    ctx = hip_renderer.Context(0)
    ctx.set_scene(s)
    acc = torch.zeros(s.width * s.height * 3, dtype=torch.float32, device="cuda")
    ref = torch.zeros_like(acc)
    ctx.render_accum_device(acc.data_ptr(), 0, 4, seed=3, flags=_abi.FLAG_TRAVERSAL_STATS)
    ctx.render_accum_device(ref.data_ptr(), 0, 4, seed=3)
    st = ctx.read_traversal_stats()
    assert torch.equal(acc, ref)                                   # counting changes nothing
    _, cn = hip_renderer.render_accum(s, seed=3, counters=True)     # BVH megakernel, event counters
    assert st["closest"]["rays"] == cn["extend_rays"]
    assert 0 < st["any"]["rays"] <= cn["shadow_rays"]             # samples that face away get no shadow ray
    for k in ("closest", "any"):
        assert st[k]["nodes"] >= st[k]["rays"] and st[k]["leaf_fetches"] > 0
    assert st["any"]["aux_rows"] == 0
    assert ctx.read_traversal_stats()["closest"]["rays"] == 0      # read and zeroed


def test_traversal_stack_spills_to_global_memory(scene20k, monkeypatch):
    """The queue pipeline keeps at most 18 stack entries per ray in LDS (8 traversal blocks per CU) and the rest in a
    global spill column; HIPPT_WFB_LDS_STACK lowers the LDS part so that the 20 000-sphere scene really spills: the
    image must not change by a bit."""
    from pathtracer_amd import hip_renderer
    s = scene20k.with_render(160, 90, 4)
    ref = hip_renderer.render_accum(s, seed=7)
    monkeypatch.setenv("HIPPT_WFB_LDS_STACK", "2")
    spilled = hip_renderer.render_accum(s, seed=7)
    assert ref.mean() > 0 and np.array_equal(ref, spilled)


def test_queue_allocation_falls_back_to_fewer_streams_when_memory_is_short(scene20k, monkeypatch, capfd):
    """The LBVH pipeline sizes its queues for throughput (here 2^26 entries, ~26 GB, for a 64-spp launch at 1080p). With
    only 16 GiB of the GPU free the allocation fails and the library halves the number of sample streams until the
    queues fit: same image up to the order in which the streams' sums are added."""
    import torch
    from pathtracer_amd import hip_renderer
    s = scene20k.with_render(1920, 1080, 64)
    ref = hip_renderer.render_accum(s, seed=9)
    free, _total = torch.cuda.mem_get_info(0)
    hog = torch.empty(free - (16 << 30), dtype=torch.uint8, device="cuda")
    monkeypatch.setenv("HIPPT_DEBUG", "1")
    try:
        got = hip_renderer.render_accum(s, seed=9)
    finally:
        del hog
        torch.cuda.empty_cache()
    assert ref.mean() > 0 and np.allclose(got, ref, rtol=2e-5, atol=1e-4)
    log = capfd.readouterr().err          # the library reports the queues it ended up with
    assert "16 sample streams per pixel" in log, log
