"""HIP integrator vs the CPU oracle with the shared counter-seeded RNG.

Both sides draw the same uniforms in the same order, so the comparison is
sample-for-sample, not statistical: pixels agree to rounding except where a
last-bit difference flips a branch (hit/miss at a grazing ray, u < R at a glass
vertex, Russian roulette), which changes that one sample's path.
Tolerances: BASELINE.json north_star — mean radiance within 1 %. Held here: 0.5 % on the raw mean of the
example scene at 64 spp (a handful of ill-conditioned mirror highlights carry the difference, see _compare),
0.01 % with the 0.5 % most different pixels left out and 0.01 % on the raw mean of the matte variant;
per-pixel differences at/below rounding for the quantiles _compare names.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compare(acc_h, acc_o, spp, q50=2e-6, q90=3e-5, q99=5e-3, mean_tol=5e-3):
    """Per-pixel |hip - oracle| relative to the pixel's radiance sum: quantile bounds.

    scene.example.json has mirror-like metals (roughness 0.01): there D = a2/(pi q^2) with
    q = nh^2 (a2 - 1) + 1 ~ 1e-4 is a cancellation that amplifies last-bit differences in n.h
    by ~1e4 on BOTH sides, so highlight pixels agree only to ~1e-3..1e-2; every other pixel
    agrees to rounding unless a branch flipped (grazing hit/miss, u < R, roulette)."""
    mean_h, mean_o = acc_h.mean(axis=(0, 1)) / spp, acc_o.mean(axis=(0, 1)) / spp
    # north star: mean radiance within 1 %. Held: 0.5 % on the raw mean — at these small sample counts
    # > 90 % of the total difference sits in ~10 highlight samples of radiance ~200 whose sign is random
    # (measured over 8 seeds: -1.3e-3 .. +3.1e-3) — and 1e-4 once the 0.5 % most different pixels are left out.
    assert np.all(np.abs(mean_h - mean_o) <= mean_tol * np.abs(mean_o) + 1e-7), (mean_h, mean_o)
    dsum = (acc_h.astype(np.float64) - acc_o).sum(axis=2)
    keep = np.abs(dsum) <= np.quantile(np.abs(dsum), 0.995)
    assert abs(dsum[keep].sum()) <= 1e-4 * acc_o.astype(np.float64).sum(axis=2)[keep].sum()
    err = np.abs(acc_h.astype(np.float64) - acc_o).max(axis=2)
    scale = np.maximum(np.abs(acc_o).max(axis=2), 1e-3 * spp)
    r = err / scale
    got = [float(np.quantile(r, q)) for q in (0.5, 0.9, 0.99)]
    assert got[0] < q50 and got[1] < q90 and got[2] < q99, got
    return got


def test_example_scene_sample_exact(oracle):
    from pathtracer_amd import Scene, hip_renderer
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 96, 72, 64)
    acc_h, cn_h = hip_renderer.render_accum(s, seed=0, counters=True)
    acc_o, cn_o = oracle.render_accum(s, seed=0, counters=True)
    assert np.isfinite(acc_h).all()
    _compare(acc_h, acc_o, 64)
    # event counters: same paths, same events up to the few flipped branches
    assert cn_h["paths"] == cn_o["paths"] == 96 * 72 * 64
    for k in ("extend_rays", "extend_hits", "shadow_rays", "nee_vertices", "lit_evals", "diffuse_lobes",
              "spec_lobes", "glass_vertices", "extend_sphere_tests", "extend_plane_tests"):
        assert abs(cn_h[k] - cn_o[k]) <= 2e-3 * cn_o[k] + 8, (k, cn_h[k], cn_o[k])
    assert cn_h["medium_vertices"] == 0 and cn_h["scatter_events"] == 0
    # shadow tests are counted as the reference executes them (scene order, up to the first occluder), whatever
    # the kernel itself runs: the algorithmic FLOP count must not depend on the kernel's shortcuts
    for k in ("shadow_sphere_tests", "shadow_plane_tests"):
        assert abs(cn_h[k] - cn_o[k]) <= 2e-3 * cn_o[k] + 8, (k, cn_h[k], cn_o[k])


def test_matte_scene_tight(oracle):
    """Same geometry with every roughness raised to >= 0.5: no ill-conditioned highlights, so
    pixels must agree to rounding except for flipped branches."""
    from pathtracer_amd import Scene, hip_renderer
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 96, 72, 32)
    s.materials["roughness"] = np.maximum(s.materials["roughness"], 0.5)
    acc_h = hip_renderer.render_accum(s, seed=4)
    acc_o = oracle.render_accum(s, seed=4)
    _compare(acc_h, acc_o, 32, q50=2e-6, q90=1e-5, q99=1e-3, mean_tol=1e-4)


def test_tables_global_variant_matches_lds(oracle):
    from pathtracer_amd import Scene, hip_renderer, _abi
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 64, 48, 16)
    a = hip_renderer.render_accum(s, seed=3)
    b = hip_renderer.render_accum(s, seed=3, flags=_abi.FLAG_TABLES_GLOBAL)
    # same arithmetic, but the two instantiations are contracted/scheduled independently by hipcc
    _compare(a, b, 16)


def test_sample_slices_change_only_the_summation_order():
    """A launch cuts its sample range into slices rendered by separate blocks (partial sums added in a fixed
    order): same paths, so the same image up to f32 summation order, and still deterministic."""
    from pathtracer_amd import Scene, hip_renderer, _abi
    for w, h, spp in ((64, 48, 37), (1920, 1080, 5)):
        s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), w, h, spp)
        a, ca = hip_renderer.render_accum(s, seed=4, counters=True)
        b, cb = hip_renderer.render_accum(s, seed=4, counters=True, flags=_abi.FLAG_NO_SLICES)
        assert ca == cb
        assert np.allclose(a, b, rtol=3e-6, atol=1e-6)
        # the plain kernel (no counters) defers part of the shadow rays and adds what they carry to the pixel
        # sums separately (Tracer::direct_light_shared): same rays, same decisions, another order of additions
        c = hip_renderer.render_accum(s, seed=4)
        _compare(a, c, spp)  # two instantiations, contracted/scheduled independently by hipcc
        assert np.array_equal(c, hip_renderer.render_accum(s, seed=4))


def test_sample_ranges_add_up():
    """Sample-range split (multi-GPU partitioning): [0,24) + [24,64) == [0,64) up to f32 summation order."""
    from pathtracer_amd import Scene, hip_renderer
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 64, 48, 64)
    full = hip_renderer.render_accum(s, 0, 64, seed=5)
    parts = hip_renderer.render_accum(s, 0, 24, seed=5) + hip_renderer.render_accum(s, 24, 40, seed=5)
    assert np.allclose(full, parts, rtol=2e-6, atol=1e-6)
    # and a different seed gives a different image
    other = hip_renderer.render_accum(s, 0, 64, seed=6)
    assert not np.allclose(full, other, rtol=1e-3)


def test_glass_fog_scene(oracle):
    from pathtracer_amd import Scene, hip_renderer
    s = Scene.load(os.path.join(ROOT, "scenes", "glass_fog.json"), 96, 72, 32)
    acc_h, cn_h = hip_renderer.render_accum(s, seed=1, counters=True)
    acc_o, cn_o = oracle.render_accum(s, seed=1, counters=True)
    assert cn_o["scatter_events"] > 1000 and cn_o["medium_vertices"] > cn_o["scatter_events"]
    _compare(acc_h, acc_o, 32)
    for k in ("medium_vertices", "scatter_events", "glass_vertices", "shadow_rays"):
        assert abs(cn_h[k] - cn_o[k]) <= 5e-3 * cn_o[k] + 8, (k, cn_h[k], cn_o[k])


def test_rgba8_matches_oracle_quantisation(oracle):
    from pathtracer_amd import Scene, hip_renderer
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 64, 48, 32)
    img_h = hip_renderer.render(s, seed=2)
    assert img_h.shape == (48, 64, 4) and (img_h[..., 3] == 255).all()
    acc_h = hip_renderer.render_accum(s, seed=2)
    img_o = oracle.resolve(acc_h, 32)  # oracle quantisation of the same sums
    d = np.abs(img_h.astype(np.int32) - img_o.astype(np.int32))
    assert d.max() <= 1 and (d == 0).mean() > 0.999


def test_zero_lights_renders_black():
    """CPU path semantics (renderer.rs:31): no lights -> black, not an error (cf. gpu_renderer.rs:116)."""
    from pathtracer_amd import Scene, hip_renderer
    import numpy as np
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 32, 24, 4)
    s.lights = s.lights[:0]
    acc = hip_renderer.render_accum(s)
    assert (acc == 0).all()
    img = hip_renderer.render(s)
    assert (img[..., :3] == 0).all()


def test_empty_scene_and_ragged_size():
    """No objects at all, and an image size that is not a multiple of the 16x16 tile."""
    from pathtracer_amd import Scene, hip_renderer
    s = Scene()
    s.width, s.height, s.samples = 37, 23, 3
    s.add_material((0.5, 0.5, 0.5))
    s.add_light((0, 5, 0), (1, 0, 0), (0, 0, 1), (1, 1, 1))
    s.set_camera((0, 0, 0), (0, 0, 1), (0, 1, 0), 60, 0.01)
    acc = hip_renderer.render_accum(s)
    assert acc.shape == (23, 37, 3) and (acc == 0).all()


def test_wavefront_variant_matches_megakernel(oracle):
    """The queue-based pipeline (HIPPT_FLAG_WAVEFRONT) runs the same device code with the same RNG streams."""
    from pathtracer_amd import Scene, hip_renderer, _abi
    # (96 / 64 samples: at the 24 / 12 of round 2 ONE roughness-0.01 highlight sample — radiance ~100, and D amplifies a
    # last-bit difference in n.h 1e4 times — was 0.7-1.1 % of these small images' totals)
    for name, spp in (("scene.example.json", 96), ("glass_fog.json", 64)):
        s = Scene.load(os.path.join(ROOT, "scenes", name), 80, 60, spp)
        a = hip_renderer.render_accum(s, seed=9)
        b = hip_renderer.render_accum(s, seed=9, flags=_abi.FLAG_WAVEFRONT)
        assert np.isfinite(b).all()
        _compare(b, a, spp)
        _compare(b, oracle.render_accum(s, seed=9), spp)
    # sample ranges compose
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 40, 30, 10)
    full = hip_renderer.render_accum(s, 0, 10, seed=1, flags=_abi.FLAG_WAVEFRONT)
    parts = (hip_renderer.render_accum(s, 0, 3, seed=1, flags=_abi.FLAG_WAVEFRONT)
             + hip_renderer.render_accum(s, 3, 7, seed=1, flags=_abi.FLAG_WAVEFRONT))
    assert np.allclose(full, parts, rtol=2e-6, atol=1e-6)


def test_independent_seeds_agree_at_the_noise_floor(oracle):
    """BASELINE north star, statistical form: GPU and CPU images from DIFFERENT random streams agree in mean radiance
    (within 1 %) and differ per pixel by what Monte-Carlo noise predicts — the RMS of (gpu - cpu) / sigma is 1, where
    sigma^2 is the sum of the two images' per-pixel variances of the mean, estimated from 16 batches of 32 samples.
    (The same-seed tests above compare sample for sample; this one would also catch an error common to both sides'
    use of the shared stream, such as correlated draws.)"""
    from pathtracer_amd import Scene, hip_renderer
    for name, mean_tol in (("scene.example.json", 1e-2), ("glass_fog.json", 1e-2)):
        s = Scene.load(os.path.join(ROOT, "scenes", name), 160, 120, 512)
        s.materials["roughness"] = np.maximum(s.materials["roughness"], 0.2)   # no roughness-0.01 fireflies in a variance estimate
        nb, per = 16, 32
        g = np.stack([hip_renderer.render_accum(s, b * per, per, seed=101).astype(np.float64) / per for b in range(nb)])
        c = np.stack([oracle.render_accum(s, b * per, per, seed=202).astype(np.float64) / per for b in range(nb)])
        mg, mc = g.mean(axis=0), c.mean(axis=0)
        tot_g, tot_c = mg.sum(), mc.sum()
        assert abs(tot_g - tot_c) <= mean_tol * tot_c, (name, tot_g, tot_c)
        var = g.var(axis=0, ddof=1) / nb + c.var(axis=0, ddof=1) / nb
        lit = var > 1e-12
        z = (mg - mc)[lit] / np.sqrt(var[lit])
        # Student-like tails (variance from 16 batches): compare the central 98 % with what a unit normal gives (0.93)
        zc = z[np.abs(z) < np.quantile(np.abs(z), 0.98)]
        rms = float(np.sqrt((zc ** 2).mean()))
        assert 0.8 < rms < 1.15, (name, rms)
        assert abs(float(np.median(z))) < 0.05, (name, float(np.median(z)))      # no systematic offset
