"""Known-answer tests that pin the CPU oracle to closed forms derived from the
reference's source lines. The reference has no tests or golden
vectors of its own ("parity unpinned"), so these are the pins.
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PI = np.pi


def _scene(w=8, h=8, spp=1):
    from pathtracer_amd import Scene
    s = Scene()
    s.width, s.height, s.samples = w, h, spp
    return s


def f32(*a):
    return np.array(a, np.float32)


# ---- sphere.rs:20-39 -----------------------------------------------------------
def test_sphere_hits(oracle):
    s = _scene()
    m = s.add_material((0.5, 0.5, 0.5))
    s.add_sphere((0, 0, 5), 1.0, m)
    ro = f32([0, 0, 0], [0, 0, 5], [0, 2, 0], [0, 0, 10], [0, 0, 0])
    rd = f32([0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 2])
    t, n, idx = oracle.unit.intersect(s, ro, rd)
    assert t[0] == 4.0 and list(n[0]) == [0, 0, -1] and idx[0] == 0
    assert idx[1] == -1                      # origin inside: near root negative (quirk Q1)
    assert idx[2] == -1                      # disc < 0
    assert idx[3] == -1                      # sphere behind the ray
    assert t[4] == 2.0 and idx[4] == 0       # un-normalised direction: t scales with 1/|rd|


# ---- plane.rs:19-56 --------------------------------------------------------------
def test_plane_hits(oracle):
    s = _scene()
    m = s.add_material((0.5, 0.5, 0.5))
    s.add_plane((0, 0, 0), (1, 0, 0), (0, 0, 1), m)
    assert np.allclose(s.planes["normal"][0], [0, -1, 0])      # u x v (scene.rs:139)
    ro = f32([0, 1, 0], [1.0, 1, 0], [1.01, 1, 0], [0, 1, 1.0], [0, 1, -1.01], [0, 1, 0], [0, 5e-5, 0], [0, -1, 0],
             [0, 1, 0])
    rd = f32([0, -1, 0], [0, -1, 0], [0, -1, 0], [0, -1, 0], [0, -1, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0],
             [0, 1, 0])
    t, n, idx = oracle.unit.intersect(s, ro, rd)
    assert t[0] == 1.0 and list(n[0]) == [0, 1, 0]             # denom > 0 -> -normal faces the ray
    assert idx[1] == 0 and idx[2] == -1                        # |d.u| <= u.u inclusive; 1.01 outside
    assert idx[3] == 0 and idx[4] == -1
    assert idx[5] == -1                                        # parallel
    assert idx[6] == -1                                        # t <= 1e-4
    assert idx[7] == 0 and list(n[7]) == [0, -1, 0]            # from below: denom < 0 -> +normal
    assert idx[8] == -1                                        # pointing away


def test_closest_is_first_minimum(oracle):
    s = _scene()
    m0, m1 = s.add_material((1, 0, 0)), s.add_material((0, 1, 0))
    s.add_plane((0, 0, 5), (2, 0, 0), (0, 2, 0), m0)
    s.add_sphere((0, 0, 6), 1.0, m1)      # also at t = 5 along +z
    s.add_plane((0, 0, 5), (2, 0, 0), (0, 2, 0), m1)
    t, n, idx = oracle.unit.intersect(s, f32([0, 0, 0]), f32([0, 0, 1]))
    assert t[0] == 5.0 and idx[0] == 0    # Iterator::min_by keeps the first of equal minima (renderer.rs:359-361)


# ---- ggx.rs ----------------------------------------------------------------------
def test_ggx_closed_forms(oracle):
    a = f32(0.01, 0.3, 1.0)
    one = np.ones(3, np.float32)
    f0 = f32([0.04, 0.5, 1.0]).repeat(3, 0)
    out = oracle.unit.ggx_terms(one, one, one, one, a, f0)
    # d_term(1,a) = 1/(pi a^2) — in f32 q = (a2-1)+1 only recovers a2 to ~6e-8 absolute, i.e. 6e-4 relative
    # at a = 0.01, so the closed form holds to that; the f32 evaluation order (ggx.rs:11) holds exactly
    assert np.allclose(out[1:, 3], 1.0 / (PI * a[1:].astype(np.float64) ** 2), rtol=2e-6)
    assert np.isclose(out[0, 3], 1.0 / (PI * 1e-4), rtol=1e-3)
    a2 = a * a
    q = (a2 - np.float32(1)) + np.float32(1)
    assert np.array_equal(out[:, 3], a2 / (np.float32(PI) * (q * q)))
    assert np.allclose(out[:, 4], 1.0, rtol=1e-6)                                      # g_term(1,1,a) = 1
    assert np.array_equal(out[:, :3], f0)                                              # F(1,f0) = f0
    out0 = oracle.unit.ggx_terms(one, one, one, 0 * one, a, f0)
    assert np.array_equal(out0[:, :3], np.ones((3, 3), np.float32))                    # F(0,f0) = 1
    # Q7: D uses alpha = roughness (not squared): d_term(0.5, 0.5)
    d = oracle.unit.ggx_terms(f32(0.5), f32(1), f32(1), f32(1), f32(0.5), f32([0, 0, 0]))[0, 3]
    assert np.isclose(d, 0.25 / (PI * (0.25 * (0.25 - 1) + 1) ** 2), rtol=1e-6)
    # G: k = a^2/2
    g = oracle.unit.ggx_terms(f32(1), f32(0.5), f32(0.25), f32(1), f32(0.8), f32([0, 0, 0]))[0, 4]
    k = 0.8 ** 2 / 2
    assert np.isclose(g, (0.5 / (0.5 * (1 - k) + k)) * (0.25 / (0.25 * (1 - k) + k)), rtol=1e-6)


def test_reflect_refract_orthonormal(oracle):
    assert np.array_equal(oracle.reflect(f32([1, -1, 0]), f32([0, 1, 0])), f32([1, 1, 0]))
    ok, r = oracle.refract(f32([0, 0, 1]), f32([0, 0, -1]), f32(1 / 1.5))
    assert ok[0] and np.allclose(r[0], [0, 0, 1])
    # total internal reflection: leaving glass (eta 1.5) at 60 degrees
    v = f32([np.sin(np.pi / 3), 0, np.cos(np.pi / 3)])
    ok, _ = oracle.refract(v, f32([0, 0, -1]), f32(1.5))
    assert not ok[0]
    # Snell at 30 degrees into ior 1.5
    v = f32([0.5, 0, np.sqrt(0.75)])
    ok, r = oracle.refract(v, f32([0, 0, -1]), f32(1 / 1.5))
    assert ok[0] and np.isclose(r[0, 0], 0.5 / 1.5, rtol=1e-6) and np.isclose(np.linalg.norm(r[0]), 1, rtol=1e-6)
    ao = oracle.any_orthonormal(f32([0, 0, 1], [1, 0, 0], [0.3, 0.4, 0.5]))
    assert np.array_equal(ao[0], f32(0, -1, 0)) and np.array_equal(ao[1], f32(0, -1, 0))
    assert np.array_equal(ao[2], f32(0.4, -0.3, 0))


def test_ggx_sample_closed_forms(oracle):
    n = f32([0, 0, 1], [0, 1, 0], [0.6, 0, 0.8])
    h = oracle.unit.ggx_sample(n, f32(0.2, 0.7, 1.0), f32(0.1, 0.6, 0.9), f32(0, 0, 0))
    assert np.allclose(h, n, atol=1e-7)                    # r2 = 0 -> cos_theta = 1 -> h = n
    # roughness 1 (a2 = 1): cos_theta = sqrt(1 - r2)
    h = oracle.unit.ggx_sample(f32([0, 0, 1]), f32(1.0), f32(0.0), f32(0.75))
    # frame for n = +z: u = normalize(any_orthonormal) = (0,-1,0), v = n x u = (1,0,0)
    assert np.allclose(h[0], [0, -np.sqrt(0.75), 0.5], atol=1e-6)


def test_glass_r0_and_branches(oracle):
    rd, n = f32([0, 0, 1]), f32([0, 0, -1])       # normal incidence: R = r0
    for ior, r0 in ((1.5, 0.04), (3.0, 0.25)):
        refl = oracle.unit.glass(rd, n, f32(ior), f32(r0 - 1e-3))
        refr = oracle.unit.glass(rd, n, f32(ior), f32(r0 + 1e-3))
        assert np.allclose(refl[0], [0, 0, -1]) and np.allclose(refr[0], [0, 0, 1])


def test_hg_closed_forms(oracle):
    rd = f32([0, 0, 1], [0, 0, 1], [0, 0, 1])
    c, d = oracle.unit.hg(rd, f32(0.0, 0.0005, 0.5), f32(0.25, 0.25, 0.5), f32(0, 0, 0))
    assert c[0] == 0.5 and c[1] == 0.5           # |g| < 1e-3 -> 1 - 2 xi
    assert np.isclose(c[2], 0.6875)              # g = .5, xi = .5: term = .75/1 -> (1 + .25 - .5625)/1
    assert np.allclose(np.linalg.norm(d, axis=1), 1, atol=1e-6)
    assert np.allclose(d[:, 2], c, atol=1e-6)    # cos of the angle to the incoming direction


def test_diffuse_sample_is_cosine_lobe(oracle):
    n = f32([0, 0, 1])
    d = oracle.unit.diffuse_sample(n, f32(0.0), f32(0.25))
    # u = (0,-1,0), v = (1,0,0): (u cos0 sqrt(r2) + w sqrt(1-r2)) = (0,-0.5, sqrt(.75))
    assert np.allclose(d[0], [0, -0.5, np.sqrt(0.75)], atol=1e-6)


# ---- tonemap.rs + renderer.rs:187-191 ------------------------------------------------
def test_aces_and_quantise(oracle):
    x = f32(0.0, 1.0, 1e6, 0.18)
    y = oracle.aces(x)
    assert y[0] == 0.0 and np.isclose(y[1], 2.54 / 3.16, rtol=1e-6) and y[2] == 1.0
    assert np.isclose(y[3], (0.18 * (2.51 * 0.18 + 0.03)) / (0.18 * (2.43 * 0.18 + 0.59) + 0.14), rtol=1e-6)
    q = oracle.unit.tonemap(f32([0, 1.0, 1e6], [np.nan, -1.0, 0.18]))
    assert list(q[0]) == [0, int(np.floor((2.54 / 3.16) ** (1 / 2.2) * 255)), 255] and q[0, 1] == 230
    assert q[1, 0] == 255                                    # NaN: f32::min(NaN, 255) = 255 -> white (Q13)
    assert q[1, 2] == int(np.floor(float(y[3]) ** (1 / 2.2) * 255))


# ---- main.rs:44-51, renderer.rs:204-234 ------------------------------------------------
def test_camera_basis(oracle):
    cam = oracle.camera_basis((0, 0, 0), (0, 0, 1), (0, 1, 0), 90.0, 200, 100)
    assert list(cam.forward) == [0, 0, 1] and list(cam.right) == [1, 0, 0] and list(cam.up) == [0, 1, 0]
    assert cam.aspect == 2.0 and np.isclose(cam.scale, 1.0, rtol=1e-6)
    # the product's host implementation (hippt_camera_basis) is the same arithmetic, bit for bit
    from pathtracer_amd import Scene
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"))
    import json
    with open(os.path.join(ROOT, "scenes", "scene.example.json")) as f:
        j = json.load(f)["camera"]
    co = oracle.camera_basis(j["pos"], j["look_at"], j["up"], j["fov"], 800, 600)
    for k in ("right", "up", "forward"):
        assert list(getattr(co, k)) == list(getattr(s.camera, k))
    assert co.aspect == s.camera.aspect and co.scale == s.camera.scale


def test_autofocus(oracle, example_scene):
    assert oracle.autofocus(example_scene) == 5.0             # no in_focus object (renderer.rs:211-213)
    s = _scene(101, 101)
    m = s.add_material((0.5, 0.5, 0.5))
    s.add_plane((0, 0, 7), (50, 0, 0), (0, 50, 0), m, in_focus=True)
    s.add_sphere((0, 0, 3), 0.5, m, in_focus=False)           # nearer, but not a focus object
    s.set_camera((0, 0, 0), (0, 0, 1), (0, 1, 0), 1.0, 0.0)   # 1 degree fov: all 25 probes have t ~ 7
    f = oracle.autofocus(s)
    assert abs(f - 7.0) < 1e-4
    assert s.camera.focus == np.float32(f)                    # host implementation agrees exactly
    s2 = _scene(64, 64)
    m = s2.add_material((0.5, 0.5, 0.5))
    s2.add_sphere((0, 50, 3), 0.5, m, in_focus=True)          # focus object never hit -> 10.0
    s2.set_camera((0, 0, 0), (0, 0, 1), (0, 1, 0), 60.0, 0.0)
    assert oracle.autofocus(s2) == 10.0 and s2.camera.focus == 10.0


def test_camera_ray_closed_form(oracle):
    s = _scene(100, 100)
    s.add_material((0.5, 0.5, 0.5))
    s.set_camera((0, 0, 0), (0, 0, 1), (0, 1, 0), 90.0, 0.0, focus=5.0)
    ro, rd = oracle.unit.camera_ray(s, [[50, 50], [99, 0]], f32([0, 0, 0.3, 0.7], [1, 0, 0.3, 0.7]))
    assert np.allclose(ro, 0) and np.allclose(rd[0], [0, 0, 1])
    # pixel (99,0) + jitter (1,0): u = +1 (right edge), v = +1 (top edge) -> (1,1,1)/sqrt3: row 0 is the TOP
    assert np.allclose(rd[1], np.ones(3) / np.sqrt(3), atol=1e-6)
    # thin lens: aperture radius r, lens sample (s,t) -> origin = r sqrt(t) (cos 2 pi s, sin 2 pi s); focal point kept
    s.camera.aperture = 0.5
    ro, rd = oracle.unit.camera_ray(s, [[50, 50]], f32([0, 0, 0.25, 0.64]))
    assert np.allclose(ro[0], [0, 0.4, 0], atol=1e-6)
    foc = ro[0] + rd[0] * np.linalg.norm(np.array([0, 0, 5.0]) - ro[0])
    assert np.allclose(foc, [0, 0, 5], atol=1e-5)


# ---- RNG stream -----------------------------------------------------
def test_rng_distribution(oracle):
    x = oracle.unit.rng(0, 123, 7, 1 << 20).astype(np.float64)
    assert x.min() >= 0.0 and x.max() < 1.0
    assert np.all(x * (1 << 24) == np.floor(x * (1 << 24)))      # 24-bit grid, like rand's Standard f32
    assert abs(x.mean() - 0.5) < 1.5e-3 and abs(x.var() - 1 / 12) < 1e-3
    hist = np.bincount((x * 64).astype(int), minlength=64)
    chi2 = ((hist - len(x) / 64) ** 2 / (len(x) / 64)).sum()
    assert chi2 < 120                                            # 63 dof: P(chi2 > 120) ~ 1e-5
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 5e-3          # lag-1
    # pairs fill the unit square evenly (light samples, lobe samples are consumed in pairs)
    h2 = np.histogram2d(x[0::2], x[1::2], bins=16, range=[[0, 1], [0, 1]])[0]
    e = len(x) / 2 / 256
    assert ((h2 - e) ** 2 / e).sum() < 400                       # 255 dof
    # an LCG's weak spot is the lattice of consecutive outputs: serial correlations up to lag 16 (one light's 8 draws
    # and the next light's) and the 3-D and 4-D cells of consecutive triples / quadruples must be evenly filled
    for lag in range(2, 17):
        assert abs(np.corrcoef(x[:-lag], x[lag:])[0, 1]) < 5e-3, lag
    n3 = len(x) // 3 * 3
    c3 = (x[:n3].reshape(-1, 3) * 8).astype(int)
    h3 = np.bincount(c3[:, 0] * 64 + c3[:, 1] * 8 + c3[:, 2], minlength=512)
    assert ((h3 - n3 / 3 / 512) ** 2 / (n3 / 3 / 512)).sum() < 700     # 511 dof: P(chi2 > 700) ~ 1e-8
    c4 = (x.reshape(-1, 4) * 4).astype(int)
    h4 = np.bincount(c4[:, 0] * 64 + c4[:, 1] * 16 + c4[:, 2] * 4 + c4[:, 3], minlength=256)
    assert ((h4 - len(x) / 4 / 256) ** 2 / (len(x) / 4 / 256)).sum() < 400
    # fine structure of pairs: 256 x 256 cells, 2 expected per cell on average -> compare the count distribution's
    # variance with Poisson's (a coarse lattice would leave most cells empty)
    h2f = np.histogram2d(x[0::2], x[1::2], bins=256, range=[[0, 1], [0, 1]])[0]
    assert abs(h2f.var() / h2f.mean() - 1.0) < 0.05
    # first draws of neighbouring streams (pixel+1, sample+1) are uncorrelated
    firsts = np.array([oracle.unit.rng(0, p, s, 1)[0] for p in range(64) for s in range(64)], np.float64)
    assert abs(firsts.mean() - 0.5) < 0.02
    g = firsts.reshape(64, 64)
    assert abs(np.corrcoef(g[:, :-1].ravel(), g[:, 1:].ravel())[0, 1]) < 0.06
    assert abs(np.corrcoef(g[:-1].ravel(), g[1:].ravel())[0, 1]) < 0.06


def _mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    return x


def _stream_keys(seed, pixel, sample):
    """numpy mirror of Rng's seeding (oracle/pt_oracle.cpp, pt_device.h): (start state, increment) per stream."""
    pixel, sample = np.asarray(pixel, np.uint32), np.asarray(sample, np.uint32)
    a = _mix32(np.array([seed], np.uint32) ^ np.uint32(0x85ebca6b))
    b = _mix32(pixel + a)
    d = _mix32(b ^ np.uint32(0x5bd1e995))
    return _mix32(sample ^ b), _mix32(sample * np.uint32(0x9e3779b1) + d) | np.uint32(1)


def _stream_draws(s0, inc, n):
    out = np.empty((n,) + s0.shape, np.float64)
    s = s0.copy()
    for k in range(n):
        s = s * np.uint32(2891336453) + inc
        out[k] = (s >> np.uint32(8)) * 2.0 ** -24
    return out


def test_rng_streams_are_keyed_on_64_bits(oracle):
    """the increment is its own hash of (pixel, sample), not a function of the start state."""
    with np.errstate(over="ignore"):
        # the numpy mirror IS the oracle's stream
        rs = np.random.default_rng(5)
        for seed, p, smp in zip(rs.integers(0, 1 << 32, 48), rs.integers(0, 1 << 25, 48), rs.integers(0, 1 << 17, 48)):
            s0, inc = _stream_keys(int(seed), np.array([p]), np.array([smp]))
            assert np.array_equal(_stream_draws(s0, inc, 8)[:, 0].astype(np.float32), oracle.unit.rng(int(seed), int(p), int(smp), 8))
        # no two of 2^24 (pixel, sample) keys share a stream: 4096 neighbouring pixels x 4096 consecutive samples,
        # and the same count for pixels of two image rows at 8K with the showcase job's last samples
        for pix0, smp0 in ((0, 0), (7680 * 2000 + 100, 131072 - 4096)):
            pix, smp = np.meshgrid(np.arange(4096, dtype=np.uint32) + np.uint32(pix0),
                                   np.arange(4096, dtype=np.uint32) + np.uint32(smp0), indexing="ij")
            s0, inc = _stream_keys(0, pix.ravel(), smp.ravel())
            key = (s0.astype(np.uint64) << np.uint64(32)) | inc.astype(np.uint64)
            assert len(np.unique(key)) == key.size
            # the old scheme's weakness, stated as a number: the start states alone DO collide (birthday in 32 bits)
            assert len(np.unique(s0)) < s0.size
            # ... and streams that share a start state have unrelated increments
            order = np.argsort(s0, kind="stable")
            same = s0[order][1:] == s0[order][:-1]
            assert same.sum() > 1000 and (inc[order][1:][same] != inc[order][:-1][same]).all()
        # one pixel, consecutive samples: draw k of stream s against draw k of stream s + 1 (k < 16)
        s0, inc = _stream_keys(0, np.full(1 << 16, 12345, np.uint32), np.arange(1 << 16, dtype=np.uint32))
        x = _stream_draws(s0, inc, 16)
        for k in range(16):
            assert abs(np.corrcoef(x[k, :-1], x[k, 1:])[0, 1]) < 0.02, k
            h2 = np.histogram2d(x[k, 0::2], x[k, 1::2], bins=16, range=[[0, 1], [0, 1]])[0]
            e = x.shape[1] / 2 / 256
            assert ((h2 - e) ** 2 / e).sum() < 400, k                  # 255 dof
            hist = np.bincount((x[k] * 64).astype(int), minlength=64)
            assert ((hist - x.shape[1] / 64) ** 2 / (x.shape[1] / 64)).sum() < 130, k
        # draws k and k + 1 ACROSS streams (what a wave's lanes consume together)
        assert abs(np.corrcoef(x[:-1].ravel(), x[1:].ravel())[0, 1]) < 0.01
        # neighbouring pixels (x + 1 and the next row at 1920), same sample: first 16 draws
        pix = np.arange(1 << 16, dtype=np.uint32)
        for dp in (1, 1920):
            xa = _stream_draws(*_stream_keys(0, pix, np.full(pix.shape, 7, np.uint32)), 16)
            xb = _stream_draws(*_stream_keys(0, pix + np.uint32(dp), np.full(pix.shape, 7, np.uint32)), 16)
            for k in range(16):
                assert abs(np.corrcoef(xa[k], xb[k])[0, 1]) < 0.02, (dp, k)
                h2 = np.histogram2d(xa[k], xb[k], bins=16, range=[[0, 1], [0, 1]])[0]
                e = len(pix) / 256
                assert ((h2 - e) ** 2 / e).sum() < 400, (dp, k)


# ---- integrator closed form (SURVEY §8c last row) ---------------------------------------
def _quirk_brdf_integral(hit, n, v, color, metallic, rough, light, m=400):
    """float64 quadrature of renderer.rs:37-77 over the light rectangle (Q5: half of each edge vector)."""
    a = (np.arange(m) + 0.5) / m - 0.5
    A, B = np.meshgrid(a, a, indexing="ij")
    pos, lu, lv, inten = (np.array(light[k], np.float64) for k in ("pos", "u", "v", "intensity"))
    lp = pos + A[..., None] * lu + B[..., None] * lv
    lvec = lp - hit
    d2 = (lvec ** 2).sum(-1)
    l = lvec / np.sqrt(d2)[..., None]
    cr = np.cross(lu, lv)
    area, ln = np.linalg.norm(cr), cr / np.linalg.norm(cr)
    ndl = np.maximum(l @ n, 0)
    cosl = np.maximum(-(l @ ln), 0)
    h = v + l
    h /= np.linalg.norm(h, axis=-1, keepdims=True)
    ndv, ndh, vdh = max(n @ v, 1e-4), np.maximum(h @ n, 0), np.maximum(h @ v, 0)
    f0 = 0.04 * (1 - metallic) + color * metallic
    F = f0 + (1 - f0) * ((1 - vdh) ** 5)[..., None]
    D = rough ** 2 / (PI * (ndh ** 2 * (rough ** 2 - 1) + 1) ** 2)
    k = rough ** 2 / 2
    G = (ndv / (ndv * (1 - k) + k)) * (ndl / (ndl * (1 - k) + k))
    spec = F * (D * G / (4 * ndv * ndl + 1e-6))[..., None]
    diff = color * (1 - metallic) * (1 - F) / PI
    rad = (diff + spec) * ndl[..., None] * inten * (area * cosl / d2)[..., None]
    rad[(ndl <= 0) | (cosl <= 0)] = 0
    return rad.mean(axis=(0, 1))


@pytest.mark.parametrize("metallic,rough", [(0.0, 1.0), (0.6, 0.4)])
def test_single_bounce_matches_quadrature(oracle, metallic, rough):
    color = np.array([0.7, 0.4, 0.2])
    s = _scene(2001, 2001)
    m = s.add_material(color, metallic=metallic, roughness=rough)
    s.add_plane((0, 0, 0), (50, 0, 0), (0, 0, 50), m)                 # floor y = 0
    light = dict(pos=(0.5, 3.0, 0.3), u=(-1.0, 0, 0), v=(0, 0, -1.0), intensity=(20, 15, 10))   # u x v = (0,-1,0): faces down
    s.add_light(light["pos"], light["u"], light["v"], light["intensity"])
    s.set_camera((0, 2, -4), (0, 0, 0), (0, 1, 0), 20.0, 0.0)
    px = 1000
    n_paths = 60000
    xys = np.stack([np.full(n_paths, px), np.full(n_paths, px), np.arange(n_paths)], 1)
    rad = oracle.paths(s, xys, seed=11, max_depth=1).astype(np.float64)   # depth 1: direct light only
    est = rad.mean(0)
    sem = rad.std(0) / np.sqrt(n_paths)
    fwd = np.array([0, -2, 4.0]) / np.sqrt(20)
    hit = np.array([0, 2, -4.0]) + fwd * np.sqrt(20)                   # centre pixel looks at the origin
    ref = _quirk_brdf_integral(hit, np.array([0, 1.0, 0]), -fwd, color, metallic, rough, light)
    assert np.all(ref > 0)
    assert np.all(np.abs(est - ref) < 4 * sem + 2e-3 * ref), (est, ref, sem)


def test_recursive_and_iterative_forms_agree(oracle, example_scene):
    s = example_scene.with_render(64, 48, 8)
    a = oracle.render_accum(s, seed=1)
    b = oracle.render_accum(s, seed=1, iterative=True)
    assert np.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_counters_bookkeeping(oracle, example_scene):
    s = example_scene.with_render(48, 36, 4)
    _, c = oracle.render_accum(s, seed=0, counters=True)
    assert c["paths"] == 48 * 36 * 4
    assert c["extend_sphere_tests"] == 3 * c["extend_rays"] and c["extend_plane_tests"] == 4 * c["extend_rays"]
    assert c["shadow_rays"] == 8 * c["nee_vertices"]                     # 2 lights x 4 samples
    assert c["nee_vertices"] >= c["diffuse_lobes"] + c["spec_lobes"]
    assert c["extend_hits"] == c["nee_vertices"] + c["glass_vertices"]   # no media in this scene
    assert c["paths"] <= c["extend_rays"] <= 5 * c["paths"]
