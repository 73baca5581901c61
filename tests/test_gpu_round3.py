"""Round-3 GPU tests: ADVICE r02 fixes and VERDICT r02 items."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _room(n_lights, n_planes, w=48, h=36, spp=4):
    """A box of `n_planes` rectangles (the last ones stacked inside the floor) lit by `n_lights` small lights."""
    from pathtracer_amd import Scene
    s = Scene()
    s.width, s.height, s.samples = w, h, spp
    m = s.add_material((0.7, 0.6, 0.5), roughness=0.8)
    g = s.add_material((0.9, 0.9, 0.9), metallic=1.0, roughness=0.2)
    s.add_sphere((0.0, 1.0, 0.0), 1.0, g)
    s.add_plane((0, 0, 0), (6, 0, 0), (0, 0, 6), m)
    s.add_plane((0, 3, 6), (6, 0, 0), (0, 3, 0), m)
    rng = np.random.default_rng(n_lights * 1000 + n_planes)
    for _ in range(n_planes - 2):
        c = rng.uniform(-4, 4, 3) * (1, 0, 1) + (0, rng.uniform(0.2, 4.0), 0)
        u, v = rng.normal(size=3), rng.normal(size=3)
        s.add_plane(tuple(c), tuple(0.3 * u), tuple(0.3 * v), m)
    for _ in range(n_lights):
        p = rng.uniform(-3, 3, 3) * (1, 0, 1) + (0, rng.uniform(4.0, 6.0), 0)
        s.add_light(tuple(p), (-0.5, 0, 0), (0, 0, -0.5), tuple(rng.uniform(1, 3, 3) * 40.0 / n_lights))
    s.set_camera((0, 2.0, -7), (0, 1, 0), (0, 1, 0), 50.0, 0.0)
    return s


def test_light_side_table_is_bounded_and_optional(oracle):
    """the light x plane table was sized with a 32-bit product. It now exists only up to 4 096
    entries; beyond that all lights share one 'never skip' row — same image as the oracle's either way."""
    from pathtracer_amd import hip_renderer
    for n_lights, n_planes in ((3, 40), (70, 70)):      # 120 entries: table; 4 900 entries: shared row
        s = _room(n_lights, n_planes)
        a = hip_renderer.render_accum(s, seed=2)
        o = oracle.render_accum(s, seed=2)
        assert np.isfinite(a).all() and a.any()
        err = np.abs(a - o).max(axis=2) / np.maximum(o.max(axis=2), 1e-2)
        assert (err < 1e-4).mean() > 0.97, (n_lights, n_planes, (err < 1e-4).mean())
        assert abs(a.sum() - o.sum()) <= 5e-3 * o.sum()
    # 66 000 lights x 66 000 planes: the product wraps 32 bits. Packing and upload succeed (9 MB of tables), nothing is rendered.
    big = _room(66000, 66000, 8, 8, 1)
    ctx = hip_renderer.Context(0)
    ctx.set_scene(big)
    ctx.close()


def test_showcase_sample_count_in_one_accumulator(oracle):
    """VERDICT r02 missing #3: the showcase job's SAMPLE COUNT (131 072 per pixel, README.md:265-268). One call sums
    all of them into one f32 accumulator (launches of at most 4 096 samples, slices inside a launch): it must equal the
    float64 sum of 128 separate 1 024-sample renders to 1e-5 per pixel, and agree in the mean with the oracle."""
    import torch
    from pathtracer_amd import Scene, hip_renderer
    W, H, SPP = 64, 48, 131072
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), W, H, SPP)
    one = hip_renderer.render_accum(s, 0, SPP, seed=4).astype(np.float64)
    ctx = hip_renderer.Context(0)
    ctx.set_scene(s)
    dev = torch.device("cuda", 0)
    buf = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    total = np.zeros(H * W * 3, np.float64)
    for k in range(128):
        buf.zero_()
        torch.cuda.synchronize(dev)
        ctx.render_accum_device(buf.data_ptr(), k * 1024, 1024, seed=4)
        ctx.kernel_time_ms()  # synchronises
        total += buf.cpu().numpy().astype(np.float64)
    total = total.reshape(H, W, 3)
    rel = np.abs(one - total) / np.maximum(total, 1e-3 * SPP)
    assert rel.max() <= 1e-5, rel.max()
    # against the CPU restatement at 4 096 spp (other samples of the same streams: a statistical comparison)
    o = oracle.render_accum(Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), W, H, 4096), seed=4).astype(np.float64)
    m_gpu, m_cpu = total.mean() / SPP, o.mean() / 4096
    assert abs(m_gpu - m_cpu) <= 1e-2 * m_cpu, (m_gpu, m_cpu)
    # and the first 4 096 samples of the same streams agree sample for sample
    first = hip_renderer.render_accum(s, 0, 4096, seed=4).astype(np.float64)
    assert abs(first.sum() - o.sum()) <= 2e-3 * o.sum()


def test_bench_two_ranks_rehearsed_on_one_gpu():
    """bench.py's N > 1 code (rank launcher, sample split, reduce, per-rank timing, JSON) with the real kernels on the
    one GPU of this box: both ranks render on device 0 and the sum goes through gloo. With two GPUs visible
    tests/test_gpu_two_devices.py runs the same through RCCL."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--spp", "64", "--width", "640", "--height", "360", "--rehearse-on-one-gpu"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["scaling"] == "strong" and "rehearsal" in d
    assert d["per_rank"]["kernel_ms_max"] >= d["per_rank"]["kernel_ms_min"] > 0 and d["per_rank"]["reduce_ms"] > 0
    assert d["config"]["workload"].count("32 spp on rank 0") == 1
