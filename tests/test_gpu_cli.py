"""The CLI driver (hippt_cli.cpp) behaves like the reference's `cargo run --release -- --gpu` (src/main.rs:89-108)."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_renders_and_saves_png(tmp_path, oracle):
    from PIL import Image
    from pathtracer_amd import Scene, hip_renderer
    exe = os.path.join(ROOT, "pathtracer_amd", "bin", "hippt")
    assert os.path.exists(exe), "build the CLI with `make -C pathtracer_amd/csrc cli`"
    scene = os.path.join(ROOT, "scenes", "scene.example.json")
    # reference behaviour: ./scene.json in the working directory, output under ./renders/
    os.symlink(scene, tmp_path / "scene.json")
    r = subprocess.run([exe, "--gpu", "-q", "--width", "64", "--height", "48", "--samples", "16", "--seed", "7"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "=== CAMERA INFO ===" in r.stdout and "=== OBJECTS (7) ===" in r.stdout and "=== LIGHTS (2) ===" in r.stdout
    assert "Running GPU renderer..." in r.stdout
    m = re.search(r"Saved → (renders/render_64x48_s16_ap0\.02_f5\.0_[A-Za-z0-9]{6}\.png)", r.stdout)
    assert m, r.stdout
    img = np.asarray(Image.open(tmp_path / m.group(1)))
    assert img.shape == (48, 64, 4) and (img[..., 3] == 255).all()
    # same pixels as the library call with the same seed
    s = Scene.load(scene, 64, 48, 16)
    ref = hip_renderer.render(s, seed=7)
    assert np.array_equal(img, ref)
    # and within one LSB of the oracle's own quantisation of the oracle's radiance on almost every value
    img_o = oracle.render_rgba8(s, seed=7)
    d = np.abs(img.astype(int) - img_o.astype(int))
    assert (d <= 1).mean() > 0.97


def test_cli_missing_scene_fails_like_expect(tmp_path):
    exe = os.path.join(ROOT, "pathtracer_amd", "bin", "hippt")
    r = subprocess.run([exe, "--gpu"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode == 101 and "scene file" in r.stderr      # scene.rs:95 expect("scene file") -> panic exit code 101


def test_cli_progressive_checkpoint_and_resume(tmp_path):
    """chunked rendering with host accumulation, checkpoint after every chunk, resume.
    10 spp in chunks of 4 (the last chunk is short, not over-rendered as gpu_renderer.rs:98-102 does)."""
    from PIL import Image
    from pathtracer_amd import Scene, hip_renderer
    exe = os.path.join(ROOT, "pathtracer_amd", "bin", "hippt")
    scene = os.path.join(ROOT, "scenes", "scene.example.json")
    base = [exe, "--gpu", "-q", "--scene", scene, "--width", "64", "--height", "48", "--seed", "3"]
    ck = str(tmp_path / "job.acc")
    r = subprocess.run(base + ["--samples", "10", "--chunk", "4", "--checkpoint", ck, "--preview", "--out", str(tmp_path / "a.png")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "4 / 10 samples" in r.stdout and "8 / 10 samples" in r.stdout and "10 / 10 samples" in r.stdout
    acc, done, seed = hip_renderer.checkpoint_load(ck, 64, 48)
    assert (done, seed) == (10, 3)
    s = Scene.load(scene, 64, 48, 10)
    one_shot = hip_renderer.render_accum(s, seed=3)
    assert np.allclose(acc, one_shot, rtol=2e-6, atol=1e-6)        # equal up to f32 summation order
    a = np.asarray(Image.open(tmp_path / "a.png"))
    ref = hip_renderer.render(s, seed=3)
    assert np.abs(a.astype(int) - ref.astype(int)).max() <= 1 and (a == ref).mean() > 0.999

    # interrupted job: 6 samples now, the remaining 4 after --resume
    ck2 = str(tmp_path / "job2.acc")
    r = subprocess.run(base + ["--samples", "6", "--chunk", "4", "--checkpoint", ck2, "--out", str(tmp_path / "b6.png")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert hip_renderer.checkpoint_load(ck2, 64, 48)[1] == 6
    r = subprocess.run(base + ["--samples", "10", "--chunk", "4", "--checkpoint", ck2, "--resume", "--out", str(tmp_path / "b.png")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Resuming at sample 6 of 10" in r.stdout
    acc2, done2, _ = hip_renderer.checkpoint_load(ck2, 64, 48)
    assert done2 == 10 and np.allclose(acc2, one_shot, rtol=2e-6, atol=1e-6)
    b = np.asarray(Image.open(tmp_path / "b.png"))
    assert np.abs(b.astype(int) - ref.astype(int)).max() <= 1
    # a checkpoint from another seed is refused
    r = subprocess.run(base[:-1] + ["4", "--samples", "12", "--checkpoint", ck2, "--resume", "--out", str(tmp_path / "c.png")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 101 and "do not fit this render" in r.stderr


def test_context_progressive_api_matches_one_shot():
    """Chunks of a sample range add up to the one-shot render to f32 summation order. This is also the tight check
    of the megakernel's wave-shared shadow-ray list (Tracer::direct_light_shared): which rays are deferred depends
    on what the wave has seen so far, i.e. on the chunking, and must not change any ray's outcome — on image sizes
    with ragged 8x8 tiles too, where some lanes skip trips."""
    from pathtracer_amd import Scene, hip_renderer
    for scene, w, h, spp, cuts in (("glass_fog.json", 50, 30, 9, (0, 2, 7, 9)), ("scene.example.json", 70, 45, 24, (0, 1, 9, 24))):
        s = Scene.load(os.path.join(ROOT, "scenes", scene), w, h, spp)
        ctx = hip_renderer.Context(0)
        ctx.set_scene(s)
        acc = np.zeros((h, w, 3), np.float32)
        for b, e in zip(cuts[:-1], cuts[1:]):
            ctx.render_accum_add(acc, b, e - b, seed=8)
        one = hip_renderer.render_accum(s, seed=8)
        assert np.allclose(acc, one, rtol=2e-6, atol=1e-6)
    s = Scene.load(os.path.join(ROOT, "scenes", "glass_fog.json"), 50, 30, 9)
    ctx = hip_renderer.Context(0)
    ctx.set_scene(s)
    acc = np.zeros((30, 50, 3), np.float32)
    for b, c in ((0, 2), (2, 5), (7, 2)):
        ctx.render_accum_add(acc, b, c, seed=8)
    img = ctx.resolve(acc, 9)
    ref = hip_renderer.render(s, seed=8)
    assert img.shape == ref.shape and np.abs(img.astype(int) - ref.astype(int)).max() <= 1


def test_multi_device_entry_points_with_one_device(tmp_path):
    """hippt_render_*_multi on the one GPU of this box: the whole path (per-device contexts and threads, librccl opened
    at run time, communicator, ncclReduce, resolve on devices[0]) with a single rank. The image must equal the
    one-device entry points' exactly (same kernel, same sample range). More than one rank needs more than one GPU:
    RCCL refuses a device listed twice, and so does the library."""
    import subprocess
    from PIL import Image
    from pathtracer_amd import Scene, hip_renderer, _abi
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 96, 64, 12)
    assert np.array_equal(hip_renderer.render_accum_multi(s, [0], seed=5), hip_renderer.render_accum(s, seed=5))
    assert np.array_equal(hip_renderer.render_accum_multi(s, [0], 3, 7, seed=5), hip_renderer.render_accum(s, 3, 7, seed=5))
    assert np.array_equal(hip_renderer.render_multi(s, [0], seed=5), hip_renderer.render(s, seed=5))
    with pytest.raises(_abi.HipptError, match="listed twice"):
        hip_renderer.render_multi(s, [0, 0])
    exe = os.path.join(ROOT, "pathtracer_amd", "bin", "hippt")
    base = [exe, "--gpu", "--quiet", "--scene", os.path.join(ROOT, "scenes", "scene.example.json"), "--width", "96", "--height", "64",
            "--samples", "12", "--seed", "5"]
    for extra, name in ((["--devices", "0"], "m.png"), ([], "s.png")):
        r = subprocess.run(base + extra + ["--out", str(tmp_path / name)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
    assert np.array_equal(np.asarray(Image.open(tmp_path / "m.png")), np.asarray(Image.open(tmp_path / "s.png")))
