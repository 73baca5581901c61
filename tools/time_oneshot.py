"""Wall time of the one-shot entry point hippt_render_rgba8 (context creation, scene upload, render, resolve, w*h*4 bytes
back over PCIe) on C2, next to the resident-context rate of bench.py. usage: python tools/time_oneshot.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import Scene, hip_renderer
s = Scene.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "scene.example.json"), 1920, 1080, 1024)
for i in range(3):
    t = time.perf_counter()
    img = hip_renderer.render(s)
    dt = time.perf_counter() - t
    print(f"call {i}: {dt * 1e3:.1f} ms -> {1920 * 1080 * 1024 / dt / 1e6:.0f} Mpaths/s including context creation and the copy back; image mean {img[..., :3].mean():.2f}")
