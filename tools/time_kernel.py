"""Kernel time only (no image checks): for ablation builds whose images are wrong on purpose.
usage: HIPPT_LIB=... python tools/time_kernel.py [spp] [scene.json]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pathtracer_amd import Scene, hip_renderer
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
name = sys.argv[2] if len(sys.argv) > 2 else "scenes/scene.example.json"
scene = Scene.load(name, 1920, 1080, spp)
ctx = hip_renderer.Context(0)
ctx.set_scene(scene)
acc = torch.zeros(1920 * 1080 * 3, dtype=torch.float32, device="cuda")
st = torch.cuda.Stream()
for i in range(3):
    ctx.render_accum_device(acc.data_ptr(), 0, spp, seed=0, stream=st.cuda_stream)
    ms, n = ctx.kernel_time_ms()
print(os.path.basename(os.environ.get("HIPPT_LIB", "libhippt.so")), f"{ms / n:.2f} ms", f"{1920 * 1080 * spp / (ms / n) / 1e3:.0f} Mpaths/s")
