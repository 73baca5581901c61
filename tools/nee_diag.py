"""Hint statistics of Tracer::direct_light_shared: build with tools/build_variant.sh diag -DPT_NEE_DIAG and run with
HIPPT_LIB=pathtracer_amd/lib/libhippt_diag.so (the plain kernel then writes its statistics into the counter array)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pathtracer_amd import Scene, _abi, hip_renderer
spp = 64
scene = Scene.load("scenes/scene.example.json", 1920, 1080, spp)
ctx = hip_renderer.Context(0)
ctx.set_scene(scene)
acc = torch.zeros(1920 * 1080 * 3, dtype=torch.float32, device="cuda")
ctx.render_accum_device(acc.data_ptr(), 0, spp, seed=0, max_depth=5, flags=0, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
c = list(ctx.read_counters().values())
for li in range(3):
    tot, n0, n1, sn = c[li*4:li*4+4]
    big = c[12+li]
    print(f"light {li}: wave-iterations {tot}, hinted n=0 {n0/tot:.3f}, hinted 0<n<=64 {n1/tot:.3f} (mean n {sn/max(n1,1):.1f}), hinted n>64 {big/tot:.3f}, unhinted {(tot-n0-n1-big)/tot:.3f}")
