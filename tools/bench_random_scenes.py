"""Mpaths/s of the table-scene megakernel on four of the randomised test scenes at 1080p x 64 spp (HIPPT_LIB selects the
build; tools/build_variant.sh plain -DPT_SHARED_NEE=0 gives the plain lights loop for comparison)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from pathtracer_amd import hip_renderer
from test_gpu_random_scenes import random_scene
for seed, ns, npl, nl in ((10, 31, 9, 2), (9, 13, 6, 4), (4, 5, 3, 3), (11, 63, 11, 1)):
    s = random_scene(seed, ns, npl, nl, w=1920, h=1080, spp=64)
    ctx = hip_renderer.Context(0); ctx.set_scene(s)
    acc = torch.zeros(1920 * 1080 * 3, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.render_accum_device(acc.data_ptr(), 0, 64, seed=0, max_depth=5, flags=8, stream=st)  # NO_BVH: table loops
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"scene {seed} ({ns} sph, {npl} pl, {nl} lights): {1920*1080*64/dt/1e6:.0f} Mpaths/s")
