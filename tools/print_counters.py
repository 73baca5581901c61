#!/usr/bin/env python3
"""Print the device event counters of one C2 launch (counting variant of the megakernel)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pathtracer_amd import Scene, _abi, hip_renderer
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene = Scene.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes", sys.argv[2] if len(sys.argv) > 2 else "scene.example.json"), 1920, 1080, spp)
ctx = hip_renderer.Context(0)
ctx.set_scene(scene)
acc = torch.zeros(1920 * 1080 * 3, dtype=torch.float32, device="cuda")
ctx.render_accum_device(acc.data_ptr(), 0, spp, seed=0, max_depth=5, flags=_abi.FLAG_COUNTERS, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for k, v in ctx.read_counters().items():
    print(f"{k:28s} {v}")
