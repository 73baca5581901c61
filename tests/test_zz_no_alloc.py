"""hippt_render_accum_device_noalloc (the entry point for callers that cannot allocate) against
hippt_render_accum_device, through the C ABI: two calls in a row on one context, the second with a smaller input;
same kernels and slice counts (2 to 4 slices: the scratch array is in use) at these sizes, so the sums must be
bit-identical, and no device memory may be allocated. Then the limit and the unsupported cases, and a 1920x1080
call where the static scratch holds 4 slices instead of 8 (equal to f32 summation order)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _both(ctx, scene, s0, cnt, seed, flags=0):
    import torch
    dev = torch.device("cuda", 0)
    n = scene.height * scene.width * 3
    a = torch.zeros(n, dtype=torch.float32, device=dev)
    b = torch.zeros(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    ctx.render_accum_device_noalloc(a.data_ptr(), s0, cnt, seed=seed, flags=flags)   # context's own stream
    ctx.render_accum_device(b.data_ptr(), s0, cnt, seed=seed, flags=flags)
    ctx.kernel_time_ms()                                                              # synchronises
    torch.cuda.synchronize(dev)
    return a.cpu().numpy(), b.cpu().numpy()


def test_noalloc_entry_point_matches_the_allocating_one():
    """Sample counts are >= 64 per launch so that the launches really cut their samples into slices (at least 32 samples
    each, hippt_api.hip: megakernel_slices) and the static scratch array — the address hipGetSymbolAddress returned —
    is written and read; a one-slice launch does not touch it."""
    import torch
    from pathtracer_amd import Scene, _abi, hip_renderer
    dev = torch.device("cuda", 0)
    ctx = hip_renderer.Context(0)
    big = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 400, 260, 128)      # 4 slices of 32 samples
    ctx.set_scene(big)
    # the very first launch of this context is the no-alloc one: the device's free memory must not move
    acc = torch.zeros(big.height * big.width * 3, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    free0 = torch.cuda.mem_get_info(dev)[0]
    ctx.render_accum_device_noalloc(acc.data_ptr(), 0, 128, seed=6)
    torch.cuda.synchronize(dev)
    ctx.kernel_time_ms()
    assert torch.cuda.mem_get_info(dev)[0] == free0, "hippt_render_accum_device_noalloc allocated device memory"
    a, b = _both(ctx, big, 0, 128, seed=6)
    assert a.any() and np.array_equal(a, b) and np.array_equal(a, acc.cpu().numpy())
    assert np.array_equal(a.reshape(260, 400, 3), hip_renderer.render_accum(big, seed=6))
    # the slices are real: one slice over the whole range adds the same samples in another order
    one, _ = _both(ctx, big, 0, 128, seed=6, flags=_abi.FLAG_NO_SLICES)
    assert not np.array_equal(one, a) and np.allclose(one, a, rtol=2e-6, atol=1e-6)
    # second call, smaller input (fewer pixels AND fewer samples, 2 slices: the scratch still holds the first call's sums)
    small = Scene.load(os.path.join(ROOT, "scenes", "glass_fog.json"), 70, 45, 80)
    ctx.set_scene(small)
    a, b = _both(ctx, small, 3, 70, seed=6)
    assert a.any() and np.array_equal(a, b)
    one, _ = _both(ctx, small, 3, 70, seed=6, flags=_abi.FLAG_NO_SLICES)
    assert not np.array_equal(one, a) and np.allclose(one, a, rtol=2e-6, atol=1e-6)
    # and the first input again (larger than what the scratch last held), counting variant, 3 slices
    ctx.set_scene(big)
    a, b = _both(ctx, big, 8, 96, seed=1, flags=_abi.FLAG_COUNTERS)
    assert np.array_equal(a, b)
    ctx.read_counters()
    # 1080p at 256 samples: 4 slices fit the static scratch against 8 in the allocated buffer — same samples, another
    # summation order
    hd = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 1920, 1080, 256)
    ctx.set_scene(hd)
    a, b = _both(ctx, hd, 0, 256, seed=3)
    assert a.any() and not np.array_equal(a, b) and np.allclose(a, b, rtol=2e-6, atol=1e-6)


def test_noalloc_entry_point_limits():
    import torch
    from pathtracer_amd import Scene, _abi, hip_renderer
    from pathtracer_amd.synthetic import synthetic_spheres
    ctx = hip_renderer.Context(0)
    dev = torch.device("cuda", 0)
    huge = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 4096, 1025, 1)   # > 4 194 304 pixels
    ctx.set_scene(huge)
    acc = torch.zeros(huge.width * huge.height * 3, dtype=torch.float32, device=dev)
    with pytest.raises(_abi.HipptError, match="exceed"):
        ctx.render_accum_device_noalloc(acc.data_ptr(), 0, 1)
    ok = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 4096, 1024, 2)     # exactly the limit: 2 slices
    ctx.set_scene(ok)
    acc2 = torch.zeros(ok.width * ok.height * 3, dtype=torch.float32, device=dev)
    ctx.render_accum_device_noalloc(acc.data_ptr(), 0, 2, seed=9)
    ctx.render_accum_device(acc2.data_ptr(), 0, 2, seed=9)
    ctx.kernel_time_ms()
    assert torch.equal(acc[:acc2.numel()], acc2)
    with pytest.raises(_abi.HipptError, match="megakernel only"):
        ctx.render_accum_device_noalloc(acc.data_ptr(), 0, 1, flags=_abi.FLAG_WAVEFRONT)
    many = synthetic_spheres(500, 3, 64, 48, 2)
    ctx.set_scene(many)
    with pytest.raises(_abi.HipptError, match="LBVH"):
        ctx.render_accum_device_noalloc(acc.data_ptr(), 0, 2)
    ctx.render_accum_device_noalloc(acc.data_ptr(), 0, 2, flags=_abi.FLAG_NO_BVH)         # brute-force loops: allowed
    torch.cuda.synchronize(dev)
