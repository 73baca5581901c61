"""SURVEY §8e / BASELINE §5 with the real kernels: N PROCESSES, each with its own HIP context (here on the one GPU of the
box), render their sample ranges of the same image; the accumulators are summed across processes (gloo here; RCCL in
bench.py) and must equal the one-process image up to f32 summation order — for the megakernel and for the LBVH pipeline.
What differs from bench.py's N > 1 path is only the transport of the final sum."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(kind):
    from pathtracer_amd import Scene
    if kind == "example":
        return Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 160, 90, 37), 37
    from pathtracer_amd.synthetic import synthetic_spheres
    return synthetic_spheres(3000, 0x5EED, 128, 72, 21), 21


def _worker(rank, world, port, kind, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from pathtracer_amd import hip_renderer
    from pathtracer_amd.partition import sample_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, spp = _scene(kind)
    b, c = sample_range(spp, rank, world)
    acc = torch.from_numpy(hip_renderer.render_accum(s, b, c, seed=3))
    dist.reduce(acc, dst=0)
    if rank == 0:
        q.put(acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,world", [("example", 2), ("example", 3), ("spheres", 2)])
def test_processes_sum_to_the_one_process_image(kind, world):
    import torch.multiprocessing as mp
    from pathtracer_amd import hip_renderer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    s, spp = _scene(kind)
    ref = hip_renderer.render_accum(s, 0, spp, seed=3)
    assert ref.mean() > 0 and np.allclose(got, ref, rtol=3e-6, atol=1e-6)
