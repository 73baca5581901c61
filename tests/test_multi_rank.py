"""N > 1 path on CPU: two gloo ranks split the sample range, reduce their
accumulators to rank 0 and get the single-rank image up to f32 summation order. The GPU
kernels cannot run here, so the per-rank renderer is the oracle; what is tested is the
partitioning + reduce + resolve logic that bench.py uses with RCCL."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from pathtracer_amd import Scene
    from pathtracer_amd.partition import sample_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spp = 12
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 48, 36, spp)
    b, c = sample_range(spp, rank, world)
    acc = torch.from_numpy(oracle_lib.Oracle().render_accum(s, b, c, seed=3, threads=2))
    dist.reduce(acc, dst=0)
    if rank == 0:
        q.put(acc.numpy())
    dist.destroy_process_group()


def test_sample_range_partition():
    from pathtracer_amd.partition import sample_range
    for spp in (1, 7, 12, 1024, 131072):
        for world in (1, 2, 3, 8):
            r = [sample_range(spp, k, world) for k in range(world)]
            assert r[0][0] == 0 and sum(c for _, c in r) == spp
            for (b0, c0), (b1, _c1) in zip(r, r[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in r) - min(c for _, c in r) <= 1


def test_two_ranks_reduce_to_single_rank_image(oracle):
    import torch.multiprocessing as mp
    from pathtracer_amd import Scene
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 48, 36, 12)
    ref = oracle.render_accum(s, 0, 12, seed=3)
    assert np.allclose(got, ref, rtol=2e-6, atol=1e-6)
