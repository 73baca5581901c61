#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric: Mpaths/s (node) at 1080p; % of FP32 VALU peak.

A "step" is one pass of the hot path over the C2 workload: scene.example.json at
1920x1080, 1024 samples per pixel per GPU, max depth 5 (BASELINE configs[1]).
With N GPUs (one process per GPU, torch.distributed/RCCL) rank r renders the
global sample range [1024 r, 1024 (r+1)) of a 1024 N spp image (weak scaling,
sample-range split of SURVEY.md §8e); the per-GPU f32 radiance accumulators are
summed on rank 0 with one RCCL reduce and resolved (average, ACES, gamma, u8)
there, all inside the timed region. Scene tables and accumulators are resident
in HBM before the timed region starts.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 FLOP/clk/CU (packed f32)
# HBM bytes of one megakernel launch at 1920x1080 x 1024 spp from the PMC passes in profiles/r01_megakernel_v7_traffic_pmc.txt
# (FETCH_SIZE 3982 KB, doubled per the gfx950 correction, + WRITE_SIZE 275421 KB). The image itself is 8 sample
# slices of partial sums, 24.9 MB each (199 MB; pt_add_partials then reads them once); the rest is what the L2
# evicts of the 4 dwords per lane the kernel keeps in scratch across the lights loop ("Spills").
# Counters cannot be read inside this process.
MEGAKERNEL_1080P_TRAFFIC_BYTES = int((2 * 3981.56 + 275421.0) * 1024)


def cpu_baseline(threads):
    """CPU restatement of the reference renderer (oracle) on the host cores: bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from pathtracer_amd import Scene
    orc = oracle_lib.Oracle()
    spp = 64
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 1920, 1080, spp)
    t0 = time.perf_counter()
    orc.render_accum(s, 0, spp, seed=0, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": round(1920 * 1080 * spp / dt / 1e6, 4), "unit": "Mpaths/s", "cores": threads, "kind": "port",
            "sample": f"scene.example.json 1920x1080 at {spp} spp (same scene/size as the step, 1/16 of its samples), "
                      f"{dt:.1f} s; C++ restatement of src/renderer.rs, one task per row as src/main.rs:126"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024, help="samples per pixel per GPU per step")
    ap.add_argument("--scene", default=None, help="scene.json path (overrides --config's scene)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4"],
                    help="BASELINE config: c2 scene.example (default, the headline), c3 glass+fog, c4 100k spheres + LBVH")
    ap.add_argument("--max-depth", type=int, default=5)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pathtracer_amd import Scene, _abi, flops, hip_renderer

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    W, H, SPP = args.width, args.height, args.spp
    if args.scene:
        scene, scene_name = Scene.load(args.scene, W, H, SPP * world), os.path.basename(args.scene)
    elif args.config == "c4":
        from pathtracer_amd.synthetic import synthetic_spheres
        scene, scene_name = synthetic_spheres(100000, 0x5EED, W, H, SPP * world), "synthetic_spheres(n=100000, seed=0x5EED), LBVH"
    else:
        scene_name = {"c2": "scene.example.json", "c3": "glass_fog.json"}[args.config]
        scene = Scene.load(os.path.join(ROOT, "scenes", scene_name), W, H, SPP * world)
    ctx = hip_renderer.Context(local_rank)
    ctx.set_scene(scene)
    accum = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    rgba = torch.zeros(H * W * 4, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    s0 = rank * SPP

    def step(flags=args.flags):
        accum.zero_()
        ctx.render_accum_device(accum.data_ptr(), s0, SPP, seed=0, max_depth=args.max_depth, flags=flags,
                                stream=stream.cuda_stream)
        if world > 1:
            dist.reduce(accum, dst=0)  # RCCL sum of the per-GPU accumulators
        if rank == 0:
            ctx.resolve_device(accum.data_ptr(), SPP * world, rgba.data_ptr(), stream=stream.cuda_stream)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    ctx.kernel_time_ms()  # drop warmup launches from the kernel timing
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms, launches = ctx.kernel_time_ms()

    # image sanity on rank 0: finite, non-trivial
    if rank == 0:
        mean_rad = float(accum.mean().item()) / (SPP * world)
        assert mean_rad == mean_rad and mean_rad > 0.0, "image is empty or NaN"

    out = None
    if rank == 0:
        # event counts of exactly this rank's step (counting variant of the megakernel, untimed; the wavefront
        # variant traces the same paths, so its events are the same)
        accum.zero_()
        ctx.render_accum_device(accum.data_ptr(), s0, SPP, seed=0, max_depth=args.max_depth,
                                flags=(args.flags & ~_abi.FLAG_WAVEFRONT) | _abi.FLAG_COUNTERS,
                                stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        cn = ctx.read_counters()
        ctx.kernel_time_ms()
        fl = flops.algorithmic_flops(cn, len(scene.lights))
        paths_step = W * H * SPP * world
        ms_per_step = elapsed / args.steps * 1e3
        k_ms = kernel_ms / max(launches, 1)
        flop_per_launch = fl / (launches / args.steps)
        achieved = flop_per_launch / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "Mpaths/s (node) at 1080p; % of FP32 VALU peak",
            "value": round(paths_step / (elapsed / args.steps) / 1e6, 2),
            "unit": "Mpaths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config.upper()}: {scene_name} {W}x{H}, {SPP} spp per GPU ({SPP * world} spp image), "
                                   f"max_depth {args.max_depth}, seed 0",
                       "parallelism": f"sample-range split over {world} GPU(s), one RCCL reduce" if world > 1 else "1 GPU",
                       "kernel": ("wavefront+lbvh" if not args.flags & _abi.FLAG_MEGAKERNEL else "megakernel+lbvh")
                                 if len(scene.spheres) >= 64 and not args.flags & _abi.FLAG_NO_BVH else
                                 ("wavefront" if args.flags & _abi.FLAG_WAVEFRONT else "megakernel"),
                       "tables": "global" if (args.flags & _abi.FLAG_TABLES_GLOBAL or len(scene.spheres) >= 64) else "lds"},
            "mrays_per_s": round(flops.rays(cn) * world / (elapsed / args.steps) / 1e6, 1),
            "pct_fp32_valu_peak": round(100.0 * achieved / FP32_VALU_PEAK_TFLOPS, 2),
            "flop_per_path": round(fl / cn["paths"], 1),
            "roofline": {"bound": "valu_fp32", "achieved": round(achieved, 3), "peak": FP32_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 4),
                         "traffic": MEGAKERNEL_1080P_TRAFFIC_BYTES if (W, H, args.spp, args.flags, args.config) == (1920, 1080, 1024, 0, "c2") and not args.scene else None,
                         "kernel_ms": round(k_ms, 3), "launches_per_step": launches // args.steps,
                         "note": "algorithmic FLOPs (SURVEY §8d event costs x device event counters of this step) / "
                                 "mean trace-kernel duration from HIP events on the launch stream; the path is "
                                 "VALU-bound (no MFMA); traffic = HBM bytes per launch from the committed PMC passes "
                                 "(profiles/r01_megakernel_v7_traffic_pmc.txt): 0.2 GB of image partial sums + 0.1 GB of evicted scratch lines, "
                                 "~140 B per pixel per launch"},
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only: the other ranks would wait at the barrier
            out["cpu_baseline"] = cpu_baseline(min(16, os.cpu_count() or 1))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
