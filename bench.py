#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric: Mpaths/s (node) at 1080p; % of FP32 VALU peak.

A "step" is one pass of the hot path over the workload of --config (default C2 = BASELINE
configs[1]: scene.example.json at 1920x1080, 1024 samples per pixel, max depth 5).

N GPUs, one process per GPU (torch.distributed, backend nccl = RCCL). The default is the
north star's split — STRONG scaling: the image's --spp samples are cut into N contiguous
sample ranges (pathtracer_amd.partition.sample_range; the reference cuts its samples into
dispatches and sums them on the host, src/gpu_renderer.rs:94-102,175-179), rank r renders
all pixels for range r, the per-GPU f32 radiance accumulators are summed on rank 0 with ONE
RCCL reduce and resolved there (average, ACES, gamma, u8), all inside the timed region.
`--scaling weak` keeps --spp per GPU instead (an N * spp image). Scene tables and
accumulators are resident in HBM before the timed region starts.

`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks itself
(N child processes, before this process touches a GPU) and exits with their status; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks are already
there and each process is one of them.

Every rank's work of one step runs on ONE explicit HIP stream (accumulator clear, trace
kernels, RCCL reduce, resolve), so the steps are ordered among themselves and the timed
region is the render -> reduce -> resolve chain.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 FLOP/clk/CU (packed f32)
L2_PEAK_TBS = 34.5             # MI355X_MICROARCH.md "L2 (per XCD)": aggregate over the 8 XCDs
HBM_PEAK_TBS = 8.0

# Workloads (BASELINE.json configs): width, height, samples per pixel of one step, scene.
CONFIGS = {
    "c2": (1920, 1080, 1024, "scene.example.json"),
    "c3": (1920, 1080, 4096, "glass_fog.json"),
    "c4": (1920, 1080, 256, None),   # synthetic_spheres(100000, 0x5EED), LBVH
    # the showcase job is 131072 spp (4.3 T paths, ~11 min on one GPU): a step is one 128-spp chunk of it
    "c5": (7680, 4320, 128, "scene.example.json"),   # "128-spp chunk of C5"
}

# Counter figures that cannot be read inside this process (rocprofv3 --pmc runs of this same command, kept under
# profiles/). Keyed by (config, width, height, spp per launch, flags, max_depth); anything else reports null.
PMC_INDEX = os.path.join(ROOT, "pathtracer_amd", "pmc_index.json")  # travels to the GPU box (profiles/ does not)


def pmc_lookup(config, W, H, spp_launch, flags, max_depth):
    try:
        with open(PMC_INDEX) as f:
            idx = json.load(f)
    except OSError:
        return None
    return idx.get(f"{config}:{W}x{H}:{spp_launch}spp:flags{flags}:depth{max_depth}")


def cpu_baseline(threads):
    """CPU restatement of the reference renderer (oracle source) on the host cores, as SURVEY §8(d) defines the CPU
    baseline: its own -O3 -march=native build (compiled here, on the machine that times it; the checker library stays
    -O2), C1 exactly (scene.example.json 320x240x16) plus 1920x1080 (at 64 spp, so that the sample lasts 10-30 s) for a per-path rate comparable with C2.
    `value` is the 1080p rate."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from pathtracer_amd import Scene
    try:
        tmp = tempfile.mkdtemp(prefix="hippt_cpu_baseline_")
        path, flags = oracle_lib.build_fast_oracle(tmp)
        orc = oracle_lib.Oracle(path)
        build = f"g++ {flags}"
    except Exception as e:  # no compiler on this machine: the checker build, said so
        orc = oracle_lib.Oracle()
        build = f"g++ -O2 -ffp-contract=off (checker library; the -O3 -march=native build failed here: {type(e).__name__})"
    ex = os.path.join(ROOT, "scenes", "scene.example.json")
    rates = {}
    # (1080p at 64 spp rather than §8d's 16: the sample should last 10-30 s; 16 spp is 3 s on 16 threads)
    for name, (w, h, spp) in (("c1_320x240x16", (320, 240, 16)), ("1080p_x64", (1920, 1080, 64))):
        s = Scene.load(ex, w, h, spp)
        orc.render_accum(s, 0, 1, seed=0, threads=threads)  # warm the threads / caches
        t0 = time.perf_counter()
        orc.render_accum(s, 0, spp, seed=0, threads=threads)
        dt = time.perf_counter() - t0
        rates[name] = (w * h * spp / dt / 1e6, dt)
    v, dt = rates["1080p_x64"]
    return {"value": round(v, 4), "unit": "Mpaths/s", "cores": threads, "kind": "port",
            "host_cores_online": os.cpu_count(), "build": build,
            "c1_mpaths_per_s": round(rates["c1_320x240x16"][0], 4), "c1_seconds": round(rates["c1_320x240x16"][1], 4),
            "sample": f"scene.example.json at 1920x1080 x 64 spp ({dt:.1f} s; C2's scene and size, 1/16 of its samples) and C1 exactly (320x240 x 16 spp, "
                      f"{rates['c1_320x240x16'][1]:.2f} s) on {threads} threads (the GPU box's CPU share), SURVEY §8(d); "
                      f"C++ restatement of src/renderer.rs, one task per row as src/main.rs:126"}


def csrc_hash():
    """Hash of the kernel and host sources the library is built from: ties the PMC figures of pmc_index.json to a build."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pathtracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None,
                    help="samples per pixel of the image (strong scaling) or per GPU (weak); default: the config's")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--scene", default=None, help="scene.json path (overrides --config's scene)")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS),
                    help="BASELINE config: c2 scene.example (default, the headline), c3 glass+fog, c4 100k spheres + LBVH, "
                         "c5 scene.example at 7680x4320")
    ap.add_argument("--max-depth", type=int, default=5)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 rehearsal on a box with ONE GPU: every rank renders on device 0 and the accumulators are "
                         "summed through gloo on the host instead of RCCL (same split, timing and JSON code; not a measurement)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: gloo ranks exchange their sample counts instead of rendering (launcher / split / reduce check)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """Starts n ranks of this script (one per GPU) and waits for them. Runs before anything touches a GPU: the
    parent never initialises HIP and never replaces itself, the ranks are fresh child processes."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with status {code}; stopping the other ranks", file=sys.stderr)
                for q in pending:
                    procs[q].terminate()  # exactly the children started above
        time.sleep(0.05)
    return rc


def dry_run(args, rank, world):
    """Launcher / partition / reduce logic without a GPU: every rank contributes its sample count per element."""
    import torch
    import torch.distributed as dist
    from pathtracer_amd.partition import sample_range
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    spp = args.spp or CONFIGS[args.config][2]
    total = spp if args.scaling == "strong" else spp * world
    b, c = sample_range(total, rank, world)
    acc = torch.full((16,), float(c), dtype=torch.float32)
    ranges = [None] * world
    if world > 1:
        dist.reduce(acc, dst=0)
        dist.all_gather_object(ranges, (b, c))
        dist.barrier()
        dist.destroy_process_group()
    else:
        ranges = [(b, c)]
    if rank == 0:
        assert float(acc[0]) == float(total), (acc, total)
        print(json.dumps({"dry_run": True, "n_gpus": world, "world_size": world, "scaling": args.scaling,
                          "spp_total": total, "sample_ranges": ranges}), flush=True)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; running {world} rank(s)", file=sys.stderr)
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch
    import torch.distributed as dist
    from pathtracer_amd import Scene, _abi, flops, hip_renderer
    from pathtracer_amd.partition import sample_range

    rehearse = args.rehearse_on_one_gpu and world > 1
    if rehearse:
        local_rank = 0
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearse else dev  # where the small bookkeeping collectives run

    cW, cH, cS, cscene = CONFIGS[args.config]
    W, H = args.width or cW, args.height or cH
    spp_arg = args.spp or cS
    total_spp = spp_arg if args.scaling == "strong" else spp_arg * world
    s0, cnt = sample_range(total_spp, rank, world)
    if args.scene:
        scene, scene_name = Scene.load(args.scene, W, H, total_spp), os.path.basename(args.scene)
    elif args.config == "c4":
        from pathtracer_amd.synthetic import synthetic_spheres
        scene, scene_name = synthetic_spheres(100000, 0x5EED, W, H, total_spp), "synthetic_spheres(n=100000, seed=0x5EED), LBVH"
    else:
        scene, scene_name = Scene.load(os.path.join(ROOT, "scenes", cscene), W, H, total_spp), cscene
    ctx = hip_renderer.Context(local_rank)
    ctx.set_scene(scene)
    accum = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    rgba = torch.zeros(H * W * 4, dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream(dev)  # a real (non-NULL) stream: NULL would mean "the context's own stream" to the C ABI
    assert stream.cuda_stream != 0

    reduce_events = []  # (start, end) around the RCCL reduce of each timed step, on the launch stream

    def step(flags=args.flags, timed=False):
        with torch.cuda.stream(stream):
            accum.zero_()
            ctx.render_accum_device(accum.data_ptr(), s0, cnt, seed=0, max_depth=args.max_depth, flags=flags,
                                    stream=stream.cuda_stream)
            if world > 1:
                if timed:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record(stream)
                if rehearse:  # one-GPU rehearsal: the sum goes through the host
                    stream.synchronize()
                    host = accum.cpu()
                    dist.reduce(host, dst=0)
                    accum.copy_(host)
                else:
                    dist.reduce(accum, dst=0)  # RCCL sum of the per-GPU accumulators, queued behind the kernels
                if timed:
                    ev[1].record(stream)
                    reduce_events.append(ev)
            if rank == 0:
                ctx.resolve_device(accum.data_ptr(), total_spp, rgba.data_ptr(), stream=stream.cuda_stream)

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
        step(timed=True)
    sync()
    elapsed = time.perf_counter() - t0
    devices = [local_rank]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ids = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([local_rank], dtype=torch.int64, device=cdev))
        devices = [int(x.item()) for x in ids]
    kernel_ms, launches = ctx.kernel_time_ms()
    per_rank = None
    if world > 1:
        # what a scaling curve needs to be read: the trace kernels' time per step on the slowest and the fastest rank,
        # and the time the launch stream spends in the reduce (which includes waiting for the slowest rank's kernels)
        k_step = kernel_ms / max(args.steps, 1)
        red = sum(a.elapsed_time(b) for a, b in reduce_events) / max(len(reduce_events), 1)
        tmax = torch.tensor([k_step, red], dtype=torch.float64, device=cdev)
        tmin = tmax.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        per_rank = {"kernel_ms_max": round(float(tmax[0]), 3), "kernel_ms_min": round(float(tmin[0]), 3),
                    "reduce_ms": round(float(tmax[1]), 3), "reduce_ms_min": round(float(tmin[1]), 3),
                    "note": "per step: trace-kernel time (HIP events on the launch stream) of the slowest / fastest rank; reduce_ms = "
                            "launch-stream time inside torch.distributed.reduce (RCCL), max / min over ranks — a rank that "
                            "finishes its kernels early waits there for the others"}

    out = None
    if rank == 0:
        # image sanity: finite, non-trivial, and exactly ONE step's worth of samples in the accumulator
        # The reference's BRDF has poles (ggx.rs:10-12: q = nh^2 (a^2 - 1) + 1 reaches 0 when a far, small sphere's
        # normal (hit - c) / r comes out 0.6 % too long and n.h = 1.0016): C4's step holds ONE such sample of 5.3e8, 9e9 in
        # the oracle's rounding and inf in the kernel's. Either way the pixel is white (renderer.rs:187-191, quirk
        # Q13); the sanity sums cap a pixel at a mean radiance of 1e4 so that it does not decide them.
        cap = 1e4 * total_spp
        robust_sum = lambda a: float(torch.nan_to_num(a, nan=0.0, posinf=cap, neginf=0.0).clamp(max=cap).double().sum().item())
        nonfinite_pixels = int((~torch.isfinite(accum.view(-1, 3)).all(dim=1)).sum().item())
        step_sum = robust_sum(accum)
        mean_rad = step_sum / accum.numel() / total_spp
        assert mean_rad == mean_rad and mean_rad > 0.0, "image is empty or NaN"

        # event counts of exactly this rank's share of a step (counting variant of the megakernel, untimed; the queue
        # variants trace the same paths, so the path-level events are the same — for LBVH scenes the sphere-test
        # counts are those of the BVH megakernel's traversal order, see "flop_note")
        with torch.cuda.stream(stream):
            accum.zero_()
            ctx.render_accum_device(accum.data_ptr(), s0, cnt, seed=0, max_depth=args.max_depth,
                                    flags=(args.flags & ~_abi.FLAG_WAVEFRONT) | _abi.FLAG_COUNTERS,
                                    stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        if world == 1:  # the counting kernel renders the same samples: a step that ran twice, or not at all, shows here
            count_sum = robust_sum(accum)
            assert abs(step_sum - count_sum) <= 2e-3 * abs(count_sum), (step_sum, count_sum)
        cn = ctx.read_counters()
        ctx.kernel_time_ms()
        fl = flops.algorithmic_flops(cn, len(scene.lights))
        paths_step = W * H * total_spp
        ms_per_step = elapsed / args.steps * 1e3
        k_ms = kernel_ms / max(launches, 1)
        launches_per_step = max(launches // max(args.steps, 1), 1)
        flop_per_launch = fl / launches_per_step
        achieved = flop_per_launch / (k_ms * 1e-3) / 1e12
        lbvh = len(scene.spheres) >= 64 and not args.flags & _abi.FLAG_NO_BVH
        kernel = (("wavefront+lbvh" if not args.flags & _abi.FLAG_MEGAKERNEL else "megakernel+lbvh") if lbvh else
                  ("wavefront" if args.flags & _abi.FLAG_WAVEFRONT else "megakernel"))
        pmc = pmc_lookup(args.config if not args.scene else "scene", W, H, cnt // launches_per_step, args.flags, args.max_depth)
        # SURVEY §8(d) credits shadow tests AS EXECUTED. `frac` above prices the step at the reference's work (the tests
        # renderer.rs:46 would run, counted by a kernel that runs the reference's loops); the timed kernel skips tests behind
        # remembered occluders and plane rows that cannot block a light. A second untimed pass runs the TIMED kernel's own
        # code with counters on its tests and evaluations (HIPPT_FLAG_COUNT_EXECUTED): `frac_as_executed`.
        fl_exec = cn_exec = None
        if kernel == "megakernel":
            with torch.cuda.stream(stream):
                accum.zero_()
                ctx.render_accum_device(accum.data_ptr(), s0, cnt, seed=0, max_depth=args.max_depth,
                                        flags=args.flags | _abi.FLAG_COUNTERS | _abi.FLAG_COUNT_EXECUTED, stream=stream.cuda_stream)
            torch.cuda.synchronize(dev)
            if world == 1:
                exec_sum = robust_sum(accum)
                assert abs(step_sum - exec_sum) <= 2e-3 * abs(exec_sum), (step_sum, exec_sum)
            cn_exec = ctx.read_counters()
            ctx.kernel_time_ms()
            fl_exec = flops.algorithmic_flops(cn_exec, len(scene.lights))
        out = {
            "metric": "Mpaths/s (node) at 1080p; % of FP32 VALU peak",
            "value": round(paths_step / (elapsed / args.steps) / 1e6, 2),
            "unit": "Mpaths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic (generated 100k-sphere scene)" if args.config == "c4" and not args.scene else
                     f"{scene_name} (the reference's example scene; no external data)"),
            "world_size": dist.get_world_size() if world > 1 else 1,
            "devices": devices,
            "config": {"workload": f"{'128-spp chunk of C5' if args.config == 'c5' and spp_arg == 128 else args.config.upper()}: {scene_name} {W}x{H}, {total_spp} spp image, "
                                   f"{cnt} spp on rank 0, max_depth {args.max_depth}, seed 0",
                       "parallelism": (f"sample-range split over {world} GPUs ({args.scaling} scaling), one RCCL reduce"
                                       if world > 1 else "1 GPU"),
                       "kernel": kernel,
                       "tables": "global" if (args.flags & _abi.FLAG_TABLES_GLOBAL or lbvh) else "lds"},
            "mrays_per_s": round(flops.rays(cn) * (paths_step / max(cn["paths"], 1)) / (elapsed / args.steps) / 1e6, 1),
            "pct_fp32_valu_peak": round(100.0 * achieved / FP32_VALU_PEAK_TFLOPS, 2),
            "flop_per_path": round(fl / cn["paths"], 1),
            "nonfinite_pixels": nonfinite_pixels,
            "roofline": {"bound": "valu_fp32", "achieved": round(achieved, 3), "peak": FP32_VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 4),
                         "traffic": None,
                         "kernel_ms": round(k_ms, 3), "launches_per_step": launches_per_step,
                         "note": "rank 0's GPU: algorithmic FLOPs (SURVEY §8d event costs x device event counters of this "
                                 "rank's share of a step) / mean trace-kernel duration from HIP events on the launch stream; "
                                 "the path is VALU-bound (no MFMA). frac credits shadow tests as the REFERENCE executes them "
                                 "(scene order up to the first occluder, renderer.rs:46); frac_as_executed credits the tests and "
                                 "BRDF evaluations the timed kernel itself executes (SURVEY §8d's rule)"},
        }
        if fl_exec is not None:
            ach_exec = fl_exec / launches_per_step / (k_ms * 1e-3) / 1e12
            out["roofline"]["frac_as_executed"] = round(ach_exec / FP32_VALU_PEAK_TFLOPS, 4)
            out["roofline"]["achieved_as_executed"] = round(ach_exec, 3)
            out["flop_per_path_as_executed"] = round(fl_exec / cn_exec["paths"], 1)
            out["shadow_tests_per_path"] = {
                "reference": round((cn["shadow_sphere_tests"] + cn["shadow_plane_tests"]) / cn["paths"], 2),
                "executed": round((cn_exec["shadow_sphere_tests"] + cn_exec["shadow_plane_tests"]) / cn_exec["paths"], 2)}
        pmc_pending = pmc
        if lbvh and kernel == "wavefront+lbvh":
            # The LBVH queue pipeline is bound by what its rays FETCH, not by arithmetic (PMC: the traversal kernels wait on
            # memory two thirds of their cycles, profiles/r0*_c4_*): its roofline is bytes against the L2 rate of the
            # guide. Algorithmic bytes of a step, from an untimed pass that counts the traversal's fetches
            # (HIPPT_FLAG_TRAVERSAL_STATS) and from the event counters above; layouts in pt_wfqueue.h / DESIGN.md §3:
            with torch.cuda.stream(stream):
                accum.zero_()
                ctx.render_accum_device(accum.data_ptr(), s0, cnt, seed=0, max_depth=args.max_depth,
                                        flags=args.flags | _abi.FLAG_TRAVERSAL_STATS, stream=stream.cuda_stream)
            torch.cuda.synchronize(dev)
            ts = ctx.read_traversal_stats()
            ctx.kernel_time_ms()
            L = len(scene.lights)
            E, V = cn["extend_rays"], cn["nee_vertices"]
            # wide nodes (64 B) that are not among the first ones kept in the block's LDS, spheres (16 B), aux rows on exact ties
            node_b, leaf_b = 64, 16
            trav = sum((ts[k]["nodes"] - ts[k]["nodes_from_lds"]) * node_b + ts[k]["leaf_fetches"] * leaf_b + ts[k]["aux_rows"] * 16
                       for k in ("closest", "any"))
            # closest ray: origin + direction in, the sphere's index out; shadow rays: the traversal kernel reads a vertex's point,
            # normal and stream state once (32 B, its 4 L samples are neighbouring lanes) and writes one result BIT per sample
            rays_io = ts["closest"]["rays"] * (24 + 4) + V * (32 + 4 * L / 8)
            # per extend ray: ray entry + sphere index in (56 + 4), next entry out (56); per NEE vertex: shade entry out (72),
            # back in with its result bits for shading (72 + 4 L / 8), which regenerates the samples from the stream state
            # (the streams' running sums live in their slices: 12 B read + written only where a vertex receives light, not counted)
            queues = E * (60 + 56) + V * (72 + 72 + L * 4 / 8)
            step_bytes = int(trav + rays_io + queues)
            gbs = step_bytes / (k_ms * launches_per_step * 1e-3) / 1e9
            out["roofline"] = {
                "bound": "l2", "achieved": round(gbs, 1), "peak": L2_PEAK_TBS * 1e3, "unit": "GB/s",
                "frac": round(gbs / (L2_PEAK_TBS * 1e3), 4), "traffic": None,
                "kernel_ms": round(k_ms, 3), "launches_per_step": launches_per_step,
                "bytes_per_step": {"traversal_fetches": int(trav), "ray_io": int(rays_io), "queue_entries": int(queues)},
                "per_ray": {k: {"nodes": round(ts[k]["nodes"] / max(ts[k]["rays"], 1), 1),
                                "nodes_from_lds": round(ts[k]["nodes_from_lds"] / max(ts[k]["rays"], 1), 1),
                                "leaf_fetches": round(ts[k]["leaf_fetches"] / max(ts[k]["rays"], 1), 1),
                                "bytes": round(((ts[k]["nodes"] - ts[k]["nodes_from_lds"]) * node_b + ts[k]["leaf_fetches"] * leaf_b + ts[k]["aux_rows"] * 16) / max(ts[k]["rays"], 1))}
                            for k in ("closest", "any")},
                "valu_fp32": {"achieved_tflops": round(achieved, 3), "frac": round(achieved / FP32_VALU_PEAK_TFLOPS, 4),
                              "note": "sphere / plane tests and shading only: the cost table of SURVEY §8d prices no box tests"},
                "limiter": "valu_issue + fetch latency (PMC: a vector instruction in flight on every SIMD cycle of the traversal kernels at 60-61 % "
                           "active lanes; sensitivity runs: more issue work and fewer resident waves both slow the step, DESIGN.md §3; "
                           "the byte roofline says how far the fetch path is from the L2's rate, not what limits the step)",
                "note": "rank 0's GPU: algorithmic bytes of a step (64 B per 4-wide BVH node visited outside the LDS-resident first levels, "
                        "16 B per sphere fetched, 16 B per aux row, ray records, queue entries read and written by the logic / shading kernels) "
                        "/ duration of the step's kernels from HIP events on the launch stream, against the aggregate L2 rate of "
                        "MI355X_MICROARCH.md (34.5 TB/s); the node and sphere fetches are per-lane 16-byte gathers, most of which "
                        "hit the vector L1",
            }
            out["flop_note"] = ("event counts come from the BVH megakernel (HIPPT_FLAG_COUNTERS); its traversal visits "
                                "spheres in another order than the queue kernels, so sphere-test counts differ slightly")
        if pmc_pending:  # figures from committed rocprofv3 --pmc passes of this command: NOT measured in this run
            r = out["roofline"]
            r["traffic"] = pmc_pending.get("traffic_bytes")
            r["traffic_source"] = f"{pmc_pending.get('source')} (rocprofv3 --pmc pass of this command on an earlier run; not measured in this run)"
            for k in ("hw_flop_frac", "lanes_active", "l2_hit_rate", "l1_hit_rate", "valu_busy"):
                if k in pmc_pending:
                    r[k] = pmc_pending[k]
            if "note" in pmc_pending:
                r["pmc_note"] = pmc_pending["note"]
            # the PMC figures belong to the build they were measured on
            r["pmc_csrc_hash"] = pmc_pending.get("csrc_hash")
            r["pmc_stale"] = pmc_pending.get("csrc_hash") != csrc_hash()
            if r.get("bound") == "l2" and r.get("traffic"):  # north star: "for wavefront queues, achieved HBM GB/s is also reported"
                r["hbm_gbs"] = round(r["traffic"] / (r["kernel_ms"] * r["launches_per_step"] * 1e-3) / 1e9, 1)
                r["hbm_frac"] = round(r["hbm_gbs"] / 8000.0, 4)
        if per_rank:
            out["per_rank"] = per_rank
        if rehearse:
            out["rehearsal"] = "every rank on device 0, host-side gloo sum: NOT a measurement of N GPUs"
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only: the other ranks would wait at the barrier
            out["cpu_baseline"] = cpu_baseline(min(16, os.cpu_count() or 1))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
