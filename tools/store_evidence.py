#!/usr/bin/env python3
"""Copy the evidence tools/collect_evidence.sh / collect_evidence_c4.sh left under gpurun_out/ into profiles/ and key
pathtracer_amd/pmc_index.json (what bench.py quotes as `roofline.traffic`, `lanes_active`, ...) to the current csrc hash.

    python tools/store_evidence.py --c4 gpurun_out/r03e_c4 --c4-name r03_c4_v2 --c2 gpurun_out/r03e_c2 --c2-name r03_megakernel_v11
"""
import argparse
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def parse(path):
    d, cur = {}, None
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith("== "):
            cur = line[3:].strip()
            d[cur] = {}
        elif cur is not None and line.startswith("   ") and not line.strip().startswith("->"):
            k, v = line.split()[:2]
            d[cur][k] = float(v)
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c4")
    ap.add_argument("--c4-name", default="r03_c4_v2")
    ap.add_argument("--c2")
    ap.add_argument("--c2-name", default="r03_megakernel_v11")
    a = ap.parse_args()
    prof = os.path.join(ROOT, "profiles")
    idx_path = os.path.join(ROOT, "pathtracer_amd", "pmc_index.json")
    idx = json.load(open(idx_path))
    h = bench.csrc_hash()
    if a.c4:
        n = a.c4_name
        shutil.copy(f"{a.c4}/bench_c4.json", f"{prof}/{n}_bench.json")
        for sub, dst in (("pmc", "pmc_summary"), ("traffic", "traffic_summary"), ("cache", "cache_summary")):
            shutil.copy(f"{a.c4}/{sub}/summary.txt", f"{prof}/{n}_{dst}.txt")
        shutil.copy(glob.glob(f"{a.c4}/stats/*/*kernel_stats.csv")[0], f"{prof}/{n}_kernel_stats.csv")
        T, C, P = (parse(f"{prof}/{n}_{k}.txt") for k in ("traffic_summary", "cache_summary", "pmc_summary"))
        clo, anyk = "wfb_traverse<false, false>", "wfb_traverse<true, false>"
        step = ["wfb_init", "wfb_logic", "wfb_shade", clo, anyk]
        F = sum(T[k]["FETCH_SIZE"] for k in step)
        W = sum(T[k]["WRITE_SIZE"] for k in step)
        e = idx["c4:1920x1080:256spp:flags0:depth5"]
        e["source"] = (f"profiles/{n}_traffic_summary.txt + profiles/{n}_cache_summary.txt + profiles/{n}_pmc_summary.txt "
                       "(64-spp runs of this command, traffic scaled x4 to the 256-spp step)")
        e["csrc_hash"] = h
        e["traffic_bytes"] = int((2 * F + W) * 1024 * 4)  # KiB units; fetch doubled per the gfx950 correction (upper bound)
        e["l2_hit_rate"] = round(C[clo]["TCC_HIT_sum"] / C[clo]["TCC_REQ_sum"], 3)
        e["l1_hit_rate"] = round(1 - C[clo]["TCP_TCC_READ_REQ_sum"] / C[clo]["TCP_TOTAL_CACHE_ACCESSES_sum"], 3)
        # active lanes per vector instruction as tools/summarize_pmc.py defines them, over the step's kernels
        e["lanes_active"] = round(sum(P[k]["SQ_THREAD_CYCLES_VALU"] for k in step) / sum(P[k]["SQ_INSTS_VALU"] for k in step) / 64, 3)
        for key, k in (("closest", clo), ("any", anyk)):  # 7 resident waves per SIMD (traversal grids, 66 registers)
            e["valu_busy"][key] = round(7 * P[k]["SQ_ACTIVE_INST_VALU"] / P[k]["SQ_WAVE_CYCLES"], 2)
        print("c4:", e["traffic_bytes"], "F+W TB", round((F + W) * 1024 * 4 / 1e12, 3), e["l2_hit_rate"], e["l1_hit_rate"], e["lanes_active"], e["valu_busy"]["closest"], e["valu_busy"]["any"])
    if a.c2:
        n = a.c2_name
        for src, dst in (("bench", "bench"), ("bench_depth12", "bench_depth12"), ("bench_c3", "bench_c3"), ("bench_c5", "bench_c5_128spp_chunk"),
                         ("bench_128spp", "bench_128spp")):
            shutil.copy(f"{a.c2}/{src}.json", f"{prof}/{n}_{dst}.json")
        shutil.copy(f"{a.c2}/pmc/summary.txt", f"{prof}/{n}_pmc_summary.txt")
        shutil.copy(f"{a.c2}/traffic/summary.txt", f"{prof}/{n}_traffic_pmc.txt")
        shutil.copy(glob.glob(f"{a.c2}/stats/*/*kernel_stats.csv")[0], f"{prof}/{n}_kernel_stats.csv")
        T, P = parse(f"{prof}/{n}_traffic_pmc.txt"), parse(f"{prof}/{n}_pmc_summary.txt")
        k = "pt_megakernel_lds<false, false>"
        e = idx["c2:1920x1080:1024spp:flags0:depth5"]
        e["source"] = f"profiles/{n}_pmc_summary.txt + profiles/{n}_traffic_pmc.txt"
        e["csrc_hash"] = h
        e["traffic_bytes"] = int((2 * T[k]["FETCH_SIZE"] + T[k]["WRITE_SIZE"]) * 1024)
        lanes = P[k]["SQ_THREAD_CYCLES_VALU"] / P[k]["SQ_INSTS_VALU"] / 64  # tools/summarize_pmc.py's definition
        e["lanes_active"] = round(lanes, 3)
        d = json.loads(open(f"{prof}/{n}_bench.json").read().strip().splitlines()[-1])
        flop = 64 * (P[k]["SQ_INSTS_VALU_ADD_F32"] + P[k]["SQ_INSTS_VALU_MUL_F32"] + P[k]["SQ_INSTS_VALU_TRANS_F32"]) + 128 * P[k]["SQ_INSTS_VALU_FMA_F32"]
        e["hw_flop_frac"] = round(flop * lanes / (d["ms_per_step"] * 1e-3) / 157.3e12, 3)
        print("c2:", e["traffic_bytes"], e["lanes_active"], e["hw_flop_frac"], d["value"], d["ms_per_step"])
    json.dump(idx, open(idx_path, "w"), indent=1)
    print("csrc hash", h)


if __name__ == "__main__":
    main()
