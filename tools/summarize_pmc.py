"""Sums rocprofv3 --pmc CSV rows per kernel and prints derived VALU metrics."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "megakernel" not in k and "wf_" not in k and "wfb_" not in k:
            continue
        short = k.split("(")[0].replace("void pt::", "").replace("pt::", "")
        tot[short][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(tot.items()):
    print(f"== {k}")
    for n, v in sorted(c.items()):
        print(f"   {n:32s} {v:.6g}")
    g = c.get
    if g("SQ_INSTS_VALU") and g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        lanes = g('SQ_THREAD_CYCLES_VALU') / g('SQ_INSTS_VALU')
        print(f"   -> active lanes per VALU instruction : {lanes:.2f} of 64 ({lanes / 64:.3f})")
    if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
        print(f"   -> VALU instructions per wave       : {g('SQ_INSTS_VALU') / g('SQ_WAVES'):.0f}")
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_WAVE_CYCLES"):
        print(f"   -> VALU-issue share of wave cycles   : {g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'):.3f}")
    if g("SQ_WAIT_INST_ANY") and g("SQ_WAVE_CYCLES"):
        print(f"   -> wait-inst / wait-any share        : {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f} / {g('SQ_WAIT_ANY', 0) / g('SQ_WAVE_CYCLES'):.3f}")
    if g("SQ_INSTS_VALU_FMA_F32") is not None and g("SQ_INSTS_VALU_ADD_F32") is not None:
        fl = 64 * (g("SQ_INSTS_VALU_ADD_F32", 0) + g("SQ_INSTS_VALU_MUL_F32", 0) + g("SQ_INSTS_VALU_TRANS_F32", 0)) + 128 * g("SQ_INSTS_VALU_FMA_F32", 0)
        print(f"   -> hardware-counted f32 FLOP (64 lanes/inst upper bound): {fl:.4g}")
