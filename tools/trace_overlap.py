"""Reads a rocprofv3 --kernel-trace CSV and prints, per kernel name, the summed duration, and for the whole trace the
union of the busy intervals (wall time with at least one kernel running) and the time with two or more kernels running.
usage: python tools/trace_overlap.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in rows)
per = collections.Counter()
for a, b, k in iv:
    per[k] += b - a
ev = sorted([(a, 1) for a, b, k in iv] + [(b, -1) for a, b, k in iv])
busy = multi = 0
depth = 0
last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: multi += t - last
    depth += d
    last = t
print("sum of kernel durations ms", sum(per.values()) / 1e6, "busy (union) ms", busy / 1e6, ">=2 kernels ms", multi / 1e6)
for k, v in per.most_common(8):
    print("  %-42s %9.2f ms" % (k, v / 1e6))
