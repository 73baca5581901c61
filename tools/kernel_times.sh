#!/bin/bash
# Per-kernel durations of one bench step under rocprofv3 (kernel trace + stats): tools/kernel_times.sh <tag> [bench args]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$TAG
(cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/err.txt) || exit 1
python3 - $R/gpurun_out/$TAG <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/stats/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 0.3:
        print(r['Name'][:60].ljust(62), r['Calls'].rjust(5), 'total ms', round(int(r['TotalDurationNs']) / 1e6, 1), 'avg', round(float(r['AverageNs']) / 1e6, 3), 'max', round(int(r['MaxNs']) / 1e6, 2))
PY
