#!/bin/bash
# HBM traffic of the trace kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (they do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).
# usage: bash tools/run_traffic.sh <tag> [bench args...]
set -u
TAG=${1:-traffic}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || { echo "pass $i failed"; exit 1; }
done
python3 "$ROOT/tools/summarize_pmc.py" "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
