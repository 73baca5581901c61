#!/bin/bash
# L2 (TCC) and vector-L1 (TCP) counters of the traversal kernels: usage tools/run_cache_pmc.sh <tag> [bench args]
set -u
TAG=${1:-cache}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for P in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || { echo "pass $i failed"; tail -3 "$OUT/pass$i.err"; exit 1; }
done
python3 "$ROOT/tools/summarize_pmc.py" "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
