#!/bin/bash
# PMC passes for the trace kernel (one rocprofv3 run per pass; no tracing domains mixed in).
# usage (on the GPU box, from the repo root): bash tools/run_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
"SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
"SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_INT32"
"SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_CVT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || { echo "pass $i failed"; exit 1; }
done
python3 "$ROOT/tools/summarize_pmc.py" "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
