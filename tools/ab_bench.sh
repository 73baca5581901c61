#!/bin/bash
# A/B timing of two builds of libhippt.so on the same box, interleaved: tools/ab_bench.sh A.so B.so [bench args]
set -u
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in "$A" "$B"; do
    HIPPT_LIB=$L timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['roofline']['kernel_ms'])" || exit 1
  done
done
