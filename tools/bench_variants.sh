#!/bin/bash
# A/B of prebuilt library variants in one GPU call: each variant is copied over libhippt.so and benched.
cd "${GRAFT_REPO_ROOT:-.}"
cp pathtracer_amd/lib/libhippt.so /tmp/libhippt_orig.so
for v in pathtracer_amd/lib/variants/*.so; do
  cp "$v" pathtracer_amd/lib/libhippt.so
  for rep in 1 2; do
    echo -n "$(basename $v) "; timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['pct_fp32_valu_peak'])"
  done
done
cp /tmp/libhippt_orig.so pathtracer_amd/lib/libhippt.so
