# A/B of libhippt variants on C4, interleaved on one box: bash tools/ab_c4.sh NAME_A NAME_B ...   (libhippt_NAME.so)
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in "$@"; do
    HIPPT_LIB=pathtracer_amd/lib/libhippt_$v.so timeout -k 10 120 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab/$v.$rep.json || exit 1
    python -c "
import json,sys
d=json.loads(open('gpurun_out/ab/$v.$rep.json').read().strip().splitlines()[-1]); print('$v', $rep, d['value'], d['ms_per_step'])"
  done
done
