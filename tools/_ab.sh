timeout -k 10 600 python -u -m pytest tests/test_gpu_bvh.py tests/test_gpu_full_size.py -x -q --timeout 300 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -1 gpurun_out/t.log
for i in 1 2; do for L in _base ""; do HIPPT_LIB=pathtracer_amd/lib/libhippt$L.so timeout -k 10 300 python bench.py --no-cpu-baseline --config c4 --steps 1 --spp 64 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$L', d['value'], d['mrays_per_s'])" || exit 1; done; done
