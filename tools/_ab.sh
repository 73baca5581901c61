timeout -k 10 300 python -u -m pytest tests/test_gpu_render_parity.py tests/test_gpu_cli.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -1
for i in 1 2 3; do for L in _base ""; do HIPPT_LIB=pathtracer_amd/lib/libhippt$L.so timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$L', d['value'])" || exit 1; done; done
bash tools/run_traffic.sh trx > /dev/null; grep -A2 "lds<false>" gpurun_out/trx/summary.txt
