# Round evidence for the megakernel on the GPU box (from the repo root): GPU tests, bench lines, rocprofv3 kernel stats,
# PMC and HBM-traffic passes, all under gpurun_out/v7/; copy what is to be kept into profiles/.
set -u
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
mkdir -p gpurun_out/v7
timeout -k 10 400 python bench.py > gpurun_out/v7/bench.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --max-depth 12 > gpurun_out/v7/bench_depth12.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --config c3 > gpurun_out/v7/bench_c3.json || exit 1
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v7/stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/v7/stats_bench.json 2> $R/gpurun_out/v7/stats.err) || exit 1
bash tools/run_pmc.sh v7/pmc > /dev/null || exit 1
bash tools/run_traffic.sh v7/traffic > /dev/null || exit 1
echo evidence done
