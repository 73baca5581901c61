# Round evidence for the megakernel on the GPU box (from the repo root): bench lines, rocprofv3 kernel stats, PMC and
# HBM-traffic passes, all under gpurun_out/<tag>/; copy what is to be kept into profiles/.
# usage: bash tools/collect_evidence.sh <tag>
set -u
TAG=${1:-r02}
mkdir -p gpurun_out/$TAG
timeout -k 10 400 python bench.py > gpurun_out/$TAG/bench.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --max-depth 12 > gpurun_out/$TAG/bench_depth12.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --config c3 > gpurun_out/$TAG/bench_c3.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --config c5 > gpurun_out/$TAG/bench_c5.json || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --spp 128 --steps 10 > gpurun_out/$TAG/bench_128spp.json || exit 1
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/$TAG/stats_bench.json 2> $R/gpurun_out/$TAG/stats.err) || exit 1
bash tools/run_pmc.sh $TAG/pmc > /dev/null || exit 1
bash tools/run_traffic.sh $TAG/traffic > /dev/null || exit 1
echo evidence done
