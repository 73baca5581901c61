# C4 evidence on the GPU box: bench line, rocprofv3 kernel stats, PMC (SQ passes, HBM traffic, cache counters).
# usage: bash tools/collect_evidence_c4.sh <tag>
set -u
TAG=${1:-r02c4}
mkdir -p gpurun_out/$TAG
timeout -k 10 400 python bench.py --config c4 --no-cpu-baseline > gpurun_out/$TAG/bench_c4.json || exit 1
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -- python3 $R/bench.py --config c4 --no-cpu-baseline --steps 1 --warmup 0 > $R/gpurun_out/$TAG/stats_bench.json 2> $R/gpurun_out/$TAG/stats.err) || exit 1
bash tools/run_pmc.sh $TAG/pmc --config c4 --spp 64 > /dev/null || exit 1
bash tools/run_traffic.sh $TAG/traffic --config c4 --spp 64 > /dev/null || exit 1
bash tools/run_cache_pmc.sh $TAG/cache --config c4 --spp 64 > /dev/null || exit 1
echo evidence done
