#!/bin/bash
# Profile the headline bench on the GPU box and reduce the results into profiles/<round>/<tag>_*.
#   tools/profile_bench.sh r02 a_cold            (run from the repo root under gpurun)
# 1. bench.py as the driver runs it (default flags, and --steps 20)         -> <tag>_bench.json, <tag>_bench_steps20.json
# 2. rocprofv3 --kernel-trace --stats of `bench.py --cache cold`            -> <tag>_kernel_stats.csv
# 3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command -> <tag>_pmc_{fetch,write}_summary.csv,
#    <tag>_pmc_traffic.json and the merged profiles/pmc_traffic.json (key = cache state)
set -e -o pipefail
ROUND=${1:-r02}; TAG=${2:-a}; STATE=${3:-cold}
OUT=gpurun_out/$ROUND; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 3000 $OUT/${TAG}_bench.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_steps20.json 2>> $OUT/${TAG}_bench.err
CMD="python3 bench.py --cache $STATE --steps 300 --warmup 20 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o s -- $CMD > $OUT/${TAG}_stats.log 2>&1
cp $(find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -o f -- $CMD > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -o w -- $CMD > $OUT/${TAG}_write.log 2>&1
cp -n profiles/pmc_traffic.json $OUT/pmc_traffic_merged.json 2>/dev/null || true
F=$(find $OUT/${TAG}_fetch -name '*counter_collection.csv' | head -1)
Wf=$(find $OUT/${TAG}_write -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py $F $Wf $OUT/${TAG}_pmc_traffic.json $STATE $OUT/pmc_traffic_merged.json
head -12 $OUT/${TAG}_kernel_stats.csv
