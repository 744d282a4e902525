#!/bin/bash
# Per-kernel device time of the plan-based chains (development tool, run under gpurun from the repo root):
#   tools/chain_stats.sh <tag> "<which> <chain>" ["<which> <chain>" ...]
# -> gpurun_out/chains/<tag>_<which>_<chain>_kernel_stats.csv and a compact table on stdout
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/chains; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg
  name=${TAG}_$1_$2
  rm -rf $OUT/$name.d
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name.d -o p -- python3 tools/chain_profile.py $1 $2 > $OUT/$name.log 2>&1 || { tail -20 $OUT/$name.log; exit 1; }
  cp $(find $OUT/$name.d -name '*kernel_stats.csv' | head -1) $OUT/${name}_kernel_stats.csv
  grep "us/step" $OUT/$name.log
  python3 - $OUT/${name}_kernel_stats.csv <<'PY'
import csv, re, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if 'advx' not in r['Name']:
        continue
    name = re.sub(r'\(.*', '', r['Name']).replace('void advx::', '').replace('advx::', '')
    per_step = float(r['TotalDurationNs']) / 60.0 / 1e3
    tot += per_step
    print(f"   {name:28s} calls/step {int(r['Calls'])/60:4.1f}  avg {float(r['AverageNs'])/1e3:7.2f} us  per step {per_step:7.2f} us")
print(f"   sum of kernel time per step {tot:7.2f} us")
PY
  rm -rf $OUT/$name.d
done
