#!/bin/bash
# PMC counters of one chain configuration (development tool): tools/chain_pmc.sh <which> <chain> "<counters>"
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$1; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc $3 --kernel-trace --output-format csv -d $OUT -o c -- python3 tools/chain_profile.py $1 $2 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - $(find $OUT -name '*counter_collection.csv' | head -1) <<'PY'
import csv, re, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if 'advx' not in r['Kernel_Name']:
        continue
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void advx::', '').replace('advx::', '')
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name, cs in sorted(agg.items()):
    print(f"{name:30s} " + "  ".join(f"{k}={sum(v)/len(v):.3e}" for k, v in sorted(cs.items())))
PY
