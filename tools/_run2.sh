export TMPDIR=/tmp
rm -rf gpurun_out/op.d; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/op.d -o p -- python3 tools/op_profile.py 512 > gpurun_out/op.log 2>&1
python3 - $(find gpurun_out/op.d -name '*kernel_stats.csv' | head -1) <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(.*', '', r['Name']).replace('void ', '')[:70]
    print(f"{name:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.2f} us min {float(r['MinNs'])/1e3:7.2f}")
PY
