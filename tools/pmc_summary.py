#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes to per-kernel HBM traffic.

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> \
        [<state> <merged.json>]

With <state> (cold | hot, the --cache mode of the profiled bench.py run) the per-kernel table is also
stored under that key in <merged.json> (profiles/pmc_traffic.json), where bench.py looks its dominant
kernel up for `roofline.traffic`.  The merged file is stamped (`_meta`: commit from $ADVX_COMMIT - .git does not
travel to the GPU box -, UTC date, command) and bench.py echoes the stamp as `roofline.traffic_source`: the
figure is a STORED counter pass, not a live counter of the benchmark run.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced
streaming read -> doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
"""
import collections
import csv
import datetime
import json
import os
import re
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            m = re.search(r"advx::(k_[a-z_]+)", r["Kernel_Name"])
            if m:
                agg[m.group(1)].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f_raw = fetch.get(k, 0.0) * 1024
        w = write.get(k, 0.0) * 1024
        out[k] = {"fetch_bytes_raw": f_raw, "fetch_bytes_corrected": 2 * f_raw, "write_bytes": w,
                  "traffic_bytes_per_launch": 2 * f_raw + w, "launches_sampled": [nf.get(k, 0), nw.get(k, 0)]}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    if len(sys.argv) >= 6:
        state, merged_path = sys.argv[4], sys.argv[5]
        try:
            merged = json.load(open(merged_path))
        except (OSError, ValueError):
            merged = {}
        if not all(k in ("cold", "hot", "_meta") for k in merged):
            merged = {}                      # a round-1 table (kernels at the top level)
        merged[state] = out
        meta = merged.setdefault("_meta", {})
        meta[state] = {"commit": os.environ.get("ADVX_COMMIT", "unknown"),
                       "date_utc": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%d %H:%M"),
                       "collected_by": "tools/profile_bench.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of "
                                       f"`bench.py --cache {state} --steps 300`; FETCH_SIZE doubled per MI355X_MICROARCH.md"}
        json.dump(merged, open(merged_path, "w"), indent=1)
    for k, v in out.items():
        print(f"{k:20s} fetch(corr) {v['fetch_bytes_corrected'] / 1e6:8.2f} MB  write {v['write_bytes'] / 1e6:8.2f} MB  "
              f"traffic {v['traffic_bytes_per_launch'] / 1e6:8.2f} MB")


if __name__ == "__main__":
    main()
