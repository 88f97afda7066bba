#!/bin/bash
# GPU box helper: one rocprofv3 counter pass over a python program; prints per-kernel sums per dispatch.
# Usage: bash tools/pmc.sh <tag> "<counters>" <script.py> [args...]      (environment variables are inherited)
TAG=$1; COUNTERS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc $COUNTERS --output-format csv -d "$OUT" -- python3 "$REPO/$1" "${@:2}" > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 1; }
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
        key = (kn, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""), r["Counter_Name"])
        agg[key] += float(r["Counter_Value"]); n[key].add(r["Dispatch_Id"])
for key in sorted(agg):
    print(f"{key[0]:60s} grid {key[1]:>9s} {key[2]:28s} {agg[key] / len(n[key]):14.6g} per dispatch ({len(n[key])} dispatches)")
PY
