#!/bin/bash
# usage (on the GPU box, via gpurun): tools/pmc.sh <tag> "<counters>" -- <command...>
# Collects PMC counters per kernel dispatch (own run, kernel-trace only) and prints per-kernel averages.
set -e
TAG=$1; CTRS=$2; shift 3
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/raw -o pmc -- "$@" > $OUT/cmd.log 2>&1 ) || true
F=$(find $OUT/raw -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    n = max(len(v) for v in cs.values())
    print(k, "dispatches", n)
    for c, v in sorted(cs.items()):
        print("    %-32s avg %.4g" % (c, sum(v) / len(v)))
PY
rm -rf $OUT/raw
