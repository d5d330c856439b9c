#!/bin/bash
# Run on the GPU box (through gpurun): VALU instruction counters of the per-op table's kernels (FPS / k-NN / ball query / 3-NN /
# nn_distance): how many vector instructions a wave issues against the ~8-op distance core, i.e. what the selection costs.
# One rocprofv3 --pmc pass (kernel-trace only), summarised per kernel by tools/pmc_summary.py.
# Usage: tools/pmc_valu_ops.sh <tag>  -> gpurun_out/<tag>/valu_summary.json
TAG=${1:-r04_valu}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/raw -o pmc -- python $GRAFT_REPO_ROOT/tools/valu_ops_driver.py > $OUT/pass.log 2>&1 ) || echo "pmc pass failed"
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/raw > $OUT/valu_raw.json
python3 - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
d = json.load(open(out + "/valu_raw.json"))
rows = {}
for k, r in d.items():
    if "SQ_INSTS_VALU" not in r or not r.get("SQ_WAVES"):
        continue
    rows[k] = {"dispatches": r["dispatches"], "valu_insts_per_wave": round(r["SQ_INSTS_VALU"] / r["SQ_WAVES"], 1),
               "waves_per_launch": round(r["SQ_WAVES"], 1), "wave_cycles_per_wave": round(r.get("SQ_WAVE_CYCLES", 0) / r["SQ_WAVES"], 1)}
json.dump(rows, open(out + "/valu_summary.json", "w"), indent=1)
print(json.dumps(rows, indent=1)[:4000])
PY
rm -rf $OUT/raw
