#!/bin/bash
# Run on the GPU box (through gpurun): HBM-traffic and MFMA-utilisation counters of the bench's kernels.
# Separate rocprofv3 passes per counter group (TCC slots: FETCH_SIZE 3 + WRITE_SIZE 2 > 4), kernel-trace only.
# Usage: tools/pmc_traffic.sh <tag>  -> gpurun_out/<tag>/pmc_summary.json (+ per-pass logs)
set -e
TAG=${1:-r01_pmc}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ops --eager --one-stream"
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/raw$i -o pmc -- $CMD > $OUT/pass$i.log 2>&1 ) || echo "pass $i ($CTRS) failed"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/raw* > $OUT/pmc_summary.json
rm -rf $OUT/raw*
head -c 6000 $OUT/pmc_summary.json
