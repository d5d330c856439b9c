#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel TRACE (start / end timestamps, queue) of a few training steps, reduced by
# tools/trace_timeline.py to the critical-path picture of the last step.  Usage: tools/trace_train.sh <tag> [batch] [dtype]
TAG=${1:-trace}; B=${2:-8}; DT=${3:-f32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o prof -- python $GRAFT_REPO_ROOT/tools/train_bench.py --batch $B --steps 4 --warmup 2 --dtype $DT $4 > $OUT/bench_profiled.json 2> $OUT/rocprof.log || true
cd $GRAFT_REPO_ROOT
F=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py $F > $OUT/timeline.txt
rm -rf $OUT/raw
head -150 $OUT/timeline.txt
