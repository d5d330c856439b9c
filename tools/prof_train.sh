#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats of the training step.  Usage: tools/prof_train.sh <tag> <batch> ["extra train_bench flags"]
set -e
TAG=${1:-r01_train}; B=${2:-8}; EXTRA=${3:-}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o prof -- python $GRAFT_REPO_ROOT/tools/train_bench.py --batch $B --steps 10 --warmup 2 $EXTRA > $OUT/bench_profiled.json 2> $OUT/rocprof.log || true
cd $GRAFT_REPO_ROOT
find $OUT/raw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python tools/train_bench.py --batch $B --steps 20 --warmup 3 $EXTRA > $OUT/bench.json 2>> $OUT/rocprof.log
rm -rf $OUT/raw
head -22 $OUT/kernel_stats.csv | cut -c1-170
cat $OUT/bench.json
