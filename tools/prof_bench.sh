#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the default bench command.
# Usage: tools/prof_bench.sh <tag>   -> gpurun_out/<tag>/{kernel_stats.csv,bench.json}
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# --one-stream: one stream, every kernel alone on the device -- the per-kernel averages then are the kernels' own times, comparable
# with bench.py's HIP-event pass (which folds the second stream back the same way); the timed step itself runs two streams
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ops --eager --one-stream > $OUT/bench_profiled.json 2> $OUT/rocprof.log || true
cd $GRAFT_REPO_ROOT
find $OUT/raw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2>> $OUT/rocprof.log
rm -rf $OUT/raw
head -25 $OUT/kernel_stats.csv
cat $OUT/bench.json
