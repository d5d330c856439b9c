cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r03_e_train_bf16
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o prof -- python $GRAFT_REPO_ROOT/tools/train_bench.py --batch 8 --steps 10 --warmup 2 --dtype bf16 > $OUT/bench_profiled.json 2> $OUT/rocprof.log || true
cd $GRAFT_REPO_ROOT
find $OUT/raw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/raw
head -3 $OUT/kernel_stats.csv | cut -c1-100
