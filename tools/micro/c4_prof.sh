# rocprofv3 kernel statistics of the two generator passes of 16x upsampling (BASELINE configs[3]); run on the GPU box
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/r03_c4_prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o prof -- python $GRAFT_REPO_ROOT/tools/micro/c4_graph.py > $OUT/log.txt 2>&1 || true
cd $GRAFT_REPO_ROOT
find $OUT/raw -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/raw
head -30 $OUT/kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
