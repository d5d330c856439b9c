# rocprofv3 kernel statistics of the bench step with the dense blocks as separate launches (0) and as one launch each (1)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/stem1
mkdir -p $OUT
cd /tmp
for m in ${MODES:-0 1}; do
DISPU_STEM_FUSED=$m rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw$m -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ops --eager > /dev/null 2> $OUT/rocprof$m.log
find $OUT/raw$m -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$m.csv
rm -rf $OUT/raw$m
python - $OUT/kernel_stats_$m.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(t in r['Name'] for t in ('edge_dense','knn_feat','skinny','small_k')):
        print(r['Name'][:75], r['Calls'], "%.1f"%(float(r['AverageNs'])/1e3))
PY
done
