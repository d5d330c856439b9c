# the distance-core rows of the per-op table, one line each (run on the GPU box)
OPS_ONLY=${OPS_ONLY:-knn_xyz,query_ball,three_nn,nn_distance} python tools/ops_bench.py 2>/dev/null | python -c "
import sys, json
for r in json.load(sys.stdin):
    print('%-32s %-22s %9.2f us  frac %.3f' % (r['op'], str(r['shape']), r['us'], r.get('frac') or 0))
"
