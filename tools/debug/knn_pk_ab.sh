# A/B inside one box: knn_xyz (32, 1024, 16) with scalar and packed distance arithmetic, alternating
for i in 1 2 3; do for pk in 0 1; do echo -n "pk=$pk "; DISPU_KNN_PK=$pk OPS_ONLY=knn_xyz python tools/ops_bench.py 2>/dev/null | python -c "
import sys, json
print(' '.join('%.2f' % r['us'] for r in json.load(sys.stdin)))"; done; done
