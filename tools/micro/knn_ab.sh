cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "knn" 2>&1 | tail -3
python - <<'PY'
import sys, os, json
sys.path.insert(0, os.path.join(os.getcwd(), "tools")); sys.path.insert(0, os.getcwd())
import torch, ops_bench
rows = ops_bench.gpu_ops(torch.device("cuda:0"), quick=False)
for r in rows:
    if "knn_xyz" in r["op"]: print(r["op"], r["shape"], r["us"], r.get("frac"))
PY
