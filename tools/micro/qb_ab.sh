cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_modules_gpu.py tests/test_train_gpu.py -m gpu -x -q -k "ball or modules or loss or repulsion" 2>&1 | tail -2
python - <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tools")); sys.path.insert(0, os.getcwd())
import torch, ops_bench
for r in ops_bench.gpu_ops(torch.device("cuda:0"), quick=False):
    if "query_ball" in r["op"]: print(r["op"], r["shape"], r["us"], r.get("frac"))
PY
