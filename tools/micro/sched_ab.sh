cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_fused_gpu.py -m gpu -x -q 2>&1 | tail -3
