cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-ops 2>/dev/null | cut -c1-240
timeout 900 python -m pytest tests/test_generator_gpu.py tests/test_headline_gpu.py -m gpu -x -q 2>&1 | tail -2
