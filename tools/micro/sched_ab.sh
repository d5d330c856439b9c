cd $GRAFT_REPO_ROOT
python tools/micro/c4_graph.py 2>&1 | grep -v amdgpu.ids | tail -5
